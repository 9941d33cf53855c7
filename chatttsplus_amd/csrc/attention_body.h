// Single-query attention over the padded KV cache: the workgroup body shared by attention.hip's kernel and the o_proj launch that carries it (skinny_gemm.hip).
// See attention.hip for the reference citations.
#pragma once
#include "kernels.h"

template <typename WT> struct KvLoad;
template <> struct KvLoad<half_t> {
    __device__ static inline void load8(const half_t* p, float (&o)[8]) {
        const half8 v = *(const half8*)p;
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = (float)v[j];
    }
};
template <> struct KvLoad<float> {
    __device__ static inline void load8(const float* p, float (&o)[8]) {
        const f32x4 a = *(const f32x4*)p, b = *(const f32x4*)(p + 4);
        o[0] = a[0]; o[1] = a[1]; o[2] = a[2]; o[3] = a[3];
        o[4] = b[0]; o[5] = b[1]; o[6] = b[2]; o[7] = b[3];
    }
};

__device__ inline float safe_exp_diff(float m, float mn) { return (m == -INFINITY) ? 0.f : expf(m - mn); }

// The body of the decode attention kernel: workgroup (bx = row * heads + head, by = key split) on the first NW waves of the block.  It is a function so that
// the o_proj launch can carry the attention workgroups in front of its tiles (skinny_gemm.hip, "attention inside the o_proj launch"): with `flags`
// the normalised output is written through (64-bit relaxed agent-scope stores) and the (row, head)'s flag word is set to `tag` once the stores are out.
template <typename WT, int NW>
__device__ inline void attn_decode_body(const int bx, const int by, const int* done_p, const RowMeta* meta_p, const float* q_p, const void* k_p, const void* v_p,
                                        const int NHp, const int Sp, const AttnArgs& a, float (*merge)[8][10], unsigned* flags, const int draw_v, const unsigned tag_lo) {
    // leading scalars = what the first loads need; preloaded into SGPRs at wave launch (see skinny_gemm.hip)
    int done_v = 0;                                   // requested with the first operand loads, tested once they are in flight (common.h)
    if (done_p != nullptr) done_v = vload_flag(done_p);
    constexpr int UN = 4;                             // keys per lane group and loop iteration (loads in flight: 2 * UN * 16 B per lane; 8 measured slower)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int grp = lane >> 3, sub = lane & 7;
    const int r = bx / NHp, h = bx % NHp, s = by;
    const RowMeta m = meta_p[r];
    const int kv0 = m.kv_start, kv1 = m.slot + 1;
    const int nsplit = Sp;
    const int chunk = (kv1 - kv0 + nsplit - 1) / nsplit;
    const int p0 = kv0 + s * chunk;
    const int p1 = min(p0 + chunk, kv1);

    float q[8];
    {
        const float* qp = q_p + ((size_t)r * NHp + h) * CTTS_HEAD_DIM + 8 * sub;
        const f32x4 q0 = *(const f32x4*)qp, q1 = *(const f32x4*)(qp + 4);
        q[0] = q0[0] * 0.125f; q[1] = q0[1] * 0.125f; q[2] = q0[2] * 0.125f; q[3] = q0[3] * 0.125f;
        q[4] = q1[0] * 0.125f; q[5] = q1[1] * 0.125f; q[6] = q1[2] * 0.125f; q[7] = q1[3] * 0.125f;
    }
    if (__builtin_amdgcn_readfirstlane(done_v)) return;   // every sequence finished: skip on device
    const size_t head_off = ((size_t)m.seq * NHp + h) * a.Lmax * CTTS_HEAD_DIM + 8 * sub;
    const WT* kb = (const WT*)k_p + head_off;
    const WT* vb = (const WT*)v_p + head_off;

    float mrun = -INFINITY, lrun = 0.f, o[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = 0.f;

    // wave w, lane-group g handle keys p0 + 8*(NW*it + w) + g; the loop bound is wave-uniform (cross-lane ops inside)
    for (int wb = p0 + 8 * wave; wb < p1; wb += 8 * NW * UN) {
        const int base = wb + grp;
        float kf[UN][8], vf[UN][8];
        bool ok[UN];
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            const int p = base + 8 * NW * u;
            ok[u] = p < p1;
            const int pc = ok[u] ? p : kv0;                 // clamp: always a valid address
            KvLoad<WT>::load8(kb + (size_t)pc * CTTS_HEAD_DIM, kf[u]);
            KvLoad<WT>::load8(vb + (size_t)pc * CTTS_HEAD_DIM, vf[u]);
        }
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            float dot = 0.f;
#pragma unroll
            for (int j = 0; j < 8; ++j) dot += q[j] * kf[u][j];
            dot += dpp_f<DPP_XOR1>(dot);                    // 8-lane group sum on DPP (quad xor1, xor2, half-mirror)
            dot += dpp_f<DPP_XOR2>(dot);
            dot += dpp_f<DPP_HALF_MIRROR>(dot);
            if (ok[u]) {
                const float mn = fmaxf(mrun, dot);
                const float sc = safe_exp_diff(mrun, mn);
                const float pe = expf(dot - mn);
                lrun = lrun * sc + pe;
#pragma unroll
                for (int j = 0; j < 8; ++j) o[j] = o[j] * sc + pe * vf[u][j];
                mrun = mn;
            }
        }
    }
    // merge the 8 key groups of this wave (lanes with equal `sub`)
#pragma unroll
    for (int off = 8; off < 64; off <<= 1) {
        const float m2 = __shfl_xor(mrun, off), l2 = __shfl_xor(lrun, off);
        const float mn = fmaxf(mrun, m2);
        const float s1 = safe_exp_diff(mrun, mn), s2 = safe_exp_diff(m2, mn);
        lrun = lrun * s1 + l2 * s2;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float o2 = __shfl_xor(o[j], off);
            o[j] = o[j] * s1 + o2 * s2;
        }
        mrun = mn;
    }
    if (grp == 0) {
        merge[wave][sub][0] = mrun;
        merge[wave][sub][1] = lrun;
#pragma unroll
        for (int j = 0; j < 8; ++j) merge[wave][sub][2 + j] = o[j];
    }
    __syncthreads();
    if (tid < 8) {
        float M = merge[0][tid][0], L = merge[0][tid][1], O[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) O[j] = merge[0][tid][2 + j];
#pragma unroll
        for (int w = 1; w < NW; ++w) {
            const float m2 = merge[w][tid][0], l2 = merge[w][tid][1];
            const float mn = fmaxf(M, m2);
            const float s1 = safe_exp_diff(M, mn), s2 = safe_exp_diff(m2, mn);
            L = L * s1 + l2 * s2;
#pragma unroll
            for (int j = 0; j < 8; ++j) O[j] = O[j] * s1 + merge[w][tid][2 + j] * s2;
            M = mn;
        }
        if (a.packed_out != nullptr) {
            // single split: finish the softmax here and hand the o_proj kernel a ready MFMA B operand (no combine prologue)
            const float inv = 1.0f / L;
            const int NBr = 16 * a.nbg, K = a.NH * CTTS_HEAD_DIM, kt = K / WTraits<WT>::KT;
            const int chunk = r / NBr, n = r % NBr, k = h * CTTS_HEAD_DIM + 8 * tid;
            WT* dst = (WT*)a.packed_out + (size_t)chunk * a.nbg * kt * 64 * WTraits<WT>::EPL;
            if (flags != nullptr) {
                // consumers inside this launch: write-through stores, drained, then the flag (cdna_hip_programming.md Guideline 16 R1: no fences)
                WT ov[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) ov[j] = (WT)(O[j] * inv);
                constexpr int EPL = WTraits<WT>::EPL;                   // EPL consecutive k are 16 contiguous bytes of the operand image
#pragma unroll
                for (int j = 0; j < 8; j += EPL) {
                    unsigned long long* d8 = (unsigned long long*)(dst + xfrag_index<WT>(n, k + j, kt));
                    unsigned long long w0, w1;
                    __builtin_memcpy(&w0, &ov[j], 8); __builtin_memcpy(&w1, &ov[j + EPL / 2], 8);
                    __hip_atomic_store(d8, w0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    __hip_atomic_store(d8 + 1, w1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                if (tid == 0) __hip_atomic_store(flags + (size_t)r * a.NH + h, ((unsigned)draw_v + 1u) * 64u + tag_lo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                return;
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) dst[xfrag_index<WT>(n, k + j, kt)] = (WT)(O[j] * inv);
            return;
        }
        const size_t pi = ((size_t)r * a.NH + h) * a.S + s;
        if (tid == 0) { a.part_ml[pi * 2] = M; a.part_ml[pi * 2 + 1] = L; }
        float* po = a.part_o + pi * CTTS_HEAD_DIM + 8 * tid;
        *(f32x4*)po = (f32x4){O[0], O[1], O[2], O[3]};
        *(f32x4*)(po + 4) = (f32x4){O[4], O[5], O[6], O[7]};
    }
}

