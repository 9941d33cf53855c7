// Single-query attention over the padded KV cache (decode step and, row by row, the prompt pass).
//
// Restates LlamaSdpaAttention's SDPA call (chattts_plus/models/llama.py:653-661: scale 1/sqrt(64),
// additive mask = causal AND key-padding with finfo.min, llama.py:1073-1087).  Masked keys contribute
// exactly 0 in the reference's fp32 softmax (exp(finfo.min - max) == 0), so they are skipped here:
// a row attends to slots [kv_start, slot] of its own sequence.
//
// HBM-bound: each K/V element is read exactly once per step, 16 B per lane, 8 lanes per key row
// (coalesced 128-B rows, 8 keys per wave-instruction).  grid = (rows*heads, S key splits) so that a
// batch-1 step still spreads one head's keys over S CUs; every block writes a flash-decoding partial
// (running max, running sum, unnormalised output) that the o_proj kernel's prologue combines.
#include <stdlib.h>

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

// Round 5 layout (first used by the persistent launch's attention workgroups, persist_layer.hip): the scores come from K rows held 8 dims per lane (key group = lane / 8,
// a 3-step DPP sum per key), the output from V rows held ONE DIM PER LANE: a key's weight reaches all lanes through v_readlane, the wave's output is one register per
// lane, and the wave rescales once per iteration of 8 * UN keys (one wave-wide maximum) instead of once per key and lane.  The former layout kept 8 output dims per lane
// and key group and merged the 8 groups at the end with 3 rounds of 10 ds_bpermute + 2 exp each, then the waves one after the other on 8 threads.
template <typename WT, int NW>
__global__ __launch_bounds__(NW * 64) void attn_decode_kernel(const int* done_p, const RowMeta* meta_p, const float* q_p, const void* k_p, const void* v_p,
                                                            const int NHp, const int Sp, const AttnArgs a) {
    // leading scalars = what the first loads need; preloaded into SGPRs at wave launch (see skinny_gemm.hip)
    int done_v = 0;                                   // requested with the first operand loads, tested once they are in flight (common.h)
    if (done_p != nullptr) done_v = vload_flag(done_p);
    constexpr int UN = 4;                             // keys per lane group and loop iteration (loads in flight: 2 * UN * 16 B per lane; 8 measured slower)
    __shared__ float mo[NW][64];
    __shared__ float mm[NW], ml[NW];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);      // (uniform on purpose: the V rows' addresses are then scalar base + lane, not 64-bit vector arithmetic per key)
    const int grp = lane >> 3, sub = lane & 7;
    const int r = blockIdx.x / NHp, h = blockIdx.x % NHp, s = blockIdx.y;
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
    const size_t head_base = ((size_t)m.seq * NHp + h) * a.Lmax * CTTS_HEAD_DIM;
    const WT* kb = (const WT*)k_p + head_base + 8 * sub;
    const WT* vb = (const WT*)v_p + head_base + lane;

    float mrun = -INFINITY, lrun = 0.f;               // the wave's running maximum (the same in every lane); the weights of this lane's key group
    float oa[4] = {0.f, 0.f, 0.f, 0.f};               // dim `lane` of the output, as four partial sums (keys g, g + 4 of every group of 8: four independent FMA chains)

    // wave w, lane-group g handle keys p0 + 8*(NW*it + w) + g; the loop bound is wave-uniform (cross-lane ops inside)
    for (int wb = p0 + 8 * wave; wb < p1; wb += 8 * NW * UN) {
        float kf[UN][8];
        typename KvElem<WT>::reg vv[UN][8];
        bool ok[UN];
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            const int pb = wb + 8 * NW * u;           // first key of the wave's group of 8
            ok[u] = pb + grp < p1;
            KvLoad<WT>::load8(kb + (size_t)(ok[u] ? pb + grp : kv0) * CTTS_HEAD_DIM, kf[u]);       // clamp: always a valid address
#pragma unroll
            for (int g = 0; g < 8; ++g) vv[u][g] = KvElem<WT>::load(vb + (size_t)min(pb + g, p1 - 1) * CTTS_HEAD_DIM);     // (a key past the share has weight 0: any valid row will do)
        }
        float sc[UN], mloc = -INFINITY;
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            float dot = 0.f;
#pragma unroll
            for (int j = 0; j < 8; ++j) dot += q[j] * kf[u][j];
            dot += dpp_f<DPP_XOR1>(dot);                    // 8-lane group sum on DPP (quad xor1, xor2, half-mirror)
            dot += dpp_f<DPP_XOR2>(dot);
            dot += dpp_f<DPP_HALF_MIRROR>(dot);
            sc[u] = ok[u] ? dot : -INFINITY;
            mloc = fmaxf(mloc, sc[u]);
        }
        const float mn = wave_max(fmaxf(mrun, mloc));       // finite: the iteration holds at least one key (wb < p1)
        const float scl = safe_exp_diff(mrun, mn);
        lrun *= scl;
#pragma unroll
        for (int i = 0; i < 4; ++i) oa[i] *= scl;
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            if (wb + 8 * NW * u >= p1) break;               // (wave-uniform: no key of the wave in this group)
            const float pe = (sc[u] == -INFINITY) ? 0.f : expf(sc[u] - mn);
            lrun += pe;
#pragma unroll
            for (int g = 0; g < 8; ++g) oa[g & 3] = fmaf(readlane_f(pe, 8 * g), KvElem<WT>::f(vv[u][g]), oa[g & 3]);
        }
        mrun = mn;
    }
    const float lw = wave_sum(lrun) * 0.125f;               // (the 8 lanes of a key group hold the same weights)
    mo[wave][lane] = (oa[0] + oa[1]) + (oa[2] + oa[3]);
    if (lane == 0) { mm[wave] = mrun; ml[wave] = lw; }
    __syncthreads();
    if (tid < 64) {
        // lane = output dim; the waves' rescale factors side by side in lanes 0..NW-1
        const float mmv = mm[lane % NW], mlv = ml[lane % NW];
        const float mj = (lane < NW) ? mmv : -INFINITY, lj = (lane < NW) ? mlv : 0.f;
        const float M = wave_max(mj);
        const float sj = safe_exp_diff(mj, M);
        const float L = wave_sum(sj * lj);
        float O = 0.f;
#pragma unroll
        for (int w = 0; w < NW; ++w) O = fmaf(readlane_f(sj, w), mo[w][lane], O);
        if (a.packed_out != nullptr) {
            // single split: finish the softmax here and hand the o_proj kernel a ready MFMA B operand (no combine prologue)
            const float inv = 1.0f / L;
            const int NBr = 16 * a.nbg, K = a.NH * CTTS_HEAD_DIM, kt = K / WTraits<WT>::KT;
            const int chunk = r / NBr, n = r % NBr, k = h * CTTS_HEAD_DIM + lane;
            if (sizeof(WT) == 4 && a.packed_split) {
                // the split decode kernels' operand: head / tail fp16 images, 24 k-tiles of 2 KiB per 16-row group (a convex combination of V rows: in range)
                half_t hi, lo;
                split_half(O * inv, hi, lo, nullptr);
                char* p = (char*)a.packed_out + (size_t)chunk * a.nbg * (K / 32) * 2048 + xfrag_split_bytes(n, k, K / 32);
                *(half_t*)p = hi;
                *(half_t*)(p + 1024) = lo;
                return;
            }
            WT* dst = (WT*)a.packed_out + (size_t)chunk * a.nbg * kt * 64 * WTraits<WT>::EPL;
            dst[xfrag_index<WT>(n, k, kt)] = (WT)(O * inv);
            return;
        }
        const size_t pi = ((size_t)r * a.NH + h) * a.S + s;
        if (lane == 0) { a.part_ml[pi * 2] = M; a.part_ml[pi * 2 + 1] = L; }
        a.part_o[pi * CTTS_HEAD_DIM + lane] = O;
    }
}

// The former layout (rounds 1-4): 8 output dims per lane and key group, a rescale per key and lane, the 8 groups merged at the end through ds_bpermute.  Kept for fp16
// engines once there is a block per CU: there a V row is 128 bytes and one-dim-per-lane loads move 2 bytes per lane and instruction (ms/step at batch 32, fp16,
// this / the layout above: 0.587 / 0.628; below that the layout above wins: batch 5 0.445 / 0.431, 16 0.499 / 0.489).
template <typename WT, int NW>
__global__ __launch_bounds__(NW * 64) void attn_decode_g8_kernel(const int* done_p, const RowMeta* meta_p, const float* q_p, const void* k_p, const void* v_p,
                                                            const int NHp, const int Sp, const AttnArgs a) {
    // leading scalars = what the first loads need; preloaded into SGPRs at wave launch (see skinny_gemm.hip)
    int done_v = 0;                                   // requested with the first operand loads, tested once they are in flight (common.h)
    if (done_p != nullptr) done_v = vload_flag(done_p);
    constexpr int UN = 4;                             // keys per lane group and loop iteration (loads in flight: 2 * UN * 16 B per lane; 8 measured slower)
    __shared__ float merge[NW][8][10];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int grp = lane >> 3, sub = lane & 7;
    const int r = blockIdx.x / NHp, h = blockIdx.x % NHp, s = blockIdx.y;
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

// ------------------------------------------------------------------------------------------------
// Prompt pass, fp16: flash-style causal attention on MFMA (llama.py:590-668 at q_len > 1, mask semantics of llama.py:1073-1087).
// The single-query kernel above spends ~8 vector instructions per (query, key) pair (a VALU variant with 8 queries per wave read K/V
// 8x less often and was no faster: 42-65 % of a 32 x 512-token prompt pass; removed, see profiles/README.md).  Here
//   block = (sequence, head, 64 consecutive queries), 4 waves x 16 queries, keys in chunks of 64;
//   S^T[key][query] = K . Q^T   v_mfma_f32_16x16x32_f16, A = K rows (16 keys x 32 dims = one 16-byte LDS read per lane; the block stages
//                               each 64-key chunk once), B = Q (fp16, pre-scaled by 1/8).  In the C layout a lane owns ONE query (lane & 15) and the
//                               keys 4 * (lane >> 4) + j of the tile -- so the online-softmax state (running max, rescale factor) is a
//                               per-lane scalar, and P^T in that layout IS the B operand of v_mfma_f32_16x16x16_f16: no transpose of P;
//   O^T[dim][query] += V^T . P^T   v_mfma_f32_16x16x16_f16, A = V^T: the block stages the chunk's V rows transposed in LDS
//                               (double-buffered, one barrier per chunk), every wave reads its A fragments with 8-byte LDS loads.
// The running max is shared by the 4 lanes of a query (lanes 16 apart) with two shuffles per chunk; the row sums stay per-lane
// partials until the end.  (Requesting the K fragments one chunk ahead into a second register set measured slower: 59 -> 68 us per
// launch at 8192 rows -- the extra 32 VGPRs cost more occupancy than the exposed load latency.)  Output: normalised rows in the o_proj kernel's fragment-major B operand, like the other kernels.
#define FA_PITCH 68        // halfs per V^T row in LDS (64 keys + pad; rows stay 8-byte aligned)
#define FA_KPITCH 72       // halfs per K row in LDS (64 dims + pad; rows stay 16-byte aligned)
__global__ __launch_bounds__(256) void attn_prefill_mfma_kernel(const RowMeta* meta_p, const float* q_p, const half_t* k_p, const half_t* v_p, const int NHp,
                                                              const int R, const AttnArgs a) {
    __shared__ __attribute__((aligned(16))) half_t vt[2][CTTS_HEAD_DIM][FA_PITCH];
    __shared__ __attribute__((aligned(16))) half_t ks[2][64][FA_KPITCH];          // the chunk's K rows as they lie in the cache (+ pad)
    __shared__ int range_s[4][3];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int qn = lane & 15, iq = lane >> 4;
    const int b = blockIdx.z, h = blockIdx.y, T = a.T;
    const int t = blockIdx.x * 64 + wave * 16 + qn;                       // this lane's query (prompt position)
    const int r = b * T + t - a.row0;                                     // its row in the current pass
    const bool live = (t < T) && (r >= 0) && (r < R);
    RowMeta m = {0, 0, -1, 0};
    if (live) m = meta_p[r];
    const int lo = live ? m.kv_start : 0x7FFFFFFF, hi = live ? m.slot : -1;
    int wlo = lo, whi = hi;                                               // wave-uniform key range
    int wsq = live ? m.seq : -1;                                          // KV lane of this block's sequence (b itself after begin; any lane after ctts_gpt_admit)
#pragma unroll
    for (int off = 1; off < 16; off <<= 1) { wlo = min(wlo, __shfl_xor(wlo, off)); whi = max(whi, __shfl_xor(whi, off)); wsq = max(wsq, __shfl_xor(wsq, off)); }
    if (lane == 0) { range_s[wave][0] = wlo; range_s[wave][1] = whi; range_s[wave][2] = wsq; }
    __syncthreads();
    const int blo = min(min(range_s[0][0], range_s[1][0]), min(range_s[2][0], range_s[3][0]));
    const int bhi = max(max(range_s[0][1], range_s[1][1]), max(range_s[2][1], range_s[3][1]));
    if (bhi < 0) return;                                                  // no live query in this block (uniform)
    const int cseq = max(max(range_s[0][2], range_s[1][2]), max(range_s[2][2], range_s[3][2]));
    wlo = __builtin_amdgcn_readfirstlane(wlo); whi = __builtin_amdgcn_readfirstlane(whi);
    const size_t head_off = ((size_t)cseq * NHp + h) * a.Lmax * CTTS_HEAD_DIM;
    const half_t* kb = k_p + head_off;
    const half_t* vb = v_p + head_off;
    // Q fragments (B operand): lane (query qn, kq = iq): dims 8 iq .. + 7 of each 32-dim half, scaled by 1/sqrt(64)
    half8 qf[2];
    {
        const float* qp = q_p + ((size_t)(live ? r : 0) * NHp + h) * CTTS_HEAD_DIM + 8 * iq;
#pragma unroll
        for (int kh = 0; kh < 2; ++kh) {
            const f32x4 q0 = *(const f32x4*)(qp + 32 * kh), q1 = *(const f32x4*)(qp + 32 * kh + 4);
            const float sc = live ? 0.125f : 0.f;
            qf[kh] = (half8){(half_t)(q0[0] * sc), (half_t)(q0[1] * sc), (half_t)(q0[2] * sc), (half_t)(q0[3] * sc),
                             (half_t)(q1[0] * sc), (half_t)(q1[1] * sc), (half_t)(q1[2] * sc), (half_t)(q1[3] * sc)};
        }
    }
    f32x4 oacc[4];                                                        // O^T: dims 16 db + 4 iq + j of this lane's query
#pragma unroll
    for (int db = 0; db < 4; ++db) oacc[db] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float mrun = -INFINITY, lpart = 0.f;
    // V staging: thread -> key (tid >> 2) of the chunk, dims 16 (tid & 3) .. + 15
    const int skey = tid >> 2, sdim = 16 * (tid & 3);
    half8 vst[2], kst[2];
    auto vload = [&](int c) {
        const int key = min(c + skey, bhi);                               // clamp: a valid slot of this sequence (masked later)
        const half8* vp = (const half8*)(vb + (size_t)key * CTTS_HEAD_DIM + sdim);
        const half8* kp = (const half8*)(kb + (size_t)key * CTTS_HEAD_DIM + sdim);
        vst[0] = vp[0]; vst[1] = vp[1];
        kst[0] = kp[0]; kst[1] = kp[1];
    };
    auto vstore = [&](int buf) {
#pragma unroll
        for (int e = 0; e < 8; ++e) { vt[buf][sdim + e][skey] = vst[0][e]; vt[buf][sdim + 8 + e][skey] = vst[1][e]; }
        *(half8*)&ks[buf][skey][sdim] = kst[0];
        *(half8*)&ks[buf][skey][sdim + 8] = kst[1];
    };
    const int c0 = blo & ~63;
    vload(c0);
    vstore(0);
    __syncthreads();
    int buf = 0;
    for (int c = c0; c <= bhi; c += 64, buf ^= 1) {
        const bool more = c + 64 <= bhi;
        if (more) vload(c + 64);
        if (c + 63 >= wlo && c <= whi) {                                   // this wave has keys in the chunk
            f32x4 sT[4];
#pragma unroll
            for (int tl = 0; tl < 4; ++tl) {
                // A operand row = key (lane & 15), dims 8 iq .. and 32 + 8 iq ..: from the chunk's K rows staged in LDS by the whole block
                // (each wave fetching them itself from the cache meant 4x the loads, consumed right after they were issued)
                const half8 k0 = *(const half8*)&ks[buf][16 * tl + qn][8 * iq], k1 = *(const half8*)&ks[buf][16 * tl + qn][32 + 8 * iq];
                f32x4 acc = {0.f, 0.f, 0.f, 0.f};
                acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(k0, qf[0], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(k1, qf[1], acc, 0, 0, 0);
                sT[tl] = acc;
            }
            float mloc = -INFINITY;
#pragma unroll
            for (int tl = 0; tl < 4; ++tl)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int kidx = c + 16 * tl + 4 * iq + j;
                    const bool ok = (kidx >= lo) && (kidx <= hi);
                    sT[tl][j] = ok ? sT[tl][j] : -INFINITY;
                    mloc = fmaxf(mloc, sT[tl][j]);
                }
            mloc = fmaxf(mloc, __shfl_xor(mloc, 16));
            mloc = fmaxf(mloc, __shfl_xor(mloc, 32));
            const float mn = fmaxf(mrun, mloc);
            const float sc = (mrun == -INFINITY) ? 0.f : __expf(mrun - mn);
            float ps = 0.f;
            half4 pT[4];
#pragma unroll
            for (int tl = 0; tl < 4; ++tl) {
                float p[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) { p[j] = (sT[tl][j] == -INFINITY) ? 0.f : __expf(sT[tl][j] - mn); ps += p[j]; }
                pT[tl] = (half4){(half_t)p[0], (half_t)p[1], (half_t)p[2], (half_t)p[3]};
            }
            lpart = lpart * sc + ps;
            mrun = mn;
#pragma unroll
            for (int db = 0; db < 4; ++db) { oacc[db][0] *= sc; oacc[db][1] *= sc; oacc[db][2] *= sc; oacc[db][3] *= sc; }
#pragma unroll
            for (int tl = 0; tl < 4; ++tl)
#pragma unroll
                for (int db = 0; db < 4; ++db) {
                    const half4 vf = *(const half4*)&vt[buf][16 * db + qn][16 * tl + 4 * iq];      // A = V^T: row = dim (lane & 15), keys 4 iq ..
                    oacc[db] = __builtin_amdgcn_mfma_f32_16x16x16f16(vf, pT[tl], oacc[db], 0, 0, 0);
                }
        }
        if (more) vstore(buf ^ 1);
        __syncthreads();
    }
    float ltot = lpart + __shfl_xor(lpart, 16);
    ltot += __shfl_xor(ltot, 32);
    if (!live) return;
    const float inv = 1.0f / ltot;
    const int NBr = 16 * a.nbg, K = a.NH * CTTS_HEAD_DIM, kt = K / 32;
    const int chunk = r / NBr, n = r % NBr;
    half_t* dst = (half_t*)a.packed_out + (size_t)chunk * a.nbg * kt * 64 * 8;
#pragma unroll
    for (int db = 0; db < 4; ++db) {
        const int k = h * CTTS_HEAD_DIM + 16 * db + 4 * iq;
        *(half4*)(dst + xfrag_index<half_t>(n, k, kt)) = (half4){(half_t)(oacc[db][0] * inv), (half_t)(oacc[db][1] * inv), (half_t)(oacc[db][2] * inv), (half_t)(oacc[db][3] * inv)};
    }
}

int launch_attention(int dtype, const AttnArgs& a, hipStream_t s) {
    dim3 grid(a.R * a.NH, a.S), block(256);
    const int* done_p = a.st ? &a.st->all_done : nullptr;
    // unsplit rows: 8 waves per (row, head) keep twice the K/V bytes in flight per CU while there are fewer blocks than CUs; from one
    // block per CU on, 4-wave blocks balance better (us/step at mean context 312: batch 8 471 vs 479, 16 512 vs 521 | 32 613 vs 598, 64 800 vs 774)
    static const int wide_env = diag_env("CTTS_ATTN_WIDE") ? atoi(diag_env("CTTS_ATTN_WIDE")) : -1;     // diagnostic builds: force 8-wave (1) / 4-wave (0) blocks
    const bool wide = (a.S == 1) && (a.st != nullptr) && (wide_env < 0 ? (a.R * a.NH < (a.wide_blocks > 0 ? a.wide_blocks : 256)) : wide_env != 0);
    static const int tiled_env = diag_env("CTTS_PREFILL_ATTN") ? atoi(diag_env("CTTS_PREFILL_ATTN")) : 1;    // diagnostic builds: 0 = prompt attention row by row
    if (a.st == nullptr && a.S == 1 && a.packed_out != nullptr && a.R >= 64 && dtype == 1 && a.T > 0 && tiled_env >= 1) {
        // prompt pass, fp16: MFMA flash attention, block = (64 queries, head, sequence)
        const int B = (a.row0 + a.R + a.T - 1) / a.T;                     // sequences 0 .. B-1 may have rows in this pass
        dim3 g3((a.T + 63) / 64, a.NH, B);
        hipLaunchKernelGGL(attn_prefill_mfma_kernel, g3, dim3(256), 0, s, a.meta, a.q, (const half_t*)a.k_cache, (const half_t*)a.v_cache, a.NH, a.R, a);
    } else if (wide) {
        if (dtype == 1) hipLaunchKernelGGL((attn_decode_kernel<half_t, 8>), grid, dim3(512), 0, s, done_p, a.meta, a.q, a.k_cache, a.v_cache, a.NH, a.S, a);
        else hipLaunchKernelGGL((attn_decode_kernel<float, 8>), grid, dim3(512), 0, s, done_p, a.meta, a.q, a.k_cache, a.v_cache, a.NH, a.S, a);
    } else if (dtype == 1) {
        if (a.S == 1 && a.st != nullptr) hipLaunchKernelGGL((attn_decode_g8_kernel<half_t, 4>), grid, block, 0, s, done_p, a.meta, a.q, a.k_cache, a.v_cache, a.NH, a.S, a);
        else hipLaunchKernelGGL((attn_decode_kernel<half_t, 4>), grid, block, 0, s, done_p, a.meta, a.q, a.k_cache, a.v_cache, a.NH, a.S, a);
    }
    else hipLaunchKernelGGL((attn_decode_kernel<float, 4>), grid, block, 0, s, done_p, a.meta, a.q, a.k_cache, a.v_cache, a.NH, a.S, a);
    CTTS_HIP_CHECK(hipGetLastError());
    return 0;
}
