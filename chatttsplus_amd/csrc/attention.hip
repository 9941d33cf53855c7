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

// one 16-row o_proj tile x one head's 64 input dims on MFMA (B operand = the head's attention output in column 0)
template <typename WT> struct HeadMma;
template <> struct HeadMma<half_t> {
    static constexpr int FR = 2;                      // 64 dims = 2 k-tiles of 32
    typedef half8 frag;
    __device__ static inline f32x4 run(const frag (&w)[FR], const float* o_s, int lane) {
        f32x4 c = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int f = 0; f < FR; ++f) {
            half8 b;
#pragma unroll
            for (int j = 0; j < 8; ++j) b[j] = ((lane & 15) == 0) ? (half_t)o_s[f * 32 + 8 * (lane >> 4) + j] : (half_t)0.f;
            c = __builtin_amdgcn_mfma_f32_16x16x32_f16(w[f], b, c, 0, 0, 0);
        }
        return c;
    }
};
template <> struct HeadMma<float> {
    static constexpr int FR = 4;                      // 64 dims = 4 k-tiles of 16
    typedef f32x4 frag;
    __device__ static inline f32x4 run(const frag (&w)[FR], const float* o_s, int lane) {
        f32x4 c = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int f = 0; f < FR; ++f)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float b = ((lane & 15) == 0) ? o_s[f * 16 + 4 * (lane >> 4) + j] : 0.f;
                c = __builtin_amdgcn_mfma_f32_16x16x4f32(w[f][j], b, c, 0, 0, 0);
            }
        return c;
    }
};
#define FUSE_TPW 3      // o_proj tiles per wave in the fused kernel: 48 tiles = JT(4) x 4 waves x 3

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

// FUSED: grid.y = JT column groups instead of key splits; every block runs the whole (row, head) attention and then
// multiplies its head's output with its share of the o_proj rows (weights prefetched at kernel entry), writing a
// per-head partial sum  opart[row][head][:]  that the next kernels add to the residual stream in head order.
// Drops one of the five dependent launches per layer in the launch-latency-bound small-batch regime.
template <typename WT, bool FUSED, int NW, int UNR = 4>
__global__ __launch_bounds__(NW * 64) void attn_decode_kernel(const int* done_p, const RowMeta* meta_p, const float* q_p, const void* k_p, const void* v_p,
                                                            const int NHp, const int Sp, const AttnArgs a) {
    // leading scalars = what the first loads need; preloaded into SGPRs at wave launch (see skinny_gemm.hip)
    static_assert(!FUSED || NW == 4, "fused o_proj phase assumes 4 waves");
    int done_v = 0;                                   // requested with the first operand loads, tested once they are in flight (common.h)
    if (done_p != nullptr) done_v = vload_flag(done_p);
    constexpr int UN = UNR;                           // keys per lane group and loop iteration (loads in flight: 2 * UN * 16 B per lane)
    __shared__ float merge[NW][8][10];
    __shared__ float o_s[CTTS_HEAD_DIM];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int grp = lane >> 3, sub = lane & 7;
    const int r = blockIdx.x / NHp, h = blockIdx.x % NHp, s = FUSED ? 0 : blockIdx.y;
    typedef HeadMma<WT> HM;
    typename HM::frag wfr[FUSE_TPW][HM::FR];
    if (FUSED) {        // this wave's o_proj tiles: issued first, consumed after the attention loop
        const int ktiles = (a.NH * CTTS_HEAD_DIM) / WTraits<WT>::KT;
#pragma unroll
        for (int t = 0; t < FUSE_TPW; ++t) {
            const int rt = (blockIdx.y * 4 + wave) * FUSE_TPW + t;
#pragma unroll
            for (int f = 0; f < HM::FR; ++f)
                wfr[t][f] = __builtin_nontemporal_load((const typename HM::frag*)a.wo + ((size_t)rt * ktiles + h * HM::FR + f) * 64 + lane);
        }
    }
    const RowMeta m = meta_p[r];
    const int kv0 = m.kv_start, kv1 = m.slot + 1;
    const int nsplit = FUSED ? 1 : Sp;
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
        if (FUSED) {
            const float inv = 1.0f / L;
#pragma unroll
            for (int j = 0; j < 8; ++j) o_s[8 * tid + j] = O[j] * inv;
        } else if (a.packed_out != nullptr) {
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
        if (!FUSED && tid == 0) { a.part_ml[pi * 2] = M; a.part_ml[pi * 2 + 1] = L; }
        float* po = a.part_o + pi * CTTS_HEAD_DIM + 8 * tid;
        if (!FUSED) {
            *(f32x4*)po = (f32x4){O[0], O[1], O[2], O[3]};
            *(f32x4*)(po + 4) = (f32x4){O[4], O[5], O[6], O[7]};
        }
    }
    if (FUSED) {
        __syncthreads();
        const int H = a.NH * CTTS_HEAD_DIM;
#pragma unroll
        for (int t = 0; t < FUSE_TPW; ++t) {
            const int rt = (blockIdx.y * 4 + wave) * FUSE_TPW + t;
            const f32x4 c = HM::run(wfr[t], o_s, lane);                 // C[row=(lane>>4)*4+reg][col=lane&15]; col 0 is ours
            if ((lane & 15) == 0) *(f32x4*)(a.opart + ((size_t)r * a.NH + h) * H + rt * 16 + (lane >> 4) * 4) = c;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Fused decode launch for one or two rows: llama.py:82-87 (input RMSNorm) + :619-633 (q/k/v projection, RoPE, cache append) +
// :653-661 (SDPA) in one kernel, so that a layer is 4 dependent launches instead of 5 and q/k/v never round-trip through HBM.
//   grid.x = (row, head);  grid.y = kind:  kind < S   q projection (4 weight tiles = 96 KB fp16) + attention over split `kind`
//                                                       of the CACHED keys [kv_start, slot)
//                                          kind == S   q and k projections of the new token: its score q.k (partial slot S: m = score,
//                                                       l = 1) and the K append
//                                          kind == S+1 v projection: V append and the output row of partial slot S
// Same arithmetic as the separate kernels (norm expression, tile order, RoPE with separately rounded products, K/V rounded to
// the cache dtype before use); the K dimension is split over NW waves x 3 k-tiles instead of 4 (8) x 6, so sums differ in the
// last bits from the 5-launch path.
template <typename WT> struct FragSel;
template <> struct FragSel<half_t> { typedef half8 type; };
template <> struct FragSel<float> { typedef f32x4 type; };
template <typename WT> struct MmaQ;
template <> struct MmaQ<half_t> { __device__ static inline f32x4 run(half8 a, half8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0); } };
template <> struct MmaQ<float> {
    __device__ static inline f32x4 run(f32x4 a, f32x4 b, f32x4 c) {
#pragma unroll
        for (int j = 0; j < 4; ++j) c = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j], b[j], c, 0, 0, 0);
        return c;
    }
};

template <typename WT, int NW, int TPP>
__global__ __launch_bounds__(NW * 64) void qkv_attn_kernel(const int* done_p, const void* Wp, const float* x_p, const RowMeta* meta_p, const int NHp, const int nS,
                                                          const QkvAttnArgs a) {
    typedef typename FragSel<WT>::type frag;
    constexpr int K = 768, KT = WTraits<WT>::KT, EPL = WTraits<WT>::EPL, KTILES = K / KT, KPW = KTILES / NW, UN = 4, HT = K / 16;
    static_assert(KPW * NW == KTILES, "K split");
    __shared__ float xn[K];
    __shared__ float red[NW][8][16];
    __shared__ float qk[2][CTTS_HEAD_DIM];
    __shared__ float merge[NW][8][10];
    int done_v = 0;
    if (done_p != nullptr) done_v = vload_flag(done_p);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = blockIdx.x / NHp, h = blockIdx.x % NHp, kind = blockIdx.y, S = nS >> 8, np = nS & 0xFF;
    const bool isK = kind == S, isV = kind == S + 1;
    const int ntile = isK ? 8 : 4;
    // weight tiles of this block: q = tiles h*4 .. h*4+3 of the first projection, k / v = the same tiles of the 2nd / 3rd
    auto tile_of = [&](int t) -> int { return isV ? 2 * HT + h * 4 + t : (t < 4 ? h * 4 + t : HT + h * 4 + (t - 4)); };
    frag wf[TPP][KPW];
    auto load_tiles = [&](int t0) {
#pragma unroll
        for (int t = 0; t < TPP; ++t)
            if (t0 + t < ntile) {
                const frag* w = (const frag*)Wp + ((size_t)tile_of(t0 + t) * KTILES + wave * KPW) * 64 + lane;
#pragma unroll
                for (int i = 0; i < KPW; ++i) wf[t][i] = __builtin_nontemporal_load(w + i * 64);
            }
    };
    const RowMeta m = meta_p[r];
    if (wave == 0) {                                  // the row's RMSNorm, same expression order as the PRO_NORM[_P] prologue
        const f32x4* xr = (const f32x4*)(x_p + (size_t)r * K);
        f32x4 v[3];
#pragma unroll
        for (int i = 0; i < 3; ++i) v[i] = xr[lane + 64 * i];
        if (np > 0) {
            f32x4 pp[4][3];
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int i = 0; i < 3; ++i)
                    pp[q][i] = (q < np) ? ((const f32x4*)(a.dpart + ((size_t)r * np + q) * K))[lane + 64 * i] : (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int i = 0; i < 3; ++i) v[i] += pp[q][i];
        }
        __builtin_amdgcn_sched_barrier(0);
        load_tiles(0);
        __builtin_amdgcn_sched_barrier(0);
        float ss = 0.f;
#pragma unroll
        for (int i = 0; i < 3; ++i) ss += v[i][0] * v[i][0] + v[i][1] * v[i][1] + v[i][2] * v[i][2] + v[i][3] * v[i][3];
        ss = wave_sum(ss);
        const float rs = 1.0f / sqrtf(ss / (float)K + a.eps);
#pragma unroll
        for (int i = 0; i < 3; ++i) *(f32x4*)(xn + 4 * (lane + 64 * i)) = (f32x4){v[i][0] * rs, v[i][1] * rs, v[i][2] * rs, v[i][3] * rs};
    } else {
        load_tiles(0);
    }
    if (__builtin_amdgcn_readfirstlane(done_v)) return;   // every sequence finished: skip on device
    __syncthreads();
    for (int t0 = 0; t0 < ntile; t0 += TPP) {
        if (t0 > 0) load_tiles(t0);                   // parity mode: second phase of the 8-tile block
        f32x4 acc[TPP];
#pragma unroll
        for (int t = 0; t < TPP; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < KPW; ++i) {
            const int kt = wave * KPW + i;
            frag b;
#pragma unroll
            for (int j = 0; j < EPL; ++j) b[j] = ((lane & 15) == 0) ? (WT)xn[kt * KT + (lane >> 4) * EPL + j] : (WT)0.f;
#pragma unroll
            for (int t = 0; t < TPP; ++t)
                if (t0 + t < ntile) acc[t] = MmaQ<WT>::run(wf[t][i], b, acc[t]);
        }
        if ((lane & 15) == 0) {                       // column 0 of C = this row: lanes 0, 16, 32, 48 hold weight rows (lane / 16) * 4 + reg
#pragma unroll
            for (int t = 0; t < TPP; ++t)
                if (t0 + t < ntile) {
#pragma unroll
                    for (int g = 0; g < 4; ++g) red[wave][t0 + t][(lane >> 4) * 4 + g] = acc[t][g];
                }
        }
    }
    __syncthreads();
    const size_t pslot = ((size_t)r * NHp + h) * (S + 1);
    if (tid < ntile * 8) {
        const int t = tid >> 3, p = tid & 7, d = (t & 3) * 8 + p;     // tile rows p, p + 8 = head dims d, d + 32
        float va = 0.f, vb = 0.f;
#pragma unroll
        for (int w = 0; w < NW; ++w) { va += red[w][t][p]; vb += red[w][t][p + 8]; }
        const size_t cslot = (((size_t)m.seq * NHp + h) * a.Lmax + m.slot) * CTTS_HEAD_DIM;
        if (isV) {
            const WT ra = (WT)va, rb = (WT)vb;
            WT* c = (WT*)a.v_cache + cslot;
            c[d] = ra; c[d + 32] = rb;
            float* po = a.part_o + (pslot + S) * CTTS_HEAD_DIM;
            po[d] = (float)ra; po[d + 32] = (float)rb;
        } else {
            const float rc = a.rope_rows[(size_t)r * 64 + d], rsn = a.rope_rows[(size_t)r * 64 + 32 + d];
            const float ya = __fadd_rn(__fmul_rn(va, rc), __fmul_rn(-vb, rsn));   // llama.py:180-181
            const float yb = __fadd_rn(__fmul_rn(vb, rc), __fmul_rn(va, rsn));
            if (t < 4) { qk[0][d] = ya; qk[0][d + 32] = yb; }
            else {
                const WT ra = (WT)ya, rb = (WT)yb;
                WT* c = (WT*)a.k_cache + cslot;
                c[d] = ra; c[d + 32] = rb;
                qk[1][d] = (float)ra; qk[1][d + 32] = (float)rb;
            }
        }
    }
    if (isV) return;
    __syncthreads();
    const int grp = lane >> 3, sub = lane & 7;
    float q[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) q[j] = qk[0][8 * sub + j] * 0.125f;
    if (isK) {                                        // the new token attends to itself: partial (m = score, l = 1, o = v_new)
        if (wave == 0) {
            float dot = 0.f;
#pragma unroll
            for (int j = 0; j < 8; ++j) dot += q[j] * qk[1][8 * sub + j];
            dot += dpp_f<DPP_XOR1>(dot);
            dot += dpp_f<DPP_XOR2>(dot);
            dot += dpp_f<DPP_HALF_MIRROR>(dot);
            if (lane == 0) { a.part_ml[(pslot + S) * 2] = dot; a.part_ml[(pslot + S) * 2 + 1] = 1.0f; }
        }
        return;
    }
    // cached keys of this split
    const int kv0 = m.kv_start, kv1 = m.slot;
    const int chunk = (max(kv1 - kv0, 0) + S - 1) / S;
    const int p0 = kv0 + kind * chunk, p1 = min(p0 + chunk, kv1);
    const size_t head_off = ((size_t)m.seq * NHp + h) * a.Lmax * CTTS_HEAD_DIM + 8 * sub;
    const WT* kb = (const WT*)a.k_cache + head_off;
    const WT* vb = (const WT*)a.v_cache + head_off;
    float mrun = -INFINITY, lrun = 0.f, o[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = 0.f;
    for (int wb = p0 + 8 * wave; wb < p1; wb += 8 * NW * UN) {
        const int base = wb + grp;
        float kf[UN][8], vf[UN][8];
        bool ok[UN];
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            const int p = base + 8 * NW * u;
            ok[u] = p < p1;
            const int pc = ok[u] ? p : kv0;
            KvLoad<WT>::load8(kb + (size_t)pc * CTTS_HEAD_DIM, kf[u]);
            KvLoad<WT>::load8(vb + (size_t)pc * CTTS_HEAD_DIM, vf[u]);
        }
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            float dot = 0.f;
#pragma unroll
            for (int j = 0; j < 8; ++j) dot += q[j] * kf[u][j];
            dot += dpp_f<DPP_XOR1>(dot);
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
        const size_t pi = pslot + kind;
        if (tid == 0) { a.part_ml[pi * 2] = M; a.part_ml[pi * 2 + 1] = L; }
        float* po = a.part_o + pi * CTTS_HEAD_DIM + 8 * tid;
        *(f32x4*)po = (f32x4){O[0], O[1], O[2], O[3]};
        *(f32x4*)(po + 4) = (f32x4){O[4], O[5], O[6], O[7]};
    }
}

int launch_qkv_attention(int dtype, const QkvAttnArgs& a, hipStream_t s) {
    if (a.S < 1 || a.S > 7 || a.np > 4) { ctts_set_error("qkv_attention: S=%d np=%d out of range", a.S, a.np); return 1; }
    const dim3 grid(a.R * a.NH, a.S + 2);
    const int* done_p = a.st ? &a.st->all_done : nullptr;
    const int nS = (a.np & 0xFF) | (a.S << 8);
    if (dtype == 1) hipLaunchKernelGGL((qkv_attn_kernel<half_t, 8, 8>), grid, dim3(512), 0, s, done_p, a.wqkv, a.x, a.meta, a.NH, nS, a);
    else hipLaunchKernelGGL((qkv_attn_kernel<float, 16, 4>), grid, dim3(1024), 0, s, done_p, a.wqkv, a.x, a.meta, a.NH, nS, a);
    CTTS_HIP_CHECK(hipGetLastError());
    return 0;
}

// ------------------------------------------------------------------------------------------------
// Prompt pass: causal attention for QT = 8 consecutive query rows per wavefront (llama.py:590-668 at q_len > 1; mask semantics of
// llama.py:1073-1087: a row attends to the key slots [kv_start, slot] of its own sequence -- left padding and the future are skipped).
// The row-by-row kernel above reads a row's whole K/V prefix once per ROW: B*T^2/2 keys per head and layer (12.9 GB per layer at
// 32 x 512 prompt tokens, 42 % of that prompt pass).  Here a lane group of 8 lanes owns one query (8 of the 64 dims per lane, like
// the decode kernel) and the 8 groups of a wave walk the keys together, so a K/V row is fetched once per 8 queries (the 8 groups
// request the same 128 bytes: one line) and no cross-lane merge is needed -- every group carries its query's complete online
// softmax state.  Keys are taken 4 at a time (one rescale per 4 keys).  Output: the normalised rows, written straight into the
// o_proj kernel's fragment-major B operand.
template <typename WT>
__global__ __launch_bounds__(256) void attn_prefill_kernel(const RowMeta* meta_p, const float* q_p, const void* k_p, const void* v_p, const int NHp, const int R,
                                                         const AttnArgs a) {
    constexpr int QT = 8, UN = 4;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int grp = lane >> 3, sub = lane & 7;
    const int h = blockIdx.y;
    const int r = (blockIdx.x * 4 + wave) * QT + grp;                    // this lane group's query row
    const bool live = r < R;
    RowMeta m = {0, 0, -1, 0};
    if (live) m = meta_p[r];
    const int lo = live ? m.kv_start : 0x7FFFFFFF, hi = live ? m.slot : -1;       // attended key slots [lo, hi]
    // wave-uniform walk over the union of the groups' ranges
    int wlo = lo, whi = hi;
#pragma unroll
    for (int off = 8; off < 64; off <<= 1) { wlo = min(wlo, __shfl_xor(wlo, off)); whi = max(whi, __shfl_xor(whi, off)); }
    wlo = __builtin_amdgcn_readfirstlane(wlo); whi = __builtin_amdgcn_readfirstlane(whi);
    if (whi < 0) return;                                                   // no live row in this wave
    float q[8];
    {
        const float* qp = q_p + ((size_t)(live ? r : 0) * NHp + h) * CTTS_HEAD_DIM + 8 * sub;
        const f32x4 q0 = *(const f32x4*)qp, q1 = *(const f32x4*)(qp + 4);
        q[0] = q0[0] * 0.125f; q[1] = q0[1] * 0.125f; q[2] = q0[2] * 0.125f; q[3] = q0[3] * 0.125f;
        q[4] = q1[0] * 0.125f; q[5] = q1[1] * 0.125f; q[6] = q1[2] * 0.125f; q[7] = q1[3] * 0.125f;
    }
    const size_t head_off = ((size_t)m.seq * NHp + h) * a.Lmax * CTTS_HEAD_DIM + 8 * sub;
    const WT* kb = (const WT*)k_p + head_off;
    const WT* vb = (const WT*)v_p + head_off;
    float mrun = -INFINITY, lrun = 0.f, o[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = 0.f;
    for (int p0 = wlo; p0 <= whi; p0 += UN) {
        float kf[UN][8], vf[UN][8], dot[UN];
        bool ok[UN];
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            const int p = p0 + u;
            ok[u] = (p >= lo) && (p <= hi);
            const int pc = ok[u] ? p : (live ? hi : 0);                    // clamp: always a valid slot of this group's sequence
            KvLoad<WT>::load8(kb + (size_t)pc * CTTS_HEAD_DIM, kf[u]);
            KvLoad<WT>::load8(vb + (size_t)pc * CTTS_HEAD_DIM, vf[u]);
        }
        float mn = mrun;
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            float d = 0.f;
#pragma unroll
            for (int j = 0; j < 8; ++j) d += q[j] * kf[u][j];
            d += dpp_f<DPP_XOR1>(d);                                       // 8-lane group sum on DPP
            d += dpp_f<DPP_XOR2>(d);
            d += dpp_f<DPP_HALF_MIRROR>(d);
            dot[u] = ok[u] ? d : -INFINITY;
            mn = fmaxf(mn, dot[u]);
        }
        const float sc = safe_exp_diff(mrun, mn);
        float pe[UN], ps = 0.f;
#pragma unroll
        for (int u = 0; u < UN; ++u) { pe[u] = safe_exp_diff(dot[u], mn); ps += pe[u]; }
        lrun = lrun * sc + ps;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            float acc = o[j] * sc;
#pragma unroll
            for (int u = 0; u < UN; ++u) acc += pe[u] * vf[u][j];
            o[j] = acc;
        }
        mrun = mn;
    }
    if (!live) return;
    const float inv = 1.0f / lrun;
    const int NBr = 16 * a.nbg, K = a.NH * CTTS_HEAD_DIM, kt = K / WTraits<WT>::KT;
    const int chunk = r / NBr, n = r % NBr, k = h * CTTS_HEAD_DIM + 8 * sub;
    WT* dst = (WT*)a.packed_out + (size_t)chunk * a.nbg * kt * 64 * WTraits<WT>::EPL;
#pragma unroll
    for (int j = 0; j < 8; ++j) dst[xfrag_index<WT>(n, k + j, kt)] = (WT)(o[j] * inv);
}

// ------------------------------------------------------------------------------------------------
// Prompt pass, fp16: flash-style causal attention on MFMA (llama.py:590-668 at q_len > 1, mask semantics of llama.py:1073-1087).
// The VALU kernels above spend ~8 vector instructions per (query, key) pair whatever the tiling (attn_prefill_kernel reads K/V
// 8x less often than the row-by-row kernel and is no faster: 42-65 % of a 32 x 512-token prompt pass).  Here
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
    __shared__ int range_s[4][2];
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
#pragma unroll
    for (int off = 1; off < 16; off <<= 1) { wlo = min(wlo, __shfl_xor(wlo, off)); whi = max(whi, __shfl_xor(whi, off)); }
    if (lane == 0) { range_s[wave][0] = wlo; range_s[wave][1] = whi; }
    __syncthreads();
    const int blo = min(min(range_s[0][0], range_s[1][0]), min(range_s[2][0], range_s[3][0]));
    const int bhi = max(max(range_s[0][1], range_s[1][1]), max(range_s[2][1], range_s[3][1]));
    if (bhi < 0) return;                                                  // no live query in this block (uniform)
    wlo = __builtin_amdgcn_readfirstlane(wlo); whi = __builtin_amdgcn_readfirstlane(whi);
    const size_t head_off = ((size_t)b * NHp + h) * a.Lmax * CTTS_HEAD_DIM;
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
    dim3 grid(a.R * a.NH, a.jt > 0 ? a.jt : a.S), block(256);
    const int* done_p = a.st ? &a.st->all_done : nullptr;
    // unsplit rows: 8 waves per (row, head) keep twice the K/V bytes in flight per CU while there are fewer blocks than CUs; from one
    // block per CU on, 4-wave blocks balance better (us/step at mean context 312: batch 8 471 vs 479, 16 512 vs 521 | 32 613 vs 598, 64 800 vs 774)
    static const int wide_env = getenv("CTTS_ATTN_WIDE") ? atoi(getenv("CTTS_ATTN_WIDE")) : -1;     // diagnostic: force 8-wave (1) / 4-wave (0) blocks
    const bool wide = (a.jt == 0) && (a.S == 1) && (a.st != nullptr) && (wide_env < 0 ? (a.R * a.NH < 256) : wide_env != 0);
    static const int un_env = getenv("CTTS_ATTN_UN") ? atoi(getenv("CTTS_ATTN_UN")) : 4;        // experiment: 8 keys per lane group in flight for unsplit 4-wave blocks
    static const int tiled_env = getenv("CTTS_PREFILL_ATTN") ? atoi(getenv("CTTS_PREFILL_ATTN")) : 1;    // 1 = MFMA flash kernel (fp16, >= 64 rows), 2 = 8-queries-per-wave VALU kernel (diagnostic: no faster than row by row), 0 = row by row
    if (a.st == nullptr && a.jt == 0 && a.S == 1 && a.packed_out != nullptr && a.R >= 64 && dtype == 1 && a.T > 0 && tiled_env >= 1 && tiled_env != 2) {
        // prompt pass, fp16: MFMA flash attention, block = (64 queries, head, sequence)
        const int B = (a.row0 + a.R + a.T - 1) / a.T;                     // sequences 0 .. B-1 may have rows in this pass
        dim3 g3((a.T + 63) / 64, a.NH, B);
        hipLaunchKernelGGL(attn_prefill_mfma_kernel, g3, dim3(256), 0, s, a.meta, a.q, (const half_t*)a.k_cache, (const half_t*)a.v_cache, a.NH, a.R, a);
    } else if (a.st == nullptr && a.jt == 0 && a.S == 1 && a.packed_out != nullptr && a.R > 16 && tiled_env == 2) {
        // prompt pass: 8 query rows per wavefront, 4 wavefronts per block
        dim3 g2((a.R + 31) / 32, a.NH);
        if (dtype == 1) hipLaunchKernelGGL(attn_prefill_kernel<half_t>, g2, dim3(256), 0, s, a.meta, a.q, a.k_cache, a.v_cache, a.NH, a.R, a);
        else hipLaunchKernelGGL(attn_prefill_kernel<float>, g2, dim3(256), 0, s, a.meta, a.q, a.k_cache, a.v_cache, a.NH, a.R, a);
    } else if (wide) {
        if (dtype == 1) hipLaunchKernelGGL((attn_decode_kernel<half_t, false, 8>), grid, dim3(512), 0, s, done_p, a.meta, a.q, a.k_cache, a.v_cache, a.NH, a.S, a);
        else hipLaunchKernelGGL((attn_decode_kernel<float, false, 8>), grid, dim3(512), 0, s, done_p, a.meta, a.q, a.k_cache, a.v_cache, a.NH, a.S, a);
    } else if (a.jt > 0) {
        if (a.jt * 4 * FUSE_TPW * 16 != a.NH * CTTS_HEAD_DIM) { ctts_set_error("fused attention: jt=%d does not tile H=%d", a.jt, a.NH * CTTS_HEAD_DIM); return 1; }
        if (dtype == 1) hipLaunchKernelGGL((attn_decode_kernel<half_t, true, 4>), grid, block, 0, s, done_p, a.meta, a.q, a.k_cache, a.v_cache, a.NH, a.S, a);
        else hipLaunchKernelGGL((attn_decode_kernel<float, true, 4>), grid, block, 0, s, done_p, a.meta, a.q, a.k_cache, a.v_cache, a.NH, a.S, a);
    } else if (dtype == 1 && a.S == 1 && a.st != nullptr && un_env == 8) {
        hipLaunchKernelGGL((attn_decode_kernel<half_t, false, 4, 8>), grid, block, 0, s, done_p, a.meta, a.q, a.k_cache, a.v_cache, a.NH, a.S, a);
    } else if (dtype == 1) hipLaunchKernelGGL((attn_decode_kernel<half_t, false, 4>), grid, block, 0, s, done_p, a.meta, a.q, a.k_cache, a.v_cache, a.NH, a.S, a);
    else hipLaunchKernelGGL((attn_decode_kernel<float, false, 4>), grid, block, 0, s, done_p, a.meta, a.q, a.k_cache, a.v_cache, a.NH, a.S, a);
    CTTS_HIP_CHECK(hipGetLastError());
    return 0;
}
