// Prompt-pass projections as an LDS-staged MFMA GEMM (fp16 weights / activations, fp32 accumulate; gfx950).
//
//   C[rows][N] = X[rows][K] . W[N][K]^T   for hundreds .. thousands of prompt rows (B * T), fused epilogues as in skinny_gemm.hip
//
// The decode-time kernel (skinny_gemm.hip) streams a weight tile per block and multiplies it with <= 32 rows: over a whole prompt
// that re-reads every weight tile once per 32-row chunk from L2 (885 MB per gate|up launch at 2048 rows; the prompt pass of
// 32 x 512 tokens spent 26 of its 48 ms there at ~10 % of the MFMA peak).  Here a block owns a 128 x 128 (gate|up on long passes: 256 x 128) output tile:
//   * both operands already live in HBM as MFMA fragment images -- weights [n tile][k tile][lane][16 B] (gpt_engine.hip pack), activations
//     [16-row group][k tile][lane][16 B] (norm_pack_kernel / the SwiGLU epilogue / the attention kernels) -- so staging a k-tile
//     is a straight copy of 1-KiB fragments into LDS (one global_load_lds_dwordx4 per wave and fragment) and a wave reads
//     its operands back with conflict-free ds_read_b128;
//   * 4 waves, each a 64 x 64 (128 x 64) sub-tile = 4 x 4 (4 x 8) v_mfma_f32_16x16x32_f16 accumulators; k-tiles travel HBM / L2 -> a ring of
//     3 LDS stages (LDS-DMA, requested three stages ahead) -> one of two register sets (read one stage ahead of its MFMAs);
//   * every output element is accumulated by one wave in k order: deterministic, no split-K, no atomics.
// Epilogues restate the same reference lines as the decode kernel: q/k/v projection + RoPE + KV append (llama.py:619-633,151-182),
// o_proj / down_proj + residual (llama.py:666,731,739), SiLU(gate) * up (llama.py:214).
#include <stdio.h>
#include <stdlib.h>

#include "kernels.h"

// Block shapes.  A wave owns 4 weight tiles (64 output features) x GR 16-row groups; a block is WR x WN waves.
//   <2, 2, 4>  128 rows x 128 features, 4 waves, 3 blocks per CU   (16 KB per k-tile stage)
//   <2, 2, 8>  256 rows x 128 features, 4 waves, 2 blocks per CU   (24 KB)
//   <2, 4, 8>  256 rows x 256 features, 8 waves, 1 block per CU    (32 KB)
// The shape hardly matters (32 x 512 tokens: 9.99 / 9.70 / 10.10 ms per prompt pass with everything on the first / gate|up on the second / everything
// on the third): operand traffic into the CU (64 / 85 / 128 flop per byte) is not what bounds these kernels.  With the epilogues compiled out
// the main loops alone take 138 (gate|up), 57 (QKV), 44 (o / down, mean) us per launch at 16384 rows = 1.0-1.1 PFLOP/s, the epilogues another
// 39 / 45 / 19 us -- about twice their HBM floor (100 MB written per launch).  Persistent workgroups started a third of a tile apart (so that
// a CU's co-resident blocks are out of step) measure the same: 9.58 vs 9.56 ms.
template <int WR, int WN, int GR>
struct PfCfg {
    static constexpr int WAVES = WR * WN, BM_G = WR * GR, BN_T = WN * 4;           // waves, row groups and n tiles per block
    static constexpr int FRAGS = BM_G + BN_T;                                      // 1-KiB fragments per k-tile stage
    static constexpr int PER_T = (FRAGS + WAVES - 1) / WAVES;                      // fragments copied per wave and stage
    static constexpr int STAGE = FRAGS * 1024;
    static constexpr int WAVES_EU = GR == 4 ? 3 : 2;                               // register budget: 168 / 256 per lane
};

#ifndef PF_RING
#define PF_RING 3                                          // LDS stages per block (diagnostic builds: 4)
#endif
#ifndef PF_LDS_PAD
#define PF_LDS_PAD 0                                       // diagnostic builds: extra LDS to lower the blocks per CU
#endif
template <int EPI, int WR, int WN, int GR>
__global__ __launch_bounds__(WR * WN * 64) __attribute__((amdgpu_waves_per_eu(PfCfg<WR, WN, GR>::WAVES_EU, 8))) void prefill_gemm_kernel(
    const void* Wq, const void* Xp, const int ktiles, const int R, const GemmArgs a) {
    typedef PfCfg<WR, WN, GR> C;
    __shared__ __attribute__((aligned(16))) char lds[PF_RING * C::STAGE + PF_LDS_PAD];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);      // uniform: fragment addresses below stay in scalar registers
    const int wr = wave / WN, wn = wave % WN;
    // XCD-aware tile order.  Workgroups are dealt round-robin to the 8 XCDs (linear id % 8), each with its own 4-MB L2; in plain
    // (x, y) order every XCD touches every weight tile AND every activation row block of the launch (22 MB for gate|up at 8192 rows):
    // the L2s thrash and the GEMM runs at the Infinity-Cache rate.  Remapped, XCD x owns the row blocks y = x (mod 8) and walks the
    // column blocks in order, so the ~64 workgroups resident on an XCD share 4 activation row blocks and ~16 weight column blocks.
    int bx = blockIdx.x, by = blockIdx.y;
    if ((gridDim.y & 7) == 0) {
        const int lin = blockIdx.x + gridDim.x * blockIdx.y, xcd = lin & 7, seq = lin >> 3, rpx = gridDim.y >> 3;
        bx = seq / rpx;
        by = (seq % rpx) * 8 + xcd;
    }
    const int nt0 = bx * C::BN_T;                          // first n tile of the block
    const int g0 = by * C::BM_G;                           // first 16-row group of the block
    // fragment f of a stage: f < BN_T -> weight tile nt0 + f, else activation group g0 + (f - BN_T); a uniform 64-bit base + lane * 16
    auto src = [&](int f, int kt) -> const char* {
        const char* base = (f < C::BN_T) ? (const char*)Wq + ((size_t)(nt0 + f) * ktiles + kt) * 1024
                                         : (const char*)Xp + ((size_t)(g0 + f - C::BN_T) * ktiles + kt) * 1024;
        return base + (unsigned)(lane * 16);
    };
    // Software pipeline: a ring of 3 LDS stages filled by LDS-DMA loads (global_load_lds_dwordx4: 1 KiB per wave-instruction straight
    // into LDS at base + lane * 16 -- exactly the fragment image -- no staging registers, no ds_write).  Staging through registers
    // cost as many LDS-pipe cycles in ds_write_b128 (13 per wave-instruction) as the MFMAs took: 24 % of the MFMA peak.  A wave keeps
    // two register sets of operand fragments: stage kt is multiplied from one while stage kt + 1 is read from LDS into the other and
    // stage kt + 3 is requested from HBM / L2 into the slot stage kt has just left.
    f32x4 acc[4][GR];                                      // [n tile][row group]
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int g = 0; g < GR; ++g) acc[t][g] = (f32x4){0.f, 0.f, 0.f, 0.f};
    typedef __attribute__((address_space(1))) const void* gptr_t;
    typedef __attribute__((address_space(3))) void* lptr_t;
#define PF_DMA(kt_, buf_)                                                                                                            \
    _Pragma("unroll") for (int i = 0; i < C::PER_T; ++i) {                                                                           \
        const int f = wave + i * C::WAVES;                                                                                           \
        if (C::FRAGS % C::WAVES == 0 || f < C::FRAGS)                                                                                \
            __builtin_amdgcn_global_load_lds((gptr_t)src(f, (kt_)), (lptr_t)(lds + (buf_) * C::STAGE + f * 1024), 16, 0, 0);         \
    }
#define PF_READ(af_, bf_, buf_)                                                                                                      \
    {                                                                                                                                \
        const char* cur = lds + (buf_) * C::STAGE;                                                                                   \
        _Pragma("unroll") for (int t = 0; t < 4; ++t) af_[t] = *(const half8*)(cur + (wn * 4 + t) * 1024 + lane * 16);               \
        _Pragma("unroll") for (int g = 0; g < GR; ++g) bf_[g] = *(const half8*)(cur + (C::BN_T + wr * GR + g) * 1024 + lane * 16);  \
    }
#define PF_MFMA(af_, bf_)                                                                                                            \
    _Pragma("unroll") for (int t = 0; t < 4; ++t) _Pragma("unroll") for (int g = 0; g < GR; ++g)                                     \
        acc[t][g] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af_[t], bf_[g], acc[t][g], 0, 0, 0);
    // One step of the pipeline (k-tile kt_ in registers rd_*, ring slot cb_): wait until stage kt + 1 has landed and every wave holds stage
    // kt in registers (s_waitcnt + a bare s_barrier: __syncthreads() carries a workgroup fence that waits for vmcnt(0), i.e. for the
    // stages just requested too), then start the LDS reads of stage kt + 1 into the other register set, refill the slot stage kt has
    // just left with stage kt + 3, and only then issue the 16 MFMAs of stage kt: the wave's own LDS latency and the DMA issue sit
    // under its MFMAs instead of in front of them.  The waits are the builtin (not inline asm) so that the compiler's own counter
    // tracking knows the register set being multiplied is complete and does not wait for the reads just issued.
    // s_waitcnt immediate (gfx9): vmcnt[3:0] and [15:14] | expcnt[6:4] = 7 (none) | lgkmcnt[11:8] = 0
#define PF_WAIT_BAR(n_) { __builtin_amdgcn_s_waitcnt(0x0070 | ((n_) & 15) | (((n_) >> 4) << 14)); __builtin_amdgcn_s_barrier(); }
#define PF_STEP(kt_, cb_, rd_a, rd_b, nx_a, nx_b, vm_, read_, dma_)                                                                  \
    {                                                                                                                                \
        PF_WAIT_BAR(vm_)                                                                                                             \
        const int n1 = ((cb_) + 1 == PF_RING) ? 0 : (cb_) + 1;                                                                             \
        if (read_) PF_READ(nx_a, nx_b, n1)                                                                                           \
        if (dma_) { PF_DMA((kt_) + PF_RING, (cb_)) }                                                                                    \
        PF_MFMA(rd_a, rd_b)                                                                                                          \
        cb_ = n1;                                                                                                                    \
    }
#pragma unroll
    for (int st = 0; st < PF_RING; ++st) { PF_DMA(st, st) }        // ktiles >= PF_RING (checked by the launcher)
    PF_WAIT_BAR((PF_RING - 1) * C::PER_T)
    half8 af0[4], bf0[GR], af1[4], bf1[GR];
    PF_READ(af0, bf0, 0)
    int cb = 0;                                            // ring slot of the stage held in registers
    int kt = 0;
    for (; kt + PF_RING < ktiles; kt += 2) {               // ktiles is even (K = 768 / 3072): PF_RING - 2 later stages stay in flight behind each step
        PF_STEP(kt, cb, af0, bf0, af1, bf1, (PF_RING - 2) * C::PER_T, true, true)
        PF_STEP(kt + 1, cb, af1, bf1, af0, bf0, (PF_RING - 2) * C::PER_T, true, kt + 1 + PF_RING < ktiles)
    }
    for (; kt < ktiles; kt += 2) {                         // the last stages: nothing left to request
        PF_STEP(kt, cb, af0, bf0, af1, bf1, 0, true, false)
        PF_STEP(kt + 1, cb, af1, bf1, af0, bf0, 0, kt + 2 < ktiles, false)
    }
#undef PF_STEP
#undef PF_WAIT_BAR
#undef PF_MFMA
#undef PF_READ
#undef PF_DMA
    // ---- epilogue.  C tile layout: lane = (iq = lane >> 4, n = lane & 15): activation row n of the group, weight rows 4 * iq + j (j = register).
    // In that layout a store instruction touches 16 rows x 32..64 B.  Each wave therefore turns its 16-row x 64-feature slab around through a
    // private 4.5-KB piece of the (now idle) LDS ring and stores whole rows: 256 B (fp32) / 128 B (fp16 K, V) per 16 / 8 lanes, or one
    // contiguous 1-KB activation fragment per instruction (about 3 % of the prompt pass; the rest of the epilogue time is the HBM write
    // burst itself).  Same-wave LDS accesses execute in order: no barrier.
    const int iq = lane >> 4, nn = lane & 15;
    constexpr int H = 768, NH = H / CTTS_HEAD_DIM, HT = H / 16;
    constexpr int P32 = 272, P16 = 144;                    // slab row pitch in bytes: 64 floats / 64 halfs + 16 B (bank spread, 16-B aligned)
    char* scr = lds + wave * (16 * 288);
    const int rt0 = nt0 + wn * 4;                          // first of the wave's 4 n tiles: one head of q / k / v, or 32 outputs of gate|up
    const bool lowh = iq < 2;
    const int p0 = 4 * (iq & 1);                           // packed tile rows [8 "a" | 8 "b"]: this lane pair holds outputs p0 .. p0 + 3 of the tile's 8
    const int which = rt0 / HT, hh = (rt0 % HT) >> 2;      // EPI_QKV: projection and head of the slab
    f32x4 x0[2][4];                                        // EPI_RESID: residual rows of the current / next group
    float* xo[2][4];
#pragma unroll
    for (int g = 0; g < GR; ++g) {
        const int rb = (g0 + wr * GR + g) * 16;            // first row of the group
        const int r = rb + nn;                             // row of the pass held by this lane
        const bool rv = r < R;
        if (EPI == EPI_RESID) {
            // the residual rows of a group are requested one group ahead (rows past the end clamped, so that the loads carry no branch)
            if (g == 0) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int ro = rb + i * 4 + iq;
                    xo[0][i] = a.x_out + (size_t)(ro < R ? ro : R - 1) * (a.n_row_tiles * 16) + rt0 * 16 + nn * 4;
                    x0[0][i] = *(const f32x4*)xo[0][i];
                }
            }
            if (g + 1 < GR) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int ro = rb + 16 + i * 4 + iq;
                    xo[(g + 1) & 1][i] = a.x_out + (size_t)(ro < R ? ro : R - 1) * (a.n_row_tiles * 16) + rt0 * 16 + nn * 4;
                    x0[(g + 1) & 1][i] = *(const f32x4*)xo[(g + 1) & 1][i];
                }
            }
#pragma unroll
            for (int t = 0; t < 4; ++t) *(f32x4*)(scr + nn * P32 + (t * 16 + 4 * iq) * 4) = acc[t][g];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int row = i * 4 + iq;
                const f32x4 c = *(const f32x4*)(scr + row * P32 + nn * 16);
                const f32x4 x = x0[g & 1][i];
                if (rb + row < R) *(f32x4*)xo[g & 1][i] = (f32x4){x[0] + c[0], x[1] + c[1], x[2] + c[2], x[3] + c[3]};   // residual + proj (llama.py:731,739)
            }
        } else if (EPI == EPI_SWIGLU) {
            // both lanes of a pair hold (gate, up) of outputs p0 .. p0 + 3: the low half finishes p0, p0 + 1, the high half p0 + 2, p0 + 3.  SiLU with
            // the hardware exp / reciprocal (relative error ~1e-6, far below the fp16 rounding of the result): the precise expf +
            // IEEE division of the decode epilogue cost as many vector instructions here as the whole 384-MFMA main loop
            typedef _Float16 half2_t __attribute__((ext_vector_type(2)));
            const int cp = (p0 >> 1) + (lowh ? 0 : 1);     // output pair within the tile's 8
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const f32x4 c = acc[t][g];
                f32x4 o;
#pragma unroll
                for (int j = 0; j < 4; ++j) o[j] = __shfl_xor(c[j], 32);
                const f32x4 va = lowh ? c : o, vb = lowh ? o : c;
                const float g0v = lowh ? va[0] : va[2], g1v = lowh ? va[1] : va[3], u0 = lowh ? vb[0] : vb[2], u1 = lowh ? vb[1] : vb[3];
                half2_t y = {(half_t)0.f, (half_t)0.f};
                if (rv) {
                    y[0] = sat_half(g0v * __frcp_rn(1.0f + __expf(-g0v)) * u0, a.sat);
                    y[1] = sat_half(g1v * __frcp_rn(1.0f + __expf(-g1v)) * u1, a.sat);
                }
                *(half2_t*)(scr + (nn + 16 * t) * 16 + cp * 4) = y;        // the fragment image: lane' = row + 16 * octet, 8 halfs each
            }
            const int ktiles_out = (a.n_row_tiles * 8) / 32;
            const half8 frag = *(const half8*)(scr + lane * 16);
            *(half8*)((half_t*)a.act_out + (((size_t)(rb >> 4) * ktiles_out + (rt0 >> 2)) * 64 + lane) * 8) = frag;
        } else {                                                             // EPI_QKV
            f32x4 cs[4], sn[4];
            const int rc = rv ? r : R - 1;                                   // rows past the end: clamped loads, nothing stored
            if (which < 2) {
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    cs[t] = *(const f32x4*)(a.rope_rows + (size_t)rc * 64 + t * 8 + p0);
                    sn[t] = *(const f32x4*)(a.rope_rows + (size_t)rc * 64 + 32 + t * 8 + p0);
                }
            }
            RowMeta mrow[2];
            if (which != 0) {
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const int ro = rb + i * 8 + (lane >> 3);
                    mrow[i] = a.meta[ro < R ? ro : R - 1];
                }
            }
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const f32x4 c = acc[t][g];
                f32x4 o;
#pragma unroll
                for (int j = 0; j < 4; ++j) o[j] = __shfl_xor(c[j], 32);
                const f32x4 va = lowh ? c : o, vb = lowh ? o : c;
                f32x4 y;
                if (which < 2) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        // q*cos + rotate_half(q)*sin (llama.py:180-181), products rounded separately like the reference
                        const float ya = __fadd_rn(__fmul_rn(va[j], cs[t][j]), __fmul_rn(-vb[j], sn[t][j]));
                        const float yb = __fadd_rn(__fmul_rn(vb[j], cs[t][j]), __fmul_rn(va[j], sn[t][j]));
                        y[j] = lowh ? ya : yb;
                    }
                } else y = lowh ? va : vb;
                const int dd = t * 8 + p0 + (lowh ? 0 : 32);                 // first of this lane's 4 head dims
                if (which == 0) *(f32x4*)(scr + nn * P32 + dd * 4) = y;
                else *(half4*)(scr + nn * P16 + dd * 2) = (half4){(half_t)y[0], (half_t)y[1], (half_t)y[2], (half_t)y[3]};
            }
            if (which == 0) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int row = i * 4 + iq, ro = rb + row;
                    const f32x4 y = *(const f32x4*)(scr + row * P32 + nn * 16);
                    if (ro < R) *(f32x4*)(a.q_out + ((size_t)ro * NH + hh) * CTTS_HEAD_DIM + nn * 4) = y;
                }
            } else {
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const int row = i * 8 + (lane >> 3), ro = rb + row;
                    const half8 y = *(const half8*)(scr + row * P16 + (lane & 7) * 16);
                    if (ro < R) {
                        const RowMeta m = mrow[i];
                        half_t* cch = (half_t*)(which == 1 ? a.k_cache : a.v_cache) + (((size_t)m.seq * NH + hh) * a.Lmax + m.slot) * CTTS_HEAD_DIM;
                        *(half8*)(cch + (lane & 7) * 8) = y;
                    }
                }
            }
        }
    }
}

template <int EPI, int WR, int WN, int GR>
static int pf_launch(const GemmArgs& a, const void* X, int ktiles, hipStream_t s) {
    typedef PfCfg<WR, WN, GR> C;
    if ((a.n_row_tiles % C::BN_T) != 0) { ctts_set_error("prefill_gemm: %d n tiles not a multiple of %d", a.n_row_tiles, C::BN_T); return 1; }
    if (ktiles < PF_RING + 1 || (ktiles & 1)) { ctts_set_error("prefill_gemm: K = %d is not a multiple of 64", ktiles * 32); return 1; }
    dim3 grid(a.n_row_tiles / C::BN_T, (a.R + C::BM_G * 16 - 1) / (C::BM_G * 16));
    hipLaunchKernelGGL((prefill_gemm_kernel<EPI, WR, WN, GR>), grid, dim3(C::WAVES * 64), 0, s, a.W, X, ktiles, a.R, a);
    CTTS_HIP_CHECK(hipGetLastError());
    return 0;
}

template <int EPI>
static int pf_shape(int shape, const GemmArgs& a, const void* X, int ktiles, hipStream_t s) {
    if (shape == 2) return pf_launch<EPI, 2, 4, 8>(a, X, ktiles, s);
    if (shape == 1) return pf_launch<EPI, 2, 2, 8>(a, X, ktiles, s);
    return pf_launch<EPI, 2, 2, 4>(a, X, ktiles, s);
}

// fp16 only.  X = packed activations (norm_packed / attn_packed / act); the operand buffers must cover whole blocks of rows
// (gpt_engine.hip allocates them for PASS_ROWS + 256 rows).
int launch_prefill_gemm(int epi, const GemmArgs& a, hipStream_t s) {
    const void* X = a.xpacked;
    const int ktiles = a.K / 32;
    // block shape per projection: 0 = 128 x 128, 1 = 256 x 128, 2 = 256 x 256.  CTTS_PF_SHAPE = "qkv,gateup,resid" overrides (diagnostic).
    static int shape_env[3] = {-1, -1, -1};
    static const bool parsed = [] {
        const char* e = diag_env("CTTS_PF_SHAPE");
        if (e) sscanf(e, "%d,%d,%d", &shape_env[0], &shape_env[1], &shape_env[2]);
        return true;
    }();
    (void)parsed;
    const int which = epi == EPI_QKV ? 0 : epi == EPI_SWIGLU ? 1 : 2;
    int shape = shape_env[which];
    // measured (32 x 512 / 8 x 2000 tokens per pass): all 128 x 128 9.99 / 13.50 ms; gate|up on 256 x 128 9.70 / 13.37; gate|up + QKV on 256 x 128
    // 9.79 / 13.64; everything on 256 x 256 10.10 / 14.08.  Below 8192 rows the 128 x 128 grid fills the chip better (4 x 512: 2.51 vs 2.57 ms).
    if (shape < 0) shape = (epi == EPI_SWIGLU && a.R >= 8192) ? 1 : 0;
    if (epi == EPI_QKV) return pf_shape<EPI_QKV>(shape, a, X, ktiles, s);
    if (epi == EPI_SWIGLU) return pf_shape<EPI_SWIGLU>(shape, a, X, ktiles, s);
    if (epi == EPI_RESID) return pf_shape<EPI_RESID>(shape, a, X, ktiles, s);
    ctts_set_error("prefill_gemm: unsupported epilogue %d", epi);
    return 1;
}
