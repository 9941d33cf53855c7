// Prompt-pass projections as an LDS-staged MFMA GEMM (fp16 weights / activations, fp32 accumulate; gfx950).
//
//   C[rows][N] = X[rows][K] . W[N][K]^T   for hundreds .. thousands of prompt rows (B * T), fused epilogues as in skinny_gemm.hip
//
// The decode-time kernel (skinny_gemm.hip) streams a weight tile per block and multiplies it with <= 32 rows: over a whole prompt
// that re-reads every weight tile once per 32-row chunk from L2 (885 MB per gate|up launch at 2048 rows; the prompt pass of
// 32 x 512 tokens spent 26 of its 48 ms there at ~10 % of the MFMA peak).  Here a block owns a 256 x 128 (or 128 x 128) output tile:
//   * both operands already live in HBM as MFMA fragment images -- weights [n tile][k tile][lane][16 B] (gpt_engine.hip pack), activations
//     [16-row group][k tile][lane][16 B] (norm_pack_kernel / the SwiGLU epilogue / the attention kernels) -- so staging a k-tile
//     is a straight copy of 1-KiB fragments into LDS (one coalesced 16-byte load + one ds_write_b128 per thread) and a wave reads
//     its operands back with conflict-free ds_read_b128;
//   * 8 (4) waves, each a 64 x 64 sub-tile = 4 x 4 v_mfma_f32_16x16x32_f16 accumulators, k-tiles double-buffered in LDS (24 KB per
//     stage), the next stage's global loads in flight under the current stage's 16 MFMAs per wave;
//   * every output element is accumulated by one wave in k order: deterministic, no split-K, no atomics.
// Epilogues restate the same reference lines as the decode kernel: q/k/v projection + RoPE + KV append (llama.py:619-633,151-182),
// o_proj / down_proj + residual (llama.py:666,731,739), SiLU(gate) * up (llama.py:214).
#include "kernels.h"

template <int WR>      // wave rows: block tile = (64 * WR) rows x 128 output features
struct PfCfg {
    static constexpr int WAVES = WR * 2, BM_G = 4 * WR, BN_T = 8;                 // row groups / n tiles per block
    static constexpr int FRAGS = BM_G + BN_T;                                      // 1-KiB fragments per k-tile stage
    static constexpr int PER_T = (FRAGS + WAVES - 1) / WAVES;                      // fragments copied per wave and stage
    static constexpr int STAGE = FRAGS * 1024;
};

template <int EPI, int WR>
__global__ __launch_bounds__(WR * 128) void prefill_gemm_kernel(const void* Wq, const void* Xp, const int ktiles, const int R, const GemmArgs a) {
    typedef PfCfg<WR> C;
    __shared__ __attribute__((aligned(16))) char lds[2 * C::STAGE];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 1, wn = wave & 1;
    const int nt0 = blockIdx.x * C::BN_T;                  // first n tile of the block
    const int g0 = blockIdx.y * C::BM_G;                   // first 16-row group of the block
    const half8* Wg = (const half8*)Wq;
    const half8* Xg = (const half8*)Xp;
    // fragment f of a stage: f < BN_T -> weight tile nt0 + f, else activation group g0 + (f - BN_T)
    auto src = [&](int f, int kt) -> const half8* {
        return (f < C::BN_T) ? Wg + ((size_t)(nt0 + f) * ktiles + kt) * 64 + lane : Xg + ((size_t)(g0 + f - C::BN_T) * ktiles + kt) * 64 + lane;
    };
    // Software pipeline: LDS double buffer + two register stages, so the global loads of stage kt + 2 are issued before the MFMAs of
    // stage kt and only have to land by the end of stage kt + 1 (with one register stage every iteration waited for the loads it had
    // just issued: 17 % of the MFMA peak).
    half8 st0[C::PER_T], st1[C::PER_T];
    f32x4 acc[4][4];                                       // [n tile][row group]
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int g = 0; g < 4; ++g) acc[t][g] = (f32x4){0.f, 0.f, 0.f, 0.f};
#define PF_LOAD(dst, kt_)                                                                                                            \
    _Pragma("unroll") for (int i = 0; i < C::PER_T; ++i) { const int f = wave + i * C::WAVES; if (f < C::FRAGS) dst[i] = *src(f, (kt_)); }
#define PF_STORE(srcr, buf_)                                                                                                         \
    _Pragma("unroll") for (int i = 0; i < C::PER_T; ++i) { const int f = wave + i * C::WAVES; if (f < C::FRAGS) *(half8*)(lds + (buf_) * C::STAGE + f * 1024 + lane * 16) = srcr[i]; }
#define PF_STEP(kt_, rnext, rfar)                                                                                                    \
    {                                                                                                                                \
        const int kt = (kt_);                                                                                                        \
        if (kt + 2 < ktiles) { PF_LOAD(rfar, kt + 2) }                                                                               \
        const char* cur = lds + (kt & 1) * C::STAGE;                                                                                 \
        half8 af[4], bf[4];                                                                                                          \
        _Pragma("unroll") for (int t = 0; t < 4; ++t) af[t] = *(const half8*)(cur + (wn * 4 + t) * 1024 + lane * 16);               \
        _Pragma("unroll") for (int g = 0; g < 4; ++g) bf[g] = *(const half8*)(cur + (C::BN_T + wr * 4 + g) * 1024 + lane * 16);     \
        _Pragma("unroll") for (int t = 0; t < 4; ++t)                                                                                \
        _Pragma("unroll") for (int g = 0; g < 4; ++g) acc[t][g] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[t], bf[g], acc[t][g], 0, 0, 0); \
        if (kt + 1 < ktiles) { PF_STORE(rnext, (kt + 1) & 1) }                                                                       \
        __syncthreads();                                                                                                             \
    }
    PF_LOAD(st0, 0)
    PF_STORE(st0, 0)
    if (ktiles > 1) { PF_LOAD(st1, 1) }
    __syncthreads();
    // stage s travels through register set s & 1: at stage kt the "next" set holds stage kt + 1, the "far" set receives stage kt + 2
    for (int kt2 = 0; kt2 < ktiles; kt2 += 2) {
        PF_STEP(kt2, st1, st0)
        if (kt2 + 1 < ktiles) PF_STEP(kt2 + 1, st0, st1)
    }
#undef PF_STEP
#undef PF_STORE
#undef PF_LOAD
    // ---- epilogue.  C tile layout: lane = (iq = lane >> 4, n = lane & 15): activation row n of the group, weight rows 4 * iq + j (j = register)
    const int iq = lane >> 4, nn = lane & 15;
    constexpr int H = 768, NH = H / CTTS_HEAD_DIM, HT = H / 16;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const int r = (g0 + wr * 4 + g) * 16 + nn;                               // row of the pass
        const bool rv = r < R;
        RowMeta m = {0, 0, 0, 0};
        if (EPI == EPI_QKV && rv) m = a.meta[r];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int rt = nt0 + wn * 4 + t;
            const f32x4 c = acc[t][g];
            if (EPI == EPI_RESID) {
                if (rv) {
                    float* xo = a.x_out + (size_t)r * (a.n_row_tiles * 16) + rt * 16 + 4 * iq;
                    const f32x4 x0 = *(const f32x4*)xo;
                    *(f32x4*)xo = (f32x4){x0[0] + c[0], x0[1] + c[1], x0[2] + c[2], x0[3] + c[3]};   // residual + proj (llama.py:731,739)
                }
            } else {
                // packed tile rows: [8 "a" rows | 8 "b" rows] (q/k/v: dims d and d + 32 of one head; gate|up: gate row and up row): the
                // partner half sits 32 lanes away.  A lane ends up with 4 consecutive outputs (j = 0..3): one vector store each.
                f32x4 o;
#pragma unroll
                for (int j = 0; j < 4; ++j) o[j] = __shfl_xor(c[j], 32);
                const bool lowh = iq < 2;
                const f32x4 va = lowh ? c : o, vb = lowh ? o : c;
                const int p0 = 4 * (iq & 1);                                    // outputs p0 .. p0 + 3 of the tile's 8 pairs
                if (EPI == EPI_SWIGLU) {
                    if (lowh) {                                                // both lanes of a pair hold it: the low half writes
                        half4 y = {(half_t)0.f, (half_t)0.f, (half_t)0.f, (half_t)0.f};
                        if (rv) {
#pragma unroll
                            for (int j = 0; j < 4; ++j) y[j] = (half_t)((va[j] / (1.0f + expf(-va[j]))) * vb[j]);
                        }
                        const int ktiles_out = (a.n_row_tiles * 8) / 32;
                        // n >> 4 = global 16-row group: the packed image is contiguous over chunks; 4 consecutive k share a fragment row
                        *(half4*)((half_t*)a.act_out + xfrag_index<half_t>(r, rt * 8 + p0, ktiles_out)) = y;
                    }
                } else if (rv) {                                             // EPI_QKV
                    const int which = rt / HT, within = rt % HT;
                    const int hh = within >> 2, d0 = ((within & 3) << 3) + p0;
                    f32x4 y;
                    if (which < 2) {
                        const f32x4 cs = *(const f32x4*)(a.rope_rows + (size_t)r * 64 + d0), sn = *(const f32x4*)(a.rope_rows + (size_t)r * 64 + 32 + d0);
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            // q*cos + rotate_half(q)*sin (llama.py:180-181), products rounded separately like the reference
                            const float ya = __fadd_rn(__fmul_rn(va[j], cs[j]), __fmul_rn(-vb[j], sn[j]));
                            const float yb = __fadd_rn(__fmul_rn(vb[j], cs[j]), __fmul_rn(va[j], sn[j]));
                            y[j] = lowh ? ya : yb;
                        }
                    } else y = lowh ? va : vb;
                    const int dd = lowh ? d0 : d0 + 32;
                    if (which == 0) *(f32x4*)(a.q_out + ((size_t)r * NH + hh) * CTTS_HEAD_DIM + dd) = y;
                    else {
                        half_t* cch = (half_t*)(which == 1 ? a.k_cache : a.v_cache) + (((size_t)m.seq * NH + hh) * a.Lmax + m.slot) * CTTS_HEAD_DIM;
                        *(half4*)(cch + dd) = (half4){(half_t)y[0], (half_t)y[1], (half_t)y[2], (half_t)y[3]};
                    }
                }
            }
        }
    }
}

template <int EPI, int WR>
static int pf_launch(const GemmArgs& a, const void* X, int ktiles, hipStream_t s) {
    const int BM = 64 * WR;
    if ((a.n_row_tiles % 8) != 0) { ctts_set_error("prefill_gemm: %d n tiles not a multiple of 8", a.n_row_tiles); return 1; }
    dim3 grid(a.n_row_tiles / 8, (a.R + BM - 1) / BM);
    hipLaunchKernelGGL((prefill_gemm_kernel<EPI, WR>), grid, dim3(WR * 128), 0, s, a.W, X, ktiles, a.R, a);
    CTTS_HIP_CHECK(hipGetLastError());
    return 0;
}

// fp16 only.  X = packed activations (norm_packed / attn_packed / act); the operand buffers must cover whole blocks of rows
// (gpt_engine.hip allocates them for PASS_ROWS + 256 rows).
int launch_prefill_gemm(int epi, const GemmArgs& a, hipStream_t s) {
    const void* X = a.xpacked;
    const int ktiles = a.K / 32;
    // 256-row blocks only where the grid still covers the chip a few times over: the N = 768 projections (6 column blocks) take 128-row blocks
    const bool big = (a.R >= 2048) && (a.n_row_tiles >= 96);
    if (epi == EPI_QKV) return big ? pf_launch<EPI_QKV, 4>(a, X, ktiles, s) : pf_launch<EPI_QKV, 2>(a, X, ktiles, s);
    if (epi == EPI_SWIGLU) return big ? pf_launch<EPI_SWIGLU, 4>(a, X, ktiles, s) : pf_launch<EPI_SWIGLU, 2>(a, X, ktiles, s);
    if (epi == EPI_RESID) return big ? pf_launch<EPI_RESID, 4>(a, X, ktiles, s) : pf_launch<EPI_RESID, 2>(a, X, ktiles, s);
    ctts_set_error("prefill_gemm: unsupported epilogue %d", epi);
    return 1;
}
