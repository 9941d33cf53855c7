// Prompt-pass projections as an LDS-staged MFMA GEMM (fp16 weights / activations, fp32 accumulate; gfx950).
//
//   C[rows][N] = X[rows][K] . W[N][K]^T   for hundreds .. thousands of prompt rows (B * T), fused epilogues as in skinny_gemm.hip
//
// The decode-time kernel (skinny_gemm.hip) streams a weight tile per block and multiplies it with <= 32 rows: over a whole prompt
// that re-reads every weight tile once per 32-row chunk from L2 (885 MB per gate|up launch at 2048 rows; the prompt pass of
// 32 x 512 tokens spent 26 of its 48 ms there at ~10 % of the MFMA peak).  Here a block owns a 128 x 128 (or 256 x 128) output tile:
//   * both operands already live in HBM as MFMA fragment images -- weights [n tile][k tile][lane][16 B] (gpt_engine.hip pack), activations
//     [16-row group][k tile][lane][16 B] (norm_pack_kernel / the SwiGLU epilogue / the attention kernels) -- so staging a k-tile
//     is a straight copy of 1-KiB fragments into LDS (one global_load_lds_dwordx4 per wave and fragment) and a wave reads
//     its operands back with conflict-free ds_read_b128;
//   * 4 (8) waves, each a 64 x 64 sub-tile = 4 x 4 v_mfma_f32_16x16x32_f16 accumulators, k-tiles in a ring of 3 LDS stages (24 KB each)
//     filled by LDS-DMA loads two stages ahead of the MFMAs;
//   * every output element is accumulated by one wave in k order: deterministic, no split-K, no atomics.
// Epilogues restate the same reference lines as the decode kernel: q/k/v projection + RoPE + KV append (llama.py:619-633,151-182),
// o_proj / down_proj + residual (llama.py:666,731,739), SiLU(gate) * up (llama.py:214).
#include <stdlib.h>

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
    __shared__ __attribute__((aligned(16))) char lds[3 * C::STAGE];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 1, wn = wave & 1;
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
    const half8* Wg = (const half8*)Wq;
    const half8* Xg = (const half8*)Xp;
    // fragment f of a stage: f < BN_T -> weight tile nt0 + f, else activation group g0 + (f - BN_T)
    auto src = [&](int f, int kt) -> const half8* {
        return (f < C::BN_T) ? Wg + ((size_t)(nt0 + f) * ktiles + kt) * 64 + lane : Xg + ((size_t)(g0 + f - C::BN_T) * ktiles + kt) * 64 + lane;
    };
    // Software pipeline: a ring of 3 LDS stages filled by LDS-DMA loads (global_load_lds_dwordx4: 1 KiB per wave-instruction straight
    // into LDS at base + lane * 16 -- exactly the fragment image -- no staging registers, no ds_write).  Staging through registers
    // cost as many LDS-pipe cycles in ds_write_b128 (13 per wave-instruction) as the MFMAs took: 24 % of the MFMA peak.  The loads
    // of stage kt + 2 are issued at the top of stage kt; stage kt + 1 must have landed by the barrier that ends stage kt.
    f32x4 acc[4][4];                                       // [n tile][row group]
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int g = 0; g < 4; ++g) acc[t][g] = (f32x4){0.f, 0.f, 0.f, 0.f};
    typedef __attribute__((address_space(1))) const void* gptr_t;
    typedef __attribute__((address_space(3))) void* lptr_t;
#define PF_DMA(kt_, buf_)                                                                                                            \
    _Pragma("unroll") for (int i = 0; i < C::PER_T; ++i) {                                                                           \
        const int f = wave + i * C::WAVES;                                                                                           \
        if (f < C::FRAGS) __builtin_amdgcn_global_load_lds((gptr_t)src(f, (kt_)), (lptr_t)(lds + (buf_) * C::STAGE + f * 1024), 16, 0, 0); \
    }
    PF_DMA(0, 0)
    if (ktiles > 1) { PF_DMA(1, 1) }
    if (ktiles > 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(C::PER_T) : "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    int cb = 0;                                            // ring slot of the current stage
    for (int kt = 0; kt < ktiles; ++kt) {
        const int nb2 = (cb + 2 >= 3) ? cb - 1 : cb + 2;
        if (kt + 2 < ktiles) { PF_DMA(kt + 2, nb2) }
        const char* cur = lds + cb * C::STAGE;
        half8 af[4], bf[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) af[t] = *(const half8*)(cur + (wn * 4 + t) * 1024 + lane * 16);
#pragma unroll
        for (int g = 0; g < 4; ++g) bf[g] = *(const half8*)(cur + (C::BN_T + wr * 4 + g) * 1024 + lane * 16);
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int g = 0; g < 4; ++g) acc[t][g] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[t], bf[g], acc[t][g], 0, 0, 0);
        // stage kt + 1 (issued one iteration ago) must be in LDS before anyone reads it: all but the loads just issued have to be back
        // (a bare s_barrier: __syncthreads() carries a workgroup fence that waits for vmcnt(0), i.e. for the stage just requested too;
        //  the asm memory clobbers keep the compiler from moving LDS accesses across it)
        if (kt + 2 < ktiles) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(C::PER_T) : "memory");
        else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
        cb = (cb + 1 == 3) ? 0 : cb + 1;
    }
#undef PF_DMA
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
                    // both lanes of a pair hold (gate, up): the low half writes outputs p0, p0 + 1, the high half p0 + 2, p0 + 3.  SiLU with
                    // the hardware exp / reciprocal (relative error ~1e-6, far below the fp16 rounding of the result): the precise expf +
                    // IEEE division of the decode epilogue cost as many vector instructions here as the whole 384-MFMA main loop
                    const int jo = lowh ? 0 : 2;
                    const float g0 = lowh ? va[0] : va[2], g1 = lowh ? va[1] : va[3], u0 = lowh ? vb[0] : vb[2], u1 = lowh ? vb[1] : vb[3];
                    typedef _Float16 half2_t __attribute__((ext_vector_type(2)));
                    half2_t y = {(half_t)0.f, (half_t)0.f};
                    if (rv) {
                        y[0] = (half_t)(g0 * __frcp_rn(1.0f + __expf(-g0)) * u0);
                        y[1] = (half_t)(g1 * __frcp_rn(1.0f + __expf(-g1)) * u1);
                    }
                    const int ktiles_out = (a.n_row_tiles * 8) / 32;
                    // n >> 4 = global 16-row group: the packed image is contiguous over chunks; consecutive k share a fragment row
                    *(half2_t*)((half_t*)a.act_out + xfrag_index<half_t>(r, rt * 8 + p0 + jo, ktiles_out)) = y;
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
    static const int wr_env = getenv("CTTS_PF_WR") ? atoi(getenv("CTTS_PF_WR")) : 0;        // diagnostic: 2 / 4 forces 128- / 256-row blocks
    // 128 x 128 blocks (4 waves, 3 blocks per CU) by default: equal to 256 x 128 at 16384 rows (10.26 ms per 32 x 512 prompt pass either way), faster
    // at 2048 rows (2.81 vs 3.21 ms)
    const bool big = wr_env == 4;
    if (epi == EPI_QKV) return big ? pf_launch<EPI_QKV, 4>(a, X, ktiles, s) : pf_launch<EPI_QKV, 2>(a, X, ktiles, s);
    if (epi == EPI_SWIGLU) return big ? pf_launch<EPI_SWIGLU, 4>(a, X, ktiles, s) : pf_launch<EPI_SWIGLU, 2>(a, X, ktiles, s);
    if (epi == EPI_RESID) return big ? pf_launch<EPI_RESID, 4>(a, X, ktiles, s) : pf_launch<EPI_RESID, 2>(a, X, ktiles, s);
    ctts_set_error("prefill_gemm: unsupported epilogue %d", epi);
    return 1;
}
