// Prompt-pass projections of the PARITY (fp32) engine on the fp16 MFMA pipes: a 3-term split GEMM with fp32-level accuracy (gfx950).
//
//   C[rows][N] = X[rows][K] . W[N][K]^T,   X = Xhi + Xlo,  W = (Whi + Wlo) / 64,   every part an fp16 image,
//   C = (Xhi.Whi + Xhi.Wlo + Xlo.Whi) / 64      (the dropped Xlo.Wlo term is 2^-22 of a product: below fp32 rounding of the sum)
//
// Why: the fp32 engine's prompt pass ran the decode kernels over 32-row chunks (every weight tile re-read from L2 once per chunk) on
// v_mfma_f32_16x16x4_f32, whose peak is 1/16 of the fp16 pipe's: 110 ms for 32 x 512 prompt tokens against 9.6 ms in the fp16 engine.
// fp16 products are exact in the fp32 accumulator, so splitting both operands into a 11-bit head and a 11-bit tail gives ~22-bit operands
// for three MFMAs instead of sixteen.  The heads / tails are plain fp16 (no block exponents): weights are pre-scaled by 64 at pack time so
// that their tails stay normal numbers (|w| ~ 0.02 -> tail ~ 6e-4); activations are O(1) rows (RMSNorm output, attention output) whose
// tails lose nothing that matters (absolute error <= 2^-24); the SwiGLU output is stored divided by 16 (range +-1e6).
// Same fusions and reference lines as prefill_gemm.hip: q/k/v projection + RoPE + KV append (llama.py:619-633,151-182), o_proj / down_proj +
// residual (llama.py:666,731,739), SiLU(gate) * up (llama.py:214, precise expf + division like the decode kernels).
//
// Structure: block = 128 rows x 128 features, 4 waves (2 x 2) of 64 x 64 = 4 x 4 accumulators; a k-tile stage = 32 one-KiB fragments
// (8 Whi, 8 Wlo, 8 Xhi, 8 Xlo) copied by LDS-DMA into a ring of SP_RING stages (2 x 32 KB: two blocks per CU; 48 MFMAs per wave and stage).
#include "kernels.h"

typedef _Float16 half2v __attribute__((ext_vector_type(2)));

#define SP_WSCALE 64.0f
#define SP_ACT_SCALE 16.0f
// LDS ring of k-tile stages: 2 x 32 KB = two blocks per CU (see cnx_gemm.h): 32 x 512-token prompt pass 26.6 -> 22.8 ms against a ring of 3 (one block per CU)
#ifndef SP_RING
#define SP_RING 2
#endif
#define SP_STAGE (32 * 1024)

// `sat` != null: a value beyond the fp16 range (or a NaN) is counted (ctts_gpt_saturations) -- passed where the operand is unbounded (SwiGLU outputs);
// normalised rows, RoPE'd q / k, v and attention outputs are bounded by construction (|w_row| * sqrt(768))
__device__ inline void split_h4(const f32x4 v, half4& hi, half4& lo, int* sat = nullptr) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const float c = fminf(fmaxf(v[j], -65504.f), 65504.f);
        if (sat != nullptr && !(c == v[j])) atomicAdd(sat, 1);
        hi[j] = (half_t)c;
        lo[j] = (half_t)(c - (float)hi[j]);
    }
}

// RMSNorm once per row (llama.py:82-87; the weight is folded into W's columns) -> head / tail fp16 fragment images [16-row group][24 k-tiles][lane][16 B]
__global__ __launch_bounds__(256) void norm_pack_split_kernel(const float* x, half_t* hi, half_t* lo, int R, float eps) {
    constexpr int K = 768, KTILES = K / 32, PER = K / 256;
    const int lane = threadIdx.x & 63, r = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= R) return;
    const f32x4* xr = (const f32x4*)(x + (size_t)r * K);
    f32x4 v[PER];
#pragma unroll
    for (int i = 0; i < PER; ++i) v[i] = xr[lane + 64 * i];
    float ss = 0.f;
#pragma unroll
    for (int i = 0; i < PER; ++i) ss += v[i][0] * v[i][0] + v[i][1] * v[i][1] + v[i][2] * v[i][2] + v[i][3] * v[i][3];
    ss = wave_sum(ss);
    const float rs = 1.0f / sqrtf(ss / (float)K + eps);
    const size_t base = (size_t)(r >> 4) * KTILES * 64 * 8;
#pragma unroll
    for (int i = 0; i < PER; ++i) {
        const int k = 4 * (lane + 64 * i);
        half4 h, l;
        split_h4((f32x4){v[i][0] * rs, v[i][1] * rs, v[i][2] * rs, v[i][3] * rs}, h, l);
        const size_t o = base + xfrag_index<half_t>(r & 15, k, KTILES);
        *(half4*)(hi + o) = h;
        *(half4*)(lo + o) = l;
    }
}

// fp32 fragment image (the attention kernel's output, [16-row group][48 k-tiles][lane][4 floats]) -> head / tail fp16 images
__global__ __launch_bounds__(256) void split_pack_kernel(const float* src, half_t* hi, half_t* lo, int R) {
    constexpr int K = 768;
    const int idx = blockIdx.x * 256 + threadIdx.x;          // (row, 4 consecutive k)
    const int r = idx / (K / 4), k = 4 * (idx % (K / 4));
    if (r >= R) return;
    const f32x4 v = *(const f32x4*)(src + (size_t)(r >> 4) * (K / 16) * 64 * 4 + xfrag_index<float>(r & 15, k, K / 16));
    half4 h, l;
    split_h4(v, h, l);
    const size_t o = (size_t)(r >> 4) * (K / 32) * 64 * 8 + xfrag_index<half_t>(r & 15, k, K / 32);
    *(half4*)(hi + o) = h;
    *(half4*)(lo + o) = l;
}

struct SplitGemm {
    const half_t *Whi, *Wlo, *Xhi, *Xlo;   // fragment images: W [n tile][k tile][head | tail][lane][8] (Wlo = Whi + 1 KiB: the tile pairs of common.h split_t), X [16-row group][k tile][lane][8]
    int ktiles, R;
    int kt_per;                             // k-tiles one block multiplies: ktiles, or ktiles / 4 when grid.z slices K (EPI_PART: short passes of the down projection, sp_launch)
    size_t part_stride;                     // EPI_PART: floats between two slices' partial outputs [slice][row][N] (GemmArgs.part_out)
    float scale;                            // applied to the accumulators: 1 / SP_WSCALE (x SP_ACT_SCALE for the down projection)
    half_t *act_hi, *act_lo;                // EPI_SWIGLU: output images [16-row group][96 k-tiles][lane][8] of silu(g) * u / SP_ACT_SCALE
};

// Epilogue of a wave tile of NT n tiles x NG 16-row groups (both split GEMM kernels).  C tile layout: lane = (iq = lane >> 4, n = lane & 15): activation row n of
// the group, weight rows 4 * iq + j (j = register).  rt0 = the wave's first n tile (q / k / v: 4 consecutive tiles = one head, tile t of it = dims 8t.. | 8t + 32..;
// gate|up: a tile = 8 outputs), G0 = its first row group.
template <int EPI, int NG, int NT>
__device__ __forceinline__ void sp_epilogue(f32x4 (&acc)[NT][NG], const SplitGemm& p, const GemmArgs& a, const int lane, const int rt0, const int G0) {
    int nsat = 0;                                          // EPI_SWIGLU: this lane's clamped outputs (one atomic per wave at the end)
    const int iq = lane >> 4, nn = lane & 15;
    constexpr int H = 768, NH = H / CTTS_HEAD_DIM, HT = H / 16;
    const bool lowh = iq < 2;
    const float sc = p.scale;
#pragma unroll
    for (int g = 0; g < NG; ++g) {
        const int G = G0 + g;                              // 16-row group
        const int row = G * 16 + nn;
        const bool rv = row < p.R;
        if (EPI == EPI_RESID) {
            const int N = a.n_row_tiles * 16;
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                if (!rv) continue;
                float* xo = a.x_out + (size_t)row * N + (rt0 + t) * 16 + 4 * iq;
                const f32x4 x = *(const f32x4*)xo, c = acc[t][g];
                f32x4 dl = {0.f, 0.f, 0.f, 0.f};             // per-utterance LoRA term of o_proj (lora.hip): part of the projection
                if (a.lora_delta != nullptr) dl = *(const f32x4*)(a.lora_delta + (size_t)row * N + (rt0 + t) * 16 + 4 * iq);
                *(f32x4*)xo = (f32x4){x[0] + (c[0] * sc + dl[0]), x[1] + (c[1] * sc + dl[1]), x[2] + (c[2] * sc + dl[2]), x[3] + (c[3] * sc + dl[3])};      // residual + proj (llama.py:731,739)
            }
        } else if (EPI == EPI_PART) {                      // a K slice's share of the product, scaled; resid_combine_kernel adds the slices in order and the residual
            const int N = a.n_row_tiles * 16;
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                if (!rv) continue;
                const f32x4 c = acc[t][g];
                *(f32x4*)(a.part_out + (size_t)blockIdx.z * p.part_stride + (size_t)row * N + (rt0 + t) * 16 + 4 * iq) = (f32x4){c[0] * sc, c[1] * sc, c[2] * sc, c[3] * sc};
            }
        } else if (EPI == EPI_SWIGLU) {
            // tile rows [8 gate | 8 up]: lanes iq 0,1 hold gate rows 4 iq + j, lanes iq 2,3 the matching up rows.  Every lane finishes TWO outputs (round 6; the gate
            // lanes used to finish all four while the up lanes idled through the expf + division: half the epilogue's VALU time): a gate lane keeps j = 0, 1 and
            // receives the up values, its partner (lane ^ 32) keeps j = 2, 3 and receives the gate values -- two exchanges instead of four, the same arithmetic per output.
            const int ktiles_out = (a.n_row_tiles * 8) / 32;
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const f32x4 c = acc[t][g];
                float gv[2], uv[2];
#pragma unroll
                for (int jj = 0; jj < 2; ++jj) {
                    const float got = __shfl_xor(lowh ? c[2 + jj] : c[jj], 32);
                    gv[jj] = (lowh ? c[jj] : got) * sc;
                    uv[jj] = (lowh ? got : c[2 + jj]) * sc;
                }
                if (rv) {
                    half2v h, l;
#pragma unroll
                    for (int jj = 0; jj < 2; ++jj) {
                        const float y = (gv[jj] / (1.0f + expf(-gv[jj]))) * uv[jj] * (1.0f / SP_ACT_SCALE);
                        const float cl = fminf(fmaxf(y, -65504.f), 65504.f);       // silu(g) * u / 16 is unbounded: clamp + report (split_h4), also in the fp32 engine's prompt pass
                        nsat += !(cl == y);
                        h[jj] = (half_t)cl;
                        l[jj] = (half_t)(cl - (float)h[jj]);
                    }
                    const size_t off = (size_t)G * ktiles_out * 64 * 8 + xfrag_index<half_t>(nn, (rt0 + t) * 8 + 4 * (iq & 1) + (lowh ? 0 : 2), ktiles_out);
                    *(half2v*)(p.act_hi + off) = h;
                    *(half2v*)(p.act_lo + off) = l;
                }
            }
        } else {                                           // EPI_QKV
            RowMeta m = {0, 0, 0, 0};
            if (rv && rt0 >= HT) m = a.meta[row];           // (a wave's tiles never straddle q | k | v: 48 tiles each, rt0 a multiple of NT = 3 or 4)
            const int which = rt0 / HT;
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const int T = rt0 + t, hh = (T % HT) >> 2;   // head of this tile
                const f32x4 c = acc[t][g];
                f32x4 o;
#pragma unroll
                for (int j = 0; j < 4; ++j) o[j] = __shfl_xor(c[j], 32);
                if (!rv) continue;
                const int d0 = (T & 3) * 8 + 4 * (iq & 1);   // first of this lane's 4 frequency indices
                f32x4 y;
                f32x4 la = {0.f, 0.f, 0.f, 0.f}, lb = {0.f, 0.f, 0.f, 0.f};      // per-utterance LoRA terms of dims d0.. and d0 + 32.. (lora.hip): before RoPE
                if (a.lora_delta != nullptr) {
                    const float* dlp = a.lora_delta + ((size_t)row * 3 + which) * H + hh * CTTS_HEAD_DIM + d0;
                    la = *(const f32x4*)dlp; lb = *(const f32x4*)(dlp + 32);
                }
                if (which < 2) {
                    const f32x4 cs = *(const f32x4*)(a.rope_rows + (size_t)row * 64 + d0), sn = *(const f32x4*)(a.rope_rows + (size_t)row * 64 + 32 + d0);
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const float va = (lowh ? c[j] : o[j]) * sc + la[j], vb = (lowh ? o[j] : c[j]) * sc + lb[j];
                        // q*cos + rotate_half(q)*sin (llama.py:180-181), products rounded separately like the reference: dims < 32 take a cos - b sin, dims >= 32 b cos + a sin
                        y[j] = __fadd_rn(__fmul_rn(lowh ? va : vb, cs[j]), __fmul_rn(lowh ? -vb : va, sn[j]));
                    }
                } else {
#pragma unroll
                    for (int j = 0; j < 4; ++j) y[j] = c[j] * sc + (lowh ? la[j] : lb[j]);
                }
                const int dd = d0 + (lowh ? 0 : 32);
                if (which == 0) *(f32x4*)(a.q_out + ((size_t)row * NH + hh) * CTTS_HEAD_DIM + dd) = y;
                else *(f32x4*)((float*)(which == 1 ? a.k_cache : a.v_cache) + (((size_t)m.seq * NH + hh) * a.Lmax + m.slot) * CTTS_HEAD_DIM + dd) = y;
            }
        }
    }
    if (EPI == EPI_SWIGLU && a.sat != nullptr && nsat != 0) atomicAdd(a.sat, nsat);
}

// RING = LDS stages of 32 KB.  2: two blocks per CU, the second block's MFMAs fill the first one's barrier / landing bubbles (long passes).  4: one block per CU with three
// stages in flight, one barrier per k-tile -- for grids of at most one block per CU (short passes), where a k-tile took ~1 us of which 0.4 are MFMAs: with a ring of two the
// copy of stage kt + 2 is issued only after stage kt is read and has one iteration to land.
// WT = n tiles and row groups per wave: 4 (128 x 128 blocks) or 2 (64 x 64 blocks, stages of 16 KB: four times the blocks for passes so short that 128 x 128 blocks leave most
// CUs idle -- a CU pulls its block's panels at ~30 GB/s whatever the ring depth, the chip has 256 of them).
template <int EPI, int RING = SP_RING, int WT = 4>
__global__ __launch_bounds__(256, RING == 2 ? 2 : 1) void prefill_split_gemm_kernel(const SplitGemm p, const GemmArgs a) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 1, wn = wave & 1;
    // XCD-aware tile order (see prefill_gemm.hip): XCD x owns the row blocks y = x (mod 8) and walks the column blocks in order
    int bx = blockIdx.x, by = blockIdx.y;
    if ((gridDim.y & 7) == 0) {
        const int lin = blockIdx.x + gridDim.x * blockIdx.y, xcd = lin & 7, seq = lin >> 3, rpx = gridDim.y >> 3;
        bx = seq / rpx;
        by = (seq % rpx) * 8 + xcd;
    }
    constexpr int BT = 2 * WT, STAGE = 4 * BT * 1024;      // tiles (= row groups) per block; bytes per stage
    const int nt0 = bx * BT, g0 = by * BT;                 // first n tile / first 16-row group of the block
    const int ktiles = p.kt_per;                           // this block's k-tiles: all of them, or slice blockIdx.z (EPI_PART)
    const int kt_all = p.ktiles, kt0 = blockIdx.z * p.kt_per;
    // fragment f of a stage: BT Whi tiles, BT Wlo tiles, BT Xhi groups, BT Xlo groups (WT = 4: 0..7, 8..15, 16..23, 24..31)
    auto src = [&](int f, int kt) -> const char* {
        const half_t* img = (f < BT) ? p.Whi : (f < 2 * BT) ? p.Wlo : (f < 3 * BT) ? p.Xhi : p.Xlo;
        const int unit = (f < 2 * BT) ? nt0 + (f & (BT - 1)) : g0 + (f & (BT - 1));
        return (const char*)img + ((size_t)unit * kt_all + kt0 + kt) * ((f < 2 * BT) ? 2048 : 1024) + (unsigned)(lane * 16);
    };
    typedef __attribute__((address_space(1))) const void* gptr_t;
    typedef __attribute__((address_space(3))) void* lptr_t;
#define SP_DMA(kt_, buf_)                                                                                              \
    _Pragma("unroll") for (int i = 0; i < BT; ++i) {                                                                    \
        const int f = wave + 4 * i;                                                                                      \
        __builtin_amdgcn_global_load_lds((gptr_t)src(f, (kt_)), (lptr_t)(lds + (buf_) * STAGE + f * 1024), 16, 0, 0);    \
    }
    // s_waitcnt immediate (gfx9): vmcnt[3:0] and [15:14] | expcnt[6:4] = 7 (none) | lgkmcnt[11:8] = 0; a bare s_barrier (no vmcnt(0) fence)
#define SP_WAIT_BAR(n_) { __builtin_amdgcn_s_waitcnt(0x0070 | ((n_) & 15) | (((n_) >> 4) << 14)); __builtin_amdgcn_s_barrier(); }
    f32x4 acc[WT][WT];                                     // [n tile][row group]
#pragma unroll
    for (int t = 0; t < WT; ++t)
#pragma unroll
        for (int g = 0; g < WT; ++g) acc[t][g] = (f32x4){0.f, 0.f, 0.f, 0.f};
    SP_DMA(0, 0)
    SP_DMA(1, 1)                                           // ktiles >= 2 (checked by the launcher)
    if (RING > 2 && ktiles > 2) SP_DMA(2, 2)
    int cb = 0;
    for (int kt = 0; kt < ktiles; ++kt) {
        // stage kt has landed for every wave (a wave's 8 loads per stage retire in order: the later stages' may stay in flight), and every wave is done with the
        // slot the next copy overwrites (RING > 2: it held stage kt - 1, whose fragments were consumed before this barrier)
        if (RING > 2 && kt + 2 < ktiles) SP_WAIT_BAR(2 * BT) else if (kt + 1 < ktiles) SP_WAIT_BAR(BT) else SP_WAIT_BAR(0)      // (a wave copies BT fragments per stage)
        const char* cur = lds + cb * STAGE;
        half8 wh[WT], wl[WT], xh[WT], xl[WT];
#pragma unroll
        for (int t = 0; t < WT; ++t) {
            wh[t] = *(const half8*)(cur + (wn * WT + t) * 1024 + lane * 16);
            wl[t] = *(const half8*)(cur + (BT + wn * WT + t) * 1024 + lane * 16);
        }
#pragma unroll
        for (int g = 0; g < WT; ++g) {
            xh[g] = *(const half8*)(cur + (2 * BT + wr * WT + g) * 1024 + lane * 16);
            xl[g] = *(const half8*)(cur + (3 * BT + wr * WT + g) * 1024 + lane * 16);
        }
        if (RING == 2) {                                   // two stages (64 KB: two blocks per CU): the stage just read is the one the next copy overwrites
            __builtin_amdgcn_s_waitcnt(0xC07F);            // lgkmcnt(0): this wave's fragments are in registers
            __builtin_amdgcn_s_barrier();
        }
        constexpr int AHEAD = RING == 2 ? 2 : 3;           // stages in flight
        if (kt + AHEAD < ktiles) {
            const int nb = (cb + AHEAD >= RING) ? cb + AHEAD - RING : cb + AHEAD;
            SP_DMA(kt + AHEAD, nb)
        }
#pragma unroll
        for (int t = 0; t < WT; ++t)
#pragma unroll
            for (int g = 0; g < WT; ++g) {
                // tails first, head product last: the small terms meet while the accumulator's low bits still see them
                acc[t][g] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl[t], xh[g], acc[t][g], 0, 0, 0);
                acc[t][g] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[t], xl[g], acc[t][g], 0, 0, 0);
                acc[t][g] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[t], xh[g], acc[t][g], 0, 0, 0);
            }
        cb = (cb + 1 == RING) ? 0 : cb + 1;
    }
#undef SP_DMA
#undef SP_WAIT_BAR
    sp_epilogue<EPI, WT, WT>(acc, p, a, lane, nt0 + wn * WT, g0 + wr * WT);
}

// ------------------------------------------------------------------------------------------------
// The same product on 256-row blocks with two wave groups in counter-phase (round 6; long prompt passes: sp_launch decides by a round count).
//
// Why: the 128 x 128 kernel above measured 0.38-0.40 MFMA-busy (profiles/r06_pmc_mfma_split.json).  A 3-term product needs FOUR fragments per 16 x 16 x 32 tile
// pair, so the operand traffic a plain fp16 GEMM has with 64 x 64 wave tiles needs 128 x 64 here: a wave holds NT = 4 n tiles x 8 row groups (24 fragments for 96
// MFMAs: 0.25 instead of 0.33 fragments per MFMA; 128 accumulator registers), 8 waves = 2 row halves x 4 feature quarters = 256 rows x 256 features per block, ONE
// block per CU (ring of 2 x 64 KB stages): half the L2 -> LDS bytes per MFMA.  With one block per CU the waves of a SIMD would read and multiply in lock step (the
// 128 x 128 kernel overlaps through its second block), so the two row halves (waves 0-3 / 4-7: one of each per SIMD) run half an iteration apart: while a group
// multiplies k-tile kt out of registers the other reads its fragments and issues its LDS-DMA copies.
//   phase p (between two block barriers):   even: group 0 reads stage p / 2, group 1 multiplies stage p / 2 - 1;   odd: group 0 multiplies, group 1 reads stage (p - 1) / 2
//   stage s lives in buffer s & 1: read by group 0 in phase 2 s, by group 1 in phase 2 s + 1; the copies of stage s + 2 are issued by group 0's waves in phase 2 s + 2
//   and by group 1's in phase 2 s + 3 (their READ phases) and waited for (vmcnt 0, every wave its own) at the end of phase 2 s + 3, before group 0 reads it.
// Accumulation order per output element = the 128 x 128 kernel's (k-tiles ascending; tail.head, head.tail, head.head): the kernels agree bit for bit
// (tests/test_gpu_split_decode.py::test_prompt_pass_block_shapes_agree_bitwise).
// Measured at 32 x 512 prompt rows (profiles/r06_ab_prefill_pp_gemm.jsonl): gate|up 429 -> 368 us, q|k|v 183 -> 170 (NT = 3), o_proj + down 287 -> 238 (NT = 3: 256
// blocks = one per CU instead of 768 on 512 slots); prompt pass 21.6 -> 19.1 ms.  Where a gate|up launch's time goes (experiment builds of the first version, 461 us):
// MFMAs alone 250 us (1.86 PFLOP/s: the chip clocks ~1.8 GHz under this load), + copies, reads and barriers 298, + epilogue 383 (after the epilogue below took all
// lanes; 461 before) -- the first version issued group 1's copies in front of its MFMAs (60-180 issue cycles per copy): 74 us; fragment reads cost nothing.
#define SPB_STAGE (64 * 1024)
// NT = n tiles per wave: 4 (256-feature blocks) or 3 (192-feature blocks: N = 768 becomes 4 x 64 = 256 blocks for 16384 rows -- one per CU -- instead of 192, and
// q|k|v 768 blocks = 3 full rounds instead of 576 = 2.25).  launch_prefill_split_gemm takes the cheaper of the shapes.
template <int EPI, int NT>
__global__ __launch_bounds__(512, 1) void prefill_split_gemm_pp_kernel(const SplitGemm p, const GemmArgs a) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wn = wave & 3;               // wr = row half = phase group
    int bx = blockIdx.x, by = blockIdx.y;                  // XCD-aware tile order as above
    if ((gridDim.y & 7) == 0) {
        const int lin = blockIdx.x + gridDim.x * blockIdx.y, xcd = lin & 7, seq = lin >> 3, rpx = gridDim.y >> 3;
        bx = seq / rpx;
        by = (seq % rpx) * 8 + xcd;
    }
    constexpr int NW = 4 * NT;                             // n tiles per block
    constexpr int NP = 2 * NW + 32;                        // 1 KiB fragments per stage: NW Whi tiles, NW Wlo tiles, 16 Xhi groups, 16 Xlo groups
    const int nt0 = bx * NW, g0 = by * 16;
    const int ktiles = p.ktiles;
    auto src = [&](int f, int kt) -> const char* {          // (f is wave-uniform: the selects are scalar)
        const half_t* img = (f < NW) ? p.Whi : (f < 2 * NW) ? p.Wlo : (f < 2 * NW + 16) ? p.Xhi : p.Xlo;
        const int unit = (f < NW) ? nt0 + f : (f < 2 * NW) ? nt0 + f - NW : g0 + ((f - 2 * NW) & 15);
        return (const char*)img + ((size_t)unit * ktiles + kt) * ((f < 2 * NW) ? 2048 : 1024) + (unsigned)(lane * 16);
    };
    typedef __attribute__((address_space(1))) const void* gptr_t;
    typedef __attribute__((address_space(3))) void* lptr_t;
    // a wave's copies of a stage: fragments wave + 8 i, i < NP / 8 (8 or 7 per wave)
#define SPB_DMA(kt_, buf_)                                                                                               \
    _Pragma("unroll") for (int i = 0; i < NP / 8; ++i) {                                                                  \
        const int f = wave + 8 * i;                                                                                        \
        __builtin_amdgcn_global_load_lds((gptr_t)src(f, (kt_)), (lptr_t)(lds + (buf_) * SPB_STAGE + f * 1024), 16, 0, 0); \
    }
    // s_waitcnt immediates (gfx9): vmcnt[3:0] | [15:14], expcnt[6:4] = 7 (none), lgkmcnt[11:8]
#define SPB_BAR_VM0_LGKM0 { __builtin_amdgcn_s_waitcnt(0x0070); __builtin_amdgcn_s_barrier(); __builtin_amdgcn_sched_barrier(0); }
#define SPB_BAR_LGKM0 { __builtin_amdgcn_s_waitcnt(0xC07F); __builtin_amdgcn_s_barrier(); __builtin_amdgcn_sched_barrier(0); }
    f32x4 acc[NT][8];                                      // [n tile][row group]
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int g = 0; g < 8; ++g) acc[t][g] = (f32x4){0.f, 0.f, 0.f, 0.f};
    half8 wh[NT], wl[NT], xh[8], xl[8];
#define SPB_READ(buf_)                                                                                 \
    {                                                                                                   \
        const char* cur = lds + (buf_) * SPB_STAGE + lane * 16;                                          \
        _Pragma("unroll") for (int t = 0; t < NT; ++t) {                                                 \
            wh[t] = *(const half8*)(cur + (wn * NT + t) * 1024);                                          \
            wl[t] = *(const half8*)(cur + (NW + wn * NT + t) * 1024);                                     \
        }                                                                                                \
        _Pragma("unroll") for (int g = 0; g < 8; ++g) {                                                  \
            xh[g] = *(const half8*)(cur + (2 * NW + wr * 8 + g) * 1024);                                  \
            xl[g] = *(const half8*)(cur + (2 * NW + 16 + wr * 8 + g) * 1024);                             \
        }                                                                                                \
    }
    // tails first, head product last: the small terms meet while the accumulator's low bits still see them (as in the 128 x 128 kernel)
#define SPB_MFMA                                                                                         \
    {                                                                                                   \
        __builtin_amdgcn_s_setprio(1);                                                                   \
        _Pragma("unroll") for (int t = 0; t < NT; ++t)                                                   \
            _Pragma("unroll") for (int g = 0; g < 8; ++g) {                                              \
                acc[t][g] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl[t], xh[g], acc[t][g], 0, 0, 0);     \
                acc[t][g] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[t], xl[g], acc[t][g], 0, 0, 0);     \
                acc[t][g] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[t], xh[g], acc[t][g], 0, 0, 0);     \
            }                                                                                            \
        __builtin_amdgcn_s_setprio(0);                                                                   \
    }
    // Every copy is issued in a READ phase (a copy costs its wave 60-180 issue cycles: in front of group 1's MFMAs it cost the kernel 16 % of its time) and is waited
    // for at the end of the phase before the stage's first read: group 0's copies have two phases to land, group 1's one.
    SPB_DMA(0, 0)
    SPB_DMA(1, 1)                                          // ktiles >= 2 (checked by the launcher)
    __builtin_amdgcn_s_waitcnt(0x0070 | (NP / 8));         // vmcnt(NP / 8): stage 0 has landed (a wave's copies retire in order)
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    if (wr == 0) {
        for (int kt = 0; kt < ktiles; ++kt) {
            // phase 2 kt: group 0 reads stage kt.  Buffer (kt + 1) & 1 (stage kt - 1) was last read by group 1 in phase 2 kt - 1: stage kt + 1 may come
            if (kt >= 1 && kt + 1 < ktiles) SPB_DMA(kt + 1, (kt + 1) & 1)
            SPB_READ(kt & 1)
            SPB_BAR_LGKM0
            // phase 2 kt + 1: group 0 multiplies
            SPB_MFMA
            SPB_BAR_VM0_LGKM0                              // own copies of stage kt + 1 landed before anyone reads it in phase 2 kt + 2
        }
    } else {
        __builtin_amdgcn_s_barrier();                      // phase 0: group 1 has nothing to multiply yet
        __builtin_amdgcn_sched_barrier(0);
        for (int kt = 0; kt < ktiles; ++kt) {
            // phase 2 kt + 1: group 1 reads stage kt; its copies of stage kt + 1 go into the other buffer (free since the end of phase 2 kt - 1) and land within this phase
            if (kt >= 1 && kt + 1 < ktiles) SPB_DMA(kt + 1, (kt + 1) & 1)
            SPB_READ(kt & 1)
            SPB_BAR_VM0_LGKM0                              // fragments in registers; own copies of stage kt + 1 landed
            // phase 2 kt + 2: group 1 multiplies
            SPB_MFMA
            if (kt + 1 < ktiles) { __builtin_amdgcn_s_barrier(); __builtin_amdgcn_sched_barrier(0); }      // (group 0 left after phase 2 ktiles - 1)
        }
    }
#undef SPB_DMA
#undef SPB_READ
#undef SPB_MFMA
#undef SPB_BAR_VM0_LGKM0
#undef SPB_BAR_LGKM0
    sp_epilogue<EPI, 8, NT>(acc, p, a, lane, nt0 + wn * NT, g0 + wr * 8);
}

// ------------------------------------------------------------------------------------------------
// Prompt attention of the parity engine: the flash-style MFMA kernel of attention.hip (attn_prefill_mfma_kernel: block = (sequence, head, 64 queries),
// S^T = K.Q^T with the lane owning ONE query so that P^T is already the B operand of the P.V product) on head / tail operands.  K and V come from
// the fp32 cache and are split while a 64-key chunk is staged in LDS; Q (pre-scaled by 1/8, exact) and P (<= 1) are split in registers; every
// product is 3 MFMAs (tail x head, head x tail, head x head).  exp is the precise expf (parity mode).  llama.py:590-668 at q_len > 1, mask semantics
// of llama.py:1073-1087 (a row attends to the key slots [kv_start, slot] of its own sequence).  The fp32 row-by-row kernel this replaces took
// 1.2 ms per layer for 32 x 512 prompt tokens (half of the prompt pass once the projections ran on the split GEMM).
// Output: the normalised rows as head / tail fp16 images = the X operand of the o_proj split GEMM (no separate conversion pass).
#define FS_PITCH 68        // halfs per V^T row in LDS (64 keys + pad; rows stay 8-byte aligned)
#define FS_KPITCH 72       // halfs per K row in LDS (64 dims + pad; rows stay 16-byte aligned)
#define FS_LDS (2 * (2 * 64 * FS_PITCH + 2 * 64 * FS_KPITCH) * 2)
#ifndef FS_WAVES
#define FS_WAVES 8          // waves (16 queries each) per block: 8 since round 6 (4 before)
#endif
// FS_EXP: the softmax exponential.  The parity engine keeps the correctly rounded expf (13 VALU instructions; half of the loop's 444 per 64-key chunk against 48 MFMAs:
// the kernel is VALU-bound on it).  -DFS_FAST_EXP (A/B builds only, never shipped: profiles/r06_ab_prefill_pp_gemm.jsonl) takes v_exp_f32 (2 instructions, ~1 ulp).
#ifdef FS_FAST_EXP
#define FS_EXP(x_) __builtin_amdgcn_exp2f((x_) * 1.44269504088896341f)
#else
#define FS_EXP(x_) expf(x_)
#endif
__global__ __launch_bounds__(64 * FS_WAVES) void attn_prefill_split_kernel(const RowMeta* meta_p, const float* q_p, const float* k_p, const float* v_p, const int NHp, const int R,
                                                               const AttnArgs a, half_t* out_hi, half_t* out_lo) {
    extern __shared__ __attribute__((aligned(16))) char fs_lds[];
    typedef half_t (*vt_t)[64][FS_PITCH];
    typedef half_t (*ks_t)[64][FS_KPITCH];
    vt_t vth = (vt_t)fs_lds, vtl = (vt_t)(fs_lds + 2 * 64 * FS_PITCH * 2);
    ks_t ksh = (ks_t)(fs_lds + 4 * 64 * FS_PITCH * 2), ksl = (ks_t)(fs_lds + 4 * 64 * FS_PITCH * 2 + 2 * 64 * FS_KPITCH * 2);
    __shared__ int range_s[FS_WAVES][3];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int qn = lane & 15, iq = lane >> 4;
    const int b = blockIdx.z, h = blockIdx.y, T = a.T;
    const int t = blockIdx.x * (16 * FS_WAVES) + wave * 16 + qn;         // this lane's query (prompt position)
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
    int blo = range_s[0][0], bhi = range_s[0][1], cseq = range_s[0][2];
#pragma unroll
    for (int w = 1; w < FS_WAVES; ++w) { blo = min(blo, range_s[w][0]); bhi = max(bhi, range_s[w][1]); cseq = max(cseq, range_s[w][2]); }
    if (bhi < 0) return;                                                  // no live query in this block (uniform)
    wlo = __builtin_amdgcn_readfirstlane(wlo); whi = __builtin_amdgcn_readfirstlane(whi);
    const size_t head_off = ((size_t)cseq * NHp + h) * a.Lmax * CTTS_HEAD_DIM;
    const float* kb = k_p + head_off;
    const float* vb = v_p + head_off;
    // Q fragments (B operand): lane (query qn, kq = iq): dims 8 iq .. + 7 of each 32-dim half, scaled by 1/sqrt(64), head / tail
    half8 qh[2], ql[2];
    {
        const float* qp = q_p + ((size_t)(live ? r : 0) * NHp + h) * CTTS_HEAD_DIM + 8 * iq;
        const float sc = live ? 0.125f : 0.f;
#pragma unroll
        for (int kh = 0; kh < 2; ++kh) {
            const f32x4 q0 = *(const f32x4*)(qp + 32 * kh), q1 = *(const f32x4*)(qp + 32 * kh + 4);
            half4 h0, l0, h1, l1;
            split_h4((f32x4){q0[0] * sc, q0[1] * sc, q0[2] * sc, q0[3] * sc}, h0, l0);
            split_h4((f32x4){q1[0] * sc, q1[1] * sc, q1[2] * sc, q1[3] * sc}, h1, l1);
            qh[kh] = (half8){h0[0], h0[1], h0[2], h0[3], h1[0], h1[1], h1[2], h1[3]};
            ql[kh] = (half8){l0[0], l0[1], l0[2], l0[3], l1[0], l1[1], l1[2], l1[3]};
        }
    }
    f32x4 oacc[4];                                                        // O^T: dims 16 db + 4 iq + j of this lane's query
#pragma unroll
    for (int db = 0; db < 4; ++db) oacc[db] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float mrun = -INFINITY, lpart = 0.f;
    // staging: thread -> key (tid / SPK) of the chunk, SPK threads per key, dims 4 (tid % SPK) + 4 SPK i .. + 3 for i < 16 / SPK.  (Round 5: the lanes of a key used to own
    // CONSECUTIVE dims; their transposing 2-byte V stores then sat 16 rows = 544 dwords = 0 or 32 banks apart and the 8-byte K stores 8 dwords apart under a 36-dword row
    // pitch: 2-way / 4-way LDS bank conflicts on every store, 47 % of the kernel's LDS cycles.  4 dims apart they land 8 / 2 banks apart.)
    // Round 6: 8 waves = 128 queries per block share a staged chunk (4 waves = 64 queries before): half the loads, splits and transposing stores per query.
    constexpr int SPK = FS_WAVES, SNI = 16 / SPK;                         // threads per key; 16-byte pieces per thread and array
    const int skey = tid / SPK, sdim = 4 * (tid % SPK);
    f32x4 vst[SNI], kst[SNI];
    auto vload = [&](int c) {
        const int key = min(c + skey, bhi);                               // clamp: a valid slot of this sequence (masked later)
        const f32x4* vp = (const f32x4*)(vb + (size_t)key * CTTS_HEAD_DIM + sdim);
        const f32x4* kp = (const f32x4*)(kb + (size_t)key * CTTS_HEAD_DIM + sdim);
#pragma unroll
        for (int i = 0; i < SNI; ++i) { vst[i] = vp[SPK * i]; kst[i] = kp[SPK * i]; }
    };
    auto vstore = [&](int buf) {
#pragma unroll
        for (int i = 0; i < SNI; ++i) {
            half4 vh, vl, kh, kl;
            split_h4(vst[i], vh, vl);
            split_h4(kst[i], kh, kl);
#pragma unroll
            for (int e = 0; e < 4; ++e) { vth[buf][sdim + 4 * SPK * i + e][skey] = vh[e]; vtl[buf][sdim + 4 * SPK * i + e][skey] = vl[e]; }
            *(half4*)&ksh[buf][skey][sdim + 4 * SPK * i] = kh;
            *(half4*)&ksl[buf][skey][sdim + 4 * SPK * i] = kl;
        }
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
                // A operand row = key (lane & 15), dims 8 iq .. and 32 + 8 iq ..
                const half8 k0h = *(const half8*)&ksh[buf][16 * tl + qn][8 * iq], k1h = *(const half8*)&ksh[buf][16 * tl + qn][32 + 8 * iq];
                const half8 k0l = *(const half8*)&ksl[buf][16 * tl + qn][8 * iq], k1l = *(const half8*)&ksl[buf][16 * tl + qn][32 + 8 * iq];
                f32x4 acc = {0.f, 0.f, 0.f, 0.f};
                acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(k0l, qh[0], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(k1l, qh[1], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(k0h, ql[0], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(k1h, ql[1], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(k0h, qh[0], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(k1h, qh[1], acc, 0, 0, 0);
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
            const float sc = (mrun == -INFINITY) ? 0.f : FS_EXP(mrun - mn);
            float ps = 0.f;
            half4 pTh[4], pTl[4];
#pragma unroll
            for (int tl = 0; tl < 4; ++tl) {
                f32x4 pv;
#pragma unroll
                for (int j = 0; j < 4; ++j) { pv[j] = (sT[tl][j] == -INFINITY) ? 0.f : FS_EXP(sT[tl][j] - mn); ps += pv[j]; }
                split_h4(pv, pTh[tl], pTl[tl]);
            }
            lpart = lpart * sc + ps;
            mrun = mn;
#pragma unroll
            for (int db = 0; db < 4; ++db) { oacc[db][0] *= sc; oacc[db][1] *= sc; oacc[db][2] *= sc; oacc[db][3] *= sc; }
            // P.V on the double-rate v_mfma_f32_16x16x32_f16 (round 6; 16x16x16 before: twice the instructions): a 32-key operand = two 16-key tiles, the lane's k-slots
            // 8 iq + e <-> keys 16 tl + 4 iq + e of tile tl = 2 pr (e < 4) and of tile 2 pr + 1 (e >= 4) -- the same assignment in A (V^T, two 8-byte reads) and B (P^T)
#pragma unroll
            for (int pr = 0; pr < 2; ++pr) {
                const half8 ph = {pTh[2 * pr][0], pTh[2 * pr][1], pTh[2 * pr][2], pTh[2 * pr][3], pTh[2 * pr + 1][0], pTh[2 * pr + 1][1], pTh[2 * pr + 1][2], pTh[2 * pr + 1][3]};
                const half8 pl = {pTl[2 * pr][0], pTl[2 * pr][1], pTl[2 * pr][2], pTl[2 * pr][3], pTl[2 * pr + 1][0], pTl[2 * pr + 1][1], pTl[2 * pr + 1][2], pTl[2 * pr + 1][3]};
#pragma unroll
                for (int db = 0; db < 4; ++db) {
                    const half4 vha = *(const half4*)&vth[buf][16 * db + qn][32 * pr + 4 * iq], vhb = *(const half4*)&vth[buf][16 * db + qn][32 * pr + 16 + 4 * iq];      // A = V^T: row = dim (lane & 15)
                    const half4 vla = *(const half4*)&vtl[buf][16 * db + qn][32 * pr + 4 * iq], vlb = *(const half4*)&vtl[buf][16 * db + qn][32 * pr + 16 + 4 * iq];
                    const half8 vh = {vha[0], vha[1], vha[2], vha[3], vhb[0], vhb[1], vhb[2], vhb[3]};
                    const half8 vl = {vla[0], vla[1], vla[2], vla[3], vlb[0], vlb[1], vlb[2], vlb[3]};
                    oacc[db] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vl, ph, oacc[db], 0, 0, 0);
                    oacc[db] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vh, pl, oacc[db], 0, 0, 0);
                    oacc[db] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vh, ph, oacc[db], 0, 0, 0);
                }
            }
        }
        if (more) vstore(buf ^ 1);
        __syncthreads();
    }
    float ltot = lpart + __shfl_xor(lpart, 16);
    ltot += __shfl_xor(ltot, 32);
    if (!live) return;
    const float inv = 1.0f / ltot;
    constexpr int KT = 768 / 32;
    const size_t base = (size_t)(r >> 4) * KT * 64 * 8;
#pragma unroll
    for (int db = 0; db < 4; ++db) {
        const int k = h * CTTS_HEAD_DIM + 16 * db + 4 * iq;
        half4 oh, ol;
        split_h4((f32x4){oacc[db][0] * inv, oacc[db][1] * inv, oacc[db][2] * inv, oacc[db][3] * inv}, oh, ol);
        const size_t o = base + xfrag_index<half_t>(r & 15, k, KT);
        *(half4*)(out_hi + o) = oh;
        *(half4*)(out_lo + o) = ol;
    }
}

// prompt attention of one pass on the split operands: a.q fp32 [R][NH][64], fp32 K / V cache of this layer -> head / tail images of the normalised output rows
int launch_attention_split(const AttnArgs& a, void* out_hi, void* out_lo, hipStream_t s) {
    static bool configured = false;
    if (!configured) {
        CTTS_HIP_CHECK(hipFuncSetAttribute((const void*)attn_prefill_split_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, FS_LDS));
        configured = true;
    }
    if (a.T <= 0 || a.S != 1) { ctts_set_error("attention_split: prompt pass only"); return 1; }
    const int B = (a.row0 + a.R + a.T - 1) / a.T;                         // sequences 0 .. B-1 may have rows in this pass
    dim3 g3((a.T + 16 * FS_WAVES - 1) / (16 * FS_WAVES), a.NH, B);
    hipLaunchKernelGGL(attn_prefill_split_kernel, g3, dim3(64 * FS_WAVES), FS_LDS, s, a.meta, a.q, (const float*)a.k_cache, (const float*)a.v_cache, a.NH, a.R, a, (half_t*)out_hi, (half_t*)out_lo);
    CTTS_HIP_CHECK(hipGetLastError());
    return 0;
}

int launch_norm_pack_split(const float* x, void* hi, void* lo, int R, float eps, hipStream_t s) {
    hipLaunchKernelGGL(norm_pack_split_kernel, dim3((R + 3) / 4), dim3(256), 0, s, x, (half_t*)hi, (half_t*)lo, R, eps);
    CTTS_HIP_CHECK(hipGetLastError());
    return 0;
}
int launch_split_pack(const float* src, void* hi, void* lo, int R, hipStream_t s) {
    hipLaunchKernelGGL(split_pack_kernel, dim3((R * (768 / 4) + 255) / 256), dim3(256), 0, s, src, (half_t*)hi, (half_t*)lo, R);
    CTTS_HIP_CHECK(hipGetLastError());
    return 0;
}

// x[row][n] += ((p0 + p1) + p2) + p3: the K slices of a short pass's down projection in slice order, then the residual (llama.py:739) -- fixed order, no atomics
__global__ __launch_bounds__(256) void resid_combine_kernel(const float* part, const size_t part_stride, float* x, const int n4) {
    const int i = blockIdx.x * 256 + threadIdx.x;          // one f32x4 of [R][N]
    if (i >= n4) return;
    const f32x4 p0 = ((const f32x4*)part)[i], p1 = ((const f32x4*)(part + part_stride))[i], p2 = ((const f32x4*)(part + 2 * part_stride))[i], p3 = ((const f32x4*)(part + 3 * part_stride))[i];
    const f32x4 xv = ((const f32x4*)x)[i];
    f32x4 o;
#pragma unroll
    for (int j = 0; j < 4; ++j) o[j] = xv[j] + (((p0[j] + p1[j]) + p2[j]) + p3[j]);
    ((f32x4*)x)[i] = o;
}

template <int EPI>
static int sp_launch(SplitGemm& p, const GemmArgs& a, const SplitGemmPolicy& pol, hipStream_t s) {
    static bool configured = false;
    if (!configured) {
        CTTS_HIP_CHECK(hipFuncSetAttribute((const void*)prefill_split_gemm_kernel<EPI>, hipFuncAttributeMaxDynamicSharedMemorySize, SP_RING * SP_STAGE));
        CTTS_HIP_CHECK(hipFuncSetAttribute((const void*)prefill_split_gemm_pp_kernel<EPI, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * SPB_STAGE));
        CTTS_HIP_CHECK(hipFuncSetAttribute((const void*)prefill_split_gemm_pp_kernel<EPI, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * SPB_STAGE));
        CTTS_HIP_CHECK(hipFuncSetAttribute((const void*)prefill_split_gemm_kernel<EPI, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, 4 * SP_STAGE));
        CTTS_HIP_CHECK(hipFuncSetAttribute((const void*)prefill_split_gemm_kernel<EPI, 4, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 4 * SP_STAGE / 2));
        if (EPI == EPI_RESID) {
            CTTS_HIP_CHECK(hipFuncSetAttribute((const void*)prefill_split_gemm_kernel<EPI_PART>, hipFuncAttributeMaxDynamicSharedMemorySize, SP_RING * SP_STAGE));
            CTTS_HIP_CHECK(hipFuncSetAttribute((const void*)prefill_split_gemm_kernel<EPI_PART, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, 4 * SP_STAGE));
            CTTS_HIP_CHECK(hipFuncSetAttribute((const void*)prefill_split_gemm_kernel<EPI_PART, 4, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 4 * SP_STAGE / 2));
        }
        configured = true;
    }
    // Block shape by a round count (256 CUs; measured per-round times at K = 768, 32 x 512 prompt rows: 128 x 128 blocks, 512 at a time: 36 us; 256 x 256 blocks,
    // one per CU: 60 us; 256 x 192: 45 us): the 128 x 128 kernel for short passes, the counter-phased kernel once its blocks fill the chip.  pp_min_blocks < 0 forces
    // the counter-phased kernel with NT = -pp_min_blocks (tests, A/B).
    const int pp_min_blocks = pol.pp_min_blocks;
    const int rb = (p.R + 255) / 256, nt = a.n_row_tiles;
    const int b_old = (nt / 8) * ((p.R + 127) / 128);
    const long c_old = (long)((b_old + 511) / 512) * 36;
    const int b4 = (nt % 16 == 0) ? (nt / 16) * rb : 0, b3 = (nt % 12 == 0) ? (nt / 12) * rb : 0;
    const long c4 = b4 ? (long)((b4 + 255) / 256) * 60 : (1L << 40), c3 = b3 ? (long)((b3 + 255) / 256) * 45 : (1L << 40);
    int shape = 0;                                          // 0: 128 x 128; 3 / 4: counter-phased, NT
    if (pp_min_blocks < 0) shape = (-pp_min_blocks == 3 && b3) ? 3 : (b4 ? 4 : (b3 ? 3 : 0));
    else if (pp_min_blocks > 0) {
        if (b4 >= pp_min_blocks && c4 < c_old && c4 <= c3) shape = 4;
        else if (b3 >= pp_min_blocks && c3 < c_old) shape = 3;
    }
    // Short passes of the down projection (K = 3072: 96 k-tiles in a row, 6 blocks per 128 rows -- 24 blocks on 256 CUs at 448 rows, 75 of the layer's 170 us): K sliced
    // four ways over grid.z, the slices' shares parked in `pol.sk_scratch`, resid_combine_kernel adds them in slice order and the residual.  Another summation order
    // than the unsliced kernel's (the usual 1e-7); a.lora_delta never rides here (the o_proj products have K = 768).
    const size_t rows_pad = (size_t)((p.R + 127) / 128) * 128;
    const bool sliced = EPI == EPI_RESID && shape == 0 && pol.sk_rows > 0 && p.R <= pol.sk_rows && p.ktiles >= 64 && (p.ktiles & 3) == 0 && a.lora_delta == nullptr &&
                        pol.sk_scratch != nullptr && 4 * rows_pad * (size_t)(nt * 16) <= pol.sk_cap_floats;
    // passes so short that even the sliced grids leave most CUs idle: 64 x 64 blocks (four times as many), 4-stage ring of 16 KB stages
    const bool small = shape == 0 && pol.small_blocks > 0 && (sliced ? 4 : 1) * b_old <= pol.small_blocks;
    const dim3 g64(nt / 4, (p.R + 63) / 64, sliced ? 4 : 1);
    if (sliced) {
        GemmArgs ap = a;
        ap.part_out = pol.sk_scratch;
        p.kt_per = p.ktiles / 4; p.part_stride = rows_pad * (size_t)(nt * 16);
        if (small) hipLaunchKernelGGL((prefill_split_gemm_kernel<EPI_PART, 4, 2>), g64, dim3(256), 4 * SP_STAGE / 2, s, p, ap);
        else if (4 * b_old <= pol.ring4_blocks) hipLaunchKernelGGL((prefill_split_gemm_kernel<EPI_PART, 4>), dim3(nt / 8, (p.R + 127) / 128, 4), dim3(256), 4 * SP_STAGE, s, p, ap);
        else hipLaunchKernelGGL((prefill_split_gemm_kernel<EPI_PART>), dim3(nt / 8, (p.R + 127) / 128, 4), dim3(256), SP_RING * SP_STAGE, s, p, ap);
        CTTS_HIP_CHECK(hipGetLastError());
        const int n4 = p.R * nt * 4;                        // f32x4 elements of [R][N]
        hipLaunchKernelGGL(resid_combine_kernel, dim3((n4 + 255) / 256), dim3(256), 0, s, (const float*)pol.sk_scratch, p.part_stride, a.x_out, n4);
        CTTS_HIP_CHECK(hipGetLastError());
        return 0;
    }
    if (shape == 4) hipLaunchKernelGGL((prefill_split_gemm_pp_kernel<EPI, 4>), dim3(nt / 16, rb), dim3(512), 2 * SPB_STAGE, s, p, a);
    else if (shape == 3) hipLaunchKernelGGL((prefill_split_gemm_pp_kernel<EPI, 3>), dim3(nt / 12, rb), dim3(512), 2 * SPB_STAGE, s, p, a);
    else if (small) hipLaunchKernelGGL((prefill_split_gemm_kernel<EPI, 4, 2>), g64, dim3(256), 4 * SP_STAGE / 2, s, p, a);
    else if (b_old <= pol.ring4_blocks) hipLaunchKernelGGL((prefill_split_gemm_kernel<EPI, 4>), dim3(nt / 8, (p.R + 127) / 128), dim3(256), 4 * SP_STAGE, s, p, a);      // at most one block per CU anyway
    else hipLaunchKernelGGL((prefill_split_gemm_kernel<EPI>), dim3(nt / 8, (p.R + 127) / 128), dim3(256), SP_RING * SP_STAGE, s, p, a);
    CTTS_HIP_CHECK(hipGetLastError());
    return 0;
}

// W / X: head and tail images; the operand buffers must cover whole 256-row blocks (gpt_engine.hip allocates PASS_ROWS + 256 rows, PASS_ROWS a multiple of 256).
// pol: block shape / K slicing policy (kernels.h SplitGemmPolicy; options "prefill_pp_blocks", "prefill_splitk_rows").
int launch_prefill_split_gemm(int epi, const GemmArgs& a, const void* Wsplit, const void* Xhi, const void* Xlo, void* act_hi, void* act_lo,
                              float scale, const SplitGemmPolicy& pol, hipStream_t s) {
    SplitGemm p;
    p.Whi = (const half_t*)Wsplit; p.Wlo = (const half_t*)Wsplit + 512; p.Xhi = (const half_t*)Xhi; p.Xlo = (const half_t*)Xlo;
    p.ktiles = a.K / 32; p.kt_per = p.ktiles; p.part_stride = 0; p.R = a.R; p.scale = scale; p.act_hi = (half_t*)act_hi; p.act_lo = (half_t*)act_lo;
    if ((a.n_row_tiles % 8) != 0 || p.ktiles < 2) { ctts_set_error("prefill_split_gemm: %d n tiles / K = %d not supported", a.n_row_tiles, a.K); return 1; }
    if (epi == EPI_QKV) return sp_launch<EPI_QKV>(p, a, pol, s);
    if (epi == EPI_SWIGLU) return sp_launch<EPI_SWIGLU>(p, a, pol, s);
    if (epi == EPI_RESID) return sp_launch<EPI_RESID>(p, a, pol, s);
    ctts_set_error("prefill_split_gemm: unsupported epilogue %d", epi);
    return 1;
}
