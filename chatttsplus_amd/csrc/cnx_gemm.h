// The two pointwise GEMMs of a ConvNeXt block (dvae.py:48-63 / Vocos' block: Linear -> GELU -> Linear -> gamma -> + residual; 90 % of the
// vocoder's flops) as LDS-tiled 3-term split GEMMs on the fp16 MFMA pipes, fp32-accurate (the scheme of prefill_split.hip):
//
//   dwconv_ln_split_kernel   depthwise conv + LayerNorm, output written as head / tail fp16 FRAGMENT IMAGES [utterance][16-frame group][k tile][lane][16 B]
//   cnx_gemm_kernel<PW1>     mid = GELU(ln . W1^T + b1)             -> head / tail fragment images of mid
//   cnx_gemm_kernel<PW2>     y  += gamma * (mid . W2^T + b2)        -> the fp32 residual stream [frame][dim]
//
// Why (round 3): the register-fragment GEMM of conv_gemm.h (64 x 64 block tiles, operands straight from L2) runs the same 23 ms per 32 x 272
// tokens whether its inner product is 8 fp32 MFMAs or 3 fp16 MFMAs, with 2 or 6 register sets in flight, staged through LDS or not: it is bound by
// the operand traffic of small tiles (every 64-row A tile and 64-row W tile is pulled into the CU twice per block).  Here a block owns 128 frames x 128
// features, its k-tile stages are straight LDS-DMA copies of 1-KiB fragments (no address arithmetic per element, no conversion in the loop), and the
// MFMA work per byte is 4x the old kernel's.
// Every output element accumulates its K range in the same order whatever the frame's place in its group, tile or batch (decode_window relies on it).
#pragma once
#include "conv_gemm.h"

#define CNX_WSCALE 256.0f
// LDS ring of k-tile stages.  2 stages = 64 KB = TWO blocks per CU (134 VGPRs): the second block's MFMAs fill the barrier / LDS-read bubbles of the
// first -- 32 x 272 tokens 9.3 -> 7.7 ms against 3 stages (96 KB, one block per CU, one barrier per stage instead of two)
#ifndef CNX_RING
#define CNX_RING 2
#endif
#define CNX_STAGE (32 * 1024)
// A launch of at most this many 128 x 128 blocks runs as four times as many 64 x 64 blocks (round 6).  Vocoder call, ms, 128 x 128 only -> with the two limits below:
// 1 x 48 tokens (a streaming window) 1.71 -> 0.85, 1 x 512: 1.79 -> 0.99, 2 x 272: 1.83 -> 1.09, 4 x 272: 2.05 -> 1.65, 8 x 272: 2.65 -> 2.6, 16 x 272 and up: unchanged.
// The first pointwise GEMM (K = 512: 16 k-tiles) gains up to ~640 blocks, the second (K = 2048: 64 k-tiles per block, N = 512) only below ~128 -- at 160 blocks it
// lost 2 %, at 320 7 % (profiles/r06_ab_prefill_pp_gemm.jsonl).
#ifndef CNX_SMALL_BLOCKS_PW1
#define CNX_SMALL_BLOCKS_PW1 768
#endif
#ifndef CNX_SMALL_BLOCKS_PW2
#define CNX_SMALL_BLOCKS_PW2 128
#endif
enum { CNX_PW1 = 0, CNX_PW2 = 1 };

struct CnxGemm {
    const half_t *Whi, *Wlo;     // weight fragment images [n tile (16 features)][k tile][lane][8] of CNX_WSCALE * W
    const half_t *Xhi, *Xlo;     // activation fragment images [utterance][16-frame group][k tile][lane][8]
    long sX;                     // halfs between utterances in X
    int ktiles;                  // K / 32
    int N;                       // output features (multiple of 128)
    const int* Ms;               // frames per utterance (device table)
    const float* bias;           // [N]
    // PW1: output images
    half_t *Ohi, *Olo; long sO; int ktiles_out;       // = N / 32
    // PW2: y[frame][N] += gamma * (acc + bias)
    const float* gamma; float* y; long sY;
};

__device__ inline void cnx_split4(const f32x4 v, half4& hi, half4& lo) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const float c = fminf(fmaxf(v[j], -65504.f), 65504.f);
        hi[j] = (half_t)c;
        lo[j] = (half_t)(c - (float)hi[j]);
    }
}

// WT = n tiles and frame groups per wave: 4 (128 x 128 blocks) or 2 (64 x 64 blocks on a ring of four 16 KB stages: four times the blocks for the small batches of
// a single request or a streaming window, whose 128 x 128 grids -- 20 to 80 blocks -- leave most CUs idle while a block walks up to 64 k-tiles; round 6, as
// prefill_split.hip's short passes).  Every output element accumulates its K range in the same order in both shapes.
template <int EPI, int RING = CNX_RING, int WT = 4>
__global__ __launch_bounds__(256, RING == 2 ? 2 : 1) void cnx_gemm_kernel(const CnxGemm p) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    constexpr int BT = 2 * WT, STAGE = 4 * BT * 1024;      // tiles (= frame groups) per block; bytes per stage
    const int z = blockIdx.z, M = p.Ms[z];
    if ((int)blockIdx.y * (16 * BT) >= M) return;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 1, wn = wave & 1;
    const int nt0 = blockIdx.x * BT, g0 = blockIdx.y * BT; // first n tile / first 16-frame group of the block
    const int ktiles = p.ktiles;
    const half_t *Xh = p.Xhi + (size_t)z * p.sX, *Xl = p.Xlo + (size_t)z * p.sX;
    // fragment f of a stage: BT Whi tiles, BT Wlo tiles, BT Xhi groups, BT Xlo groups
    auto src = [&](int f, int kt) -> const char* {
        const half_t* img = (f < BT) ? p.Whi : (f < 2 * BT) ? p.Wlo : (f < 3 * BT) ? Xh : Xl;
        const int unit = (f < 2 * BT) ? nt0 + (f & (BT - 1)) : g0 + (f & (BT - 1));
        return (const char*)img + ((size_t)unit * ktiles + kt) * 1024 + (unsigned)(lane * 16);
    };
    typedef __attribute__((address_space(1))) const void* gptr_t;
    typedef __attribute__((address_space(3))) void* lptr_t;
#define CNX_DMA(kt_, buf_)                                                                                              \
    _Pragma("unroll") for (int i = 0; i < BT; ++i) {                                                                     \
        const int f = wave + 4 * i;                                                                                       \
        __builtin_amdgcn_global_load_lds((gptr_t)src(f, (kt_)), (lptr_t)(lds + (buf_) * STAGE + f * 1024), 16, 0, 0);     \
    }
#define CNX_WAIT_BAR(n_) { __builtin_amdgcn_s_waitcnt(0x0070 | ((n_) & 15) | (((n_) >> 4) << 14)); __builtin_amdgcn_s_barrier(); }
    f32x4 acc[WT][WT];                                     // [n tile][frame group]
#pragma unroll
    for (int t = 0; t < WT; ++t)
#pragma unroll
        for (int g = 0; g < WT; ++g) acc[t][g] = (f32x4){0.f, 0.f, 0.f, 0.f};
    CNX_DMA(0, 0)
    CNX_DMA(1, 1)                                          // ktiles >= 2 (checked by the launcher)
    if (RING > 2 && ktiles > 2) CNX_DMA(2, 2)              // a ring of four keeps three stages in flight and needs one barrier per k-tile
    int cb = 0;
    for (int kt = 0; kt < ktiles; ++kt) {
        if (RING > 2 && kt + 2 < ktiles) CNX_WAIT_BAR(2 * BT) else if (kt + 1 < ktiles) CNX_WAIT_BAR(BT) else CNX_WAIT_BAR(0)      // (a wave copies BT fragments per stage)
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
            CNX_DMA(kt + AHEAD, nb)
        }
#pragma unroll
        for (int t = 0; t < WT; ++t)
#pragma unroll
            for (int g = 0; g < WT; ++g) {
                acc[t][g] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl[t], xh[g], acc[t][g], 0, 0, 0);
                acc[t][g] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[t], xl[g], acc[t][g], 0, 0, 0);
                acc[t][g] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[t], xh[g], acc[t][g], 0, 0, 0);
            }
        cb = (cb + 1 == RING) ? 0 : cb + 1;
    }
#undef CNX_DMA
#undef CNX_WAIT_BAR
    // C tile layout: lane = (iq = lane >> 4, n = lane & 15): frame n of the group, features 4 * iq + j of the n tile (j = register)
    const int iq = lane >> 4, nn = lane & 15;
#pragma unroll
    for (int g = 0; g < WT; ++g) {
        const int G = g0 + wr * WT + g, frame = G * 16 + nn;
        if (frame >= M) continue;
#pragma unroll
        for (int t = 0; t < WT; ++t) {
            const int f0 = (nt0 + wn * WT + t) * 16 + 4 * iq;         // first of this lane's 4 features
            const f32x4 b = *(const f32x4*)(p.bias + f0), c = acc[t][g];
            f32x4 v = {c[0] * (1.0f / CNX_WSCALE) + b[0], c[1] * (1.0f / CNX_WSCALE) + b[1], c[2] * (1.0f / CNX_WSCALE) + b[2], c[3] * (1.0f / CNX_WSCALE) + b[3]};
            if (EPI == CNX_PW1) {
                half4 h, l;
                cnx_split4((f32x4){gelu_erf(v[0]), gelu_erf(v[1]), gelu_erf(v[2]), gelu_erf(v[3])}, h, l);
                const size_t o = (size_t)z * p.sO + (size_t)G * p.ktiles_out * 64 * 8 + xfrag_index<half_t>(nn, f0, p.ktiles_out);
                *(half4*)(p.Ohi + o) = h;
                *(half4*)(p.Olo + o) = l;
            } else {
                float* yo = p.y + (size_t)z * p.sY + (size_t)frame * p.N + f0;
                const f32x4 r = *(const f32x4*)yo, gm = *(const f32x4*)(p.gamma + f0);
                *(f32x4*)yo = (f32x4){__fadd_rn(__fmul_rn(v[0], gm[0]), r[0]), __fadd_rn(__fmul_rn(v[1], gm[1]), r[1]),
                                      __fadd_rn(__fmul_rn(v[2], gm[2]), r[2]), __fadd_rn(__fmul_rn(v[3], gm[3]), r[3])};
            }
        }
    }
}

template <int EPI>
static int launch_cnx_gemm_t(const CnxGemm& p, int Fmax, int nb, hipStream_t s) {
    static bool configured = false;
    if (!configured) {
        CTTS_HIP_CHECK(hipFuncSetAttribute((const void*)cnx_gemm_kernel<EPI>, hipFuncAttributeMaxDynamicSharedMemorySize, CNX_RING * CNX_STAGE));
        CTTS_HIP_CHECK(hipFuncSetAttribute((const void*)cnx_gemm_kernel<EPI, 4, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 4 * CNX_STAGE / 2));
        configured = true;
    }
    if (p.N % 128 || p.ktiles < 2) { ctts_set_error("cnx_gemm: N=%d K=%d not supported", p.N, p.ktiles * 32); return 1; }
    const int blocks = (p.N / 128) * ((Fmax + 127) / 128) * nb;
    if (blocks <= (EPI == CNX_PW1 ? CNX_SMALL_BLOCKS_PW1 : CNX_SMALL_BLOCKS_PW2))          // grids that leave CUs idle: four times as many 64 x 64 blocks
        hipLaunchKernelGGL((cnx_gemm_kernel<EPI, 4, 2>), dim3(p.N / 64, (Fmax + 63) / 64, nb), dim3(256), 4 * CNX_STAGE / 2, s, p);
    else hipLaunchKernelGGL((cnx_gemm_kernel<EPI>), dim3(p.N / 128, (Fmax + 127) / 128, nb), dim3(256), CNX_RING * CNX_STAGE, s, p);
    CTTS_HIP_CHECK(hipGetLastError());
    return 0;
}
static int launch_cnx_gemm(int epi, const CnxGemm& p, int Fmax, int nb, hipStream_t s) {
    return epi == CNX_PW1 ? launch_cnx_gemm_t<CNX_PW1>(p, Fmax, nb, s) : launch_cnx_gemm_t<CNX_PW2>(p, Fmax, nb, s);
}

// depthwise conv (k7, dilation d, zero padding) + LayerNorm over C = 64 * CPL channels, one wave per frame (as dwconv_ln_kernel), output =
// head / tail fp16 fragment images [utterance][16-frame group][C / 32 k tiles][lane][8]: a lane's CPL consecutive channels are one (CPL = 8) or
// half of one (CPL = 4) 16-byte fragment piece
template <int CPL>
__global__ __launch_bounds__(256) void dwconv_ln_split_kernel(const float* x, half_t* ohi, half_t* olo, const float* w /*[taps][C]: transposed at load time*/, const float* b,
                                                              const float* lnw, const float* lnb, const int* Ts, long sx, long so, int C, int dil, int taps) {
    static_assert(CPL == 4 || CPL == 8, "4 or 8 channels per lane");
    const int lane = threadIdx.x & 63, t = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int T = Ts[blockIdx.y];
    if (t >= T) return;
    x += (size_t)blockIdx.y * sx;
    float v[CPL];
    const int c0 = lane * CPL;
#pragma unroll
    for (int j = 0; j < CPL; ++j) v[j] = b[c0 + j];
    for (int k = 0; k < taps; ++k) {
        const int tt = t + (k - taps / 2) * dil;
        if (tt < 0 || tt >= T) continue;
        // tap-major weights: a lane's CPL channels are contiguous (the [C][7] layout cost 7 * CPL strided loads per lane: 131 us per launch at
        // 17408 frames, a quarter of the vocoder once the GEMMs were off the fp32 pipe)
#pragma unroll
        for (int q4 = 0; q4 < CPL / 4; ++q4) {
            const f32x4 pv = *(const f32x4*)(x + (size_t)tt * C + c0 + 4 * q4), wv = *(const f32x4*)(w + (size_t)k * C + c0 + 4 * q4);
#pragma unroll
            for (int j = 0; j < 4; ++j) v[4 * q4 + j] += wv[j] * pv[j];
        }
    }
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < CPL; ++j) s += v[j];
    const float mean = wave_sum(s) / (float)C;
    float ss = 0.f;
#pragma unroll
    for (int j = 0; j < CPL; ++j) { const float d = v[j] - mean; ss += d * d; }
    const float rstd = 1.0f / sqrtf(wave_sum(ss) / (float)C + 1e-6f);
    const int KT = C / 32;
    const size_t base = (size_t)blockIdx.y * so + (size_t)(t >> 4) * KT * 64 * 8;
#pragma unroll
    for (int q4 = 0; q4 < CPL / 4; ++q4) {
        f32x4 o;
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = (v[4 * q4 + j] - mean) * rstd * lnw[c0 + 4 * q4 + j] + lnb[c0 + 4 * q4 + j];
        half4 h, l;
        cnx_split4(o, h, l);
        const size_t off = base + xfrag_index<half_t>(t & 15, c0 + 4 * q4, KT);
        *(half4*)(ohi + off) = h;
        *(half4*)(olo + off) = l;
    }
}

// torch Linear weight [N][K] -> head / tail fragment images [N / 16][K / 32][64][8] of CNX_WSCALE * W
static inline void cnx_pack_weights(const std::vector<float>& w, int N, int K, std::vector<half_t>& hi, std::vector<half_t>& lo) {
    const int ktiles = K / 32;
    hi.assign((size_t)N * K, (half_t)0.f); lo.assign((size_t)N * K, (half_t)0.f);
    for (int n = 0; n < N; ++n)
        for (int k = 0; k < K; ++k) {
            const size_t o = (((size_t)(n >> 4) * ktiles + (k >> 5)) * 64 + (n & 15) + 16 * ((k >> 3) & 3)) * 8 + (k & 7);
            const float v = CNX_WSCALE * w[(size_t)n * K + k];
            const half_t h = (half_t)v;
            hi[o] = h;
            lo[o] = (half_t)(v - (float)h);
        }
}
