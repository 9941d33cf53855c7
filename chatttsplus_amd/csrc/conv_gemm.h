// Shared by vocoder.hip (DVAE decoder + Vocos) and encoder.hip (zero-shot DVAE encoder): the batched fp32 MFMA GEMM with
// fused epilogues that every Conv1d / Linear of these stacks is lowered to, the depthwise-conv + LayerNorm kernel, and the
// host-side weight re-layout helpers.
#pragma once
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include <vector>

#include "../../include/ctts_hip.h"
#include "common.h"

enum { EP_NONE = 0, EP_BIAS = 1, EP_BIAS_GELU = 2, EP_GAMMA_RESID = 3, EP_SCALE_T = 4, EP_SCALE = 5, EP_LOGCLIP_DIV = 6 };

// Every kernel below is batched over utterances with grid.z (or grid.y): utterance z owns its own zero-guarded region
// of each workspace (batch stride s*), its frame count comes from a device table Ms[z].  One launch sequence serves the
// whole batch: the reference's per-utterance loop (pipeline:298-304) left the chip ~3/4 idle and paid ~1.8 ms of fixed
// latency (serial K loops, 70 launches) per utterance -- 71 ms for 32 x 272 tokens vs 24 ms batched.
struct GemmF32Args {
    const float* A; int lda; long sA;   // activations [M][lda] (row windows may overlap: conv-as-GEMM); batch stride in floats
    const float* W; int ldw;            // weights [Npad][ldw], k-contiguous (shared by the batch)
    float* C; int ldc; long sC;
    int M, N, K;                        // M = max rows over the batch (grid.y); K multiple of 16
    const int* Ms;                      // device table of rows per utterance (null: M)
    const float* bias;                  // [N]
    const float* gamma;                 // [N]   EP_GAMMA_RESID
    const float* resid; int ldr; long sR;
    const float* scale;                 // [N]   EP_SCALE / EP_SCALE_T
    float* const* Cptrs;                // EP_SCALE_T: per-utterance output base (written transposed C[n*M_z + m]); null -> C
    const half_t *Whi, *Wlo;            // head / tail fp16 images of VOC_WSCALE * W, same [Npad][ldw] layout (null: the fp32 MFMA kernel)
};
#define VOC_WSCALE 256.0f

__device__ inline float gelu_erf(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }

// block = 4 waves (2x2); wave tile (16 MI) x (16 NJ): 32x32 (block 64x64) by default; the 64x64 wave tile (block 128x128,
// half the L2 bytes per flop) is kept as a diagnostic variant (CTTS_VOC_TILE=128) -- it measured slower, see launch_gemm_f32.
// Fragments are loaded straight from global/L2.  Every output element accumulates its K range in
// the same order whatever the tiling, so results do not depend on M, on the tile shape or on the batch (decode_window relies
// on this).
template <int EPI, int MI, int NJ>
__global__ __launch_bounds__(256) void gemm_f32_kernel(const GemmF32Args a) {
    const int z = blockIdx.z;
    const int M = a.Ms ? a.Ms[z] : a.M;
    if ((int)blockIdx.y * (32 * MI) >= M) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int m0 = blockIdx.y * (32 * MI) + (wave >> 1) * (16 * MI), n0 = blockIdx.x * (32 * NJ) + (wave & 1) * (16 * NJ);
    const float* Ap = a.A + (size_t)z * a.sA + (size_t)(m0 + (lane & 15)) * a.lda + 4 * (lane >> 4);
    const float* Wp = a.W + (size_t)(n0 + (lane & 15)) * a.ldw + 4 * (lane >> 4);
    const size_t a16 = (size_t)16 * a.lda, w16 = (size_t)16 * a.ldw;
    f32x4 acc[MI][NJ];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    // software-pipelined K loop (two register sets, order pinned with sched_barrier -- hipcc otherwise rotates the loop
    // back into load -> wait -> MFMA): the fragments of the next 16-deep step are in flight during the MFMAs of the current one.
#define GEMM_LOAD(FA, FB, KO)                                                                             \
    _Pragma("unroll") for (int i = 0; i < MI; ++i) FA[i] = *(const f32x4*)(Ap + i * a16 + (KO));          \
    _Pragma("unroll") for (int j = 0; j < NJ; ++j) FB[j] = *(const f32x4*)(Wp + j * w16 + (KO));
#define GEMM_STEP(FA, FB)                                                                                 \
    _Pragma("unroll") for (int e = 0; e < 4; ++e)                                                         \
    _Pragma("unroll") for (int i = 0; i < MI; ++i)                                                        \
    _Pragma("unroll") for (int j = 0; j < NJ; ++j)                                                        \
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(FA[i][e], FB[j][e], acc[i][j], 0, 0, 0);
    f32x4 fa0[MI], fb0[NJ], fa1[MI], fb1[NJ];
    GEMM_LOAD(fa0, fb0, 0)
    for (int k0 = 0; k0 < a.K; k0 += 32) {
        const int k1 = (k0 + 16 < a.K) ? k0 + 16 : k0;
        GEMM_LOAD(fa1, fb1, k1)
        __builtin_amdgcn_sched_barrier(0);
        GEMM_STEP(fa0, fb0)
        __builtin_amdgcn_sched_barrier(0);
        if (k0 + 16 >= a.K) break;
        const int k2 = (k0 + 32 < a.K) ? k0 + 32 : k0;
        GEMM_LOAD(fa0, fb0, k2)
        __builtin_amdgcn_sched_barrier(0);
        GEMM_STEP(fa1, fb1)
        __builtin_amdgcn_sched_barrier(0);
    }
#undef GEMM_STEP
#undef GEMM_LOAD
    float* Cb = (EPI == EP_SCALE_T && a.Cptrs) ? a.Cptrs[z] : a.C + (size_t)z * a.sC;
    const float* Rb = (EPI == EP_GAMMA_RESID) ? a.resid + (size_t)z * a.sR : nullptr;
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < NJ; ++ni) {
            const int n = n0 + ni * 16 + (lane & 15);
            if (n >= a.N) continue;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = m0 + mi * 16 + (lane >> 4) * 4 + r;
                if (m >= M) continue;
                float v = acc[mi][ni][r];
                if (EPI == EP_BIAS || EPI == EP_BIAS_GELU || EPI == EP_GAMMA_RESID) v += a.bias[n];
                if (EPI == EP_BIAS_GELU) v = gelu_erf(v);
                if (EPI == EP_GAMMA_RESID) v = __fadd_rn(__fmul_rn(v, a.gamma[n]), Rb[(size_t)m * a.ldr + n]);
                if (EPI == EP_SCALE_T) { Cb[(size_t)n * M + m] = v * a.scale[n]; continue; }     // mel [n_mels][F_z]
                if (EPI == EP_SCALE) v *= a.scale[n];
                if (EPI == EP_LOGCLIP_DIV) v = logf(fmaxf(v, 1e-5f)) / a.scale[n];      // log(clip(mel, 1e-5)) / coef (dvae.py:196-198,264-266)
                Cb[(size_t)m * a.ldc + n] = v;
            }
        }
}

// The same GEMM on the fp16 pipes with fp32-level accuracy (round 3; the idea of prefill_split.hip): operands split into an 11-bit head and an
// 11-bit tail, C = (Ah.Wh + Ah.Wl + Al.Wh) / VOC_WSCALE -- 3 v_mfma_f32_16x16x32_f16 (48 pipe cycles per 32-deep step and 16 x 16 tile) instead of
// 8 v_mfma_f32_16x16x4_f32 (256 cycles).  The fp32 kernel above sits at 37 % of the fp32 MFMA peak whatever the tile shape, the staging (LDS or
// straight from L2) or the prefetch depth (2 / 3 / 4 / 6 register sets: 23.6 / 23.8 / 24.0 / 24.8 ms per 32 x 272 tokens): the pipe itself is the limit.
// Weights are split once at load time (scaled by 256 so that the tails stay normal fp16 numbers); activation fragments -- 8 consecutive k of one row
// per lane, the A layout of the 32-deep MFMA -- are split in registers.  K % 16 == 0: the upper half of a last, half-filled step is zeroed.
// Every output element still accumulates its K range in the same order whatever M, the tiling or the batch.
template <int EPI, int MI, int NJ>
__global__ __launch_bounds__(256) void gemm_split_kernel(const GemmF32Args a) {
    const int z = blockIdx.z;
    const int M = a.Ms ? a.Ms[z] : a.M;
    if ((int)blockIdx.y * (32 * MI) >= M) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int m0 = blockIdx.y * (32 * MI) + (wave >> 1) * (16 * MI), n0 = blockIdx.x * (32 * NJ) + (wave & 1) * (16 * NJ);
    const int kq = lane >> 4;                                         // this lane's 8-wide k group of a 32-deep step
    const float* Ap = a.A + (size_t)z * a.sA + (size_t)(m0 + (lane & 15)) * a.lda + 8 * kq;
    const size_t woff = (size_t)(n0 + (lane & 15)) * a.ldw + 8 * kq;
    const half_t *Wh = a.Whi + woff, *Wl = a.Wlo + woff;
    const size_t a16 = (size_t)16 * a.lda, w16 = (size_t)16 * a.ldw;
    f32x4 acc[MI][NJ];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    struct Frags { f32x4 a0[MI], a1[MI]; half8 wh[NJ], wl[NJ]; };
    auto load = [&](Frags& f, int k0) {
        const bool on = (k0 + 8 * kq) < a.K;                           // K % 32 == 16: lanes of the upper half step hold zeros
#pragma unroll
        for (int i = 0; i < MI; ++i) {
            f.a0[i] = on ? *(const f32x4*)(Ap + i * a16 + k0) : (f32x4){0.f, 0.f, 0.f, 0.f};
            f.a1[i] = on ? *(const f32x4*)(Ap + i * a16 + k0 + 4) : (f32x4){0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            f.wh[j] = on ? *(const half8*)(Wh + j * w16 + k0) : (half8){0, 0, 0, 0, 0, 0, 0, 0};
            f.wl[j] = on ? *(const half8*)(Wl + j * w16 + k0) : (half8){0, 0, 0, 0, 0, 0, 0, 0};
        }
    };
    auto step = [&](const Frags& f) {
#pragma unroll
        for (int i = 0; i < MI; ++i) {
            half8 ah, al;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float v = (e < 4) ? f.a0[i][e] : f.a1[i][e - 4];
                const float c = fminf(fmaxf(v, -65504.f), 65504.f);
                ah[e] = (half_t)c;
                al[e] = (half_t)(c - (float)ah[e]);
            }
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, f.wh[j], acc[i][j], 0, 0, 0);
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, f.wl[j], acc[i][j], 0, 0, 0);
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, f.wh[j], acc[i][j], 0, 0, 0);
            }
        }
    };
    Frags f0, f1;
    load(f0, 0);
    for (int k0 = 0; k0 < a.K; k0 += 64) {
        if (k0 + 32 < a.K) load(f1, k0 + 32);
        __builtin_amdgcn_sched_barrier(0);
        step(f0);
        __builtin_amdgcn_sched_barrier(0);
        if (k0 + 32 >= a.K) break;
        if (k0 + 64 < a.K) load(f0, k0 + 64);
        __builtin_amdgcn_sched_barrier(0);
        step(f1);
        __builtin_amdgcn_sched_barrier(0);
    }
    float* Cb = (EPI == EP_SCALE_T && a.Cptrs) ? a.Cptrs[z] : a.C + (size_t)z * a.sC;
    const float* Rb = (EPI == EP_GAMMA_RESID) ? a.resid + (size_t)z * a.sR : nullptr;
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < NJ; ++ni) {
            const int n = n0 + ni * 16 + (lane & 15);
            if (n >= a.N) continue;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = m0 + mi * 16 + (lane >> 4) * 4 + r;
                if (m >= M) continue;
                float v = acc[mi][ni][r] * (1.0f / VOC_WSCALE);
                if (EPI == EP_BIAS || EPI == EP_BIAS_GELU || EPI == EP_GAMMA_RESID) v += a.bias[n];
                if (EPI == EP_BIAS_GELU) v = gelu_erf(v);
                if (EPI == EP_GAMMA_RESID) v = __fadd_rn(__fmul_rn(v, a.gamma[n]), Rb[(size_t)m * a.ldr + n]);
                if (EPI == EP_SCALE_T) { Cb[(size_t)n * M + m] = v * a.scale[n]; continue; }     // mel [n_mels][F_z]
                if (EPI == EP_SCALE) v *= a.scale[n];
                if (EPI == EP_LOGCLIP_DIV) v = logf(fmaxf(v, 1e-5f)) / a.scale[n];
                Cb[(size_t)m * a.ldc + n] = v;
            }
        }
}

template <int EPI>
static int launch_gemm_f32_epi(const GemmF32Args& a, int nb, hipStream_t s, bool big) {
    if (a.Whi != nullptr) {
        hipLaunchKernelGGL((gemm_split_kernel<EPI, 2, 2>), dim3((a.N + 63) / 64, (a.M + 63) / 64, nb), dim3(256), 0, s, a);
        CTTS_HIP_CHECK(hipGetLastError());
        return 0;
    }
    if (big) hipLaunchKernelGGL((gemm_f32_kernel<EPI, 4, 4>), dim3((a.N + 127) / 128, (a.M + 127) / 128, nb), dim3(256), 0, s, a);
    else hipLaunchKernelGGL((gemm_f32_kernel<EPI, 2, 2>), dim3((a.N + 63) / 64, (a.M + 63) / 64, nb), dim3(256), 0, s, a);
    CTTS_HIP_CHECK(hipGetLastError());
    return 0;
}

// `row_pad`: rows the caller guarantees to be readable past M in A (the weights are padded to a multiple of 64 rows): the
// 128-row tiles need row_pad >= 127 and N % 128 == 0.
static int launch_gemm_f32(int epi, const GemmF32Args& a, int nb, hipStream_t s, int row_pad = 63) {
    if (a.K % 16 || a.lda % 4 || a.ldw % 4 || (a.Whi != nullptr && a.ldw % 8)) { ctts_set_error("gemm_f32: K=%d lda=%d ldw=%d alignment", a.K, a.lda, a.ldw); return 1; }
    static const int force = diag_env("CTTS_VOC_TILE") ? atoi(diag_env("CTTS_VOC_TILE")) : 0;      // 64 / 128: diagnostic override
    // measured (32 utterances x 272 tokens): the pointwise-conv GEMMs already run at ~110 TFLOP/s (70 % of the fp32 MFMA peak);
    // 128x128 register tiles 28.8 ms vs 64x64 25.6 ms, an LDS-staged 128x128 variant 25.7 ms (bit-identical, no gain: removed)
    // measured (32 utterances x 272 tokens): 128x128 tiles 28.8 ms vs 64x64 25.6 ms -- the fragment loads (16 rows x 64 B per
    // wave instruction) are what limits this kernel, not the L2 bytes per flop, so the small tile stays the default
    const bool big = (force == 128) && row_pad >= 127 && a.N % 128 == 0;
    switch (epi) {
        case EP_NONE: return launch_gemm_f32_epi<EP_NONE>(a, nb, s, big);
        case EP_BIAS: return launch_gemm_f32_epi<EP_BIAS>(a, nb, s, big);
        case EP_BIAS_GELU: return launch_gemm_f32_epi<EP_BIAS_GELU>(a, nb, s, big);
        case EP_GAMMA_RESID: return launch_gemm_f32_epi<EP_GAMMA_RESID>(a, nb, s, big);
        case EP_SCALE_T: return launch_gemm_f32_epi<EP_SCALE_T>(a, nb, s, false);
        case EP_SCALE: return launch_gemm_f32_epi<EP_SCALE>(a, nb, s, false);
        case EP_LOGCLIP_DIV: return launch_gemm_f32_epi<EP_LOGCLIP_DIV>(a, nb, s, false);
        default: ctts_set_error("gemm_f32: bad epilogue"); return 1;
    }
}

// depthwise conv (k7, dilation d, zero padding) fused with LayerNorm over C = 64 * CPL channels (CPL = 8: the 512-wide
// decoder / Vocos blocks, CPL = 4: the 256-wide encoder blocks); one wave per frame, grid.y = utterance.
// taps == 0 -> plain LayerNorm of the input row (Vocos' post-embed / final norms).  sx / so: batch strides of in / out.
template <int CPL>
__global__ __launch_bounds__(256) void dwconv_ln_kernel(const float* x, float* out, const float* w /*[C][7]*/, const float* b,
                                                        const float* lnw, const float* lnb, const int* Ts, long sx, long so, int C, int dil, int taps) {
    static_assert(CPL == 4 || CPL == 8, "4 or 8 channels per lane");
    const int lane = threadIdx.x & 63, t = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int T = Ts[blockIdx.y];
    if (t >= T) return;
    x += (size_t)blockIdx.y * sx; out += (size_t)blockIdx.y * so;
    float v[CPL];
    const int c0 = lane * CPL;
    if (taps == 0) {
#pragma unroll
        for (int q4 = 0; q4 < CPL / 4; ++q4) {
            const f32x4 p = *(const f32x4*)(x + (size_t)t * C + c0 + 4 * q4);
            v[4 * q4] = p[0]; v[4 * q4 + 1] = p[1]; v[4 * q4 + 2] = p[2]; v[4 * q4 + 3] = p[3];
        }
    } else {
#pragma unroll
        for (int j = 0; j < CPL; ++j) v[j] = b[c0 + j];
        for (int k = 0; k < taps; ++k) {
            const int tt = t + (k - taps / 2) * dil;
            if (tt < 0 || tt >= T) continue;
            float xv[CPL];
#pragma unroll
            for (int q4 = 0; q4 < CPL / 4; ++q4) {
                const f32x4 p = *(const f32x4*)(x + (size_t)tt * C + c0 + 4 * q4);
                xv[4 * q4] = p[0]; xv[4 * q4 + 1] = p[1]; xv[4 * q4 + 2] = p[2]; xv[4 * q4 + 3] = p[3];
            }
#pragma unroll
            for (int j = 0; j < CPL; ++j) v[j] += w[(c0 + j) * taps + k] * xv[j];
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
#pragma unroll
    for (int q4 = 0; q4 < CPL / 4; ++q4) {
        f32x4 o;
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = (v[4 * q4 + j] - mean) * rstd * lnw[c0 + 4 * q4 + j] + lnb[c0 + 4 * q4 + j];
        *(f32x4*)(out + (size_t)t * C + c0 + 4 * q4) = o;
    }
}

// torch Conv1d weight [Cout][Cin][taps] -> GEMM weight [Npad][taps*ld_in], k = tap*ld_in + ci, zero padded
static inline std::vector<float> conv_to_gemm(const std::vector<float>& w, int cout, int cin, int taps, int ld_in, int npad) {
    std::vector<float> o((size_t)npad * taps * ld_in, 0.f);
    for (int co = 0; co < cout; ++co)
        for (int ci = 0; ci < cin; ++ci)
            for (int t = 0; t < taps; ++t) o[(size_t)co * taps * ld_in + (size_t)t * ld_in + ci] = w[((size_t)co * cin + ci) * taps + t];
    return o;
}
static inline std::vector<float> pad_rows(const std::vector<float>& w, int rows, int cols, int npad) {
    std::vector<float> o((size_t)npad * cols, 0.f);
    memcpy(o.data(), w.data(), (size_t)rows * cols * 4);
    return o;
}
static inline int r64(int n) { return (n + 63) / 64 * 64; }
// head / tail fp16 images of VOC_WSCALE * w (gemm_split_kernel)
static inline void split_weights(const std::vector<float>& w, std::vector<half_t>& hi, std::vector<half_t>& lo) {
    hi.resize(w.size()); lo.resize(w.size());
    for (size_t i = 0; i < w.size(); ++i) {
        const float v = VOC_WSCALE * w[i];
        const half_t h = (half_t)v;
        hi[i] = h;
        lo[i] = (half_t)(v - (float)h);
    }
}

