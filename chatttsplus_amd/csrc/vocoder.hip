// DVAE decoder + Vocos synthesis on gfx950 (fp32, v_mfma_f32_16x16x4_f32).
//
// Reference arithmetic:
//   DVAE.forward(mode="decode")     chattts_plus/models/dvae.py:272-291
//   DVAEDecoder.forward             dvae.py:161-168 (conv_in k3 -> GELU -> k3; 12 ConvNeXt; 1x1 conv_out)
//   ConvNeXtBlock.forward           dvae.py:48-63  (dw-conv k7 [dilation 2] -> LN -> Linear -> GELU -> Linear -> gamma -> +res)
//   Vocos (third-party vocos 0.1.0, parity unpinned): VocosBackbone + ISTFTHead, called at pipeline:303
//
// Everything is kept channels-last [frame][channel] so that
//   * the DVAE input "view(1,2,384,n).permute(0,2,3,1).flatten(2)" (dvae.py:277-283) is the identity:
//     hidden[n][768] row-major == frames[2n][384] row-major;
//   * a k-tap Conv1d is a GEMM whose A rows are overlapping windows of the (zero guarded) input:
//     A[m][tap*Cin + ci] = x[m + tap - pad][ci]  ->  row stride Cin, K = taps*Cin, no im2col copy;
//   * pointwise convs / Linear layers are plain GEMMs with fused bias / GELU / gamma+residual epilogues;
//   * the ISTFT is a real DFT-as-GEMM against a precomputed windowed basis followed by a gather-form
//     overlap-add (no atomics) with torch.istft's window-envelope normalisation (center=True).
// These stages are MFMA-bound but <3 % of the wall time of a 512-token utterance.
#include <math.h>
#include <string.h>

#include <map>
#include <string>
#include <vector>

#include "../../include/ctts_hip.h"
#include "common.h"

enum { EP_NONE = 0, EP_BIAS = 1, EP_BIAS_GELU = 2, EP_GAMMA_RESID = 3, EP_SCALE_T = 4 };

struct GemmF32Args {
    const float* A; int lda;     // activations [M][lda] (row windows may overlap: conv-as-GEMM)
    const float* W; int ldw;     // weights [Npad][ldw], k-contiguous
    float* C; int ldc;
    int M, N, K;                 // K multiple of 16; A rows readable up to roundup(M,64); W rows up to roundup(N,64)
    const float* bias;           // [N]
    const float* gamma;          // [N]   EP_GAMMA_RESID
    const float* resid; int ldr; // [M][ldr]
    const float* scale;          // [N]   EP_SCALE_T (C is written transposed: C[n*ldc + m])
};

__device__ inline float gelu_erf(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }

// block = 4 waves (2x2), wave tile 32x32, block tile 64x64; fragments are loaded straight from global/L2
template <int EPI>
__global__ __launch_bounds__(256) void gemm_f32_kernel(const GemmF32Args a) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int m0 = blockIdx.y * 64 + (wave >> 1) * 32, n0 = blockIdx.x * 64 + (wave & 1) * 32;
    const float* Ap = a.A + (size_t)(m0 + (lane & 15)) * a.lda + 4 * (lane >> 4);
    const float* Wp = a.W + (size_t)(n0 + (lane & 15)) * a.ldw + 4 * (lane >> 4);
    const size_t a16 = (size_t)16 * a.lda, w16 = (size_t)16 * a.ldw;
    f32x4 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int k0 = 0; k0 < a.K; k0 += 16) {
        const f32x4 a0 = *(const f32x4*)(Ap + k0), a1 = *(const f32x4*)(Ap + a16 + k0);
        const f32x4 b0 = *(const f32x4*)(Wp + k0), b1 = *(const f32x4*)(Wp + w16 + k0);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            acc[0][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[j], b0[j], acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[j], b1[j], acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[j], b0[j], acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[j], b1[j], acc[1][1], 0, 0, 0);
        }
    }
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) {
            const int n = n0 + ni * 16 + (lane & 15);
            if (n >= a.N) continue;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = m0 + mi * 16 + (lane >> 4) * 4 + r;
                if (m >= a.M) continue;
                float v = acc[mi][ni][r];
                if (EPI == EP_BIAS || EPI == EP_BIAS_GELU || EPI == EP_GAMMA_RESID) v += a.bias[n];
                if (EPI == EP_BIAS_GELU) v = gelu_erf(v);
                if (EPI == EP_GAMMA_RESID) v = __fadd_rn(__fmul_rn(v, a.gamma[n]), a.resid[(size_t)m * a.ldr + n]);
                if (EPI == EP_SCALE_T) { a.C[(size_t)n * a.ldc + m] = v * a.scale[n]; continue; }
                a.C[(size_t)m * a.ldc + n] = v;
            }
        }
}

static int launch_gemm_f32(int epi, const GemmF32Args& a, hipStream_t s) {
    if (a.K % 16 || a.lda % 4 || a.ldw % 4) { ctts_set_error("gemm_f32: K=%d lda=%d ldw=%d alignment", a.K, a.lda, a.ldw); return 1; }
    dim3 grid((a.N + 63) / 64, (a.M + 63) / 64), block(256);
    switch (epi) {
        case EP_NONE: hipLaunchKernelGGL(gemm_f32_kernel<EP_NONE>, grid, block, 0, s, a); break;
        case EP_BIAS: hipLaunchKernelGGL(gemm_f32_kernel<EP_BIAS>, grid, block, 0, s, a); break;
        case EP_BIAS_GELU: hipLaunchKernelGGL(gemm_f32_kernel<EP_BIAS_GELU>, grid, block, 0, s, a); break;
        case EP_GAMMA_RESID: hipLaunchKernelGGL(gemm_f32_kernel<EP_GAMMA_RESID>, grid, block, 0, s, a); break;
        case EP_SCALE_T: hipLaunchKernelGGL(gemm_f32_kernel<EP_SCALE_T>, grid, block, 0, s, a); break;
        default: ctts_set_error("gemm_f32: bad epilogue"); return 1;
    }
    CTTS_HIP_CHECK(hipGetLastError());
    return 0;
}

// depthwise conv (k7, dilation d, zero padding) fused with LayerNorm over C=512; one wave per frame.
// taps == 0 -> plain LayerNorm of the input row (Vocos' post-embed / final norms).
__global__ __launch_bounds__(256) void dwconv_ln_kernel(const float* x, float* out, const float* w /*[C][7]*/, const float* b,
                                                        const float* lnw, const float* lnb, int T, int C, int dil, int taps) {
    const int lane = threadIdx.x & 63, t = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (t >= T) return;
    float v[8];
    const int c0 = lane * 8;                                  // C == 512: 8 channels per lane
    if (taps == 0) {
        const f32x4 p = *(const f32x4*)(x + (size_t)t * C + c0), q = *(const f32x4*)(x + (size_t)t * C + c0 + 4);
        v[0] = p[0]; v[1] = p[1]; v[2] = p[2]; v[3] = p[3]; v[4] = q[0]; v[5] = q[1]; v[6] = q[2]; v[7] = q[3];
    } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = b[c0 + j];
        for (int k = 0; k < taps; ++k) {
            const int tt = t + (k - taps / 2) * dil;
            if (tt < 0 || tt >= T) continue;
            const f32x4 p = *(const f32x4*)(x + (size_t)tt * C + c0), q = *(const f32x4*)(x + (size_t)tt * C + c0 + 4);
            const float xv[8] = {p[0], p[1], p[2], p[3], q[0], q[1], q[2], q[3]};
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] += w[(c0 + j) * taps + k] * xv[j];
        }
    }
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) s += v[j];
    const float mean = wave_sum(s) / (float)C;
    float ss = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) { const float d = v[j] - mean; ss += d * d; }
    const float rstd = 1.0f / sqrtf(wave_sum(ss) / (float)C + 1e-6f);
    float o[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = (v[j] - mean) * rstd * lnw[c0 + j] + lnb[c0 + j];
    *(f32x4*)(out + (size_t)t * C + c0) = (f32x4){o[0], o[1], o[2], o[3]};
    *(f32x4*)(out + (size_t)t * C + c0 + 4) = (f32x4){o[4], o[5], o[6], o[7]};
}

// mel [n_mels][F] (API layout) -> channels-last, channel-padded, zero-guarded [F + 2*guard][ldc]
__global__ void mel_to_cl_kernel(const float* mel, float* out, int n_mels, int F, int ldc, int guard) {
    const int f = blockIdx.x, c = threadIdx.x;
    if (c < ldc) out[(size_t)(f + guard) * ldc + c] = (c < n_mels) ? mel[(size_t)c * F + f] : 0.f;
}

// ISTFTHead: x [F][ldx] = Linear output (mag | phase)  ->  spec [F][lds] = (mag cos p | mag sin p | 0 pad)
__global__ void head_spec_kernel(const float* x, float* spec, int F, int ldx, int lds, int nb /*513*/) {
    const int f = blockIdx.x;
    for (int k = threadIdx.x; k < lds; k += blockDim.x) {
        float v = 0.f;
        if (k < 2 * nb) {
            const int kk = (k < nb) ? k : k - nb;
            const float mag = fminf(expf(x[(size_t)f * ldx + kk]), 100.0f);      // clip(exp(mag), max=1e2)
            const float p = x[(size_t)f * ldx + nb + kk];
            v = (k < nb) ? mag * cosf(p) : mag * sinf(p);
        }
        spec[(size_t)f * lds + k] = v;
    }
}

// torch.istft(center=True): out[s] = sum_f frames[f][s + n/2 - hop f] / sum_f w^2[s + n/2 - hop f]
__global__ void overlap_add_kernel(const float* frames, const float* win, float* wav, int F, int n_fft, int hop) {
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    const int len = hop * (F - 1);
    if (s >= len) return;
    const int t = s + n_fft / 2;
    float acc = 0.f, env = 0.f;
    const int f_hi = min(F - 1, t / hop);
    for (int f = f_hi; f >= 0; --f) {
        const int o = t - f * hop;
        if (o >= n_fft) break;
        acc += frames[(size_t)f * n_fft + o];
        env += win[o] * win[o];
    }
    wav[s] = acc / env;
}

// ------------------------------------------------------------------------------------------------
struct ConvNext { float *dw_w, *dw_b, *ln_w, *ln_b, *w1, *b1, *w2, *b2, *gamma; };

struct ctts_voc {
    ctts_voc_cfg cfg;
    std::map<std::string, std::vector<float>> host;
    bool finalized = false;
    std::vector<void*> allocs;
    // DVAE
    float *ci0_w, *ci0_b, *ci2_w, *ci2_b, *co_w, *oc_w, *coef;
    std::vector<ConvNext> dblocks;
    // Vocos
    float *em_w, *em_b, *n0_w, *n0_b, *nf_w, *nf_b, *hd_w, *hd_b, *win, *basis;
    std::vector<ConvNext> vblocks;
    // workspaces
    float *in384, *b128, *y, *ln, *mid, *co384, *mcl, *hbuf, *spec, *frames;
    int Fp;
    int mel_ld, spec_ld, head_ld;
};

static int valloc(ctts_voc* h, float** p, size_t n) {
    CTTS_HIP_CHECK(hipMalloc((void**)p, n * 4));
    CTTS_HIP_CHECK(hipMemset(*p, 0, n * 4));
    h->allocs.push_back(*p);
    return 0;
}
static int upload(ctts_voc* h, float** p, const std::vector<float>& v) {
    if (valloc(h, p, v.size())) return 1;
    CTTS_HIP_CHECK(hipMemcpy(*p, v.data(), v.size() * 4, hipMemcpyHostToDevice));
    return 0;
}
static const std::vector<float>* vneed(ctts_voc* h, const std::string& k, size_t n) {
    auto it = h->host.find(k);
    if (it == h->host.end()) { ctts_set_error("missing weight %s", k.c_str()); return nullptr; }
    if (it->second.size() != n) { ctts_set_error("weight %s has %zu elements, expected %zu", k.c_str(), it->second.size(), n); return nullptr; }
    return &it->second;
}
// torch Conv1d weight [Cout][Cin][taps] -> GEMM weight [Npad][taps*ld_in], k = tap*ld_in + ci, zero padded
static std::vector<float> conv_to_gemm(const std::vector<float>& w, int cout, int cin, int taps, int ld_in, int npad) {
    std::vector<float> o((size_t)npad * taps * ld_in, 0.f);
    for (int co = 0; co < cout; ++co)
        for (int ci = 0; ci < cin; ++ci)
            for (int t = 0; t < taps; ++t) o[(size_t)co * taps * ld_in + (size_t)t * ld_in + ci] = w[((size_t)co * cin + ci) * taps + t];
    return o;
}
static std::vector<float> pad_rows(const std::vector<float>& w, int rows, int cols, int npad) {
    std::vector<float> o((size_t)npad * cols, 0.f);
    memcpy(o.data(), w.data(), (size_t)rows * cols * 4);
    return o;
}
static inline int r64(int n) { return (n + 63) / 64 * 64; }

static int load_convnext(ctts_voc* h, const std::string& p, int dim, int inter, ConvNext* cb) {
    const std::vector<float>*dw = vneed(h, p + "dwconv.weight", (size_t)dim * 7), *db = vneed(h, p + "dwconv.bias", dim),
                            *lw = vneed(h, p + "norm.weight", dim), *lb = vneed(h, p + "norm.bias", dim),
                            *w1 = vneed(h, p + "pwconv1.weight", (size_t)inter * dim), *b1 = vneed(h, p + "pwconv1.bias", inter),
                            *w2 = vneed(h, p + "pwconv2.weight", (size_t)dim * inter), *b2 = vneed(h, p + "pwconv2.bias", dim),
                            *g = vneed(h, p + "gamma", dim);
    if (!dw || !db || !lw || !lb || !w1 || !b1 || !w2 || !b2 || !g) return 1;
    return upload(h, &cb->dw_w, *dw) || upload(h, &cb->dw_b, *db) || upload(h, &cb->ln_w, *lw) || upload(h, &cb->ln_b, *lb) ||
           upload(h, &cb->w1, *w1) || upload(h, &cb->b1, *b1) || upload(h, &cb->w2, *w2) || upload(h, &cb->b2, *b2) ||
           upload(h, &cb->gamma, *g);
}

extern "C" int ctts_voc_create(const ctts_voc_cfg* c, ctts_voc** out) {
    if (!c || !out) { ctts_set_error("null argument"); return 1; }
    if (c->dvae_hidden != 512 || c->vocos_dim != 512 || c->dvae_idim % 64 || c->dvae_bn % 64 || c->n_fft != 1024 || c->hop != 256 ||
        c->vocos_inter % 64 || c->n_mels > 112 || c->max_frames < 2) {
        ctts_set_error("unsupported vocoder configuration");
        return 1;
    }
    ctts_voc* h = new ctts_voc();
    h->cfg = *c;
    *out = h;
    return 0;
}
extern "C" void ctts_voc_destroy(ctts_voc* h) {
    if (!h) return;
    for (void* p : h->allocs) (void)hipFree(p);
    delete h;
}
extern "C" int ctts_voc_set_weight(ctts_voc* h, const char* name, const float* data, size_t numel) {
    if (!h || !name || !data) { ctts_set_error("null argument"); return 1; }
    if (h->finalized) { ctts_set_error("weights already finalized"); return 1; }
    h->host[std::string(name)].assign(data, data + numel);
    return 0;
}

extern "C" int ctts_voc_finalize(ctts_voc* h) {
    if (!h) { ctts_set_error("null handle"); return 1; }
    if (h->finalized) return 0;
    const ctts_voc_cfg& c = h->cfg;
    const int ID = c.dvae_idim, BN = c.dvae_bn, HD = c.dvae_hidden, NM = c.n_mels, VD = c.vocos_dim, VI = c.vocos_inter;
    const int NB = c.n_fft / 2 + 1;
    h->mel_ld = 112; h->spec_ld = (2 * NB + 15) / 16 * 16; h->head_ld = r64(2 * NB);
    // ---- DVAE
    const std::vector<float>*w = vneed(h, "dvae.decoder.conv_in.0.weight", (size_t)BN * ID * 3), *b = vneed(h, "dvae.decoder.conv_in.0.bias", BN);
    if (!w || !b || upload(h, &h->ci0_w, conv_to_gemm(*w, BN, ID, 3, ID, r64(BN))) || upload(h, &h->ci0_b, *b)) return 1;
    w = vneed(h, "dvae.decoder.conv_in.2.weight", (size_t)HD * BN * 3); b = vneed(h, "dvae.decoder.conv_in.2.bias", HD);
    if (!w || !b || upload(h, &h->ci2_w, conv_to_gemm(*w, HD, BN, 3, BN, r64(HD))) || upload(h, &h->ci2_b, *b)) return 1;
    h->dblocks.resize(c.dvae_layers);
    for (int i = 0; i < c.dvae_layers; ++i)
        if (load_convnext(h, "dvae.decoder.decoder_block." + std::to_string(i) + ".", HD, HD * 4, &h->dblocks[i])) return 1;
    w = vneed(h, "dvae.decoder.conv_out.weight", (size_t)ID * HD);
    if (!w || upload(h, &h->co_w, pad_rows(*w, ID, HD, r64(ID)))) return 1;
    w = vneed(h, "dvae.out_conv.weight", (size_t)NM * ID * 3);
    if (!w || upload(h, &h->oc_w, conv_to_gemm(*w, NM, ID, 3, ID, r64(NM)))) return 1;
    w = vneed(h, "dvae.coef", NM);
    if (!w || upload(h, &h->coef, *w)) return 1;
    // ---- Vocos
    w = vneed(h, "vocos.backbone.embed.weight", (size_t)VD * NM * 7); b = vneed(h, "vocos.backbone.embed.bias", VD);
    if (!w || !b || upload(h, &h->em_w, conv_to_gemm(*w, VD, NM, 7, h->mel_ld, r64(VD))) || upload(h, &h->em_b, *b)) return 1;
    w = vneed(h, "vocos.backbone.norm.weight", VD); b = vneed(h, "vocos.backbone.norm.bias", VD);
    if (!w || !b || upload(h, &h->n0_w, *w) || upload(h, &h->n0_b, *b)) return 1;
    h->vblocks.resize(c.vocos_layers);
    for (int i = 0; i < c.vocos_layers; ++i)
        if (load_convnext(h, "vocos.backbone.convnext." + std::to_string(i) + ".", VD, VI, &h->vblocks[i])) return 1;
    w = vneed(h, "vocos.backbone.final_layer_norm.weight", VD); b = vneed(h, "vocos.backbone.final_layer_norm.bias", VD);
    if (!w || !b || upload(h, &h->nf_w, *w) || upload(h, &h->nf_b, *b)) return 1;
    w = vneed(h, "vocos.head.out.weight", (size_t)2 * NB * VD); b = vneed(h, "vocos.head.out.bias", 2 * NB);
    if (!w || !b || upload(h, &h->hd_w, pad_rows(*w, 2 * NB, VD, h->head_ld)) || upload(h, &h->hd_b, *b)) return 1;
    const std::vector<float>* win = vneed(h, "vocos.head.istft.window", c.n_fft);
    if (!win || upload(h, &h->win, *win)) return 1;
    {   // windowed inverse real-DFT basis: frame[n] = w[n]/N * sum_k c_k (Re_k cos(2 pi k n/N) - Im_k sin(2 pi k n/N))
        const int N = c.n_fft, LD = h->spec_ld;
        std::vector<float> B((size_t)N * LD, 0.f);
        for (int n = 0; n < N; ++n)
            for (int k = 0; k < NB; ++k) {
                const double ck = (k == 0 || k == N / 2) ? 1.0 : 2.0;
                const double ang = 2.0 * M_PI * (double)((long long)k * n % N) / N;
                B[(size_t)n * LD + k] = (float)((*win)[n] * ck * cos(ang) / N);
                B[(size_t)n * LD + NB + k] = (float)(-(double)(*win)[n] * ck * sin(ang) / N);
            }
        if (upload(h, &h->basis, B)) return 1;
    }
    // ---- workspaces (rows padded so every 64-row GEMM tile stays in bounds)
    const int Fp = r64(c.max_frames) + 64;
    h->Fp = Fp;
    const int maxmid = (HD * 4 > VI) ? HD * 4 : VI;
    if (valloc(h, &h->in384, (size_t)Fp * ID) || valloc(h, &h->b128, (size_t)Fp * BN) || valloc(h, &h->y, (size_t)Fp * HD) ||
        valloc(h, &h->ln, (size_t)Fp * HD) || valloc(h, &h->mid, (size_t)Fp * maxmid) || valloc(h, &h->co384, (size_t)Fp * ID) ||
        valloc(h, &h->mcl, (size_t)Fp * h->mel_ld) || valloc(h, &h->hbuf, (size_t)Fp * h->head_ld) ||
        valloc(h, &h->spec, (size_t)Fp * h->spec_ld) || valloc(h, &h->frames, (size_t)Fp * c.n_fft))
        return 1;
    h->host.clear();
    h->finalized = true;
    return 0;
}

static int run_convnext(ctts_voc* h, const ConvNext& cb, int F, int dim, int inter, int dil, hipStream_t s) {
    hipLaunchKernelGGL(dwconv_ln_kernel, dim3((F + 3) / 4), dim3(256), 0, s, h->y, h->ln, cb.dw_w, cb.dw_b, cb.ln_w, cb.ln_b, F, dim, dil, 7);
    CTTS_HIP_CHECK(hipGetLastError());
    GemmF32Args g = {};
    g.A = h->ln; g.lda = dim; g.W = cb.w1; g.ldw = dim; g.C = h->mid; g.ldc = inter; g.M = F; g.N = inter; g.K = dim; g.bias = cb.b1;
    if (launch_gemm_f32(EP_BIAS_GELU, g, s)) return 1;
    GemmF32Args g2 = {};
    g2.A = h->mid; g2.lda = inter; g2.W = cb.w2; g2.ldw = inter; g2.C = h->y; g2.ldc = dim; g2.M = F; g2.N = dim; g2.K = inter;
    g2.bias = cb.b2; g2.gamma = cb.gamma; g2.resid = h->y; g2.ldr = dim;
    return launch_gemm_f32(EP_GAMMA_RESID, g2, s);
}

extern "C" int ctts_dvae_decode(ctts_voc* h, const float* hidden, int n_tokens, float* mel, void* stream) {
    if (!h || !h->finalized || !hidden || !mel) { ctts_set_error("dvae_decode: bad argument"); return 1; }
    const ctts_voc_cfg& c = h->cfg;
    const int F = 2 * n_tokens, ID = c.dvae_idim, BN = c.dvae_bn, HD = c.dvae_hidden;
    if (n_tokens < 1 || F > c.max_frames) { ctts_set_error("dvae_decode: %d frames exceed max_frames=%d", F, c.max_frames); return 1; }
    hipStream_t s = (hipStream_t)stream;
    // hidden [n][2*ID] row-major IS frames [2n][ID] row-major (dvae.py:277-283); row 0 and row F+1 are conv guards
    CTTS_HIP_CHECK(hipMemsetAsync(h->in384, 0, (size_t)ID * 4, s));
    CTTS_HIP_CHECK(hipMemcpyAsync(h->in384 + ID, hidden, (size_t)F * ID * 4, hipMemcpyDeviceToDevice, s));
    CTTS_HIP_CHECK(hipMemsetAsync(h->in384 + (size_t)(F + 1) * ID, 0, (size_t)ID * 4, s));
    GemmF32Args g = {};
    g.A = h->in384; g.lda = ID; g.W = h->ci0_w; g.ldw = 3 * ID; g.C = h->b128 + BN; g.ldc = BN; g.M = F; g.N = BN; g.K = 3 * ID; g.bias = h->ci0_b;
    CTTS_HIP_CHECK(hipMemsetAsync(h->b128, 0, (size_t)BN * 4, s));
    CTTS_HIP_CHECK(hipMemsetAsync(h->b128 + (size_t)(F + 1) * BN, 0, (size_t)BN * 4, s));
    if (launch_gemm_f32(EP_BIAS_GELU, g, s)) return 1;                               // conv_in.0 + GELU (dvae.py:143-145)
    GemmF32Args g2 = {};
    g2.A = h->b128; g2.lda = BN; g2.W = h->ci2_w; g2.ldw = 3 * BN; g2.C = h->y; g2.ldc = HD; g2.M = F; g2.N = HD; g2.K = 3 * BN; g2.bias = h->ci2_b;
    if (launch_gemm_f32(EP_BIAS, g2, s)) return 1;                                    // conv_in.2 (dvae.py:146)
    for (int i = 0; i < c.dvae_layers; ++i)
        if (run_convnext(h, h->dblocks[i], F, HD, HD * 4, 2, s)) return 1;            // dvae.py:147-158,164-165
    GemmF32Args g3 = {};
    g3.A = h->y; g3.lda = HD; g3.W = h->co_w; g3.ldw = HD; g3.C = h->co384 + ID; g3.ldc = ID; g3.M = F; g3.N = ID; g3.K = HD;
    CTTS_HIP_CHECK(hipMemsetAsync(h->co384, 0, (size_t)ID * 4, s));
    CTTS_HIP_CHECK(hipMemsetAsync(h->co384 + (size_t)(F + 1) * ID, 0, (size_t)ID * 4, s));
    if (launch_gemm_f32(EP_NONE, g3, s)) return 1;                                    // conv_out 1x1, no bias (dvae.py:159,167)
    GemmF32Args g4 = {};
    g4.A = h->co384; g4.lda = ID; g4.W = h->oc_w; g4.ldw = 3 * ID; g4.C = mel; g4.ldc = F; g4.M = F; g4.N = c.n_mels; g4.K = 3 * ID;
    g4.scale = h->coef;
    return launch_gemm_f32(EP_SCALE_T, g4, s);                                        // out_conv k3 * coef -> [100][F] (dvae.py:285-291)
}

extern "C" int ctts_vocos_decode(ctts_voc* h, const float* mel, int F, float* wav, void* stream) {
    if (!h || !h->finalized || !mel || !wav) { ctts_set_error("vocos_decode: bad argument"); return 1; }
    const ctts_voc_cfg& c = h->cfg;
    if (F < 2 || F > c.max_frames) { ctts_set_error("vocos_decode: %d frames exceed max_frames=%d", F, c.max_frames); return 1; }
    hipStream_t s = (hipStream_t)stream;
    const int VD = c.vocos_dim, LD = h->mel_ld, NB = c.n_fft / 2 + 1;
    CTTS_HIP_CHECK(hipMemsetAsync(h->mcl, 0, (size_t)3 * LD * 4, s));
    CTTS_HIP_CHECK(hipMemsetAsync(h->mcl + (size_t)(F + 3) * LD, 0, (size_t)3 * LD * 4, s));
    hipLaunchKernelGGL(mel_to_cl_kernel, dim3(F), dim3(128), 0, s, mel, h->mcl, c.n_mels, F, LD, 3);
    CTTS_HIP_CHECK(hipGetLastError());
    GemmF32Args g = {};
    g.A = h->mcl; g.lda = LD; g.W = h->em_w; g.ldw = 7 * LD; g.C = h->mid; g.ldc = VD; g.M = F; g.N = VD; g.K = 7 * LD; g.bias = h->em_b;
    if (launch_gemm_f32(EP_BIAS, g, s)) return 1;                                     // embed conv k7 p3
    hipLaunchKernelGGL(dwconv_ln_kernel, dim3((F + 3) / 4), dim3(256), 0, s, h->mid, h->y, nullptr, nullptr, h->n0_w, h->n0_b, F, VD, 1, 0);
    CTTS_HIP_CHECK(hipGetLastError());
    for (int i = 0; i < c.vocos_layers; ++i)
        if (run_convnext(h, h->vblocks[i], F, VD, c.vocos_inter, 1, s)) return 1;
    hipLaunchKernelGGL(dwconv_ln_kernel, dim3((F + 3) / 4), dim3(256), 0, s, h->y, h->ln, nullptr, nullptr, h->nf_w, h->nf_b, F, VD, 1, 0);
    CTTS_HIP_CHECK(hipGetLastError());
    GemmF32Args g2 = {};
    g2.A = h->ln; g2.lda = VD; g2.W = h->hd_w; g2.ldw = VD; g2.C = h->hbuf; g2.ldc = h->head_ld; g2.M = F; g2.N = 2 * NB; g2.K = VD; g2.bias = h->hd_b;
    if (launch_gemm_f32(EP_BIAS, g2, s)) return 1;                                    // ISTFTHead.out
    hipLaunchKernelGGL(head_spec_kernel, dim3(F), dim3(256), 0, s, h->hbuf, h->spec, F, h->head_ld, h->spec_ld, NB);
    CTTS_HIP_CHECK(hipGetLastError());
    GemmF32Args g3 = {};
    g3.A = h->spec; g3.lda = h->spec_ld; g3.W = h->basis; g3.ldw = h->spec_ld; g3.C = h->frames; g3.ldc = c.n_fft; g3.M = F; g3.N = c.n_fft; g3.K = h->spec_ld;
    if (launch_gemm_f32(EP_NONE, g3, s)) return 1;                                    // windowed irfft as GEMM
    const int len = c.hop * (F - 1);
    hipLaunchKernelGGL(overlap_add_kernel, dim3((len + 255) / 256), dim3(256), 0, s, h->frames, h->win, wav, F, c.n_fft, c.hop);
    CTTS_HIP_CHECK(hipGetLastError());
    return 0;
}
