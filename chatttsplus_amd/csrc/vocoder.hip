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

#include "conv_gemm.h"
#include "cnx_gemm.h"
#include "roctx_range.h"

// Stage the batch: copy each utterance's hidden rows ([n][768] == frames [2n][384], dvae.py:277-283) behind a zero guard
// row and zero the guard rows / channel padding of every conv input (grid = (Fmax + 6, nb), one row per block).
__global__ void prep_kernel(const float* const* hidden, const int* Fs, float* in384, float* b128, float* co384, float* mcl,
                            long s_in, long s_b, long s_mcl, int ID, int BN, int LD, int n_mels) {
    const int u = blockIdx.y, r = blockIdx.x, F = Fs[u], tid = threadIdx.x;
    if (r <= F + 1) {
        const bool guard = (r == 0 || r == F + 1);
        if (guard) {
            for (int c = tid; c < ID; c += blockDim.x) { in384[(size_t)u * s_in + (size_t)r * ID + c] = 0.f; co384[(size_t)u * s_in + (size_t)r * ID + c] = 0.f; }
            for (int c = tid; c < BN; c += blockDim.x) b128[(size_t)u * s_b + (size_t)r * BN + c] = 0.f;
        } else if (hidden != nullptr) {
            const float* src = hidden[u] + (size_t)(r - 1) * ID;
            for (int c = tid; c < ID; c += blockDim.x) in384[(size_t)u * s_in + (size_t)r * ID + c] = src[c];
        }
    }
    if (r < 3 || (r >= F + 3 && r < F + 6))
        for (int c = tid; c < LD; c += blockDim.x) mcl[(size_t)u * s_mcl + (size_t)r * LD + c] = 0.f;
    else if (r < F + 3)                                        // channel padding n_mels..LD-1 of the mel rows
        for (int c = n_mels + tid; c < LD; c += blockDim.x) mcl[(size_t)u * s_mcl + (size_t)r * LD + c] = 0.f;
}

// GFSQ._embed (dvae.py:85-96) -> GroupedResidualFSQ.get_output_from_indices (vector_quantize_pytorch 1.17.8, restated): frame 2t+g of
// utterance u = project_out_g( sum_r code(ids[t][g*R + r]) * (levels - 1)^-r ), code_d(i) = ((i / basis_d) % L_d - L_d/2) / (L_d/2);
// written behind the zero guard row of the decoder's input image exactly where prep_kernel copies the hidden rows in the other mode
// (the "view(1,2,C/2,n).permute(0,2,3,1).flatten(2)" of dvae.py:277-283 interleaves the two groups frame by frame).
struct VqCfg { int G, R; int levels[4]; };
__global__ void embed_codes_kernel(const int* const* ids, const int* Fs, float* in_img, long s_in, int ID, const float* po_w /*[G][ID][4]*/,
                                   const float* po_b /*[G][ID]*/, VqCfg vq) {
    const int u = blockIdx.y, f = blockIdx.x, F = Fs[u];
    if (f >= F) return;
    const int t = f / vq.G, g = f % vq.G;
    float lat[4] = {0.f, 0.f, 0.f, 0.f};
    for (int r = 0; r < vq.R; ++r) {
        int idx = ids[u][(size_t)t * vq.G * vq.R + g * vq.R + r], basis = 1;
#pragma unroll
        for (int d = 0; d < 4; ++d) {
            const int L = vq.levels[d], hw = L / 2;
            const float code = ((float)((idx / basis) % L) - (float)hw) / (float)hw;
            float sc = 1.f;
            for (int q = 0; q < r; ++q) sc /= (float)(L - 1);           // (levels - 1)^-r
            lat[d] += code * sc;
            basis *= L;
        }
    }
    float* dst = in_img + (size_t)u * s_in + (size_t)(f + 1) * ID;
    for (int c = threadIdx.x; c < ID; c += blockDim.x) {
        const float* w = po_w + ((size_t)g * ID + c) * 4;
        dst[c] = ((w[0] * lat[0] + w[1] * lat[1]) + (w[2] * lat[2] + w[3] * lat[3])) + po_b[(size_t)g * ID + c];
    }
}

// mel [n_mels][F] (API layout) -> channels-last rows guard..F+guard-1 of mcl (guards / padding are zeroed by prep_kernel)
__global__ void mel_to_cl_kernel(const float* mel, float* out, int n_mels, int F, int ldc, int guard) {
    const int f = blockIdx.x, c = threadIdx.x;
    if (c < n_mels) out[(size_t)(f + guard) * ldc + c] = mel[(size_t)c * F + f];
}

// ISTFTHead: x [F][ldx] = Linear output (mag | phase)  ->  spec [F][lds] = (mag cos p | mag sin p | 0 pad)
__global__ void head_spec_kernel(const float* x, float* spec, const int* Fs, long sx, long ss, int ldx, int lds, int nb /*513*/) {
    const int f = blockIdx.x, u = blockIdx.y;
    if (f >= Fs[u]) return;
    x += (size_t)u * sx; spec += (size_t)u * ss;
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
__global__ void overlap_add_kernel(const float* frames, const float* win, float* const* wavs, const int* Fs, long sf, int n_fft, int hop) {
    const int u = blockIdx.y, F = Fs[u];
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    const int len = hop * (F - 1);
    if (s >= len) return;
    frames += (size_t)u * sf;
    const int t = s + n_fft / 2;
    float acc = 0.f, env = 0.f;
    const int f_hi = min(F - 1, t / hop);
    for (int f = f_hi; f >= 0; --f) {
        const int o = t - f * hop;
        if (o >= n_fft) break;
        acc += frames[(size_t)f * n_fft + o];
        env += win[o] * win[o];
    }
    wavs[u][s] = acc / env;
}

// ------------------------------------------------------------------------------------------------
struct SplitW { half_t *hi = nullptr, *lo = nullptr; };       // head / tail fp16 images of a GEMM weight (conv_gemm.h gemm_split_kernel)
struct ConvNext { float *dw_w, *dw_b, *ln_w, *ln_b, *b1, *b2, *gamma; SplitW s1, s2; };      // s1 / s2: head / tail FRAGMENT images of pwconv1 / pwconv2 (cnx_gemm.h)

struct ctts_voc {
    ctts_voc_cfg cfg;
    std::map<std::string, std::vector<float>> host;
    bool finalized = false;
    std::vector<void*> allocs;
    // DVAE
    float *ci0_w, *ci0_b, *ci2_w, *ci2_b, *co_w, *oc_w, *coef;
    float *po_w = nullptr, *po_b = nullptr;       // quantiser project_out [G][idim][4] / [G][idim] (vq_groups > 0)
    SplitW s_ci0, s_ci2, s_co, s_oc, s_em, s_hd, s_basis;
    std::vector<ConvNext> dblocks;
    // Vocos
    float *em_w, *em_b, *n0_w, *n0_b, *nf_w, *nf_b, *hd_w, *hd_b, *win, *basis;
    std::vector<ConvNext> vblocks;
    // workspaces
    float *in384, *b128, *y, *ln, *mid, *co384, *mcl, *hbuf, *spec, *frames;
    half_t *ln_hi = nullptr, *ln_lo = nullptr, *mid_hi = nullptr, *mid_lo = nullptr;      // ConvNeXt operands as head / tail fragment images (cnx_gemm.h)
    int Fp;
    int mel_ld, spec_ld, head_ld, mid_ld;
    // per-call tables: frames per utterance, hidden / wav / mel pointers (device copy + ring of pinned staging slots)
    void* d_tab = nullptr; void* pin = nullptr;
    int* d_F = nullptr; const float* const* d_hid = nullptr; float* const* d_wav = nullptr; float* const* d_mel = nullptr;      // (d_hid doubles as the table of code-id pointers in the codes mode)
    std::vector<int> frames_host;
    hipEvent_t pin_ev[8]; bool pin_used[8] = {false, false, false, false, false, false, false, false}; int pin_next = 0;
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
// the head / tail images hold 256 * w as fp16: |w| >= 255.9 would become inf -- refuse such a checkpoint loudly instead of computing with inf
static int check_split_range(const std::vector<float>& v) {
    for (float x : v)
        if (!(fabsf(x) * VOC_WSCALE <= 65504.0f)) { ctts_set_error("vocoder weight %g is outside the +-255 range of the split fp16 weight images", (double)x); return 1; }
    return 0;
}
// a GEMM weight: the fp32 array (kept for reference / the fp32 kernel) + its head / tail fp16 images
static int upload_w(ctts_voc* h, float** p, SplitW* sw, const std::vector<float>& v) {
    if (check_split_range(v) || upload(h, p, v)) return 1;
    std::vector<half_t> hi, lo;
    split_weights(v, hi, lo);
    CTTS_HIP_CHECK(hipMalloc((void**)&sw->hi, hi.size() * 2)); h->allocs.push_back(sw->hi);
    CTTS_HIP_CHECK(hipMalloc((void**)&sw->lo, lo.size() * 2)); h->allocs.push_back(sw->lo);
    CTTS_HIP_CHECK(hipMemcpy(sw->hi, hi.data(), hi.size() * 2, hipMemcpyHostToDevice));
    CTTS_HIP_CHECK(hipMemcpy(sw->lo, lo.data(), lo.size() * 2, hipMemcpyHostToDevice));
    return 0;
}
// a ConvNeXt pointwise weight [N][K]: head / tail fragment images for cnx_gemm_kernel
static int upload_frag(ctts_voc* h, SplitW* sw, const std::vector<float>& w, int N, int K) {
    if (check_split_range(w)) return 1;
    std::vector<half_t> hi, lo;
    cnx_pack_weights(w, N, K, hi, lo);
    CTTS_HIP_CHECK(hipMalloc((void**)&sw->hi, hi.size() * 2)); h->allocs.push_back(sw->hi);
    CTTS_HIP_CHECK(hipMalloc((void**)&sw->lo, lo.size() * 2)); h->allocs.push_back(sw->lo);
    CTTS_HIP_CHECK(hipMemcpy(sw->hi, hi.data(), hi.size() * 2, hipMemcpyHostToDevice));
    CTTS_HIP_CHECK(hipMemcpy(sw->lo, lo.data(), lo.size() * 2, hipMemcpyHostToDevice));
    return 0;
}
static const std::vector<float>* vneed(ctts_voc* h, const std::string& k, size_t n) {
    auto it = h->host.find(k);
    if (it == h->host.end()) { ctts_set_error("missing weight %s", k.c_str()); return nullptr; }
    if (it->second.size() != n) { ctts_set_error("weight %s has %zu elements, expected %zu", k.c_str(), it->second.size(), n); return nullptr; }
    return &it->second;
}
static int load_convnext(ctts_voc* h, const std::string& p, int dim, int inter, ConvNext* cb) {
    const std::vector<float>*dw = vneed(h, p + "dwconv.weight", (size_t)dim * 7), *db = vneed(h, p + "dwconv.bias", dim),
                            *lw = vneed(h, p + "norm.weight", dim), *lb = vneed(h, p + "norm.bias", dim),
                            *w1 = vneed(h, p + "pwconv1.weight", (size_t)inter * dim), *b1 = vneed(h, p + "pwconv1.bias", inter),
                            *w2 = vneed(h, p + "pwconv2.weight", (size_t)dim * inter), *b2 = vneed(h, p + "pwconv2.bias", dim),
                            *g = vneed(h, p + "gamma", dim);
    if (!dw || !db || !lw || !lb || !w1 || !b1 || !w2 || !b2 || !g) return 1;
    std::vector<float> dwT((size_t)7 * dim);                               // depthwise weights tap-major [7][C] (dwconv_ln_split_kernel)
    for (int c = 0; c < dim; ++c) for (int k = 0; k < 7; ++k) dwT[(size_t)k * dim + c] = (*dw)[(size_t)c * 7 + k];
    return upload(h, &cb->dw_w, dwT) || upload(h, &cb->dw_b, *db) || upload(h, &cb->ln_w, *lw) || upload(h, &cb->ln_b, *lb) ||
           upload_frag(h, &cb->s1, *w1, inter, dim) || upload(h, &cb->b1, *b1) || upload_frag(h, &cb->s2, *w2, dim, inter) || upload(h, &cb->b2, *b2) ||
           upload(h, &cb->gamma, *g);
}

extern "C" int ctts_voc_create(const ctts_voc_cfg* c, ctts_voc** out) {
    if (!c || !out) { ctts_set_error("null argument"); return 1; }
    if ((c->dvae_hidden != 512 && c->dvae_hidden != 256) || c->vocos_dim != 512 || c->dvae_idim % 64 || c->dvae_bn % 64 || c->n_fft != 1024 || c->hop != 256 ||
        c->vocos_inter % 128 || c->n_mels > 112 || c->max_frames < 2 || c->max_batch < 1 || c->max_batch > 64) {
        ctts_set_error("unsupported vocoder configuration");
        return 1;
    }
    if (c->vq_groups < 0 || c->vq_groups > 4 || (c->vq_groups > 0 && (c->vq_residuals < 1 || c->vq_residuals > 4))) { ctts_set_error("unsupported quantiser configuration"); return 1; }
    for (int d = 0; d < 4 && c->vq_groups > 0; ++d)
        if (c->vq_levels[d] < 2 || c->vq_levels[d] > 64) { ctts_set_error("unsupported quantiser levels"); return 1; }
    ctts_voc* h = new ctts_voc();
    h->cfg = *c;
    *out = h;
    return 0;
}
extern "C" void ctts_voc_destroy(ctts_voc* h) {
    if (!h) return;
    for (void* p : h->allocs) (void)hipFree(p);
    if (h->d_tab) (void)hipFree(h->d_tab);
    if (h->pin) (void)hipHostFree(h->pin);
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
    if (!w || !b || upload_w(h, &h->ci0_w, &h->s_ci0, conv_to_gemm(*w, BN, ID, 3, ID, r64(BN))) || upload(h, &h->ci0_b, *b)) return 1;
    w = vneed(h, "dvae.decoder.conv_in.2.weight", (size_t)HD * BN * 3); b = vneed(h, "dvae.decoder.conv_in.2.bias", HD);
    if (!w || !b || upload_w(h, &h->ci2_w, &h->s_ci2, conv_to_gemm(*w, HD, BN, 3, BN, r64(HD))) || upload(h, &h->ci2_b, *b)) return 1;
    h->dblocks.resize(c.dvae_layers);
    for (int i = 0; i < c.dvae_layers; ++i)
        if (load_convnext(h, "dvae.decoder.decoder_block." + std::to_string(i) + ".", HD, HD * 4, &h->dblocks[i])) return 1;
    w = vneed(h, "dvae.decoder.conv_out.weight", (size_t)ID * HD);
    if (!w || upload_w(h, &h->co_w, &h->s_co, pad_rows(*w, ID, HD, r64(ID)))) return 1;
    w = vneed(h, "dvae.out_conv.weight", (size_t)NM * ID * 3);
    if (!w || upload_w(h, &h->oc_w, &h->s_oc, conv_to_gemm(*w, NM, ID, 3, ID, r64(NM)))) return 1;
    w = vneed(h, "dvae.coef", NM);
    if (!w || upload(h, &h->coef, *w)) return 1;
    if (c.vq_groups > 0) {      // GroupedResidualFSQ.rvqs[g].project_out: Linear(4 -> idim)
        std::vector<float> pw((size_t)c.vq_groups * ID * 4), pb((size_t)c.vq_groups * ID);
        for (int g = 0; g < c.vq_groups; ++g) {
            const std::string p = "dvae.vq_layer.quantizer.rvqs." + std::to_string(g) + ".project_out.";
            w = vneed(h, p + "weight", (size_t)ID * 4); b = vneed(h, p + "bias", ID);
            if (!w || !b) return 1;
            memcpy(&pw[(size_t)g * ID * 4], w->data(), (size_t)ID * 4 * 4);
            memcpy(&pb[(size_t)g * ID], b->data(), (size_t)ID * 4);
        }
        if (upload(h, &h->po_w, pw) || upload(h, &h->po_b, pb)) return 1;
    }
    // ---- Vocos
    w = vneed(h, "vocos.backbone.embed.weight", (size_t)VD * NM * 7); b = vneed(h, "vocos.backbone.embed.bias", VD);
    if (!w || !b || upload_w(h, &h->em_w, &h->s_em, conv_to_gemm(*w, VD, NM, 7, h->mel_ld, r64(VD))) || upload(h, &h->em_b, *b)) return 1;
    w = vneed(h, "vocos.backbone.norm.weight", VD); b = vneed(h, "vocos.backbone.norm.bias", VD);
    if (!w || !b || upload(h, &h->n0_w, *w) || upload(h, &h->n0_b, *b)) return 1;
    h->vblocks.resize(c.vocos_layers);
    for (int i = 0; i < c.vocos_layers; ++i)
        if (load_convnext(h, "vocos.backbone.convnext." + std::to_string(i) + ".", VD, VI, &h->vblocks[i])) return 1;
    w = vneed(h, "vocos.backbone.final_layer_norm.weight", VD); b = vneed(h, "vocos.backbone.final_layer_norm.bias", VD);
    if (!w || !b || upload(h, &h->nf_w, *w) || upload(h, &h->nf_b, *b)) return 1;
    w = vneed(h, "vocos.head.out.weight", (size_t)2 * NB * VD); b = vneed(h, "vocos.head.out.bias", 2 * NB);
    if (!w || !b || upload_w(h, &h->hd_w, &h->s_hd, pad_rows(*w, 2 * NB, VD, h->head_ld)) || upload(h, &h->hd_b, *b)) return 1;
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
        if (upload_w(h, &h->basis, &h->s_basis, B)) return 1;
    }
    // ---- workspaces: max_batch regions of Fp rows (rows padded so every 64-row GEMM tile stays in bounds)
    const int Fp = r64(c.max_frames) + 192;             // 128-row GEMM tiles + conv guard rows stay in bounds
    h->Fp = Fp;
    const size_t MB = c.max_batch;
    h->mid_ld = (HD * 4 > VI) ? HD * 4 : VI;
    const int YD = HD > VD ? HD : VD;                   // y / ln hold the DVAE stream (HD wide) and then the Vocos stream (VD wide)
    if (valloc(h, &h->in384, MB * Fp * ID) || valloc(h, &h->b128, MB * Fp * BN) || valloc(h, &h->y, MB * Fp * YD) ||
        valloc(h, &h->ln, MB * Fp * YD) || valloc(h, &h->mid, MB * Fp * h->mid_ld) || valloc(h, &h->co384, MB * Fp * ID) ||
        valloc(h, &h->mcl, MB * Fp * h->mel_ld) || valloc(h, &h->hbuf, MB * Fp * h->head_ld) ||
        valloc(h, &h->spec, MB * Fp * h->spec_ld) || valloc(h, &h->frames, MB * Fp * c.n_fft))
        return 1;
    {
        float *a = nullptr, *b2 = nullptr, *c2 = nullptr, *d2 = nullptr;                 // (valloc counts floats: half as many for fp16 images)
        if (valloc(h, &a, MB * Fp * YD / 2) || valloc(h, &b2, MB * Fp * YD / 2) || valloc(h, &c2, MB * Fp * h->mid_ld / 2) || valloc(h, &d2, MB * Fp * h->mid_ld / 2)) return 1;
        h->ln_hi = (half_t*)a; h->ln_lo = (half_t*)b2; h->mid_hi = (half_t*)c2; h->mid_lo = (half_t*)d2;
    }
    const size_t tab_bytes = MB * 4 + MB * 24;
    CTTS_HIP_CHECK(hipMalloc(&h->d_tab, tab_bytes));
    CTTS_HIP_CHECK(hipHostMalloc(&h->pin, tab_bytes * 8));
    for (int i = 0; i < 8; ++i) CTTS_HIP_CHECK(hipEventCreateWithFlags(&h->pin_ev[i], hipEventDisableTiming));
    h->d_F = (int*)h->d_tab;
    h->d_hid = (const float* const*)((char*)h->d_tab + MB * 4);
    h->d_wav = (float* const*)((char*)h->d_tab + MB * 4 + MB * 8);
    h->d_mel = (float* const*)((char*)h->d_tab + MB * 4 + MB * 16);
    h->frames_host.assign(MB, 0);
    h->host.clear();
    h->finalized = true;
    return 0;
}

static int run_convnext(ctts_voc* h, const ConvNext& cb, int nb, int Fmax, int dim, int inter, int dil, hipStream_t s) {
    // dw-conv + LayerNorm -> head / tail fragment images; Linear + GELU -> images of mid; Linear, gamma, + residual -> y (cnx_gemm.h)
    const long sd = (long)h->Fp * dim, sm = (long)h->Fp * inter;
    if (dim == 512) hipLaunchKernelGGL(dwconv_ln_split_kernel<8>, dim3((Fmax + 3) / 4, nb), dim3(256), 0, s, h->y, h->ln_hi, h->ln_lo, cb.dw_w, cb.dw_b, cb.ln_w, cb.ln_b, h->d_F, sd, sd, dim, dil, 7);
    else hipLaunchKernelGGL(dwconv_ln_split_kernel<4>, dim3((Fmax + 3) / 4, nb), dim3(256), 0, s, h->y, h->ln_hi, h->ln_lo, cb.dw_w, cb.dw_b, cb.ln_w, cb.ln_b, h->d_F, sd, sd, dim, dil, 7);     // DVAE_full decoder: 256 wide
    CTTS_HIP_CHECK(hipGetLastError());
    CnxGemm g = {};
    g.Whi = cb.s1.hi; g.Wlo = cb.s1.lo; g.Xhi = h->ln_hi; g.Xlo = h->ln_lo; g.sX = sd; g.ktiles = dim / 32; g.N = inter; g.Ms = h->d_F; g.bias = cb.b1;
    g.Ohi = h->mid_hi; g.Olo = h->mid_lo; g.sO = sm; g.ktiles_out = inter / 32;
    if (launch_cnx_gemm(CNX_PW1, g, Fmax, nb, s)) return 1;
    CnxGemm g2 = {};
    g2.Whi = cb.s2.hi; g2.Wlo = cb.s2.lo; g2.Xhi = h->mid_hi; g2.Xlo = h->mid_lo; g2.sX = sm; g2.ktiles = inter / 32; g2.N = dim; g2.Ms = h->d_F; g2.bias = cb.b2;
    g2.gamma = cb.gamma; g2.y = h->y; g2.sY = sd;
    return launch_cnx_gemm(CNX_PW2, g2, Fmax, nb, s);
}

// upload the per-call tables (frames, pointers): a ring of pinned staging slots guarded by events, no stream sync
static int set_tables(ctts_voc* h, const float* const* hidden, const int* frames, float* const* wavs, float* const* mels, int nb, hipStream_t s) {
    const size_t MB = h->cfg.max_batch;
    const size_t tab_bytes = MB * 4 + MB * 24;
    const int slot = h->pin_next++ % 8;
    if (h->pin_used[slot]) CTTS_HIP_CHECK(hipEventSynchronize(h->pin_ev[slot]));
    char* p = (char*)h->pin + slot * tab_bytes;
    memcpy(p, frames, nb * 4);
    if (hidden) memcpy(p + MB * 4, hidden, nb * sizeof(void*));
    if (wavs) memcpy(p + MB * 4 + MB * 8, wavs, nb * sizeof(void*));
    if (mels) memcpy(p + MB * 4 + MB * 16, mels, nb * sizeof(void*));
    CTTS_HIP_CHECK(hipMemcpyAsync(h->d_tab, p, tab_bytes, hipMemcpyHostToDevice, s));
    CTTS_HIP_CHECK(hipEventRecord(h->pin_ev[slot], s));
    h->pin_used[slot] = true;
    return 0;
}

static int stage(ctts_voc* h, bool with_hidden, int nb, int Fmax, hipStream_t s, bool codes = false) {
    const ctts_voc_cfg& c = h->cfg;
    hipLaunchKernelGGL(prep_kernel, dim3(Fmax + 6, nb), dim3(128), 0, s, (with_hidden && !codes) ? h->d_hid : nullptr, h->d_F, h->in384, h->b128, h->co384, h->mcl,
                       (long)h->Fp * c.dvae_idim, (long)h->Fp * c.dvae_bn, (long)h->Fp * h->mel_ld, c.dvae_idim, c.dvae_bn, h->mel_ld, c.n_mels);
    CTTS_HIP_CHECK(hipGetLastError());
    if (codes) {
        VqCfg vq; vq.G = c.vq_groups; vq.R = c.vq_residuals;
        for (int d = 0; d < 4; ++d) vq.levels[d] = c.vq_levels[d];
        hipLaunchKernelGGL(embed_codes_kernel, dim3(Fmax, nb), dim3(128), 0, s, (const int* const*)h->d_hid, h->d_F, h->in384, (long)h->Fp * c.dvae_idim,
                           c.dvae_idim, h->po_w, h->po_b, vq);
        CTTS_HIP_CHECK(hipGetLastError());
    }
    return 0;
}

// DVAE decoder chain for nb staged utterances; the mel goes to per-utterance [100][F] buffers (d_mel) or straight into the
// Vocos input image mcl (channels-last, 3 guard rows)
static int run_dvae(ctts_voc* h, int nb, int Fmax, bool to_mcl, hipStream_t s) {
    const ctts_voc_cfg& c = h->cfg;
    const int ID = c.dvae_idim, BN = c.dvae_bn, HD = c.dvae_hidden;
    const long Fp = h->Fp;
    GemmF32Args g = {};
    g.A = h->in384; g.lda = ID; g.sA = Fp * ID; g.W = h->ci0_w; g.ldw = 3 * ID; g.C = h->b128 + BN; g.ldc = BN; g.sC = Fp * BN;
    g.M = Fmax; g.Ms = h->d_F; g.N = BN; g.K = 3 * ID; g.bias = h->ci0_b; g.Whi = h->s_ci0.hi; g.Wlo = h->s_ci0.lo;
    if (launch_gemm_f32(EP_BIAS_GELU, g, nb, s, 127)) return 1;                               // conv_in.0 + GELU (dvae.py:143-145)
    GemmF32Args g2 = {};
    g2.A = h->b128; g2.lda = BN; g2.sA = Fp * BN; g2.W = h->ci2_w; g2.ldw = 3 * BN; g2.C = h->y; g2.ldc = HD; g2.sC = Fp * HD;
    g2.M = Fmax; g2.Ms = h->d_F; g2.N = HD; g2.K = 3 * BN; g2.bias = h->ci2_b; g2.Whi = h->s_ci2.hi; g2.Wlo = h->s_ci2.lo;
    if (launch_gemm_f32(EP_BIAS, g2, nb, s, 127)) return 1;                                    // conv_in.2 (dvae.py:146)
    for (int i = 0; i < c.dvae_layers; ++i)
        if (run_convnext(h, h->dblocks[i], nb, Fmax, HD, HD * 4, 2, s)) return 1;         // dvae.py:147-158,164-165
    GemmF32Args g3 = {};
    g3.A = h->y; g3.lda = HD; g3.sA = Fp * HD; g3.W = h->co_w; g3.ldw = HD; g3.C = h->co384 + ID; g3.ldc = ID; g3.sC = Fp * ID;
    g3.M = Fmax; g3.Ms = h->d_F; g3.N = ID; g3.K = HD; g3.Whi = h->s_co.hi; g3.Wlo = h->s_co.lo;
    if (launch_gemm_f32(EP_NONE, g3, nb, s, 127)) return 1;                                    // conv_out 1x1, no bias (dvae.py:159,167)
    GemmF32Args g4 = {};
    g4.A = h->co384; g4.lda = ID; g4.sA = Fp * ID; g4.W = h->oc_w; g4.ldw = 3 * ID; g4.M = Fmax; g4.Ms = h->d_F; g4.N = c.n_mels; g4.K = 3 * ID;
    g4.scale = h->coef; g4.Whi = h->s_oc.hi; g4.Wlo = h->s_oc.lo;
    if (to_mcl) {
        g4.C = h->mcl + 3 * h->mel_ld; g4.ldc = h->mel_ld; g4.sC = Fp * h->mel_ld;
        return launch_gemm_f32(EP_SCALE, g4, nb, s, 127);                                      // out_conv k3 * coef (dvae.py:285-291)
    }
    g4.Cptrs = h->d_mel;
    return launch_gemm_f32(EP_SCALE_T, g4, nb, s, 127);                                        //   ... -> [100][F] API layout
}

static int run_vocos(ctts_voc* h, int nb, int Fmax, hipStream_t s) {
    const ctts_voc_cfg& c = h->cfg;
    const int VD = c.vocos_dim, LD = h->mel_ld, NB = c.n_fft / 2 + 1;
    const long Fp = h->Fp;
    GemmF32Args g = {};
    g.A = h->mcl; g.lda = LD; g.sA = Fp * LD; g.W = h->em_w; g.ldw = 7 * LD; g.C = h->mid; g.ldc = VD; g.sC = Fp * h->mid_ld;
    g.M = Fmax; g.Ms = h->d_F; g.N = VD; g.K = 7 * LD; g.bias = h->em_b; g.Whi = h->s_em.hi; g.Wlo = h->s_em.lo;
    if (launch_gemm_f32(EP_BIAS, g, nb, s, 127)) return 1;                                     // embed conv k7 p3
    hipLaunchKernelGGL(dwconv_ln_kernel<8>, dim3((Fmax + 3) / 4, nb), dim3(256), 0, s, h->mid, h->y, nullptr, nullptr, h->n0_w, h->n0_b,
                       h->d_F, Fp * h->mid_ld, Fp * VD, VD, 1, 0);                        // post-embed LayerNorm -> residual stream
    CTTS_HIP_CHECK(hipGetLastError());
    for (int i = 0; i < c.vocos_layers; ++i)
        if (run_convnext(h, h->vblocks[i], nb, Fmax, VD, c.vocos_inter, 1, s)) return 1;
    hipLaunchKernelGGL(dwconv_ln_kernel<8>, dim3((Fmax + 3) / 4, nb), dim3(256), 0, s, h->y, h->ln, nullptr, nullptr, h->nf_w, h->nf_b,
                       h->d_F, Fp * VD, Fp * VD, VD, 1, 0);                               // final LayerNorm
    CTTS_HIP_CHECK(hipGetLastError());
    GemmF32Args g2 = {};
    g2.A = h->ln; g2.lda = VD; g2.sA = Fp * VD; g2.W = h->hd_w; g2.ldw = VD; g2.C = h->hbuf; g2.ldc = h->head_ld; g2.sC = Fp * h->head_ld;
    g2.M = Fmax; g2.Ms = h->d_F; g2.N = 2 * NB; g2.K = VD; g2.bias = h->hd_b; g2.Whi = h->s_hd.hi; g2.Wlo = h->s_hd.lo;
    if (launch_gemm_f32(EP_BIAS, g2, nb, s, 127)) return 1;                                    // ISTFTHead.out
    hipLaunchKernelGGL(head_spec_kernel, dim3(Fmax, nb), dim3(256), 0, s, h->hbuf, h->spec, h->d_F, Fp * h->head_ld, Fp * h->spec_ld, h->head_ld, h->spec_ld, NB);
    CTTS_HIP_CHECK(hipGetLastError());
    GemmF32Args g3 = {};
    g3.A = h->spec; g3.lda = h->spec_ld; g3.sA = Fp * h->spec_ld; g3.W = h->basis; g3.ldw = h->spec_ld; g3.C = h->frames; g3.ldc = c.n_fft;
    g3.sC = Fp * c.n_fft; g3.M = Fmax; g3.Ms = h->d_F; g3.N = c.n_fft; g3.K = h->spec_ld; g3.Whi = h->s_basis.hi; g3.Wlo = h->s_basis.lo;
    if (launch_gemm_f32(EP_NONE, g3, nb, s, 127)) return 1;                                    // windowed irfft as GEMM
    const int lenmax = c.hop * (Fmax - 1);
    hipLaunchKernelGGL(overlap_add_kernel, dim3((lenmax + 255) / 256, nb), dim3(256), 0, s, h->frames, h->win, h->d_wav, h->d_F, Fp * c.n_fft, c.n_fft, c.hop);
    CTTS_HIP_CHECK(hipGetLastError());
    return 0;
}

extern "C" int ctts_dvae_decode(ctts_voc* h, const float* hidden, int n_tokens, float* mel, void* stream) {
    if (!h || !h->finalized || !hidden || !mel) { ctts_set_error("dvae_decode: bad argument"); return 1; }
    const int F = 2 * n_tokens;
    if (n_tokens < 1 || F > h->cfg.max_frames) { ctts_set_error("dvae_decode: %d frames exceed max_frames=%d", F, h->cfg.max_frames); return 1; }
    CTTS_RANGE("ctts_dvae_decode");
    hipStream_t s = (hipStream_t)stream;
    if (set_tables(h, &hidden, &F, nullptr, &mel, 1, s) || stage(h, true, 1, F, s)) return 1;
    return run_dvae(h, 1, F, false, s);
}

static int codes_ready(ctts_voc* h, const char* who) {
    if (!h || !h->finalized) { ctts_set_error("%s: handle not finalized", who); return 1; }
    if (h->cfg.vq_groups < 1 || !h->po_w) { ctts_set_error("%s: this handle has no quantiser (create it with vq_groups > 0 from the DVAE_full checkpoint)", who); return 1; }
    if (h->cfg.vq_groups != 2) { ctts_set_error("%s: the decoder input interleaves exactly 2 groups (dvae.py:277-283)", who); return 1; }
    return 0;
}

extern "C" int ctts_dvae_decode_codes(ctts_voc* h, const int32_t* ids, int n_tokens, float* mel, void* stream) {
    if (codes_ready(h, "dvae_decode_codes")) return 1;
    if (!ids || !mel) { ctts_set_error("dvae_decode_codes: null argument"); return 1; }
    const int F = 2 * n_tokens;
    if (n_tokens < 1 || F > h->cfg.max_frames) { ctts_set_error("dvae_decode_codes: %d frames exceed max_frames=%d", F, h->cfg.max_frames); return 1; }
    CTTS_RANGE("ctts_dvae_decode_codes");
    hipStream_t s = (hipStream_t)stream;
    const float* idp = (const float*)ids;                   // pointer table slot shared with the hidden-row pointers
    if (set_tables(h, &idp, &F, nullptr, &mel, 1, s) || stage(h, true, 1, F, s, true)) return 1;
    return run_dvae(h, 1, F, false, s);
}

extern "C" int ctts_synth_batch_codes(ctts_voc* h, const int32_t* const* ids_ptrs, const int32_t* n_tokens, int B, float* const* wav_ptrs, void* stream) {
    if (codes_ready(h, "synth_batch_codes")) return 1;
    if (!ids_ptrs || !n_tokens || !wav_ptrs) { ctts_set_error("synth_batch_codes: null argument"); return 1; }
    if (B < 1 || B > h->cfg.max_batch) { ctts_set_error("synth_batch_codes: B=%d exceeds max_batch=%d", B, h->cfg.max_batch); return 1; }
    CTTS_RANGE("ctts_synth_batch_codes");
    int Fmax = 0;
    for (int u = 0; u < B; ++u) {
        const int F = 2 * n_tokens[u];
        if (n_tokens[u] < 1 || F > h->cfg.max_frames) { ctts_set_error("synth_batch_codes: utterance %d has %d frames (max_frames=%d)", u, F, h->cfg.max_frames); return 1; }
        h->frames_host[u] = F;
        Fmax = F > Fmax ? F : Fmax;
    }
    hipStream_t s = (hipStream_t)stream;
    if (set_tables(h, (const float* const*)ids_ptrs, h->frames_host.data(), wav_ptrs, nullptr, B, s) || stage(h, true, B, Fmax, s, true)) return 1;
    if (run_dvae(h, B, Fmax, true, s)) return 1;
    return run_vocos(h, B, Fmax, s);
}

extern "C" int ctts_vocos_decode(ctts_voc* h, const float* mel, int F, float* wav, void* stream) {
    if (!h || !h->finalized || !mel || !wav) { ctts_set_error("vocos_decode: bad argument"); return 1; }
    if (F < 2 || F > h->cfg.max_frames) { ctts_set_error("vocos_decode: %d frames exceed max_frames=%d", F, h->cfg.max_frames); return 1; }
    CTTS_RANGE("ctts_vocos_decode");
    hipStream_t s = (hipStream_t)stream;
    if (set_tables(h, nullptr, &F, &wav, nullptr, 1, s) || stage(h, false, 1, F, s)) return 1;
    hipLaunchKernelGGL(mel_to_cl_kernel, dim3(F), dim3(128), 0, s, mel, h->mcl, h->cfg.n_mels, F, h->mel_ld, 3);
    CTTS_HIP_CHECK(hipGetLastError());
    return run_vocos(h, 1, F, s);
}

extern "C" int ctts_synth_batch(ctts_voc* h, const float* const* hidden_ptrs, const int32_t* n_tokens, int B, float* const* wav_ptrs, void* stream) {
    if (!h || !h->finalized || !hidden_ptrs || !n_tokens || !wav_ptrs) { ctts_set_error("synth_batch: bad argument"); return 1; }
    if (B < 1 || B > h->cfg.max_batch) { ctts_set_error("synth_batch: B=%d exceeds max_batch=%d", B, h->cfg.max_batch); return 1; }
    CTTS_RANGE("ctts_synth_batch");
    int Fmax = 0;
    for (int u = 0; u < B; ++u) {
        const int F = 2 * n_tokens[u];
        if (n_tokens[u] < 1 || F > h->cfg.max_frames) { ctts_set_error("synth_batch: utterance %d has %d frames (max_frames=%d)", u, F, h->cfg.max_frames); return 1; }
        h->frames_host[u] = F;
        Fmax = F > Fmax ? F : Fmax;
    }
    hipStream_t s = (hipStream_t)stream;
    if (set_tables(h, hidden_ptrs, h->frames_host.data(), wav_ptrs, nullptr, B, s) || stage(h, true, B, Fmax, s)) return 1;
    if (run_dvae(h, B, Fmax, true, s)) return 1;
    return run_vocos(h, B, Fmax, s);
}
