// Zero-shot speaker path on gfx950: waveform -> log-mel -> DVAE encoder -> GFSQ codes (audio-prompt tokens).
//
// Reference arithmetic:
//   ChatTTSPlusPipeline.sample_audio_speaker      pipelines/chattts_plus_pipeline.py:279-284 (called from :486-499)
//   DVAE.forward(mode="encode")                   models/dvae.py:263-270
//   MelSpectrogramFeatures                        dvae.py:171-199  (third-party torchaudio MelSpectrogram: parity unpinned)
//   downsample_conv (k3 + GELU, k4 stride 2 + GELU), encoder = DVAEDecoder(512 -> 1024, hidden 256)   dvae.py:224-231,130-168
//   GFSQ.forward                                  dvae.py:94-126   (third-party vector_quantize_pytorch 1.17.8: parity unpinned)
//
// Same lowering as vocoder.hip (conv_gemm.h): channels-last rows, every Conv1d is a GEMM over overlapping row windows of a
// zero-guarded buffer.  Two more instances of that trick:
//   * the STFT: frame f is the 1024-sample window starting at sample 256 f of the reflect-padded signal -> A = padded signal,
//     row stride 256 (the hop), K = 1024, against a windowed real-DFT basis [cos | -sin]: no frame buffer;
//   * the stride-2 k4 conv: output t reads input rows 2t-1 .. 2t+2 -> row stride 2 C, K = 4 C.
#include <map>
#include <string>

#include "conv_gemm.h"
#include "roctx_range.h"

struct EncConvNext { float *dw_w, *dw_b, *ln_w, *ln_b, *w1, *b1, *w2, *b2, *gamma; };

struct ctts_enc {
    ctts_enc_cfg cfg;
    std::map<std::string, std::vector<float>> host;
    bool finalized = false;
    std::vector<void*> allocs;
    float *basis, *fbw, *coef, *ds0_w, *ds0_b, *ds2_w, *ds2_b, *ci0_w, *ci0_b, *ci2_w, *ci2_b, *co_w, *pin_w, *pin_b;
    std::vector<EncConvNext> blocks;
    float *xp, *spec, *mag, *melcl, *d1, *x2, *b128, *y, *ln, *mid, *feat;
    int* d_T;
    int Fmax, spec_ld, mag_ld, mel_ld;
};

static int ealloc(ctts_enc* h, float** p, size_t n) {
    CTTS_HIP_CHECK(hipMalloc((void**)p, n * 4));
    CTTS_HIP_CHECK(hipMemset(*p, 0, n * 4));
    h->allocs.push_back(*p);
    return 0;
}
static int eupload(ctts_enc* h, float** p, const std::vector<float>& v) {
    if (ealloc(h, p, v.size())) return 1;
    CTTS_HIP_CHECK(hipMemcpy(*p, v.data(), v.size() * 4, hipMemcpyHostToDevice));
    return 0;
}
static const std::vector<float>* eneed(ctts_enc* h, const std::string& k, size_t n) {
    auto it = h->host.find(k);
    if (it == h->host.end()) { ctts_set_error("missing weight %s", k.c_str()); return nullptr; }
    if (it->second.size() != n) { ctts_set_error("weight %s has %zu elements, expected %zu", k.c_str(), it->second.size(), n); return nullptr; }
    return &it->second;
}

// reflect padding of torch.stft(center=True, pad_mode="reflect"): xp[i] = x[reflect(i - n_fft/2)]
__global__ void reflect_pad_kernel(const float* x, float* xp, int n, int half, int total) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    int j = i - half;
    if (j < 0) j = -j;
    if (j >= n) j = 2 * (n - 1) - j;
    xp[i] = (j >= 0 && j < n) ? x[j] : 0.f;
}

// |X_k| from the [re | im] GEMM output; columns nb..ld-1 are zero padding of the mel GEMM's K
__global__ void magnitude_kernel(const float* spec, float* mag, int F, int nb, int lds, int ldm) {
    const int f = blockIdx.x;
    for (int k = threadIdx.x; k < ldm; k += blockDim.x) {
        float v = 0.f;
        if (k < nb) {
            const float re = spec[(size_t)f * lds + k], im = spec[(size_t)f * lds + nb + k];
            v = sqrtf(re * re + im * im);
        }
        mag[(size_t)f * ldm + k] = v;
    }
}

// mel [F][ld] channels-last -> API layout [n_mels][F] (test hook output)
__global__ void mel_out_kernel(const float* melcl, float* mel, int F, int n_mels, int ld) {
    const int f = blockIdx.x, c = threadIdx.x;
    if (c < n_mels) mel[(size_t)c * F + f] = melcl[(size_t)f * ld + c];
}

// GFSQ.forward (dvae.py:94-126) -> GroupedResidualFSQ: group g = channels [g*per, (g+1)*per); project_in to 4 dims; two
// residual FSQ layers, levels 5^4: code = rint(tanh(z) * 2.002) / 2, index = sum (2 code + 2) * 5^j, scale_r = 4^-r.
// One wave per (frame, group).
__global__ __launch_bounds__(64) void gfsq_kernel(const float* feat, const float* w, const float* b, int* ids, int T, int per, int ld, int R, int pre_bound) {
    const int t = blockIdx.x, g = blockIdx.y, lane = threadIdx.x;
    const float* x = feat + (size_t)t * ld + (size_t)g * per;
    float z[4] = {0.f, 0.f, 0.f, 0.f};
    for (int c = lane; c < per; c += 64) {
        const float xv = x[c];
#pragma unroll
        for (int j = 0; j < 4; ++j) z[j] += w[((size_t)g * 4 + j) * per + c] * xv;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) z[j] = wave_sum(z[j]) + b[g * 4 + j];
    if (lane != 0) return;
    const float half_l = 2.002f;                                   // (levels - 1) * (1 + 1e-3) / 2 in fp32
    float res[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) res[j] = pre_bound ? tanhf(z[j]) * half_l : z[j];
    float scale = 1.0f;
    for (int r = 0; r < R; ++r) {
        int idx = 0, basis = 1;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float q = rintf(tanhf(res[j] / scale) * half_l);  // torch.round: half to even
            const float code = q / 2.0f;
            idx += (int)(code * 2.0f + 2.0f) * basis;
            basis *= 5;
            res[j] -= code * scale;
        }
        ids[(size_t)(g * R + r) * T + t] = idx;
        scale *= 0.25f;                                            // (levels - 1) ** -(r + 1)
    }
}

extern "C" int ctts_enc_create(const ctts_enc_cfg* c, ctts_enc** out) {
    if (!c || !out) { ctts_set_error("null argument"); return 1; }
    if (c->dim != 512 || c->enc_hidden != 256 || c->enc_bn % 64 || c->enc_odim % 64 || c->n_fft != 1024 || c->hop != 256 || c->n_mels > 112 ||
        c->vq_groups < 1 || c->vq_groups > 8 || c->enc_odim % c->vq_groups || c->vq_residuals < 1 || c->vq_residuals > 8 || c->max_samples < c->n_fft) {
        ctts_set_error("unsupported encoder configuration");
        return 1;
    }
    ctts_enc* h = new ctts_enc();
    h->cfg = *c;
    *out = h;
    return 0;
}
extern "C" void ctts_enc_destroy(ctts_enc* h) {
    if (!h) return;
    for (void* p : h->allocs) (void)hipFree(p);
    delete h;
}
extern "C" int ctts_enc_set_weight(ctts_enc* h, const char* name, const float* data, size_t numel) {
    if (!h || !name || !data) { ctts_set_error("null argument"); return 1; }
    if (h->finalized) { ctts_set_error("weights already finalized"); return 1; }
    h->host[std::string(name)].assign(data, data + numel);
    return 0;
}

extern "C" int ctts_enc_finalize(ctts_enc* h) {
    if (!h) { ctts_set_error("null handle"); return 1; }
    if (h->finalized) return 0;
    const ctts_enc_cfg& c = h->cfg;
    const int NM = c.n_mels, D = c.dim, HD = c.enc_hidden, BN = c.enc_bn, OD = c.enc_odim, N = c.n_fft, NB = N / 2 + 1;
    h->mel_ld = 112; h->mag_ld = (NB + 15) / 16 * 16; h->spec_ld = r64(2 * NB);
    // windowed forward real-DFT basis rows: k < NB: w[n] cos(2 pi k n / N);  NB + k: -w[n] sin(2 pi k n / N)
    const std::vector<float>* win = eneed(h, "mel.window", N);
    const std::vector<float>* fb = eneed(h, "mel.fb", (size_t)NB * NM);
    if (!win || !fb) return 1;
    {
        std::vector<float> B((size_t)h->spec_ld * N, 0.f);
        for (int k = 0; k < NB; ++k)
            for (int n = 0; n < N; ++n) {
                const double ang = 2.0 * M_PI * (double)((long long)k * n % N) / N;
                B[(size_t)k * N + n] = (float)((*win)[n] * cos(ang));
                B[(size_t)(NB + k) * N + n] = (float)(-(double)(*win)[n] * sin(ang));
            }
        if (eupload(h, &h->basis, B)) return 1;
        std::vector<float> W((size_t)r64(NM) * h->mag_ld, 0.f);       // mel filterbank transposed: [n_mels][n_freqs]
        for (int k = 0; k < NB; ++k)
            for (int m = 0; m < NM; ++m) W[(size_t)m * h->mag_ld + k] = (*fb)[(size_t)k * NM + m];
        if (eupload(h, &h->fbw, W)) return 1;
    }
    const std::vector<float>*w, *b;
    w = eneed(h, "coef", NM);
    if (!w || eupload(h, &h->coef, *w)) return 1;
    w = eneed(h, "downsample_conv.0.weight", (size_t)D * NM * 3); b = eneed(h, "downsample_conv.0.bias", D);
    if (!w || !b || eupload(h, &h->ds0_w, conv_to_gemm(*w, D, NM, 3, h->mel_ld, r64(D))) || eupload(h, &h->ds0_b, *b)) return 1;
    w = eneed(h, "downsample_conv.2.weight", (size_t)D * D * 4); b = eneed(h, "downsample_conv.2.bias", D);
    if (!w || !b || eupload(h, &h->ds2_w, conv_to_gemm(*w, D, D, 4, D, r64(D))) || eupload(h, &h->ds2_b, *b)) return 1;
    w = eneed(h, "encoder.conv_in.0.weight", (size_t)BN * D * 3); b = eneed(h, "encoder.conv_in.0.bias", BN);
    if (!w || !b || eupload(h, &h->ci0_w, conv_to_gemm(*w, BN, D, 3, D, r64(BN))) || eupload(h, &h->ci0_b, *b)) return 1;
    w = eneed(h, "encoder.conv_in.2.weight", (size_t)HD * BN * 3); b = eneed(h, "encoder.conv_in.2.bias", HD);
    if (!w || !b || eupload(h, &h->ci2_w, conv_to_gemm(*w, HD, BN, 3, BN, r64(HD))) || eupload(h, &h->ci2_b, *b)) return 1;
    h->blocks.resize(c.enc_layers);
    for (int i = 0; i < c.enc_layers; ++i) {
        const std::string p = "encoder.decoder_block." + std::to_string(i) + ".";
        const int inter = HD * 4;
        const std::vector<float>*dw = eneed(h, p + "dwconv.weight", (size_t)HD * 7), *db = eneed(h, p + "dwconv.bias", HD),
                                *lw = eneed(h, p + "norm.weight", HD), *lb = eneed(h, p + "norm.bias", HD),
                                *w1 = eneed(h, p + "pwconv1.weight", (size_t)inter * HD), *b1 = eneed(h, p + "pwconv1.bias", inter),
                                *w2 = eneed(h, p + "pwconv2.weight", (size_t)HD * inter), *b2 = eneed(h, p + "pwconv2.bias", HD),
                                *g = eneed(h, p + "gamma", HD);
        if (!dw || !db || !lw || !lb || !w1 || !b1 || !w2 || !b2 || !g) return 1;
        EncConvNext& cb = h->blocks[i];
        if (eupload(h, &cb.dw_w, *dw) || eupload(h, &cb.dw_b, *db) || eupload(h, &cb.ln_w, *lw) || eupload(h, &cb.ln_b, *lb) ||
            eupload(h, &cb.w1, *w1) || eupload(h, &cb.b1, *b1) || eupload(h, &cb.w2, *w2) || eupload(h, &cb.b2, *b2) || eupload(h, &cb.gamma, *g))
            return 1;
    }
    w = eneed(h, "encoder.conv_out.weight", (size_t)OD * HD);
    if (!w || eupload(h, &h->co_w, pad_rows(*w, OD, HD, r64(OD)))) return 1;
    {
        const int G = c.vq_groups, per = OD / G;
        std::vector<float> pw((size_t)G * 4 * per), pb((size_t)G * 4);
        for (int g = 0; g < G; ++g) {
            const std::string p = "vq_layer.quantizer.rvqs." + std::to_string(g) + ".project_in.";
            w = eneed(h, p + "weight", (size_t)4 * per); b = eneed(h, p + "bias", 4);
            if (!w || !b) return 1;
            memcpy(pw.data() + (size_t)g * 4 * per, w->data(), (size_t)4 * per * 4);
            memcpy(pb.data() + (size_t)g * 4, b->data(), 16);
        }
        if (eupload(h, &h->pin_w, pw) || eupload(h, &h->pin_b, pb)) return 1;
    }
    // workspaces: F = 1 + n / hop frames, T = (F - 2) / 2 + 1 codes; rows padded so that every 64-row GEMM tile stays in bounds
    const int Fmax = 1 + c.max_samples / c.hop;
    h->Fmax = Fmax;
    const size_t Fp = r64(Fmax) + 64 + 8, Tp = r64(Fmax / 2 + 1) + 64 + 8;
    if (ealloc(h, &h->xp, (size_t)(r64(Fmax) + 64) * c.hop + N) || ealloc(h, &h->spec, Fp * h->spec_ld) || ealloc(h, &h->mag, Fp * h->mag_ld) ||
        ealloc(h, &h->melcl, Fp * h->mel_ld) || ealloc(h, &h->d1, (Fp + 136) * D) || ealloc(h, &h->x2, Tp * D) || ealloc(h, &h->b128, Tp * BN) ||
        ealloc(h, &h->y, Tp * HD) || ealloc(h, &h->ln, Tp * HD) || ealloc(h, &h->mid, Tp * HD * 4) || ealloc(h, &h->feat, Tp * OD))
        return 1;
    CTTS_HIP_CHECK(hipMalloc((void**)&h->d_T, 16));
    h->allocs.push_back(h->d_T);
    h->host.clear();
    h->finalized = true;
    return 0;
}

extern "C" int ctts_dvae_encode(ctts_enc* h, const float* wav, int n_samples, int32_t* ids, float* mel_out, float* feat_out, void* stream) {
    if (!h || !h->finalized || !wav || !ids) { ctts_set_error("dvae_encode: bad argument"); return 1; }
    CTTS_RANGE("ctts_dvae_encode");
    const ctts_enc_cfg& c = h->cfg;
    const int N = c.n_fft, hop = c.hop, NB = N / 2 + 1, D = c.dim, HD = c.enc_hidden, BN = c.enc_bn, OD = c.enc_odim;
    if (n_samples <= N / 2 || n_samples > c.max_samples) {        // reflect padding needs n > n_fft / 2 (torch.stft raises otherwise)
        ctts_set_error("dvae_encode: %d samples outside (%d, %d]", n_samples, N / 2, c.max_samples);
        return 1;
    }
    const int F = 1 + n_samples / hop, T = (F - 2) / 2 + 1;
    if (F < 2) { ctts_set_error("dvae_encode: too short"); return 1; }
    hipStream_t s = (hipStream_t)stream;
    // 1. reflect-padded signal (rows of the STFT GEMM overlap: frame f = xp[256 f .. 256 f + 1023])
    const int total = n_samples + N;
    hipLaunchKernelGGL(reflect_pad_kernel, dim3((total + 255) / 256), dim3(256), 0, s, wav, h->xp, n_samples, N / 2, total);
    CTTS_HIP_CHECK(hipGetLastError());
    GemmF32Args g = {};
    g.A = h->xp; g.lda = hop; g.W = h->basis; g.ldw = N; g.C = h->spec; g.ldc = h->spec_ld; g.M = F; g.N = 2 * NB; g.K = N;
    if (launch_gemm_f32(EP_NONE, g, 1, s)) return 1;                                       // torch.stft (dvae.py:186-193)
    hipLaunchKernelGGL(magnitude_kernel, dim3(F), dim3(256), 0, s, h->spec, h->mag, F, NB, h->spec_ld, h->mag_ld);   // power = 1
    CTTS_HIP_CHECK(hipGetLastError());
    // zero guard rows of this call's conv inputs (a longer previous call may have left data there)
    CTTS_HIP_CHECK(hipMemsetAsync(h->melcl, 0, (size_t)h->mel_ld * 4, s));
    CTTS_HIP_CHECK(hipMemsetAsync(h->melcl + (size_t)(F + 1) * h->mel_ld, 0, (size_t)h->mel_ld * 4, s));
    CTTS_HIP_CHECK(hipMemsetAsync(h->d1, 0, (size_t)D * 4, s));
    CTTS_HIP_CHECK(hipMemsetAsync(h->d1 + (size_t)(F + 1) * D, 0, (size_t)3 * D * 4, s));
    CTTS_HIP_CHECK(hipMemsetAsync(h->x2, 0, (size_t)D * 4, s));
    CTTS_HIP_CHECK(hipMemsetAsync(h->x2 + (size_t)(T + 1) * D, 0, (size_t)D * 4, s));
    CTTS_HIP_CHECK(hipMemsetAsync(h->b128, 0, (size_t)BN * 4, s));
    CTTS_HIP_CHECK(hipMemsetAsync(h->b128 + (size_t)(T + 1) * BN, 0, (size_t)BN * 4, s));
    CTTS_HIP_CHECK(hipMemcpyAsync(h->d_T, &T, 4, hipMemcpyHostToDevice, s));               // pageable source: staged before return
    GemmF32Args gm = {};
    gm.A = h->mag; gm.lda = h->mag_ld; gm.W = h->fbw; gm.ldw = h->mag_ld; gm.C = h->melcl + h->mel_ld; gm.ldc = h->mel_ld;
    gm.M = F; gm.N = c.n_mels; gm.K = h->mag_ld; gm.scale = h->coef;
    if (launch_gemm_f32(EP_LOGCLIP_DIV, gm, 1, s)) return 1;                               // MelScale, log(clip), / coef (dvae.py:196-198,264-266)
    if (mel_out) {                                                                         // test hook: log-mel / coef, [n_mels][F]
        hipLaunchKernelGGL(mel_out_kernel, dim3(F), dim3(128), 0, s, h->melcl + h->mel_ld, mel_out, F, c.n_mels, h->mel_ld);
        CTTS_HIP_CHECK(hipGetLastError());
    }
    GemmF32Args g1 = {};
    g1.A = h->melcl; g1.lda = h->mel_ld; g1.W = h->ds0_w; g1.ldw = 3 * h->mel_ld; g1.C = h->d1 + D; g1.ldc = D; g1.M = F; g1.N = D; g1.K = 3 * h->mel_ld;
    g1.bias = h->ds0_b;
    if (launch_gemm_f32(EP_BIAS_GELU, g1, 1, s)) return 1;                                 // downsample_conv.0 k3 p1 + GELU
    GemmF32Args g2 = {};
    g2.A = h->d1; g2.lda = 2 * D; g2.W = h->ds2_w; g2.ldw = 4 * D; g2.C = h->x2 + D; g2.ldc = D; g2.M = T; g2.N = D; g2.K = 4 * D; g2.bias = h->ds2_b;
    if (launch_gemm_f32(EP_BIAS_GELU, g2, 1, s)) return 1;                                 // downsample_conv.2 k4 stride 2 p1 + GELU
    GemmF32Args g3 = {};
    g3.A = h->x2; g3.lda = D; g3.W = h->ci0_w; g3.ldw = 3 * D; g3.C = h->b128 + BN; g3.ldc = BN; g3.M = T; g3.N = BN; g3.K = 3 * D; g3.bias = h->ci0_b;
    if (launch_gemm_f32(EP_BIAS_GELU, g3, 1, s)) return 1;                                 // encoder.conv_in.0 + GELU
    GemmF32Args g4 = {};
    g4.A = h->b128; g4.lda = BN; g4.W = h->ci2_w; g4.ldw = 3 * BN; g4.C = h->y; g4.ldc = HD; g4.M = T; g4.N = HD; g4.K = 3 * BN; g4.bias = h->ci2_b;
    if (launch_gemm_f32(EP_BIAS, g4, 1, s)) return 1;                                      // encoder.conv_in.2
    for (int i = 0; i < c.enc_layers; ++i) {                                               // ConvNeXt, dilation 2 (dvae.py:147-158)
        const EncConvNext& cb = h->blocks[i];
        hipLaunchKernelGGL(dwconv_ln_kernel<4>, dim3((T + 3) / 4, 1), dim3(256), 0, s, h->y, h->ln, cb.dw_w, cb.dw_b, cb.ln_w, cb.ln_b, h->d_T, 0L, 0L, HD, 2, 7);
        CTTS_HIP_CHECK(hipGetLastError());
        GemmF32Args a1 = {};
        a1.A = h->ln; a1.lda = HD; a1.W = cb.w1; a1.ldw = HD; a1.C = h->mid; a1.ldc = HD * 4; a1.M = T; a1.N = HD * 4; a1.K = HD; a1.bias = cb.b1;
        if (launch_gemm_f32(EP_BIAS_GELU, a1, 1, s)) return 1;
        GemmF32Args a2 = {};
        a2.A = h->mid; a2.lda = HD * 4; a2.W = cb.w2; a2.ldw = HD * 4; a2.C = h->y; a2.ldc = HD; a2.M = T; a2.N = HD; a2.K = HD * 4; a2.bias = cb.b2;
        a2.gamma = cb.gamma; a2.resid = h->y; a2.ldr = HD;
        if (launch_gemm_f32(EP_GAMMA_RESID, a2, 1, s)) return 1;
    }
    GemmF32Args g5 = {};
    g5.A = h->y; g5.lda = HD; g5.W = h->co_w; g5.ldw = HD; g5.C = h->feat; g5.ldc = OD; g5.M = T; g5.N = OD; g5.K = HD;
    if (launch_gemm_f32(EP_NONE, g5, 1, s)) return 1;                                      // encoder.conv_out 1x1
    if (feat_out) CTTS_HIP_CHECK(hipMemcpyAsync(feat_out, h->feat, (size_t)T * OD * 4, hipMemcpyDeviceToDevice, s));
    hipLaunchKernelGGL(gfsq_kernel, dim3(T, c.vq_groups), dim3(64), 0, s, h->feat, h->pin_w, h->pin_b, (int*)ids, T, OD / c.vq_groups, OD,
                       c.vq_residuals, c.pre_bound);
    CTTS_HIP_CHECK(hipGetLastError());
    return 0;
}
