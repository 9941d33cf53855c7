// Per-utterance LoRA: adapters that stay SEPARATE from the packed weights, one (or none) per sequence of a batch.
//
// The reference serves an adapter by merging it into a deep copy of the whole Llama for the call (peft merge_and_unload,
// pipelines/chattts_plus_pipeline.py:420-432: W' = W + (alpha / r) B A on q/k/v/o of every layer) -- one adapter per batch.
// Here  y = W x + scale * B (A x)  is evaluated per row: a small kernel per projection group computes the low-rank term of every
// row with the row's own adapter (rows of sequences without one get zeros) and the projection kernels add it in their epilogues
// (before RoPE / the cache append for q/k/v, with the residual for o_proj).  Two extra launches per layer, paid only by batches
// that carry adapters; the algebra equals the merged weights up to fp32 rounding.
#include "kernels.h"

#define LORA_RMAX 16

__device__ inline float block_sum_256(float v, float* red, int tid) {
    v = wave_sum(v);
    if ((tid & 63) == 0) red[tid >> 6] = v;
    __syncthreads();
    const float s = (red[0] + red[1]) + (red[2] + red[3]);
    __syncthreads();
    return s;
}

// One block = one (row, target): 4 waves x 4 rank components, every lane owns 12 of the 768 input columns.  All loads of a phase are
// issued together (a first version walked the 16 dot products one after the other, each waiting for its own loads: 20-40 us per
// launch, 3.7x the step time at batch 32).
__device__ inline void lora_low_rank(const float* hw, const float* A_t, const float* B_t, float scale, float* drow, int H, float* u, int tid) {
    const int lane = tid & 63, wave = tid >> 6;
    float hv[12];
#pragma unroll
    for (int i = 0; i < 12; ++i) hv[i] = hw[lane + 64 * i];
    float acc[4];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {                      // rank components 4 * wave + kk
        const float* ar = A_t + (size_t)(4 * wave + kk) * H + lane;
        float a = 0.f;
#pragma unroll
        for (int i = 0; i < 12; ++i) a += ar[64 * i] * hv[i];
        acc[kk] = a;
    }
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) { const float v = wave_sum(acc[kk]); if (lane == 0) u[4 * wave + kk] = v; }
    __syncthreads();
    float uu[LORA_RMAX];
#pragma unroll
    for (int k = 0; k < LORA_RMAX; ++k) uu[k] = u[k];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const int n = tid + 256 * i;
        float b[LORA_RMAX];                                   // B is stored rank-major [16][768]
#pragma unroll
        for (int k = 0; k < LORA_RMAX; ++k) b[k] = B_t[(size_t)k * H + n];
        float a = 0.f;
        a += b[0] * uu[0] + b[1] * uu[1] + b[2] * uu[2] + b[3] * uu[3];
        a += b[4] * uu[4] + b[5] * uu[5] + b[6] * uu[6] + b[7] * uu[7];
        a += b[8] * uu[8] + b[9] * uu[9] + b[10] * uu[10] + b[11] * uu[11];
        a += b[12] * uu[12] + b[13] * uu[13] + b[14] * uu[14] + b[15] * uu[15];
        drow[n] = scale * a;
    }
}

// q/k/v targets: input = input_layernorm(x) (llama.py:726-731: the projections see the normalised hidden states)
//   x [rows][768] fp32 residual stream, lnw [768], meta row -> sequence, slot_of_seq [max_batch] (-1 = no adapter),
//   A_l / B_l: this layer's adapters, [slot][target 0..3][16][768], both rank-major (B transposed; zero padded to r = 16), scale_l [slot][4]
//   delta [rows][3][768];  grid = (rows, 3 targets)
__global__ __launch_bounds__(256) void lora_delta_qkv_kernel(const float* x, const float* lnw, float eps, const RowMeta* meta, const int* slot_of_seq,
                                                           const float* A_l, const float* B_l, const float* scale_l, float* delta, int H) {
    __shared__ float hw[768];
    __shared__ float u[LORA_RMAX];
    __shared__ float red[4];
    const int r = blockIdx.x, t = blockIdx.y, tid = threadIdx.x;
    const int slot = slot_of_seq[meta[r].seq];
    float* drow = delta + ((size_t)r * 3 + t) * H;
    if (slot < 0) {
        for (int i = tid; i < H; i += 256) drow[i] = 0.f;
        return;
    }
    float xv[3], ss = 0.f;
#pragma unroll
    for (int i = 0; i < 3; ++i) { xv[i] = x[(size_t)r * H + tid + 256 * i]; ss += xv[i] * xv[i]; }
    ss = block_sum_256(ss, red, tid);
    const float rs = 1.0f / sqrtf(ss / (float)H + eps);                  // LlamaRMSNorm (llama.py:82-87)
#pragma unroll
    for (int i = 0; i < 3; ++i) hw[tid + 256 * i] = lnw[tid + 256 * i] * (xv[i] * rs);
    __syncthreads();
    const size_t off = ((size_t)slot * 4 + t) * LORA_RMAX * H;
    lora_low_rank(hw, A_l + off, B_l + off, scale_l[slot * 4 + t], drow, H, u, tid);
}

// o_proj target: input = the attention output rows, read back from the o_proj kernel's fragment-major B operand (S = 1 path)
template <typename WT>
__global__ __launch_bounds__(256) void lora_delta_o_kernel(const void* attn_packed, int nbg, const RowMeta* meta, const int* slot_of_seq, const float* A_l,
                                                         const float* B_l, const float* scale_l, float* delta, int H) {
    __shared__ float o[768];
    __shared__ float u[LORA_RMAX];
    const int r = blockIdx.x, tid = threadIdx.x;
    const int slot = slot_of_seq[meta[r].seq];
    float* drow = delta + (size_t)r * H;
    if (slot < 0) {
        for (int i = tid; i < H; i += 256) drow[i] = 0.f;
        return;
    }
    constexpr int KT = WTraits<WT>::KT, EPL = WTraits<WT>::EPL;
    const int NB = 16 * nbg, kt = H / KT;
    const WT* base = (const WT*)attn_packed + (size_t)(r / NB) * nbg * kt * 64 * EPL;
#pragma unroll
    for (int i = 0; i < 3; ++i) { const int c = tid + 256 * i; o[c] = (float)base[xfrag_index<WT>(r % NB, c, kt)]; }
    __syncthreads();
    const size_t off = ((size_t)slot * 4 + 3) * LORA_RMAX * H;
    lora_low_rank(o, A_l + off, B_l + off, scale_l[slot * 4 + 3], drow, H, u, tid);
}

// the same for the parity engine's split prompt pass (prefill_split.hip): the attention output rows exist as head / tail fp16 images [16-row group][24 k-tiles][lane][16 B]
__global__ __launch_bounds__(256) void lora_delta_o_split_kernel(const half_t* hi, const half_t* lo, const RowMeta* meta, const int* slot_of_seq, const float* A_l,
                                                               const float* B_l, const float* scale_l, float* delta, int H) {
    __shared__ float o[768];
    __shared__ float u[LORA_RMAX];
    const int r = blockIdx.x, tid = threadIdx.x;
    const int slot = slot_of_seq[meta[r].seq];
    float* drow = delta + (size_t)r * H;
    if (slot < 0) {
        for (int i = tid; i < H; i += 256) drow[i] = 0.f;
        return;
    }
    const size_t base = (size_t)(r >> 4) * 24 * 64 * 8;
#pragma unroll
    for (int i = 0; i < 3; ++i) { const int c = tid + 256 * i; const size_t q = base + xfrag_index<half_t>(r & 15, c, 24); o[c] = (float)hi[q] + (float)lo[q]; }
    __syncthreads();
    const size_t off = ((size_t)slot * 4 + 3) * LORA_RMAX * H;
    lora_low_rank(o, A_l + off, B_l + off, scale_l[slot * 4 + 3], drow, H, u, tid);
}
int launch_lora_delta_o_split(const void* hi, const void* lo, const RowMeta* meta, const int* slot_of_seq, const float* A_l, const float* B_l,
                              const float* scale_l, float* delta, int rows, int H, hipStream_t s) {
    if (H != 768) { ctts_set_error("lora: hidden %d != 768", H); return 1; }
    hipLaunchKernelGGL(lora_delta_o_split_kernel, dim3(rows), dim3(256), 0, s, (const half_t*)hi, (const half_t*)lo, meta, slot_of_seq, A_l, B_l, scale_l, delta, H);
    CTTS_HIP_CHECK(hipGetLastError());
    return 0;
}

int launch_lora_delta_qkv(const float* x, const float* lnw, float eps, const RowMeta* meta, const int* slot_of_seq, const float* A_l, const float* B_l,
                          const float* scale_l, float* delta, int rows, int H, hipStream_t s) {
    if (H != 768) { ctts_set_error("lora: hidden %d != 768", H); return 1; }
    hipLaunchKernelGGL(lora_delta_qkv_kernel, dim3(rows, 3), dim3(256), 0, s, x, lnw, eps, meta, slot_of_seq, A_l, B_l, scale_l, delta, H);
    CTTS_HIP_CHECK(hipGetLastError());
    return 0;
}
int launch_lora_delta_o(int dtype, const void* attn_packed, int nbg, const RowMeta* meta, const int* slot_of_seq, const float* A_l, const float* B_l,
                        const float* scale_l, float* delta, int rows, int H, hipStream_t s) {
    if (dtype == 1) hipLaunchKernelGGL(lora_delta_o_kernel<half_t>, dim3(rows), dim3(256), 0, s, attn_packed, nbg, meta, slot_of_seq, A_l, B_l, scale_l, delta, H);
    else hipLaunchKernelGGL(lora_delta_o_kernel<float>, dim3(rows), dim3(256), 0, s, attn_packed, nbg, meta, slot_of_seq, A_l, B_l, scale_l, delta, H);
    CTTS_HIP_CHECK(hipGetLastError());
    return 0;
}
