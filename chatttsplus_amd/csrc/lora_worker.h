// Per-utterance LoRA folded into the projection launches (decode steps).
//
// lora.hip evaluates  scale * B (A x)  of every row in two extra launches per layer -- 13 us per layer on a chain whose launches cost ~5 us whatever
// they do (27-30 % of a batch-32 step).  Here the same arithmetic runs in WORKER workgroups that ride in front of every chunk's tiles of the QKV and
// o_proj launches (skinny_gemm.hip: blockIdx.x < workers), and the projection's epilogue picks the low-rank term up as tagged 8-byte granules
// {tag, value} -- the hand-off of persist_layer.hip: the data is the flag, no fences, no counters.  A worker has a LOWER linear block index than every
// tile that waits for it, so it is resident (or done) before any of its consumers starts: the wait cannot deadlock, and it is bounded anyway.
// tag = (draw + 1) * 64 + layer * 2 + which, `draw` = the sampler's running draw counter (DevState, advanced once per decode step on the device, so a
// replayed graph needs no host-side argument): two consecutive writes of a granule always differ in the tag.
//
// Reference: peft merge_and_unload as the pipeline applies it (pipelines/chattts_plus_pipeline.py:420-432), per row instead of per batch.
#pragma once
#include "kernels.h"

#define LORA_RMAX 16
#define LORA_SPIN_LIMIT (1u << 20)


__device__ inline void lora_publish(lora_u64* p, unsigned tag, float v) {
    __hip_atomic_store(p, ((lora_u64)tag << 32) | (lora_u64)__builtin_bit_cast(unsigned, v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// tag of a step's granules: the draw counter is requested with a worker's / tile's first loads and only turned into the tag where it is used (the
// arithmetic on a loaded value is a wait on the load)
__device__ inline unsigned lora_tag_of(int draw_v, unsigned tag_lo) { return ((unsigned)draw_v + 1u) * 64u + tag_lo; }

// one lane waits for its granule; 0.f + the error word after LORA_SPIN_LIMIT passes (a worker can only be late, never absent)
__device__ inline float lora_take(const lora_u64* p, unsigned tag, int* err) {
#pragma unroll 1
    for (unsigned spins = 0; spins < LORA_SPIN_LIMIT; ++spins) {
        const lora_u64 x = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if ((unsigned)(x >> 32) == tag) return __builtin_bit_cast(float, (unsigned)x);
        __builtin_amdgcn_s_sleep(1);
    }
    if (err != nullptr) atomicCAS(err, 0, 7);
    return 0.f;
}

// a first look at a granule, issued where its round trip hides behind the tile's own cross-wave reduction (skinny_gemm.hip): lora_peek early,
// lora_take_peeked in the epilogue -- which only polls if the early look came too soon
__device__ inline lora_u64 lora_peek(const lora_u64* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ inline float lora_take_peeked(lora_u64 x, const lora_u64* p, unsigned tag, int* err) {
    if ((unsigned)(x >> 32) == tag) return __builtin_bit_cast(float, (unsigned)x);
    return lora_take(p, tag, err);
}

// two granules of one lane (q/k/v epilogue: dims d and d + 32), polled together: one round trip per pass instead of two
__device__ inline void lora_take2(const lora_u64* p0, const lora_u64* p1, unsigned tag, int* err, float& v0, float& v1) {
#pragma unroll 1
    for (unsigned spins = 0; spins < LORA_SPIN_LIMIT; ++spins) {
        const lora_u64 x0 = __hip_atomic_load(p0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const lora_u64 x1 = __hip_atomic_load(p1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if ((unsigned)(x0 >> 32) == tag && (unsigned)(x1 >> 32) == tag) {
            v0 = __builtin_bit_cast(float, (unsigned)x0); v1 = __builtin_bit_cast(float, (unsigned)x1);
            return;
        }
        __builtin_amdgcn_s_sleep(1);
    }
    if (err != nullptr) atomicCAS(err, 0, 7);
    v0 = 0.f; v1 = 0.f;
}

// One worker = one (row, target).  Everything it reads is requested in ONE batch at entry (a first version loaded the row, reduced it, then loaded A, then B:
// three dependent round trips -- the tiles behind it reached their epilogues first and waited), and it reads as little as it can, because its loads share the
// queues of the launch's weight stream: the row and the RMSNorm weight once per workgroup (through LDS), and only the adapter's OWN rank components of A and B
// (adapters are stored zero-padded to r = 16; r = 8 halves the worker's bytes).  The arithmetic per element is lora.hip's:
// u_k = sum_c A[k][c] * (lnw[c] * (x[c] * rs)), delta_n = scale * sum_k B[n][k] * u_k.
template <int NW, bool NORM, typename LoadIn>
__device__ inline void lora_worker_core(LoadIn load_in, const float* lnw, float eps, const float* A_t, const float* B_t, const float* scale_p, int rk, lora_u64* grow,
                                        int draw_v, unsigned tag_lo, float* lds, int tid) {
    static_assert(NW == 4 || NW == 8 || NW == 16, "256 / 512 / 1024 threads");
    constexpr int RPW = LORA_RMAX / NW;                 // rank components per wave
    constexpr int NT = NW * 64, NPT = (768 + NT - 1) / NT;
    const int lane = tid & 63, wave = tid >> 6;
    float* const xs = lds;                              // [768] the row
    float* const ws = lds + 768;                        // [768] RMSNorm weight
    float* const u = lds + 1536;                        // [16]
    float xin[NPT], lwv[NPT], av[RPW][12];
    float bv[NPT][LORA_RMAX];
#pragma unroll
    for (int j = 0; j < NPT; ++j) {
        const int c = tid + NT * j;
        xin[j] = (c < 768) ? load_in(c) : 0.f;
        lwv[j] = (NORM && c < 768) ? lnw[c] : 0.f;
    }
    const float scale = *scale_p;
#pragma unroll
    for (int kk = 0; kk < RPW; ++kk) {
        const float* ar = A_t + (size_t)(RPW * wave + kk) * 768 + lane;
#pragma unroll
        for (int i = 0; i < 12; ++i) av[kk][i] = (RPW * wave + kk < rk) ? ar[64 * i] : 0.f;
    }
#pragma unroll
    for (int j = 0; j < NPT; ++j) {
        const int n = tid + NT * j;
        const float* bc = B_t + (n < 768 ? n : 0);                       // B is stored rank-major [16][768] like A: an adapter of rank r costs r rows, not 16
#pragma unroll
        for (int k = 0; k < LORA_RMAX; ++k) bv[j][k] = (k < rk) ? bc[(size_t)k * 768] : 0.f;
    }
#pragma unroll
    for (int j = 0; j < NPT; ++j) {
        const int c = tid + NT * j;
        if (c < 768) { xs[c] = xin[j]; if (NORM) ws[c] = lwv[j]; }
    }
    __syncthreads();
    float hv[12];
#pragma unroll
    for (int i = 0; i < 12; ++i) hv[i] = xs[lane + 64 * i];
    if (NORM) {
        float ss = 0.f;
#pragma unroll
        for (int i = 0; i < 12; ++i) ss += hv[i] * hv[i];
        ss = wave_sum(ss);
        const float rs = 1.0f / sqrtf(ss / 768.0f + eps);                  // LlamaRMSNorm (llama.py:82-87)
#pragma unroll
        for (int i = 0; i < 12; ++i) hv[i] = ws[lane + 64 * i] * (hv[i] * rs);
    }
#pragma unroll
    for (int kk = 0; kk < RPW; ++kk) {
        float a = 0.f;
#pragma unroll
        for (int i = 0; i < 12; ++i) a += av[kk][i] * hv[i];
        a = wave_sum(a);
        if (lane == 0) u[RPW * wave + kk] = a;
    }
    __syncthreads();
    float uu[LORA_RMAX];
#pragma unroll
    for (int k = 0; k < LORA_RMAX; ++k) uu[k] = u[k];
    const unsigned tag = lora_tag_of(draw_v, tag_lo);
#pragma unroll
    for (int j = 0; j < NPT; ++j) {
        const int n = tid + NT * j;
        const float* b = bv[j];
        float a = 0.f;
        a += b[0] * uu[0] + b[1] * uu[1] + b[2] * uu[2] + b[3] * uu[3];
        a += b[4] * uu[4] + b[5] * uu[5] + b[6] * uu[6] + b[7] * uu[7];
        a += b[8] * uu[8] + b[9] * uu[9] + b[10] * uu[10] + b[11] * uu[11];
        a += b[12] * uu[12] + b[13] * uu[13] + b[14] * uu[14] + b[15] * uu[15];
        if (n < 768) lora_publish(grow + n, tag, scale * a);
    }
}

// q/k/v worker: (row r, target t) of lora.hip's lora_delta_qkv_kernel
// (Measured and dropped: workers that also warm their XCD's L2 with the adapters the NEXT launch's workers will read -- +70 us per batch-32 step instead of
//  less: every byte a worker moves shares the queues of the launch's weight stream.)
template <int NW>
__device__ inline void lora_worker_qkv(const LoraFold& f, const float* x, float eps, int r, int t, int draw_v, unsigned tag_lo, float* lds, int tid) {
    lora_u64* const grow = f.g + ((size_t)r * 3 + t) * 768;
    const int slot = f.slots[r];
    if (slot < 0 || f.diag == 3) {
        for (int i = tid; i < 768; i += NW * 64) lora_publish(grow + i, lora_tag_of(draw_v, tag_lo), 0.f);
        return;
    }
    const float* xr = x + (size_t)r * 768;
    const size_t off = ((size_t)slot * 4 + t) * LORA_RMAX * 768;
    lora_worker_core<NW, true>([xr](int c) { return xr[c]; }, f.lnw, eps, f.A + off, f.B + off, f.scale + slot * 4 + t, f.ranks[slot * 4 + t], grow, draw_v, tag_lo, lds, tid);
}

// o_proj worker: row r of lora.hip's lora_delta_o_kernel (input = the attention output, read back from o_proj's fragment-major B operand)
template <typename WT, int NW>
__device__ inline void lora_worker_o(const LoraFold& f, const void* attn_packed, int nbg, int r, int draw_v, unsigned tag_lo, float* lds, int tid) {
    lora_u64* const grow = f.g_o + (size_t)r * 768;
    const int slot = f.slots[r];
    if (slot < 0 || f.diag == 3) {
        for (int i = tid; i < 768; i += NW * 64) lora_publish(grow + i, lora_tag_of(draw_v, tag_lo), 0.f);
        return;
    }
    constexpr int KT = WTraits<WT>::KT, EPL = WTraits<WT>::EPL;
    const int NB = 16 * nbg, kt = 768 / KT, n = r % NB;
    const WT* base = (const WT*)attn_packed + (size_t)(r / NB) * nbg * kt * 64 * EPL;
    const size_t off = ((size_t)slot * 4 + 3) * LORA_RMAX * 768;
    lora_worker_core<NW, false>([base, n, kt](int c) { return (float)base[xfrag_index<WT>(n, c, kt)]; }, nullptr, 0.f, f.A + off, f.B + off, f.scale + slot * 4 + 3,
                                f.ranks[slot * 4 + 3], grow, draw_v, tag_lo, lds, tid);
}
