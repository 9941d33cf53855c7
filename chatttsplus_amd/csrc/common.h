// Shared device/host definitions for libctts_hip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define CTTS_HEAD_DIM 64
#define CTTS_NUM_VQ 4
#define CTTS_MAX_B 32

typedef _Float16 half_t;
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// One activation row of the engine ("row" = one new token of one sequence).
//   decode : row b  <-> sequence b, written at cache slot T+step-1
//   prefill: row b*T+t <-> token t of sequence b
struct RowMeta {
    int seq;        // sequence index (KV cache batch slot)
    int pos;        // RoPE position id = cumsum(mask)-1, pad -> 1     (gpt.py:238-245)
    int slot;       // KV cache slot this row's K/V are written to
    int kv_start;   // first attended key slot (left padding is skipped; causal end = slot)
};

// Per-generate() device state.  Everything that changes from step to step lives here so a captured
// hipGraph of one decode step can be replayed unchanged.
struct DevState {
    int step;            // sample steps executed so far (i of gpt.py:389)
    int draw;            // noise draws consumed (keeps running across ensure_non_empty restarts)
    int all_done;        // finish.all()  (gpt.py:545)
    int ticket;          // last-block detection in the sampler kernel
    int B;
    int T;
    int pad[CTTS_MAX_B]; // left-pad count per sequence
};

// fragment-major ("xfrag") activation layout used for every MFMA B operand:
//   element (n, k) of a [NB rows][K] chunk lives at
//     fp16: ((g*KT + k/32)*64 + (n%16) + 16*((k/8)%4))*8 + k%8      halfs
//     fp32: ((g*KT + k/16)*64 + (n%16) + 16*((k/4)%4))*4 + k%4      floats
//   with g = n/16, KT = K/32 (fp16) or K/16 (fp32): one (g, k-tile) = 64 lanes x 16 B = 1 KiB,
//   i.e. exactly what one wave loads as its B fragment with one 16-byte access per lane.
template <typename WT> struct WTraits;
template <> struct WTraits<half_t> {
    static constexpr int KT = 32;       // k per 16-row tile (v_mfma_f32_16x16x32_f16)
    static constexpr int EPL = 8;       // elements per lane per tile (16 bytes)
    typedef half8 frag;
};
template <> struct WTraits<float> {
    static constexpr int KT = 16;       // 4 x v_mfma_f32_16x16x4_f32 per 16-byte fragment
    static constexpr int EPL = 4;
    typedef float4 frag;
};

template <typename WT>
__host__ __device__ inline size_t xfrag_index(int n, int k, int ktiles) {
    constexpr int KT = WTraits<WT>::KT, EPL = WTraits<WT>::EPL;
    int g = n >> 4;
    return ((size_t)(g * ktiles + k / KT) * 64 + (n & 15) + 16 * ((k / EPL) & 3)) * EPL + (k % EPL);
}

__device__ inline float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
__device__ inline double wave_sum_d(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
__device__ inline float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    return v;
}

// host-side error plumbing (gpt_engine.cpp)
void ctts_set_error(const char* fmt, ...);
#define CTTS_HIP_CHECK(expr)                                                                   \
    do {                                                                                       \
        hipError_t _e = (expr);                                                                \
        if (_e != hipSuccess) {                                                                \
            ctts_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
            return 1;                                                                          \
        }                                                                                      \
    } while (0)
