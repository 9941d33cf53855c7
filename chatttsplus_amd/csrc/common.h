// Shared device/host definitions for libctts_hip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define CTTS_HEAD_DIM 64
#define CTTS_NUM_VQ 4
#define CTTS_MAX_B 128

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

// Per decode ROW state the sampler keeps in engine memory (one 32-byte record, fetched with the kernel's first batch of loads):
// the engine-side mirror of {finish, end_idx} (gpt.py:339-342,486-487,530-531), and what keys the row's device noise stream --
// the caller's global utterance id, the row's own regenerate attempt (ensure_non_empty, gpt.py:496-525) -- plus the row's own token
// limit (<= max_new_token).  Rows are re-packed by ctts_gpt_compact and re-used by ctts_gpt_admit; `seq` of RowMeta names the KV lane,
// `out` the utterance's place in the caller's output arrays (equal until a lane is handed to another utterance).
struct RowState {
    int fin;             // 0 = live; bit 0 = finished (no more tokens counted); bit 1 = ... by a sampled EOS (gpt.py:486-487) rather than by the row's limit
    int end;             // end_idx = the row's OWN step: tokens it has sampled so far (rows admitted later run behind the batch's step counter)
    int attempt;         // first-step-EOS regenerations of THIS row so far
    int limit;           // the row stops after this many tokens
    unsigned uid_lo, uid_hi;   // global utterance id
    int out;             // index of the utterance in ids / hiddens / finish / end_idx (and in the caller-supplied noise rows)
    int pad1;
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
    static constexpr int TILE_BYTES = 1024;
    typedef half8 frag;
    typedef half_t cache_t;             // element type of the K / V cache the engine keeps
};
template <> struct WTraits<float> {
    static constexpr int KT = 16;       // 4 x v_mfma_f32_16x16x4_f32 per 16-byte fragment
    static constexpr int EPL = 4;
    static constexpr int TILE_BYTES = 1024;
    typedef float4 frag;
    typedef float cache_t;
};
// Operand format "split" (round 6): the fp32 PARITY engine's decode projections on the fp16 matrix pipes.  Every operand is a head / tail pair of fp16 images,
// v = hi + lo with hi = fp16(v), lo = fp16(v - hi) (weights pre-scaled by 64 so that their tails stay normal numbers); a product is three
// v_mfma_f32_16x16x32_f16 (lo.hi, hi.lo, hi.hi: the dropped lo.lo term is 2^-22 of a product) instead of eight exact-f32 v_mfma_f32_16x16x4_f32 -- 1/5 of the
// matrix-pipe time at fp32-level accuracy, the same bytes from HBM (2 x 2 bytes per weight).  A (tile, k-tile) pair is stored [hi: 64 lanes x 16 B | lo: 64 lanes x 16 B]
// = 2 KiB, so one base pointer addresses both images; the tile order is the fp16 engine's.  prefill_split.hip reads the same weight images in the prompt pass.
struct split_t { half_t hi, lo; };
template <> struct WTraits<split_t> {
    static constexpr int KT = 32;
    static constexpr int EPL = 8;
    static constexpr int TILE_BYTES = 2048;
    typedef float cache_t;              // the engine itself is an fp32 engine: fp32 K / V
};
#define CTTS_SPLIT_WSCALE 64.0f         // weight images hold 64 * W (== SP_WSCALE of prefill_split.hip)
#define CTTS_SPLIT_ACT_SCALE 16.0f      // SwiGLU outputs are stored divided by 16 (== SP_ACT_SCALE)

template <typename WT>
__host__ __device__ inline size_t xfrag_index(int n, int k, int ktiles) {
    constexpr int KT = WTraits<WT>::KT, EPL = WTraits<WT>::EPL;
    int g = n >> 4;
    return ((size_t)(g * ktiles + k / KT) * 64 + (n & 15) + 16 * ((k / EPL) & 3)) * EPL + (k % EPL);
}

// fp16 stores of values the model produces without bound (SwiGLU outputs, the scaled residual copy): SATURATE at the
// largest finite half and REPORT (a device counter the host reads at the end of generate()), instead of the silent +-inf -> NaN a plain
// conversion gives -- the reference's own .half() path has that failure mode on checkpoints with outlier channels (pipeline:37-41).
// A NaN input is reported too (it stays NaN).
__device__ inline half_t sat_half(float v, int* sat) {
    const float c = fminf(fmaxf(v, -65504.f), 65504.f);
    if (!(c == v) && sat != nullptr) atomicAdd(sat, 1);
    return (half_t)((v != v) ? v : c);
}
// head / tail of one value (clamped to the fp16 range and reported like sat_half)
__device__ inline void split_half(float v, half_t& hi, half_t& lo, int* sat) {
    const float c = fminf(fmaxf(v, -65504.f), 65504.f);
    if (!(c == v) && sat != nullptr) atomicAdd(sat, 1);
    hi = (half_t)((v != v) ? v : c);
    lo = (half_t)(c - (float)hi);
}
// byte offset of the HEAD element (n, k) in a split fragment image whose 16-row groups hold `ktiles` k-tiles of 32 (the tail sits 1024 bytes behind it)
__host__ __device__ inline size_t xfrag_split_bytes(int n, int k, int ktiles) {
    return ((size_t)((n >> 4) * ktiles + (k >> 5)) * 128 + (n & 15) + 16 * ((k >> 3) & 3)) * 16 + (size_t)(k & 7) * 2;
}
template <typename WT> __device__ inline WT sat_store(float v, int* sat);
template <> __device__ inline half_t sat_store<half_t>(float v, int* sat) { return sat_half(v, sat); }
template <> __device__ inline float sat_store<float>(float v, int*) { return v; }

// One cache element held in a register of its own (a V row one dim per lane: attention.hip, persist_layer.hip).  A half is loaded as a zero-extended 16-bit word into a
// full register: a plain `half` load is a d16 load that keeps the register's other half, i.e. depends on whatever wrote that register before.
template <typename WT> struct KvElem;
template <> struct KvElem<float> {
    typedef float reg;
    __device__ static inline reg load(const float* p) { return *p; }
    __device__ static inline float f(reg v) { return v; }
};
template <> struct KvElem<half_t> {
    typedef unsigned reg;
    __device__ static inline reg load(const half_t* p) { return (unsigned)*(const unsigned short*)p; }
    __device__ static inline float f(reg v) { return (float)__builtin_bit_cast(half_t, (unsigned short)v); }
};

// ---- wave64 reductions on DPP (no LDS crossbar): 4 intra-row steps (quad xor1, quad xor2, half-mirror, mirror)
// leave the 16-lane row result in every lane; the 4 rows are combined through v_readlane.  ~10x cheaper than a
// 6-step ds_bpermute butterfly (the sampler's 20+ dependent arg-max rounds were 30 us with __shfl_xor).
template <int CTRL> __device__ inline int dpp_i(int v) { return __builtin_amdgcn_update_dpp(v, v, CTRL, 0xF, 0xF, false); }
template <int CTRL> __device__ inline float dpp_f(float v) { return __builtin_bit_cast(float, dpp_i<CTRL>(__builtin_bit_cast(int, v))); }
template <int CTRL> __device__ inline unsigned long long dpp_u64(unsigned long long v) {
    const unsigned lo = (unsigned)dpp_i<CTRL>((int)(unsigned)v), hi = (unsigned)dpp_i<CTRL>((int)(unsigned)(v >> 32));
    return ((unsigned long long)hi << 32) | lo;
}
template <int CTRL> __device__ inline double dpp_d(double v) {
    return __builtin_bit_cast(double, dpp_u64<CTRL>(__builtin_bit_cast(unsigned long long, v)));
}
#define DPP_XOR1 0xB1
#define DPP_XOR2 0x4E
#define DPP_HALF_MIRROR 0x141
#define DPP_MIRROR 0x140
__device__ inline float readlane_f(float v, int l) { return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), l)); }
__device__ inline unsigned long long readlane_u64(unsigned long long v, int l) {
    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)v, l), hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(v >> 32), l);
    return ((unsigned long long)hi << 32) | lo;
}
__device__ inline float wave_sum(float v) {
    v += dpp_f<DPP_XOR1>(v); v += dpp_f<DPP_XOR2>(v); v += dpp_f<DPP_HALF_MIRROR>(v); v += dpp_f<DPP_MIRROR>(v);
    return (readlane_f(v, 0) + readlane_f(v, 16)) + (readlane_f(v, 32) + readlane_f(v, 48));
}
__device__ inline double wave_sum_d(double v) {
    v += dpp_d<DPP_XOR1>(v); v += dpp_d<DPP_XOR2>(v); v += dpp_d<DPP_HALF_MIRROR>(v); v += dpp_d<DPP_MIRROR>(v);
    const unsigned long long b = __builtin_bit_cast(unsigned long long, v);
    return (__builtin_bit_cast(double, readlane_u64(b, 0)) + __builtin_bit_cast(double, readlane_u64(b, 16))) +
           (__builtin_bit_cast(double, readlane_u64(b, 32)) + __builtin_bit_cast(double, readlane_u64(b, 48)));
}
__device__ inline float wave_max(float v) {
    v = fmaxf(v, dpp_f<DPP_XOR1>(v)); v = fmaxf(v, dpp_f<DPP_XOR2>(v)); v = fmaxf(v, dpp_f<DPP_HALF_MIRROR>(v)); v = fmaxf(v, dpp_f<DPP_MIRROR>(v));
    return fmaxf(fmaxf(readlane_f(v, 0), readlane_f(v, 16)), fmaxf(readlane_f(v, 32), readlane_f(v, 48)));
}
__device__ inline unsigned long long umax64(unsigned long long a, unsigned long long b) { return a > b ? a : b; }
__device__ inline unsigned long long wave_max_u64(unsigned long long v) {
    v = umax64(v, dpp_u64<DPP_XOR1>(v)); v = umax64(v, dpp_u64<DPP_XOR2>(v));
    v = umax64(v, dpp_u64<DPP_HALF_MIRROR>(v)); v = umax64(v, dpp_u64<DPP_MIRROR>(v));
    return umax64(umax64(readlane_u64(v, 0), readlane_u64(v, 16)), umax64(readlane_u64(v, 32), readlane_u64(v, 48)));
}
// total order on floats as unsigned keys (larger float -> larger key; -inf -> small but non-zero key)
__device__ inline unsigned f32_key(float f) {
    const unsigned u = __builtin_bit_cast(unsigned, f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ inline float key_f32(unsigned k) {
    const unsigned u = (k & 0x80000000u) ? (k & 0x7FFFFFFFu) : ~k;
    return __builtin_bit_cast(float, u);
}

// Diagnostic switches (A/B experiments, kernel-variant overrides) exist only in builds with -DCTTS_DIAG
// (`python -m chatttsplus_amd.build --diag`); in the product library this is a constant null and no launch path reads the environment.
#include <stdlib.h>
static inline const char* diag_env(const char* name) {
#ifdef CTTS_DIAG
    return getenv(name);
#else
    (void)name;
    return nullptr;
#endif
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

// The "every sequence finished" flag is fetched with a VECTOR load (opaque zero lane offset) so that it travels together
// with the kernel's first operand loads; a scalar load + branch at kernel entry costs one extra serial cache-miss round trip
// in each of the ~100 dependent launches of a decode step.  The caller tests the value after its loads have been issued.
__device__ inline int vload_flag(const int* p) {
    int z;
    asm volatile("v_mov_b32 %0, 0" : "=v"(z));
    return p[z];
}
