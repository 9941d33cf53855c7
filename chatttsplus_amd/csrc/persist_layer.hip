// The decoder stack of a small decode batch (<= 5 rows; fp32 or fp16 engine) as ONE persistent launch (gfx950): all 20 layers of a step, or `n_layers` of them per
// launch (option "persistent_layers_per_launch"; the first version of the structure ran one layer per launch).
//
// Reference arithmetic: LlamaDecoderLayer.forward, chattts_plus/models/llama.py:719-749 (RMSNorm :82-87, q/k/v + RoPE + cache append :619-633,
// SDPA :653-661, o_proj + residual :666,731, SwiGLU MLP + residual :214,737-739) -- the loop it serves is gpt.py:389-546.
//
// Why: a batch-1 decode step of the launch path is 102 dependent launches of ~4.8 us each although a layer's 37.7 MB of fp32 weights stream
// in ~6 us: every launch pays boundary + ramp + one load round trip + drain.  Here the stack is ONE launch of 256 resident workgroups:
//   * 192 GEMV workgroups own a fixed slice of every projection of every layer (12 q/k/v rows, 4 o_proj rows, 16 gate|up pairs, 4 down rows = 192 KB of
//     weights per layer); their 8 compute waves keep the slice in registers and request each array for the NEXT use one wait ahead, paced (SCHED 3: one 1 KB
//     fragment per wave every ~0.2 us while the edge waves gather), so the weight stream runs AHEAD of the layer's dependency edges
//     (MI355X_MICROARCH.md price list, row prefetch-credit) and every product is VALU work on registers;
//   * 64 attention workgroups own one (row, head, key share) each and pull its cached K / V rows into registers before the query exists;
//   * the activations travel between workgroups as 8-byte {tag, value} granules (one sc1 store each, cdna_hip_programming.md Guideline 16 R2):
//     the data is the flag, a consumer wave re-reads its granules until every tag equals (launch counter, layer) -- no fences, no counters,
//     placement-independent.  Four "edge" waves per workgroup (two until round 5) do all gathering, epilogues and publishing, so the compute waves never poll
//     (a poll's result would queue behind their in-flight weight loads: vmcnt retires in order);
//   * five edges per layer: q|k|v -> attention, attention -> o_proj, (x + attention) -> gate|up, silu(gate) * up -> down, layer output -> next layer;
//     the residual stream enters the launch and leaves it through plain global memory (the sampler wrote it, the heads read it: launch boundaries).
// Every spin is bounded; a give-up sets a device error word that makes this and every later launch return at once (ctts_gpt_progress
// reports it).  The tag's launch counter is a device word the launch itself advances (graph replay freezes kernel arguments).
// Residency: the launch is a plain one and assumes its 256 workgroups are co-resident (one per CU, nothing else running persistent launches on the device):
// a per-device file lock keeps other processes off the mode, and decode calls of one process that launch persistent kernels take turns (gpt_engine.hip PersistTurn).
#include "kernels.h"
#include "persist.h"

typedef unsigned long long u64;

#ifndef PL_LB
#define PL_LB PL_THREADS_MAX
#endif
#define PL_SPIN_LIMIT (1u << 18)      // ~0.3 s of polling before a wave gives up

// the workgroup's give-up flag lives in LDS and is read / written as LDS (a `volatile int*` access stays a FLAT access to an address-space-cast constant: slower, and
// this hipcc emits an illegal V_CMP against src_shared_base for it in some of the 6..8-row kernels)
typedef __attribute__((address_space(3))) int pl_lds_int;
#define PL_ABORT_GET(p_) __hip_atomic_load((pl_lds_int*)(p_), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)
#define PL_ABORT_SET(p_) __hip_atomic_store((pl_lds_int*)(p_), 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)
__device__ inline void store_granule(u64* g, unsigned tag, float v) {
    __hip_atomic_store(g, ((u64)tag << 32) | (u64)__builtin_bit_cast(unsigned, v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// One wave re-reads its N granules (lane's granule k = base[lane_elem + OFF(k)], OFF(k) a compile-time constant) every pass until every tag equals `tag`;
// values -> v.  The lane's pointer is formed ONCE per sweep and made opaque: the granule addresses are then that register pair + immediates (the first
// version let hipcc hoist a separate 64-bit address pair per granule of every edge out of the layer loop: ~100 VGPRs alive through the loop, spills).
// Returns false after PL_SPIN_LIMIT passes or once the workgroup / the engine has given up.
template <int N, typename OffFn>
__device__ inline bool sweep(const u64* base, unsigned lane_elem, OffFn off, unsigned tag, float (&v)[N], int* err, int code, int* abort_s, int nap = 1) {
    const u64* p = base + lane_elem;
    asm volatile("" : "+v"(p));
#pragma unroll 1
    for (unsigned spins = 0;; ++spins) {
        bool ok = true;
#pragma unroll
        for (int k = 0; k < N; ++k) {
            const u64 x = __hip_atomic_load(p + off(k), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            v[k] = __builtin_bit_cast(float, (unsigned)x);
            ok = ok && ((unsigned)(x >> 32) == tag);
        }
        if (__all(ok)) return true;
        bool giveup = spins >= PL_SPIN_LIMIT || PL_ABORT_GET(abort_s) != 0;
        if (!giveup && (spins & 1023u) == 1023u) giveup = __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0;
        if (giveup) {
            if ((threadIdx.x & 63) == 0) { atomicCAS(err, 0, code); PL_ABORT_SET(abort_s); }
            return false;
        }
        for (int z = 0; z < nap; ++z) __builtin_amdgcn_s_sleep(2);
    }
}

// The same sweep through a buffer descriptor (6..8 rows).  The granules of a sweep lie 2 KB (columns) and 6 / 24 KB (rows) apart -- beyond the 4 KB immediate of a global
// load, so the flat form needs a 64-bit address pair per granule: 24 granules = 48 address registers beside the 48 data registers, and the 8-row kernel wanted 190 VGPRs
// (it has 168: 20 spills, a scratch reload in front of every publish, 0.70 ms per step instead of ~0.57).  A buffer load takes the lane's byte offset in ONE register and
// the granule's offset as a scalar: no per-granule address registers at all (cdna_hip_programming.md T8).  aux: sc1 (the relaxed agent-scope load's cache policy) + the
// intrinsic's volatile bit (the loop re-reads memory another workgroup writes).
template <int N, typename OffFn>
__device__ inline bool sweep_buf(const u64* base, unsigned n_granules, unsigned lane_elem, OffFn off, unsigned tag, float (&v)[N], int* err, int code, int* abort_s, int nap = 1) {
    typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, (int)(n_granules * 8u), 0x00020000);
    const unsigned voff = lane_elem * 8u;
#pragma unroll 1
    for (unsigned spins = 0;; ++spins) {
        bool ok = true;
#pragma unroll
        for (int k = 0; k < N; ++k) {
            const u32x2 x = __builtin_amdgcn_raw_buffer_load_b64(rsrc, (int)voff, off(k) * 8, (int)(16u | 0x80000000u));
            v[k] = __builtin_bit_cast(float, x[0]);
            ok = ok && (x[1] == tag);
        }
        if (__all(ok)) return true;
        bool giveup = spins >= PL_SPIN_LIMIT || PL_ABORT_GET(abort_s) != 0;
        if (!giveup && (spins & 1023u) == 1023u) giveup = __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0;
        if (giveup) {
            if ((threadIdx.x & 63) == 0) { atomicCAS(err, 0, code); PL_ABORT_SET(abort_s); }
            return false;
        }
        for (int z = 0; z < nap; ++z) __builtin_amdgcn_s_sleep(2);
    }
}
// the GEMV edge waves' sweeps: flat loads up to 5 rows (the tuned kernels stay as they are), buffer loads from 6 rows on
#ifndef PL_BUF_MIN_R
#define PL_BUF_MIN_R (PL_MAXR_ONE + 1)     // (A/B builds: tools/build_variant.sh bufall -DPL_BUF_MIN_R=1)
#endif
#define PL_SWEEP(N_, base_, count_, lane_elem_, off_, tag_, v_, code_) \
    ((R >= PL_BUF_MIN_R) ? sweep_buf<N_>(base_, (unsigned)(count_), lane_elem_, off_, tag_, v_, a.error, code_, abort_s, a.nap) \
                       : sweep<N_>(base_, lane_elem_, off_, tag_, v_, a.error, code_, abort_s, a.nap))

// ---- 16-byte act granules (round 6, PL_ACT16).  The act edge is the fattest gather of a layer: every GEMV workgroup needs all 3072 R SwiGLU outputs, as 8-byte
// {tag, value} granules that is 12 R loads per edge lane and 24.6 KB R per workgroup and layer (4.7 MB R across the chip per poll pass).  Here a producer workgroup
// packs its 16 outputs of a row into SIX 16-byte granules {v0, v1, v2, check} (the sixth holds one value and two zeros), check = tag + bits(v0) + bits(v1) + bits(v2):
// 4.5 R loads per lane, 18.4 KB R per workgroup.  Nothing in the ISA promises that a lane's 16-byte store is seen whole by another CU, so the tag is not a
// plain word: a reader accepts a granule only if check - bits(v0) - bits(v1) - bits(v2) == tag, which a mix of two generations fails (unless the mixed-in words are
// equal anyway); tools/mb/tear16.hip measures how often that would matter on this chip.  Layout: g_act16[(row * 192 + producer) * 6 + slot], i.e. granule f of the flat
// array holds columns 16 (f / 6) + 3 (f % 6) + {0, 1, 2} of the [R][3072] act rows -- edge lane e reads f = e + NE k, consecutive lanes consecutive granules.
#ifndef PL_ACT16
#define PL_ACT16 1
#endif
#define PL_A16_SLOTS 6
#ifndef PL_A16_MAXR
#define PL_A16_MAXR PL_MAXR_ONE
#endif
#ifndef PL_A16_CHUNK
#define PL_A16_CHUNK 12
#endif
typedef unsigned pl_u32x4 __attribute__((ext_vector_type(4)));
__device__ inline void store_granule16(void* base, unsigned n_granules, unsigned idx, unsigned tag, float v0, float v1, float v2) {
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(base, 0, (int)(n_granules * 16u), 0x00020000);
    const unsigned w0 = __builtin_bit_cast(unsigned, v0), w1 = __builtin_bit_cast(unsigned, v1), w2 = __builtin_bit_cast(unsigned, v2);
    const pl_u32x4 g = {w0, w1, w2, tag + w0 + w1 + w2};
    __builtin_amdgcn_raw_buffer_store_b128(g, rsrc, (int)(idx * 16u), 0, 16);           // sc1, like the 8-byte granules' agent-scope stores
}
// one wave re-reads its N granules f0 + stride k (k < N) until every check word matches; TAIL: the last of them exists only in the lanes with f0 + stride (N - 1) < total
template <int N, bool TAIL>
__device__ __forceinline__ bool sweep16(const void* base, unsigned total, unsigned f0, unsigned stride, unsigned tag, pl_u32x4 (&v)[N], int* err, int code, int* abort_s, int nap) {
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, (int)(total * 16u), 0x00020000);
    const unsigned voff = f0 * 16u;
#pragma unroll 1
    for (unsigned spins = 0;; ++spins) {
        bool ok = true;
#pragma unroll
        for (int k = 0; k < N; ++k) {
            const pl_u32x4 x = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)voff, (int)(k * stride * 16u), (int)(16u | 0x80000000u));
            v[k] = x;
            if (TAIL && k == N - 1) ok = ok && ((x[3] - x[0] - x[1] - x[2] == tag) || (f0 + k * stride >= total));
            else ok = ok && (x[3] - x[0] - x[1] - x[2] == tag);
        }
        if (__all(ok)) return true;
        bool giveup = spins >= PL_SPIN_LIMIT || PL_ABORT_GET(abort_s) != 0;
        if (!giveup && (spins & 1023u) == 1023u) giveup = __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0;
        if (giveup) {
            if ((threadIdx.x & 63) == 0) { atomicCAS(err, 0, code); PL_ABORT_SET(abort_s); }
            return false;
        }
        for (int z = 0; z < nap; ++z) __builtin_amdgcn_s_sleep(2);
    }
}
// granules [K0 NE, (K0 + N) NE) of the act gather -> xs
template <int K0, int N, int NE, int TOT>
__device__ __forceinline__ void act16_chunk(const void* g_act, int e, unsigned tag, __attribute__((address_space(3))) float* xs, int* err, int* abort_s, int nap) {
    pl_u32x4 v[N];
    constexpr bool TAIL = (NE * (K0 + N) > TOT);
    const bool got = sweep16<N, TAIL>(g_act, (unsigned)TOT, (unsigned)(e + NE * K0), (unsigned)NE, tag, v, err, 5, abort_s, nap);
    (void)got;
    int ee = e;
    asm volatile("" : "+v"(ee));                 // (the column arithmetic below is redone per layer instead of living in registers through the layer loop)
#pragma unroll
    for (int k = 0; k < N; ++k) {
        const int f = ee + NE * (K0 + k);
        if (!(TAIL && k == N - 1) || f < TOT) {
            const int P = f / PL_A16_SLOTS, s = f - PL_A16_SLOTS * P;
            __attribute__((address_space(3))) float* const d = xs + 16 * P + 3 * s;
            const unsigned w0 = v[k][0], w1 = v[k][1], w2 = v[k][2];      // (scalars first: __builtin_bit_cast of a vector ELEMENT reads element 0 with this hipcc)
            d[0] = __builtin_bit_cast(float, w0);
            if (s < 5) { d[1] = __builtin_bit_cast(float, w1); d[2] = __builtin_bit_cast(float, w2); }
        }
    }
}

// Before a wave sweeps ALL its granules of an edge it watches one "sentinel" granule per producer workgroup that feeds it (the last one that producer
// stores): 96 eight-byte loads per pass instead of up to 3072.  A hint only -- stores of different lanes land in any order -- the sweep that follows checks
// every tag; bounded, silent (the sweep reports).
template <typename OffFn>
__device__ inline void watch_sentinels(const u64* base, OffFn off, int n, unsigned tag, int lane, int* abort_s) {
    const u64* p0 = base + ((lane < n) ? off(lane) : 0);
    const u64* p1 = base + ((lane + 64 < n) ? off(lane + 64) : 0);
    asm volatile("" : "+v"(p0), "+v"(p1));
#pragma unroll 1
    for (unsigned spins = 0; spins < 8192u; ++spins) {
        bool ok = true;
        if (lane < n) ok = (unsigned)(__hip_atomic_load(p0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >> 32) == tag;
        if (lane + 64 < n) ok = ok && ((unsigned)(__hip_atomic_load(p1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >> 32) == tag);
        if (__all(ok) || PL_ABORT_GET(abort_s) != 0) return;
        __builtin_amdgcn_s_sleep(2);
    }
}

// 6..8 rows: the granule store addresses are loop invariants that hipcc hoists out of the layer loop as 64-bit pairs and then SPILLS (a scratch reload in front of every
// publish: the 7- and 8-row kernels ran 0.63 / 0.70 ms per step).  The element index made opaque at the use site keeps the address formation there: one 32-bit index per
// store stays live instead of a pointer pair.
template <bool ON> __device__ inline int pl_opq(int idx) {
    if constexpr (ON) asm volatile("" : "+v"(idx));
    return idx;
}
__device__ inline float dot4(const f32x4 w, const f32x4 x, float acc) {
    return fmaf(w[3], x[3], fmaf(w[2], x[2], fmaf(w[1], x[1], fmaf(w[0], x[0], acc))));
}
// fp16 engines (round 5): the same per-workgroup image with half weights -- a lane's fragment is its 4 weights of a row as 8 bytes; activations, granules and
// accumulation stay fp32 (no activation is rounded to fp16 on this path, unlike the launch chain's fp16 MFMA operands)
__device__ inline float dot4(const half4 w, const f32x4 x, float acc) {
    return fmaf((float)w[3], x[3], fmaf((float)w[2], x[2], fmaf((float)w[1], x[1], fmaf((float)w[0], x[0], acc))));
}
template <typename WT> struct PlW;
template <> struct PlW<float> { typedef f32x4 frag; };
template <> struct PlW<half_t> { typedef half4 frag; };
// 8 dims of a cached K / V row as a lane holds them
template <typename WT> struct PlKV;
template <> struct PlKV<float> {
    f32x4 a, b;
    __device__ inline void load(const float* p) { a = *(const f32x4*)p; b = *(const f32x4*)(p + 4); }
    __device__ inline float at(int j) const { return j < 4 ? a[j & 3] : b[j & 3]; }
};
template <> struct PlKV<half_t> {
    half8 h;
    __device__ inline void load(const half_t* p) { h = *(const half8*)p; }
    __device__ inline float at(int j) const { return (float)h[j]; }
};
__device__ inline float pl_exp_diff(float m, float mn) { return (m == -INFINITY) ? 0.f : expf(m - mn); }

// ---- the compute waves' row sums, several at a time (round 5).  A wave finishes up to 16 dot products per phase (weight rows x decode rows); each was reduced by its
// own wave_sum (4 DPP steps + 4 v_readlane + the lane-0 store: ~45 instructions, one after the other -- 16 of them were 1.4 us of the 2.5 us gate|up phase at 4 rows).
// Here the P partial sums of a lane are reduced TOGETHER: at each of the 4 DPP steps a lane keeps one half of its values and gives the other half to its partner
// (who keeps that half), so the work halves every step: P/2 + P/4 + ... exchanges instead of 4 P.  The partners are wave_sum's (quad xor 1, quad xor 2, half-mirror,
// mirror) and every value is added in wave_sum's order, so the sums are bit-identical; which half a lane keeps at step s is a parity f_s of its lane bits chosen so that
// the partners of the LATER steps hold the same values: f0 = b0^b2, f1 = b1^b2, f2 = b2^b3, f3 = b3.  After the 4 steps a lane holds the 16-lane row sum of value
// idx = f0 P/2 + f1 P/4 + ...; the 4 rows go to LDS as red[wave][base + idx][row] and the reader adds them (r0 + r1) + (r2 + r3), again wave_sum's order.
constexpr int pl_pow2(int n) { return n <= 1 ? 1 : n <= 2 ? 2 : n <= 4 ? 4 : n <= 8 ? 8 : 16; }
template <int CTRL, int C> __device__ inline void pl_rs_step(float* v, bool f) {
    if constexpr (C > 1) {
#pragma unroll
        for (int i = 0; i < C / 2; ++i) {
            const float keep = f ? v[i + C / 2] : v[i];
            const float give = f ? v[i] : v[i + C / 2];
            v[i] = keep + dpp_f<CTRL>(give);
        }
    } else v[0] += dpp_f<CTRL>(v[0]);
}
#define PL_FRONT_BYTES 4096                 // 16 + (384 + 512 + 16 + 16) * 4 = 3728, rounded
#ifndef PL_TAIL_ALL
#define PL_TAIL_ALL 1
#endif
#ifndef PL_TAIL_IT
#define PL_TAIL_IT 4                        // 6..8 rows: iterations of 8 keys per compute wave that wait in LDS (4 KB each) -- 4 x 8 x 4 = 128 keys per item behind the 192 in registers
#endif
#define PL_RED_FLOATS (8 * 32 * 4)          // [compute wave][value < 32][16-lane row]: two groups of <= 16 values per wave (gate|up: pair 0 | pair 1)
template <int P> __device__ inline void pl_reduce_store(float (&v)[P], float* red, int wave, int lane, int base = 0) {
    static_assert(P == 1 || P == 2 || P == 4 || P == 8 || P == 16, "pad to a power of two");
    constexpr int C1 = P > 1 ? P / 2 : 1, C2 = P > 2 ? P / 4 : 1, C3 = P > 4 ? P / 8 : 1;
    const bool f0 = ((lane ^ (lane >> 2)) & 1) != 0, f1 = (((lane >> 1) ^ (lane >> 2)) & 1) != 0, f2 = (((lane >> 2) ^ (lane >> 3)) & 1) != 0, f3 = ((lane >> 3) & 1) != 0;
    pl_rs_step<DPP_XOR1, P>(v, f0);
    pl_rs_step<DPP_XOR2, C1>(v, f1);
    pl_rs_step<DPP_HALF_MIRROR, C2>(v, f2);
    pl_rs_step<DPP_MIRROR, C3>(v, f3);
    int idx = 0;
    bool writer = true;                               // one lane per (value, row): the lanes whose unused parities are 0
    if (P > 1) idx += f0 ? P / 2 : 0; else writer = writer && !f0;
    if (C1 > 1) idx += f1 ? C1 / 2 : 0; else writer = writer && !f1;
    if (C2 > 1) idx += f2 ? C2 / 2 : 0; else writer = writer && !f2;
    if (C3 > 1) idx += f3 ? C3 / 2 : 0; else writer = writer && !f3;
    if (writer) red[(wave * 32 + base + idx) * 4 + (lane >> 4)] = v[0];
}
__device__ inline float pl_red(const float* red, int wave, int n) {
    const f32x4 p = *(const f32x4*)(red + (wave * 32 + n) * 4);
    return (p[0] + p[1]) + (p[2] + p[3]);
}
#define PL_PV(N_) float pv[pl_pow2(N_)]; _Pragma("unroll") for (int n_ = 0; n_ < pl_pow2(N_); ++n_) pv[n_] = 0.f

// ---- per-utterance LoRA adapters inside the launch (round 6; LORA kernels, paced schedule).  Reference: peft's merge as the pipeline applies it
// (pipelines/chattts_plus_pipeline.py:420-432), per row instead of per batch: row r adds scale * B (A h) of its adapter to q / k / v (h = the normalised input)
// and to o_proj (h = the attention output).  A merged weight image per adapter would be 755 MB and a different image per row; the low-rank form is 2 x 16 x 768 MACs
// per target and row -- work for the two compute waves (6, 7) that own no q|k|v and no o_proj rows:
//   * wave 2 b + w of the 384 such waves computes ONE u = A[k] . h (16 ranks x 3 targets x R rows <= 384; o_proj: 16 R <= 384), its A row requested a phase ahead
//     into the registers that hold q|k|v / o_proj rows in the other waves, and publishes it as a granule;
//   * the same waves request the B entries of their workgroup's OWN 12 q|k|v and 4 o_proj rows a phase ahead and park them in LDS (the scale rides on u);
//   * the edge lanes that finish a row gather the row's 16 u (one more hop per projection: the u must cross workgroups) and add sum_k B[k][n] u[k].
// The launch chain computes the same terms in worker workgroups of its q|k|v / o_proj launches (lora_worker.h).
__device__ inline int pl_slot(unsigned long long slots, int r) { return (int)(signed char)(unsigned char)(slots >> (8 * r)); }
template <typename WT> struct PlLoraRegs {                // fp16 engines: the adapters are fp32 -- registers of their own (those kernels have room)
    f32x4 qa_[3], qb_[3], oa_[3], ob_;
    template <typename Q> __device__ inline f32x4& qa(Q&, int j) { return qa_[j]; }
    template <typename Q> __device__ inline f32x4& qb(Q&, int j) { return qb_[j]; }
    template <typename O> __device__ inline f32x4& oa(O&, int j) { return oa_[j]; }
    template <typename Q> __device__ inline f32x4& ob(Q&) { return ob_; }
};
template <> struct PlLoraRegs<float> {                    // fp32 engines: waves 6 and 7 never hold q|k|v or o_proj rows -- their q_w / o_w registers carry the adapter operands
    __device__ inline f32x4& qa(f32x4 (&q_w)[2][3], int j) { return q_w[0][j]; }
    __device__ inline f32x4& qb(f32x4 (&q_w)[2][3], int j) { return q_w[1][j]; }
    __device__ inline f32x4& oa(f32x4 (&o_w)[3], int j) { return o_w[j]; }
    __device__ inline f32x4& ob(f32x4 (&q_w)[2][3]) { return q_w[0][0]; }      // (requested after phase A, when the q|k|v operands are spent)
};
// one wave re-reads N consecutive granules per lane until the tags match; lanes with valid == false pass (every lane of the wave calls this)
template <int N>
__device__ inline bool sweep_masked(const u64* base, unsigned lane_elem, bool valid, unsigned tag, float (&v)[N], int* err, int code, int* abort_s, int nap) {
    const u64* p = base + (valid ? lane_elem : 0u);
    asm volatile("" : "+v"(p));
#pragma unroll
    for (int k = 0; k < N; ++k) v[k] = 0.f;
    if (!__any(valid)) return true;                       // (a wave without such a lane: no loads at all -- 192 workgroups poll the same few lines)
#pragma unroll 1
    for (unsigned spins = 0;; ++spins) {
        bool ok = true;
        if (valid) {
#pragma unroll
            for (int k = 0; k < N; ++k) {
                const u64 x = __hip_atomic_load(p + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                v[k] = __builtin_bit_cast(float, (unsigned)x);
                ok = ok && ((unsigned)(x >> 32) == tag);
            }
        }
        if (__all(ok)) return true;
        bool giveup = spins >= PL_SPIN_LIMIT || PL_ABORT_GET(abort_s) != 0;
        if (!giveup && (spins & 1023u) == 1023u) giveup = __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0;
        if (giveup) {
            if ((threadIdx.x & 63) == 0) { atomicCAS(err, 0, code); PL_ABORT_SET(abort_s); }
            return false;
        }
        for (int z = 0; z < nap; ++z) __builtin_amdgcn_s_sleep(2);
    }
}

// SCHED = when a compute wave requests its weight arrays (each array is needed one phase per layer: q|k|v rows in A, o_proj rows in C, gate|up in D, down in E):
//   1: gate|up(l) + q|k|v(l+1) when layer l's attention wait begins, down(l) + o_proj(l+1) when its (x + attention) wait begins;
//   2: o_proj(l) + gate|up(l) at the attention wait, down(l) at the (x + attention) wait, q|k|v(l+1) at the end of the layer -- nothing is requested
//      more than one wait ahead, so at most gate|up + down (18 of the 27 fragments) are live at once: 36 fewer VGPRs (the 2- to 4-row kernels spill under 1).
// (Re-requesting every array right after its use, a whole layer ahead, put the gate|up burst in front of the act gather's polls and the down burst in
//  front of the next layer's x gather: +6.7 us per layer, profiles/r04_persist_probe_v2_one_launch.jsonl.)
template <int R, int SCHED, typename WT, bool LORA = false>
__global__ __launch_bounds__(PL_LB) void persist_layer_kernel(const PersistArgs a) {
    static_assert(!LORA || SCHED == 3, "per-utterance adapters: the paced schedule only");
    typedef typename PlW<WT>::frag wfrag;
    // Edge waves per workgroup: 2 at one row, 4 at 2..4 rows (round 5).  A gather is 768 R (3072 R for the act rows) granules over the edge lanes; with two waves the
    // per-lane share grew with the rows (6 R and 24 R loads per lane, the act rows one sweep after the other) -- +4.3 us of the +9.2 us per layer between 1 and 4 rows.
    constexpr bool A16 = PL_ACT16 != 0 && R <= PL_A16_MAXR;      // 16-byte act granules (measured: -1.5 % per step at 1..5 rows, +1 % / +4.5 % at 6 / 8 rows)
    constexpr int EW = PL_EDGE_WAVES(R), NE = 64 * EW, GX = 768 / NE;      // edge lanes; granules per lane and row of a 768-wide gather (6 or 3)
    constexpr size_t BLOCK_BYTES = (size_t)PL_BLOCK_BYTES / 4 * sizeof(WT), LAYER_BYTES = PL_LAYER_BYTES / 4 * sizeof(WT);
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // LDS: [front block: give-up flag + the attention workgroups' scratch][GEMV workgroups: xs | red | ssq | xres   /   attention workgroups of the 6..8-row kernels: the K / V tail]
    int* const abort_s = (int*)smem;                      // [4]: [0] give-up flag, [1] SCHED 3: gathers completed by the edge waves (2 per phase)
    float* const att_s = (float*)(abort_s + 4);           // attention workgroups: q[2][64] | k_new[2][64] | v_new[2][64] | the waves' outputs [8][64], maxima [8], sums [8]
    float* const xs = (float*)(smem + PL_FRONT_BYTES);    // activations of the current phase [R][768] ([R][3072] for the down projection)
    float* const red = xs + R * PL_I;                     // compute waves' results [8 waves][16 values][4 rows of 16 lanes] (pl_reduce_store / pl_red)
    float* const ssq = red + PL_RED_FLOATS;                   // sums of squares of the gathered rows [edge wave <= 4][R]
    float* const xres = ssq + 4 * R;                      // this workgroup's 4 columns of the residual stream [R][4] (x, later x + attention, then the layer output)
    float* const lbs = xres + 4 * R;                      // LORA: B of this workgroup's own rows [R][12 q|k|v rows][16] | [R][4 o_proj rows][16]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int b = blockIdx.x;
    // the three words every workgroup needs first travel together (vector loads with an opaque offset: a scalar load + branch each would be three
    // serial cache-miss round trips in front of everything else -- measured 4.4 us before the first useful load in the first version)
    const int done_v = vload_flag(a.done), err_v = vload_flag(a.error), ep_v = vload_flag((const int*)a.epoch);
    const int NL = a.n_layers;
    unsigned long long t_mark[10];
#define PL_MARK(i) do { if (a.ts != nullptr) t_mark[i] = wall_clock64(); } while (0)
    PL_MARK(0);
    if (tid == 0) { abort_s[0] = 0; abort_s[1] = 0; }

    if (b < PL_GEMV_BLOCKS) {
        const char* wb = a.w + (size_t)b * BLOCK_BYTES;
        if (wave < 8) {
            // ------------------------------------------------ compute wave: weights in registers, products on the VALU.
            // The four weight arrays form a ring over the layers: each is re-requested for layer l + 1 right after its last use in layer l, so the
            // weight stream runs a whole layer ahead of the dependency edges without a single extra register.
            const size_t oq = (size_t)wave * (2 * 3 * 64) + lane;                                                  // [row 2][j 3][lane]
            const size_t oo = (size_t)PL_QKV_BYTES / 16 + (size_t)wave * (3 * 64) + lane;                          // [j 3][lane]
            const size_t og = (size_t)(PL_QKV_BYTES + PL_O_BYTES) / 16 + (size_t)wave * (4 * 3 * 64) + lane;       // [pair 2][gate|up][j 3][lane]
            const size_t od = (size_t)(PL_QKV_BYTES + PL_O_BYTES + PL_GU_BYTES) / 16 + (size_t)wave * (6 * 64) + lane;   // [j 6][lane]
            wfrag q_w[2][3], o_w[3], g_w[4][3], d_w[6];
#define PL_LOAD_Q(base_) do { if (wave < 6) { _Pragma("unroll") for (int row = 0; row < 2; ++row) _Pragma("unroll") for (int j = 0; j < 3; ++j) \
                q_w[row][j] = __builtin_nontemporal_load((const wfrag*)(base_) + oq + (row * 3 + j) * 64); } } while (0)
#define PL_LOAD_O(base_) do { if (wave < 4) { _Pragma("unroll") for (int j = 0; j < 3; ++j) o_w[j] = __builtin_nontemporal_load((const wfrag*)(base_) + oo + j * 64); } } while (0)
#define PL_LOAD_G(base_) do { _Pragma("unroll") for (int s = 0; s < 4; ++s) _Pragma("unroll") for (int j = 0; j < 3; ++j) \
                g_w[s][j] = __builtin_nontemporal_load((const wfrag*)(base_) + og + (s * 3 + j) * 64); } while (0)
#define PL_LOAD_D(base_) do { _Pragma("unroll") for (int j = 0; j < 6; ++j) d_w[j] = __builtin_nontemporal_load((const wfrag*)(base_) + od + j * 64); } while (0)
            PL_LOAD_Q(wb);
            if (SCHED == 1) PL_LOAD_O(wb);
            // ---- LORA (waves 6, 7): this wave's (row, target, rank) of the q|k|v terms, (row, rank) of the o_proj terms; the workgroup's own B entries
            PlLoraRegs<WT> lr;
            const unsigned tag0c = (unsigned)ep_v * 32u;
            const int lw = 2 * b + (wave - 6), ll = (wave - 6) * 64 + lane;             // one of 384 such waves; one of the workgroup's 128 such lanes
            const int lq_r = lw / 48, lq_tk = lw - 48 * lq_r;                            // q|k|v: row, 16 target + rank
            const int lq_sl = (LORA && wave >= 6 && lq_r < R) ? pl_slot(a.lslots, lq_r) : -1;
            const int lo_r = lw >> 4, lo_k = lw & 15;                                    // o_proj: row, rank
            const int lo_sl = (LORA && wave >= 6 && lo_r < R) ? pl_slot(a.lslots, lo_r) : -1;
            constexpr int LQM = (192 * R + 127) / 128, LOM = (64 * R + 127) / 128;       // B entries per lane: q|k|v (<= 12), o_proj (<= 4)
            // element i = ll + 128 m of [R][12 own rows][16 ranks]: own row j = (pw, half) as the edge lanes finish them (pw < 4: q / k dims 2 jj + (pw & 1) (+ 32); else v dims 4 jj + 2 (pw - 4) (+ 1))
#define PL_LORA_REQ_QKV(l_) do { if (wave >= 6) { \
                if (lq_sl >= 0) { const f32x4* ap_ = (const f32x4*)(a.la_qkv + (size_t)(l_) * a.la_qkv_stride + (size_t)(lq_sl * 48 + lq_tk) * PL_H); \
                    _Pragma("unroll") for (int j = 0; j < 3; ++j) lr.qa(q_w, j) = ap_[64 * j + lane]; \
                    lq_scale = a.lscale[((l_) * 8 + lq_sl) * 4 + (lq_tk >> 4)]; } \
                int li_ = ll; asm volatile("" : "+v"(li_)); \
                _Pragma("unroll") for (int m = 0; m < LQM; ++m) { const int i_ = li_ + 128 * m; float v_ = 0.f; \
                    if (i_ < 192 * R) { const int r_ = i_ / 192, rem_ = i_ - 192 * r_, j_ = rem_ >> 4, k_ = rem_ & 15, sl_ = pl_slot(a.lslots, r_); \
                        if (sl_ >= 0) { const int pw_ = j_ >> 1, hf_ = j_ & 1, hh_ = b >> 4, jj_ = b & 15; \
                            const int t_ = pw_ < 4 ? (pw_ >> 1) : 2, d_ = pw_ < 4 ? 2 * jj_ + (pw_ & 1) + 32 * hf_ : 4 * jj_ + 2 * (pw_ - 4) + hf_; \
                            v_ = a.lb[(size_t)(l_) * a.la_stride + (size_t)((sl_ * 4 + t_) * 16 + k_) * PL_H + hh_ * 64 + d_]; } } \
                    lr.qb(q_w, m >> 2)[m & 3] = v_; } } } while (0)
#define PL_LORA_REQ_O(l_) do { if (wave >= 6) { \
                if (lo_sl >= 0) { const f32x4* ap_ = (const f32x4*)(a.la + (size_t)(l_) * a.la_stride + (size_t)((lo_sl * 4 + 3) * 16 + lo_k) * PL_H); \
                    _Pragma("unroll") for (int j = 0; j < 3; ++j) lr.oa(o_w, j) = ap_[64 * j + lane]; \
                    lo_scale = a.lscale[((l_) * 8 + lo_sl) * 4 + 3]; } \
                int li_ = ll; asm volatile("" : "+v"(li_)); \
                _Pragma("unroll") for (int m = 0; m < LOM; ++m) { const int i_ = li_ + 128 * m; float v_ = 0.f; \
                    if (i_ < 64 * R) { const int r_ = i_ >> 6, ii_ = (i_ >> 4) & 3, k_ = i_ & 15, sl_ = pl_slot(a.lslots, r_); \
                        if (sl_ >= 0) v_ = a.lb[(size_t)(l_) * a.la_stride + (size_t)((sl_ * 4 + 3) * 16 + k_) * PL_H + 4 * b + ii_]; } \
                    lr.ob(q_w)[m] = v_; } } } while (0)
            float lq_scale = 0.f, lo_scale = 0.f;          // (the scale rides on u: nothing is computed on a requested value before its use, a request never waits)
            if constexpr (LORA) PL_LORA_REQ_QKV(0);
            __builtin_amdgcn_sched_barrier(0);
            if (__builtin_amdgcn_readfirstlane(done_v | err_v)) return;      // every sequence finished (gpt.py:545) / an earlier launch gave up: the same for every workgroup
            __syncthreads();                                  // S0
            if constexpr (SCHED == 3) {
                // Paced requests: instead of sleeping in a barrier while the edge waves gather, a compute wave requests ONE weight fragment (1 KB) every
                // ~0.27 us until the gather is complete (then the rest at once).  8 waves x 1 KB per interval = the rate one CU drains anyway, so the
                // queue in front of the edge waves' polls stays a few KB deep instead of a 50-100 KB burst.  Each array is requested during the wait
                // that precedes the phase BEFORE its use: o_proj(l) while x is gathered, gate|up(l) during the attention wait, down(l) during the
                // (x + attention) wait, q|k|v(l+1) during the act wait -- at most 18 of the 27 fragments are live at once.
                typedef __attribute__((address_space(3))) int lds_int;
                lds_int* const arrive = (lds_int*)(abort_s + 1);
                int phase = 0;
                const int pace = a.pace;                      // x 128 cycles between two requests of a wave
#define PL_PACE_BEGIN() const int tgt_ = EW * (++phase); bool ready_ = __hip_atomic_load(arrive, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) >= tgt_
#define PL_PIECE(stmt_) do { stmt_; if (!ready_) { for (int z_ = 0; z_ < pace; ++z_) __builtin_amdgcn_s_sleep(2); ready_ = __hip_atomic_load(arrive, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) >= tgt_; } } while (0)
#define PL_PACE_END() do { while (!ready_) { __builtin_amdgcn_s_sleep(1); ready_ = __hip_atomic_load(arrive, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) >= tgt_; } asm volatile("" ::: "memory"); } while (0)
                for (int l = 0; l < NL; ++l) {
                    const char* const wn = wb + LAYER_BYTES;
                    const bool more = l + 1 < NL;
                    // ---- wait for x, requesting o_proj(l); phase A
                    {
                        PL_PACE_BEGIN();
                        if (wave < 4) {
#pragma unroll
                            for (int j = 0; j < 3; ++j) PL_PIECE(o_w[j] = __builtin_nontemporal_load((const wfrag*)wb + oo + j * 64));
                        }
                        PL_PACE_END();
                    }
                    if constexpr (LORA) if (wave >= 6) {
                        if (lq_sl >= 0) {
                            float acc = 0.f;
#pragma unroll
                            for (int j = 0; j < 3; ++j) acc = dot4(lr.qa(q_w, j), ((const f32x4*)(xs + lq_r * PL_H))[64 * j + lane], acc);
                            acc = wave_sum(acc);
                            const float rs = 1.0f / sqrtf((EW == 2 ? ssq[lq_r] + ssq[R + lq_r] : (ssq[lq_r] + ssq[R + lq_r]) + (ssq[2 * R + lq_r] + ssq[3 * R + lq_r])) / (float)PL_H + a.eps);
                            if (lane == 0) store_granule(a.g_u + lq_r * 64 + lq_tk, tag0c + (unsigned)l, (acc * rs) * lq_scale);
                        }
#pragma unroll
                        for (int m = 0; m < LQM; ++m) if (ll + 128 * m < 192 * R) lbs[ll + 128 * m] = lr.qb(q_w, m >> 2)[m & 3];
                    }
                    if (wave < 6) {
                        PL_PV(2 * R);
#pragma unroll
                        for (int r = 0; r < R; ++r) {              // (one decode row's operands at a time: R x 12 registers of activations otherwise)
                            f32x4 xr[3];
#pragma unroll
                            for (int j = 0; j < 3; ++j) xr[j] = ((const f32x4*)(xs + r * PL_H))[64 * j + lane];
#pragma unroll
                            for (int row = 0; row < 2; ++row) {
                                float acc = 0.f;
#pragma unroll
                                for (int j = 0; j < 3; ++j) acc = dot4(q_w[row][j], xr[j], acc);
                                pv[row * R + r] = acc;
                            }
                        }
                        pl_reduce_store<pl_pow2(2 * R)>(pv, red, wave, lane);
                    }
                    __syncthreads();                          // B2(A)
                    if constexpr (LORA) PL_LORA_REQ_O(l);     // (the q|k|v operands are spent; the attention takes > 2 us)
                    // ---- wait for the attention output, requesting gate|up(l); phase C
                    {
                        PL_PACE_BEGIN();
#pragma unroll
                        for (int s = 0; s < 4; ++s)
#pragma unroll
                            for (int j = 0; j < 3; ++j) PL_PIECE(g_w[s][j] = __builtin_nontemporal_load((const wfrag*)wb + og + (s * 3 + j) * 64));
                        PL_PACE_END();
                    }
                    if constexpr (LORA) if (wave >= 6) {
                        if (lo_sl >= 0) {
                            float acc = 0.f;
#pragma unroll
                            for (int j = 0; j < 3; ++j) acc = dot4(lr.oa(o_w, j), ((const f32x4*)(xs + lo_r * PL_H))[64 * j + lane], acc);
                            acc = wave_sum(acc);
                            if (lane == 0) store_granule(a.g_u + lo_r * 64 + 48 + lo_k, tag0c + (unsigned)l, acc * lo_scale);
                        }
#pragma unroll
                        for (int m = 0; m < LOM; ++m) if (ll + 128 * m < 64 * R) lbs[192 * R + ll + 128 * m] = lr.ob(q_w)[m];
                    }
                    if (wave < 4) {
                        PL_PV(R);
#pragma unroll
                        for (int r = 0; r < R; ++r) {
                            float acc = 0.f;
#pragma unroll
                            for (int j = 0; j < 3; ++j) acc = dot4(o_w[j], ((const f32x4*)(xs + r * PL_H))[64 * j + lane], acc);
                            pv[r] = acc;
                        }
                        pl_reduce_store<pl_pow2(R)>(pv, red, wave, lane);
                    }
                    __syncthreads();                          // B2(C)
                    // ---- wait for x + attention, requesting down(l); phase D
                    {
                        PL_PACE_BEGIN();
#pragma unroll
                        for (int j = 0; j < 6; ++j) PL_PIECE(d_w[j] = __builtin_nontemporal_load((const wfrag*)wb + od + j * 64));
                        PL_PACE_END();
                    }
#ifdef PL_EXP_E2
                    if constexpr (R == 5) {
#else
                    if constexpr (R >= 5) {
#endif
                        // (register allocation, measured: at 5 rows this form compiles without spills and the one below with 27-55; at 4 rows it is the other way round)
                        float pv[2][pl_pow2(2 * R)];
#pragma unroll
                        for (int n = 0; n < pl_pow2(2 * R); ++n) { pv[0][n] = 0.f; pv[1][n] = 0.f; }
#pragma unroll
                        for (int r = 0; r < R; ++r) {
                            f32x4 xr[3];
#pragma unroll
                            for (int j = 0; j < 3; ++j) xr[j] = ((const f32x4*)(xs + r * PL_H))[64 * j + lane];
#pragma unroll
                            for (int s = 0; s < 4; ++s) {
                                float acc = 0.f;
#pragma unroll
                                for (int j = 0; j < 3; ++j) acc = dot4(g_w[s][j], xr[j], acc);
                                pv[s >> 1][(s & 1) * R + r] = acc;
                            }
                        }
                        pl_reduce_store<pl_pow2(2 * R)>(pv[0], red, wave, lane, 0);
                        pl_reduce_store<pl_pow2(2 * R)>(pv[1], red, wave, lane, 16);
                    } else
#pragma unroll
                    for (int g = 0; g < 2; ++g) {
                        // the wave's two (gate, up) pairs, one after the other: a group of 2 R values each (value (s, r) = red[wave][16 (s / 2) + (s % 2) R + r])
                        PL_PV(2 * R);
#pragma unroll
                        for (int r = 0; r < R; ++r) {
                            f32x4 xr[3];
#pragma unroll
                            for (int j = 0; j < 3; ++j) xr[j] = ((const f32x4*)(xs + r * PL_H))[64 * j + lane];
#pragma unroll
                            for (int t = 0; t < 2; ++t) {
                                float acc = 0.f;
#pragma unroll
                                for (int j = 0; j < 3; ++j) acc = dot4(g_w[2 * g + t][j], xr[j], acc);
                                pv[t * R + r] = acc;
                            }
                        }
                        pl_reduce_store<pl_pow2(2 * R)>(pv, red, wave, lane, 16 * g);
                    }
                    __syncthreads();                          // B2(D)
                    // ---- wait for silu(gate) * up, requesting q|k|v(l+1); phase E
                    if constexpr (LORA) if (more) PL_LORA_REQ_QKV(l + 1);
                    {
                        PL_PACE_BEGIN();
                        if (more && wave < 6) {
#pragma unroll
                            for (int row = 0; row < 2; ++row)
#pragma unroll
                                for (int j = 0; j < 3; ++j) PL_PIECE(q_w[row][j] = __builtin_nontemporal_load((const wfrag*)wn + oq + (row * 3 + j) * 64));
                        } else if (!more && a.heads && wave < 7) {
                            // the stack ends here: the q|k|v registers take this workgroup's 14 head rows (2 per wave) for the phase behind the last layer
                            const wfrag* const hwb = (const wfrag*)a.hw + (size_t)b * PL_HEAD_FRAGS + oq;
#pragma unroll
                            for (int row = 0; row < 2; ++row)
#pragma unroll
                                for (int j = 0; j < 3; ++j) PL_PIECE(q_w[row][j] = __builtin_nontemporal_load(hwb + (row * 3 + j) * 64));
                        }
                        PL_PACE_END();
                    }
                    const int half = wave & 1;
                    {
                        PL_PV(R);
#pragma unroll
                        for (int r = 0; r < R; ++r) {
                            float acc = 0.f;
#pragma unroll
                            for (int j = 0; j < 6; ++j) acc = dot4(d_w[j], ((const f32x4*)(xs + r * PL_I))[384 * half + 64 * j + lane], acc);
                            pv[r] = acc;
                        }
                        pl_reduce_store<pl_pow2(R)>(pv, red, wave, lane);
                    }
                    __syncthreads();                          // B2(E)
                    wb = wn;
                }
                if (a.heads) {
                    // ---- phase H: final RMSNorm + the 4 folded heads (gpt.py:422-447) on the last layer's output, gathered like a layer input
                    {
                        PL_PACE_BEGIN();
                        PL_PACE_END();
                    }
                    f32x4 xr[R][3];
#pragma unroll
                    for (int r = 0; r < R; ++r)
#pragma unroll
                        for (int j = 0; j < 3; ++j) xr[r][j] = ((const f32x4*)(xs + r * PL_H))[64 * j + lane];
                    if (wave < 7) {
                        PL_PV(2 * R);
#pragma unroll
                        for (int row = 0; row < 2; ++row)
#pragma unroll
                            for (int r = 0; r < R; ++r) {
                                float acc = 0.f;
#pragma unroll
                                for (int j = 0; j < 3; ++j) acc = dot4(q_w[row][j], xr[r][j], acc);
                                pv[row * R + r] = acc;
                            }
                        pl_reduce_store<pl_pow2(2 * R)>(pv, red, wave, lane);
                    }
                    __syncthreads();                          // B2(H)
                }
                return;
            }
            for (int l = 0; l < NL; ++l) {
                const char* const wn = wb + LAYER_BYTES;      // next layer's image
                const bool more = l + 1 < NL;
                if (SCHED == 2) {
                    // these three are requested and used inside one layer: say so (the conditional requests otherwise make them loop-carried and all 27
                    // fragments stay allocated through the whole loop)
                    const wfrag z = {};
#pragma unroll
                    for (int j = 0; j < 3; ++j) o_w[j] = z;
#pragma unroll
                    for (int s = 0; s < 4; ++s)
#pragma unroll
                        for (int j = 0; j < 3; ++j) g_w[s][j] = z;
#pragma unroll
                    for (int j = 0; j < 6; ++j) d_w[j] = z;
                }
                // ---- phase A: q | k | v rows
                __syncthreads();                              // B1(A): xs = x
                f32x4 xr[R][3];
#pragma unroll
                for (int r = 0; r < R; ++r)
#pragma unroll
                    for (int j = 0; j < 3; ++j) xr[r][j] = ((const f32x4*)(xs + r * PL_H))[64 * j + lane];
                if (wave < 6) {
                    PL_PV(2 * R);
#pragma unroll
                    for (int row = 0; row < 2; ++row)
#pragma unroll
                        for (int r = 0; r < R; ++r) {
                            float acc = 0.f;
#pragma unroll
                            for (int j = 0; j < 3; ++j) acc = dot4(q_w[row][j], xr[r][j], acc);
                            pv[row * R + r] = acc;
                        }
                    pl_reduce_store<pl_pow2(2 * R)>(pv, red, wave, lane);
                }
                if (SCHED == 2) PL_LOAD_O(wb);
                PL_LOAD_G(wb);
                if (SCHED == 1 && more) PL_LOAD_Q(wn);
                __builtin_amdgcn_sched_barrier(0);
                __syncthreads();                              // B2(A)
                // ---- phase C: o_proj rows
                __syncthreads();                              // B1(C): xs = attention output
#pragma unroll
                for (int r = 0; r < R; ++r)
#pragma unroll
                    for (int j = 0; j < 3; ++j) xr[r][j] = ((const f32x4*)(xs + r * PL_H))[64 * j + lane];
                if (wave < 4) {
                    PL_PV(R);
#pragma unroll
                    for (int r = 0; r < R; ++r) {
                        float acc = 0.f;
#pragma unroll
                        for (int j = 0; j < 3; ++j) acc = dot4(o_w[j], xr[r][j], acc);
                        pv[r] = acc;
                    }
                    pl_reduce_store<pl_pow2(R)>(pv, red, wave, lane);
                }
                PL_LOAD_D(wb);
                if (SCHED == 1 && more) PL_LOAD_O(wn);
                __builtin_amdgcn_sched_barrier(0);
                __syncthreads();                              // B2(C)
                // ---- phase D: gate | up pairs
                __syncthreads();                              // B1(D): xs = x + attention
#pragma unroll
                for (int r = 0; r < R; ++r)
#pragma unroll
                    for (int j = 0; j < 3; ++j) xr[r][j] = ((const f32x4*)(xs + r * PL_H))[64 * j + lane];
                {
                    float pv[2][pl_pow2(2 * R)];
#pragma unroll
                    for (int n = 0; n < pl_pow2(2 * R); ++n) { pv[0][n] = 0.f; pv[1][n] = 0.f; }
#pragma unroll
                    for (int s = 0; s < 4; ++s)
#pragma unroll
                        for (int r = 0; r < R; ++r) {
                            float acc = 0.f;
#pragma unroll
                            for (int j = 0; j < 3; ++j) acc = dot4(g_w[s][j], xr[r][j], acc);
                            pv[s >> 1][(s & 1) * R + r] = acc;
                        }
                    pl_reduce_store<pl_pow2(2 * R)>(pv[0], red, wave, lane, 0);
                    pl_reduce_store<pl_pow2(2 * R)>(pv[1], red, wave, lane, 16);
                }
                __builtin_amdgcn_sched_barrier(0);
                __syncthreads();                              // B2(D)
                // ---- phase E: down rows, two K halves per row
                __syncthreads();                              // B1(E): xs = silu(gate) * up, [R][3072]
                const int half = wave & 1;
                {
                    PL_PV(R);
#pragma unroll
                    for (int r = 0; r < R; ++r) {
                        float acc = 0.f;
#pragma unroll
                        for (int j = 0; j < 6; ++j) acc = dot4(d_w[j], ((const f32x4*)(xs + r * PL_I))[384 * half + 64 * j + lane], acc);
                        pv[r] = acc;
                    }
                    pl_reduce_store<pl_pow2(R)>(pv, red, wave, lane);
                }
                if (SCHED == 2 && more) PL_LOAD_Q(wn);
                __builtin_amdgcn_sched_barrier(0);
                __syncthreads();                              // B2(E)
                wb = wn;
            }
        } else {
            // ------------------------------------------------ edge wave: gathers, epilogues, publishing
#define PL_B1() do { if (SCHED == 3) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); \
        if (lane == 0) __hip_atomic_fetch_add((__attribute__((address_space(3))) int*)(abort_s + 1), 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); } \
        else __syncthreads(); } while (0)

            const int ew = wave - 8, e = ew * 64 + lane;      // 0..127
            const int hh = b >> 4, jj = b & 15;               // q | k | v rows of this workgroup: head hh, dims (2jj, 2jj+1, +32) / v dims 4jj..4jj+3
            // epilogue operands of phase A (the same in every layer), requested now
            const int tA = e, pwA = tA % 6, rA = (tA / 6 < R) ? tA / 6 : 0;
            const bool doA = tA < 6 * R;
            RowMeta mA = {0, 0, 0, 0};
            float cA = 1.f, sA = 0.f;
            if (doA) {
                mA = a.meta[rA];
                if (pwA < 4) {
                    const int d = 2 * jj + (pwA & 1);
                    cA = a.rope_rows[(size_t)rA * 64 + d];
                    sA = a.rope_rows[(size_t)rA * 64 + 32 + d];
                }
            }
            // the residual stream: [R][768] from global memory (written by the previous launch) -> xs, sums of squares, own 4 columns
            constexpr int XIT = (R * 192 + NE - 1) / NE;
            f32x4 xv[XIT];
#pragma unroll
            for (int i = 0; i < XIT; ++i) {
                const int idx = e + NE * i;
                xv[i] = (idx < R * 192) ? ((const f32x4*)a.x)[idx] : (f32x4){0.f, 0.f, 0.f, 0.f};
            }
            f32x4 xown = {0.f, 0.f, 0.f, 0.f};
            if (e < R) xown = *(const f32x4*)(a.x + (size_t)e * PL_H + 4 * b);
            if (__builtin_amdgcn_readfirstlane(done_v | err_v)) return;
            const unsigned tag0 = (unsigned)ep_v * 32u;       // + layer: one tag per (launch, layer); the launch counter advances at the very end
            if (e < R) *(f32x4*)(xres + 4 * e) = xown;
            float ssp[R];
#pragma unroll
            for (int r = 0; r < R; ++r) ssp[r] = 0.f;
#pragma unroll
            for (int i = 0; i < XIT; ++i) {
                const int idx = e + NE * i;
                if (idx < R * 192) ((f32x4*)xs)[idx] = xv[i];
                const float d = xv[i][0] * xv[i][0] + xv[i][1] * xv[i][1] + xv[i][2] * xv[i][2] + xv[i][3] * xv[i][3];
#pragma unroll
                for (int r = 0; r < R; ++r) ssp[r] += (idx / 192 == r) ? d : 0.f;
            }
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const float s = wave_sum(ssp[r]);
                if (lane == 0) ssq[ew * R + r] = s;
            }
            __syncthreads();                                  // S0
            PL_MARK(1);
            const size_t kv_per = a.kv_per;                   // floats between K and V of a layer (and half the distance between layers)
            for (int l = 0; l < NL; ++l) {
                const unsigned tag = tag0 + (unsigned)l;
                const bool last = l + 1 == NL;
                if (l > 0) {
                    // ---- edge 1: the previous layer's output, published by the 192 GEMV workgroups -> xs, sums of squares
                    for (int z = 0; z < a.delay_x; ++z) __builtin_amdgcn_s_sleep(2);
                    // (producers of this wave's columns e + 128 k: workgroups 16 (2 m + ew) + t; each stores its 4 columns of every row in one instruction)
                    if (EW == 2 && (a.poll & 1)) watch_sentinels(a.g_x, [ew](int i) { return (R - 1) * PL_H + 4 * (16 * (2 * (i >> 4) + ew) + (i & 15)) + 3; }, 96, tag - 1u, lane, abort_s);
                    float v[GX * R];
                    const bool got = PL_SWEEP(GX * R, a.g_x, PL_G_X, (unsigned)e, [](int k) { return (k / GX) * PL_H + NE * (k % GX); }, tag - 1u, v, 1);
                    (void)got;
#pragma unroll
                    for (int r = 0; r < R; ++r) ssp[r] = 0.f;
#pragma unroll
                    for (int k = 0; k < GX * R; ++k) {
                        xs[(k / GX) * PL_H + NE * (k % GX) + e] = v[k];
                        ssp[k / GX] += v[k] * v[k];
                    }
#pragma unroll
                    for (int r = 0; r < R; ++r) {
                        const float s = wave_sum(ssp[r]);
                        if (lane == 0) ssq[ew * R + r] = s;
                    }
                }
                if (last) PL_MARK(2);
                PL_B1();                                      // B1(A): the gather is in LDS
                float lda = 0.f, ldb = 0.f;                   // LORA: this lane's two rows' scale * B (A h) (pipeline:420-432, per row)
                float uq[LORA ? 16 : 1];
                const int slA = (LORA && doA) ? pl_slot(a.lslots, rA) : -1;
                if constexpr (LORA) {                         // (polled while the compute waves work on phase A: the u cross workgroups)
                    for (int z = 0; z < a.delay_u; ++z) __builtin_amdgcn_s_sleep(2);
                    const bool got = sweep_masked<16>(a.g_u, (unsigned)(rA * 64 + (pwA < 4 ? (pwA >> 1) : 2) * 16), slA >= 0, tag, uq, a.error, 9, abort_s, a.nap);
                    (void)got;
                }
                __syncthreads();                              // B2(A)
                if constexpr (LORA) {
                    if (slA >= 0) {
                        const float* const bq = lbs + rA * 192 + 2 * pwA * 16;
#pragma unroll
                        for (int k = 0; k < 16; ++k) { lda = fmaf(bq[k], uq[k], lda); ldb = fmaf(bq[16 + k], uq[k], ldb); }
                    }
                }
                if (doA) {
                    const float rs = 1.0f / sqrtf((EW == 2 ? ssq[rA] + ssq[R + rA] : (ssq[rA] + ssq[R + rA]) + (ssq[2 * R + rA] + ssq[3 * R + rA])) / (float)PL_H + a.eps);          // llama.py:82-87 (the weight is folded into W's columns)
                    const float va = pl_red(red, pwA, rA) * rs + lda, vb = pl_red(red, pwA, R + rA) * rs + ldb;
                    float ya = va, yb = vb;
                    int which, dA, dB;
                    if (pwA < 4) {                               // q / k: RoPE pair (d, d + 32), products rounded separately like the reference (llama.py:180-181)
                        which = pwA >> 1; dA = 2 * jj + (pwA & 1); dB = dA + 32;
                        ya = __fadd_rn(__fmul_rn(va, cA), __fmul_rn(-vb, sA));
                        yb = __fadd_rn(__fmul_rn(vb, cA), __fmul_rn(va, sA));
                    } else {
                        which = 2; dA = 4 * jj + 2 * (pwA - 4); dB = dA + 1;
                    }
                    u64* const gq = a.g_qkv + (size_t)pl_opq<(R > PL_MAXR_ONE)>((rA * PL_NH + hh) * 192 + which * 64);
                    store_granule(gq + dA, tag, ya);
                    store_granule(gq + dB, tag, yb);
                    if (which >= 1) {                            // KV append (llama.py:633): later steps read it from the cache
                        WT* c = (WT*)a.kv + (size_t)l * 2 * kv_per + (which == 2 ? kv_per : 0) + (((size_t)mA.seq * PL_NH + hh) * a.Lmax + mA.slot) * CTTS_HEAD_DIM;
                        c[dA] = (WT)ya; c[dB] = (WT)yb;              // (fp16 engines: plain conversion, like the launch chain's K / V stores)
                    }
                }
                if (last) PL_MARK(3);
                // ---- phase C: attention output -> o_proj + residual
                for (int z = 0; z < a.delay_att; ++z) __builtin_amdgcn_s_sleep(2);      // the attention cannot have published yet: do not poll into the weight stream
                {
                    // (this wave's columns belong to the heads 2 m + ew: one sentinel per (row, head))
                    if (EW == 2 && (a.poll & 1)) watch_sentinels(a.g_att, [ew](int i) { return (i / 6) * PL_H + 64 * (2 * (i % 6) + ew) + 63; }, 6 * R, tag, lane, abort_s);
                    float v[GX * R];
                    const bool got = PL_SWEEP(GX * R, a.g_att, PL_G_ATT, (unsigned)e, [](int k) { return (k / GX) * PL_H + NE * (k % GX); }, tag, v, 3);
                    (void)got;
#pragma unroll
                    for (int k = 0; k < GX * R; ++k) xs[(k / GX) * PL_H + NE * (k % GX) + e] = v[k];
                }
                if (last) PL_MARK(4);
                PL_B1();                                      // B1(C): the gather is in LDS
                float ldo = 0.f;
                float uo[LORA ? 16 : 1];
                const int slO = (LORA && e < 4 * R) ? pl_slot(a.lslots, e >> 2) : -1;
                if constexpr (LORA) {
                    for (int z = 0; z < a.delay_u; ++z) __builtin_amdgcn_s_sleep(2);
                    const bool got = sweep_masked<16>(a.g_u, (unsigned)((e >> 2) * 64 + 48), slO >= 0, tag, uo, a.error, 10, abort_s, a.nap);
                    (void)got;
                }
                __syncthreads();                              // B2(C)
                if constexpr (LORA) {
                    if (slO >= 0) {
                        const float* const bo = lbs + 192 * R + (e >> 2) * 64 + (e & 3) * 16;
#pragma unroll
                        for (int k = 0; k < 16; ++k) ldo = fmaf(bo[k], uo[k], ldo);
                    }
                }
                if (e < 4 * R) {
                    const int i = e & 3, r = e >> 2;
                    const float x1 = xres[4 * r + i] + (pl_red(red, i, r) + ldo);                       // llama.py:731
                    xres[4 * r + i] = x1;
                    if (!(a.fault > 0 && b == 5 && l + 1 == a.fault))          // (test hook "persistent_fault": workgroup 5 withholds its columns in layer fault - 1)
                        store_granule(a.g_x1 + (size_t)pl_opq<(R > PL_MAXR_ONE)>(r * PL_H + 4 * b + i), tag, x1);
                }
                if (last) PL_MARK(5);
                // ---- phase D: x + attention -> RMSNorm, gate | up, SiLU * up
                for (int z = 0; z < a.delay; ++z) __builtin_amdgcn_s_sleep(2);
                {
                    if (EW == 2 && (a.poll & 1)) watch_sentinels(a.g_x1, [ew](int i) { return (R - 1) * PL_H + 4 * (16 * (2 * (i >> 4) + ew) + (i & 15)) + 3; }, 96, tag, lane, abort_s);
                    float v[GX * R];
                    const bool got = PL_SWEEP(GX * R, a.g_x1, PL_G_X1, (unsigned)e, [](int k) { return (k / GX) * PL_H + NE * (k % GX); }, tag, v, 4);
                    (void)got;
#pragma unroll
                    for (int r = 0; r < R; ++r) ssp[r] = 0.f;
#pragma unroll
                    for (int k = 0; k < GX * R; ++k) {
                        xs[(k / GX) * PL_H + NE * (k % GX) + e] = v[k];
                        ssp[k / GX] += v[k] * v[k];
                    }
#pragma unroll
                    for (int r = 0; r < R; ++r) {
                        const float s = wave_sum(ssp[r]);
                        if (lane == 0) ssq[ew * R + r] = s;
                    }
                }
                if (last) PL_MARK(6);
                PL_B1();                                      // B1(D): the gather is in LDS
                __syncthreads();                              // B2(D)
                if constexpr (A16) {
                    float av = 0.f;
                    const int pi = e & 15, r = e >> 4;
                    if (e < 16 * R) {
                        const int w = pi >> 1, p = pi & 1;
                        const float rs = 1.0f / sqrtf((EW == 2 ? ssq[r] + ssq[R + r] : (ssq[r] + ssq[R + r]) + (ssq[2 * R + r] + ssq[3 * R + r])) / (float)PL_H + a.eps);
                        const float gv = pl_red(red, w, 16 * p + r) * rs, uv = pl_red(red, w, 16 * p + R + r) * rs;
                        av = (gv / (1.0f + expf(-gv))) * uv;                                                                    // llama.py:214
                    }
                    // the 16 lanes of a decode row: lane pi = 3 s collects (pi, pi + 1, pi + 2) and stores granule s
                    const float a1 = __shfl_down(av, 1), a2 = __shfl_down(av, 2);
                    if (e < 16 * R && pi % 3 == 0)
                        store_granule16(a.g_act, PL_GEMV_BLOCKS * PL_A16_SLOTS * R, (unsigned)pl_opq<(R > PL_MAXR_ONE)>((r * PL_GEMV_BLOCKS + b) * PL_A16_SLOTS + pi / 3), tag,
                                        av, (pi + 1 < 16) ? a1 : 0.f, (pi + 2 < 16) ? a2 : 0.f);
                } else if (e < 16 * R) {
                    const int pi = e & 15, r = e >> 4, w = pi >> 1, p = pi & 1;
                    const float rs = 1.0f / sqrtf((EW == 2 ? ssq[r] + ssq[R + r] : (ssq[r] + ssq[R + r]) + (ssq[2 * R + r] + ssq[3 * R + r])) / (float)PL_H + a.eps);
                    const float gv = pl_red(red, w, 16 * p + r) * rs, uv = pl_red(red, w, 16 * p + R + r) * rs;
                    store_granule(a.g_act + (size_t)pl_opq<(R > PL_MAXR_ONE)>(r * PL_I + 16 * b + pi), tag, (gv / (1.0f + expf(-gv))) * uv);       // llama.py:214
                }
                if (last) PL_MARK(7);
                // ---- phase E: silu(gate) * up [R][3072] -> down + residual
                for (int z = 0; z < a.delay_act; ++z) __builtin_amdgcn_s_sleep(2);
                // (producers of this wave's columns e + 128 k: workgroups 8 k + 4 ew + t, t < 4; each stores its 16 columns of every row in one instruction)
                if constexpr (A16) {
                    constexpr int TOT = PL_GEMV_BLOCKS * PL_A16_SLOTS * R, NK = (TOT + NE - 1) / NE;      // granules; per lane
                    constexpr int NC = (NK + PL_A16_CHUNK - 1) / PL_A16_CHUNK, CH = (NK + NC - 1) / NC;       // sweeps; granules per lane and sweep (4 registers each)
                    act16_chunk<0, (NK < CH ? NK : CH), NE, TOT>(a.g_act, e, tag, (__attribute__((address_space(3))) float*)xs, a.error, abort_s, a.nap);
                    if constexpr (NC > 1) act16_chunk<CH, (NK - CH < CH ? NK - CH : CH), NE, TOT>(a.g_act, e, tag, (__attribute__((address_space(3))) float*)xs, a.error, abort_s, a.nap);
                    if constexpr (NC > 2) act16_chunk<2 * CH, (NK - 2 * CH < CH ? NK - 2 * CH : CH), NE, TOT>(a.g_act, e, tag, (__attribute__((address_space(3))) float*)xs, a.error, abort_s, a.nap);
                    if constexpr (NC > 3) act16_chunk<3 * CH, (NK - 3 * CH < CH ? NK - 3 * CH : CH), NE, TOT>(a.g_act, e, tag, (__attribute__((address_space(3))) float*)xs, a.error, abort_s, a.nap);
                    if constexpr (NC > 4) act16_chunk<4 * CH, NK - 4 * CH, NE, TOT>(a.g_act, e, tag, (__attribute__((address_space(3))) float*)xs, a.error, abort_s, a.nap);
                    static_assert(NC <= 5, "act gather: more chunks");
                } else {
                if (EW == 2 && (a.poll & 1)) watch_sentinels(a.g_act, [ew](int i) { return (R - 1) * PL_I + 16 * (8 * (i >> 2) + 4 * ew + (i & 3)) + 15; }, 96, tag, lane, abort_s);
                if constexpr (EW == 2) {
#pragma unroll
                    for (int r = 0; r < R; ++r) {
                        float v[24];
                        const bool got = sweep<24>(a.g_act, (unsigned)(r * PL_I + e), [](int k) { return 128 * k; }, tag, v, a.error, 5, abort_s, a.nap);
                        (void)got;
#pragma unroll
                        for (int k = 0; k < 24; ++k) xs[r * PL_I + 128 * k + e] = v[k];
                    }
                } else {
                    // four edge waves: 12 granules per lane and row -> two rows per sweep (24 granules in flight per lane, as before), R / 2 round trips instead of R
#ifdef PL_EXP_E1
                    if constexpr (R > PL_MAXR_ONE) {
#pragma unroll
                        for (int r = 0; r < R; ++r) {
                            float v[12];
                            const bool got = PL_SWEEP(12, a.g_act, PL_G_ACT, (unsigned)(r * PL_I + e), [](int k) { return 256 * k; }, tag, v, 5);
                            (void)got;
#pragma unroll
                            for (int k = 0; k < 12; ++k) xs[r * PL_I + 256 * k + e] = v[k];
                        }
                    } else
#endif
#pragma unroll
                    for (int r = 0; r + 1 < R; r += 2) {
                        float v[24];
                        const bool got = PL_SWEEP(24, a.g_act, PL_G_ACT, (unsigned)(r * PL_I + e), [](int k) { return (k / 12) * PL_I + 256 * (k % 12); }, tag, v, 5);
                        (void)got;
#pragma unroll
                        for (int k = 0; k < 24; ++k) xs[(r + k / 12) * PL_I + 256 * (k % 12) + e] = v[k];
                    }
#ifdef PL_EXP_E1
                    if constexpr ((R & 1) != 0 && R <= PL_MAXR_ONE) {
#else
                    if constexpr ((R & 1) != 0) {
#endif
                        constexpr int r = R - 1;
                        float v[12];
                        const bool got = PL_SWEEP(12, a.g_act, PL_G_ACT, (unsigned)(r * PL_I + e), [](int k) { return 256 * k; }, tag, v, 5);
                        (void)got;
#pragma unroll
                        for (int k = 0; k < 12; ++k) xs[r * PL_I + 256 * k + e] = v[k];
                    }
                }
                }
                if (last) PL_MARK(8);
                PL_B1();                                      // B1(E): the gather is in LDS
                __syncthreads();                              // B2(E)
                if (e < 4 * R) {
                    const int i = e & 3, r = e >> 2;
                    const float x2 = xres[4 * r + i] + (pl_red(red, 2 * i, r) + pl_red(red, 2 * i + 1, r));      // llama.py:739
                    if (last && !a.heads) a.x[(size_t)r * PL_H + 4 * b + i] = x2;          // the heads read it after the launch boundary
                    else {
                        xres[4 * r + i] = x2;
                        store_granule(a.g_x + (size_t)pl_opq<(R > PL_MAXR_ONE)>(r * PL_H + 4 * b + i), tag, x2);
                    }
                }
            }
            if (a.heads) {
                // ---- phase H: the stack's output rows, gathered like a layer's input -> final RMSNorm factor, logits of this workgroup's 14 head rows, hidden columns
                const unsigned tagh = tag0 + (unsigned)(NL - 1);
                // what the hidden row of this thread's (row, column) needs: requested here, consumed behind the gather (held through the layer loop they cost spills)
                int4 hst = make_int4(1, 0, 0, 0);
                int hout = 0;
                float lnf_v = 0.f;
                if (e < 4 * R) {
                    hst = *(const int4*)(a.rows + (e >> 2));         // {fin, end, attempt, limit}
                    hout = a.rows[e >> 2].out;
                    lnf_v = a.lnf[4 * b + (e & 3)];
                }
                for (int z = 0; z < a.delay_x; ++z) __builtin_amdgcn_s_sleep(2);
                float v[GX * R];
                const bool got = PL_SWEEP(GX * R, a.g_x, PL_G_X, (unsigned)e, [](int k) { return (k / GX) * PL_H + NE * (k % GX); }, tagh, v, 1);
                (void)got;
#pragma unroll
                for (int r = 0; r < R; ++r) ssp[r] = 0.f;
#pragma unroll
                for (int k = 0; k < GX * R; ++k) {
                    xs[(k / GX) * PL_H + NE * (k % GX) + e] = v[k];
                    ssp[k / GX] += v[k] * v[k];
                }
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    const float s = wave_sum(ssp[r]);
                    if (lane == 0) ssq[ew * R + r] = s;
                }
                PL_B1();                                      // B1(H)
                __syncthreads();                              // B2(H)
                if (e < PL_HEAD_ROWS * R) {
                    const int rr = e % PL_HEAD_ROWS, r = e / PL_HEAD_ROWS, col = PL_HEAD_ROWS * b + rr;
                    const float rs = 1.0f / sqrtf((EW == 2 ? ssq[r] + ssq[R + r] : (ssq[r] + ssq[R + r]) + (ssq[2 * R + r] + ssq[3 * R + r])) / (float)PL_H + a.eps);          // llama.py:1002 (the weight is folded into the heads' columns)
                    if (col < a.n_valid) a.logits[(size_t)r * a.n_valid + col] = pl_red(red, rr >> 1, (rr & 1) * R + r) * rs;
                }
                if (e < 4 * R) {
                    // hidden = weight * (x * rs) (llama.py:87, gpt.py:422-423) -> hiddens[utterance][its own step] while the row is live
                    float* const hid_out = ((SamplerDynPtr)a.dyn)->hidden_out;
                    if (hid_out != nullptr && hst.x == 0) {
                        const int i = e & 3, r = e >> 2;
                        const float rs = 1.0f / sqrtf((EW == 2 ? ssq[r] + ssq[R + r] : (ssq[r] + ssq[R + r]) + (ssq[2 * R + r] + ssq[3 * R + r])) / (float)PL_H + a.eps);
                        hid_out[(size_t)hout * ((SamplerDynPtr)a.dyn)->hidden_stride + (size_t)hst.y * PL_H + 4 * b + i] = lnf_v * (xres[4 * r + i] * rs);
                    }
                }
            }
            PL_MARK(9);
            if (a.ts != nullptr && e == 0) {
#pragma unroll
                for (int i = 0; i < 10; ++i) a.ts[(size_t)b * 10 + i] = t_mark[i];
            }
            // Advance the launch counter.  Safe although other workgroups may still be running: workgroup 0 gets here only after its last gather, i.e.
            // after every attention item of the last layer was published, i.e. after every workgroup has long read its copy at entry.
            if (b == 0 && e == 0) __hip_atomic_fetch_add(a.epoch, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        return;
    }

    // ---------------------------------------------------- attention workgroup: one (row, head) in every layer
    // a.S key splits per (row, head) for long contexts (host: decode_persist): split s owns a contiguous share of the cached keys; the splits s > 0 publish
    // their partial (max, sum, unnormalised output) and split 0 merges them with this step's own key -- one more hop, but every share stays within the
    // 384 keys a workgroup holds in registers before the query exists (a longer share streams behind the query: +0.6 us per 64 keys and layer)
    // Round 6, 6..8 rows (PAIR): 12 R items exceed the 64 attention workgroups, so a workgroup serves TWO items -- compute waves 0..3 and edge wave 8 the first,
    // waves 4..7 and edge wave 9 the second: 4 x 8 x 6 = 192 keys per item requested before the query exists, the rest of a share streams behind it (no key splits).
    constexpr bool PAIR = R > PL_MAXR_ONE;
    constexpr int NWI = PAIR ? 4 : 8;                         // compute waves per item
    const int S = PAIR ? 1 : a.S;
    // this wave's item: compute waves by their half, edge waves 8 / 9 one each (waves 10, 11 -- and wave 9 outside PAIR -- only keep the barriers)
    const int isub = PAIR ? ((wave < 8) ? (wave >> 2) : ((wave - 8) & 1)) : 0;
    const int iw = PAIR ? (wave & 3) : wave;                  // the wave's index among its item's compute waves
    const int item = PAIR ? 2 * (b - PL_GEMV_BLOCKS) + isub : b - PL_GEMV_BLOCKS;
    if ((PAIR ? 2 * (b - PL_GEMV_BLOCKS) : item) >= PL_NH * R * S) return;      // (12 R is even: a workgroup holds two items or none)
    const int rh = item / S, sp = item - rh * S;
    const int r = rh / PL_NH, hh = rh % PL_NH;
    float* const qs = att_s + 64 * isub;                       // [2][64] each: q (x 1/8), this step's k, v of the wave's item
    float* const ks = att_s + 128 + 64 * isub;
    float* const vs = att_s + 256 + 64 * isub;
    float* const mo = att_s + 384;                            // [8 waves][64]: the waves' unnormalised outputs, one dim per lane
    float* const mm = mo + 512;                               // [8] the waves' maxima
    float* const ml = mm + 8;                                 // [8] the waves' sums
    const size_t kv_per = a.kv_per;
    if (wave < 8) {
        // Round 5 layout: the scores come from K rows held 8 dims per lane (key group = lane / 8, a 3-step DPP sum per key), the output from V rows held ONE DIM PER
        // LANE (8 keys x PRE registers): a key's weight reaches all lanes through v_readlane and the wave's output is one register per lane -- no cross-lane sums of
        // 9 values over the key groups (27 ds_bpermute round trips: 0.44 us of the 1.9 us phase, profiles/r05_persist_attention_fine_marks.jsonl).
        const RowMeta m = a.meta[r];
        const int kvc = (m.slot - m.kv_start + S - 1) / S;      // cached keys [kv_start, slot) in S shares; this step's key / value arrive with the query
        const int kv0 = m.kv_start + sp * kvc, kv1 = min(kv0 + kvc, m.slot);
        const int grp = lane >> 3, sub = lane & 7;
        const size_t head_base = ((size_t)m.seq * PL_NH + hh) * a.Lmax * CTTS_HEAD_DIM;
        constexpr int PRE = 6;                                // iterations requested before the query exists: 8 (PAIR: 4) waves x 8 keys x 6 = 384 (192) keys
        // Round 6, PAIR: NT more iterations per wave wait in LDS.  With two items per workgroup the registers hold 192 keys of an item; at the bench window's context
        // (~310) the other ~120 streamed BEHIND the query, one dependent HBM round trip per layer (+2.5 us: the 5 -> 6 row step was +24 %).  LDS-DMA (global_load_lds_dwordx4)
        // needs no registers: a lane fetches exactly the 16-byte pieces it will read back itself (K: its 8 dims of its key group; V: rows of 64 floats, read one dim per
        // lane), so the image is lane-linear and every read is conflict-free -- LDS as an asynchronously filled extension of the register file.  The GEMV workgroups' xs
        // block is idle in an attention workgroup: 8 waves x NT x 4 KB.
        constexpr int NT = (PAIR || PL_TAIL_ALL) ? PL_TAIL_IT : 0, PT = PRE + NT;      // (PL_TAIL_ALL: also the one-item workgroups of 1..5 rows: 8 waves x 4 x 8 = 256 more keys)
        constexpr int TB = sizeof(WT) == 4 ? 4096 : 2048;     // bytes of one LDS iteration: K 8 keys x 64 dims | V 8 keys x 64 dims
        typedef __attribute__((address_space(3))) char lds_c;
        typedef __attribute__((address_space(1))) const void* gptr_t;
        lds_c* const tl = (lds_c*)smem + PL_FRONT_BYTES + wave * (NT * TB);
        PlKV<WT> kf[PRE];
        typename KvElem<WT>::reg vv[PRE][8];
        bool ok[PT];
        bool any_ok[PT];
#pragma unroll
        for (int u = 0; u < PT; ++u) {
            const int p = kv0 + 8 * (iw + NWI * u) + grp;
            ok[u] = p < kv1;
            any_ok[u] = kv0 + 8 * (iw + NWI * u) < kv1;       // wave-uniform: some lane group of this wave has a key in iteration u
        }
#define PL_LOAD_KV(l_) do { const WT* const kb_ = (const WT*)a.kv + (size_t)(l_) * 2 * kv_per + head_base; const WT* const vb_ = kb_ + kv_per + lane; \
        _Pragma("unroll") for (int u = 0; u < PRE; ++u) if (any_ok[u]) { const int p0_ = kv0 + 8 * (iw + NWI * u); const int pc_ = ok[u] ? p0_ + grp : m.kv_start; \
            kf[u].load(kb_ + (size_t)pc_ * CTTS_HEAD_DIM + 8 * sub); \
            _Pragma("unroll") for (int g = 0; g < 8; ++g) vv[u][g] = KvElem<WT>::load(vb_ + (size_t)min(p0_ + g, kv1 - 1) * CTTS_HEAD_DIM); } } while (0)
        // the LDS iterations of layer l_: fp32: K dims 8 sub .. + 3 | K dims 8 sub + 4 .. + 7 | V keys 0..3 | V keys 4..7 (1 KB per instruction); fp16: K | V
#define PL_DMA_KV(l_) do { if constexpr (NT > 0) { const WT* const kb_ = (const WT*)a.kv + (size_t)(l_) * 2 * kv_per + head_base; const WT* const vb_ = kb_ + kv_per; \
        int ln_ = lane; asm volatile("" : "+v"(ln_));      /* (the lane's source offsets are re-derived per layer: hoisted out of the layer loop they are 11 spilled address pairs) */ \
        const int grp = ln_ >> 3, sub = ln_ & 7, lane = ln_; \
        _Pragma("unroll") for (int t = 0; t < NT; ++t) if (any_ok[PRE + t]) { const int p0_ = kv0 + 8 * (iw + NWI * (PRE + t)); \
            const WT* const ks_ = kb_ + (size_t)min(p0_ + grp, kv1 - 1) * CTTS_HEAD_DIM + 8 * sub; \
            if constexpr (sizeof(WT) == 4) { \
                __builtin_amdgcn_global_load_lds((gptr_t)ks_, (__attribute__((address_space(3))) void*)(tl + t * TB), 16, 0, 0); \
                __builtin_amdgcn_global_load_lds((gptr_t)(ks_ + 4), (__attribute__((address_space(3))) void*)(tl + t * TB + 1024), 16, 0, 0); \
                const WT* const v0_ = vb_ + (size_t)min(p0_ + (lane >> 4), kv1 - 1) * CTTS_HEAD_DIM + 4 * (lane & 15); \
                const WT* const v1_ = vb_ + (size_t)min(p0_ + 4 + (lane >> 4), kv1 - 1) * CTTS_HEAD_DIM + 4 * (lane & 15); \
                __builtin_amdgcn_global_load_lds((gptr_t)v0_, (__attribute__((address_space(3))) void*)(tl + t * TB + 2048), 16, 0, 0); \
                __builtin_amdgcn_global_load_lds((gptr_t)v1_, (__attribute__((address_space(3))) void*)(tl + t * TB + 3072), 16, 0, 0); \
            } else { \
                __builtin_amdgcn_global_load_lds((gptr_t)ks_, (__attribute__((address_space(3))) void*)(tl + t * TB), 16, 0, 0); \
                const WT* const v0_ = vb_ + (size_t)min(p0_ + grp, kv1 - 1) * CTTS_HEAD_DIM + 8 * sub; \
                __builtin_amdgcn_global_load_lds((gptr_t)v0_, (__attribute__((address_space(3))) void*)(tl + t * TB + 1024), 16, 0, 0); \
            } } } } while (0)
        PL_DMA_KV(0);
        PL_LOAD_KV(0);
        __builtin_amdgcn_sched_barrier(0);
        if (__builtin_amdgcn_readfirstlane(done_v | err_v)) return;
        __syncthreads();                                      // S0
        for (int l = 0; l < NL; ++l) {
            const WT* const kb = (const WT*)a.kv + (size_t)l * 2 * kv_per + head_base + 8 * sub;
            const WT* const vb = (const WT*)a.kv + (size_t)l * 2 * kv_per + kv_per + head_base + lane;
            __syncthreads();                                  // B1: q (x 1/8), k_new, v_new in LDS
#define PL_AMARK(i) do { if (a.ts != nullptr && l + 1 == NL && tid == 0) a.ts[(size_t)b * 10 + (i)] = wall_clock64(); } while (0)
            PL_AMARK(3);
            const f32x4 q0 = *(const f32x4*)(qs + 8 * sub), q1 = *(const f32x4*)(qs + 8 * sub + 4);
            // two passes over the keys held in registers: scores -> the wave's maximum -> ONE exponential per key (no running rescale)
            float sc[PT];
            float mw = -INFINITY;
#pragma unroll
            for (int u = 0; u < PRE; ++u) {
                sc[u] = -INFINITY;
                if (any_ok[u]) {
                    float dot = q0[0] * kf[u].at(0) + q0[1] * kf[u].at(1) + q0[2] * kf[u].at(2) + q0[3] * kf[u].at(3) +
                                q1[0] * kf[u].at(4) + q1[1] * kf[u].at(5) + q1[2] * kf[u].at(6) + q1[3] * kf[u].at(7);
                    dot += dpp_f<DPP_XOR1>(dot);
                    dot += dpp_f<DPP_XOR2>(dot);
                    dot += dpp_f<DPP_HALF_MIRROR>(dot);
                    sc[u] = ok[u] ? dot : -INFINITY;
                    mw = fmaxf(mw, sc[u]);
                }
            }
            if constexpr (NT > 0) {
                // the iterations that waited in LDS: requested a layer ago, long complete (the wait costs nothing; the compiler does not know that LDS depends on vmcnt)
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    sc[PRE + t] = -INFINITY;
                    if (any_ok[PRE + t]) {
                        PlKV<WT> kt;
                        if constexpr (sizeof(WT) == 4) { kt.a = *(const __attribute__((address_space(3))) f32x4*)(tl + t * TB + lane * 16); kt.b = *(const __attribute__((address_space(3))) f32x4*)(tl + t * TB + 1024 + lane * 16); }
                        else kt.h = *(const __attribute__((address_space(3))) half8*)(tl + t * TB + lane * 16);
                        float dot = q0[0] * kt.at(0) + q0[1] * kt.at(1) + q0[2] * kt.at(2) + q0[3] * kt.at(3) + q1[0] * kt.at(4) + q1[1] * kt.at(5) + q1[2] * kt.at(6) + q1[3] * kt.at(7);
                        dot += dpp_f<DPP_XOR1>(dot);
                        dot += dpp_f<DPP_XOR2>(dot);
                        dot += dpp_f<DPP_HALF_MIRROR>(dot);
                        sc[PRE + t] = ok[PRE + t] ? dot : -INFINITY;
                        mw = fmaxf(mw, sc[PRE + t]);
                    }
                }
            }
            float mrun = wave_max(mw);                        // the same in every lane (-inf: this wave holds no key)
            PL_AMARK(4);
            float lrun = 0.f;                                 // the weights of this lane's key group
            float oa[4] = {0.f, 0.f, 0.f, 0.f};               // dim `lane` of the wave's output as four partial sums (four independent FMA chains instead of one of 8 per iteration)
#pragma unroll
            for (int u = 0; u < PRE; ++u) {
                if (any_ok[u]) {
                    const float pe = (sc[u] == -INFINITY) ? 0.f : expf(sc[u] - mrun);
                    lrun += pe;
#pragma unroll
                    for (int g = 0; g < 8; ++g) oa[g & 3] = fmaf(readlane_f(pe, 8 * g), KvElem<WT>::f(vv[u][g]), oa[g & 3]);
                }
            }
            if constexpr (NT > 0) {
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    if (any_ok[PRE + t]) {
                        const float pe = (sc[PRE + t] == -INFINITY) ? 0.f : expf(sc[PRE + t] - mrun);
                        lrun += pe;
#pragma unroll
                        for (int g = 0; g < 8; ++g) {
                            float vt;
                            if constexpr (sizeof(WT) == 4) vt = *(const __attribute__((address_space(3))) float*)(tl + t * TB + 2048 + g * 256 + lane * 4);
                            else vt = (float)*(const __attribute__((address_space(3))) half_t*)(tl + t * TB + 1024 + g * 128 + lane * 2);
                            oa[g & 3] = fmaf(readlane_f(pe, 8 * g), vt, oa[g & 3]);
                        }
                    }
                }
            }
            // Shares beyond the PT * 8 NWI prefetched keys stream behind the query, UNS steps of 8 NWI keys per round trip
            constexpr int UNS = 4, STEP = 8 * NWI;
            for (int wb0 = kv0 + 8 * (iw + NWI * PT); wb0 < kv1; wb0 += STEP * UNS) {       // (wave-uniform bound)
                PlKV<WT> ks_[UNS];
                typename KvElem<WT>::reg vs_[UNS][8];
                bool live_[UNS];
#pragma unroll
                for (int u = 0; u < UNS; ++u) {
                    const int p0 = wb0 + STEP * u;
                    live_[u] = p0 + grp < kv1;
                    ks_[u].load(kb + (size_t)(live_[u] ? p0 + grp : m.kv_start) * CTTS_HEAD_DIM);
#pragma unroll
                    for (int g = 0; g < 8; ++g) vs_[u][g] = KvElem<WT>::load(vb + (size_t)min(p0 + g, kv1 - 1) * CTTS_HEAD_DIM);      // (a key past the share: weight 0, any valid row will do)
                }
#pragma unroll
                for (int u = 0; u < UNS; ++u) {
                    if (wb0 + STEP * u >= kv1) break;                                        // (wave-uniform: the step does not exist)
                    const bool live = live_[u];
                    float dot = q0[0] * ks_[u].at(0) + q0[1] * ks_[u].at(1) + q0[2] * ks_[u].at(2) + q0[3] * ks_[u].at(3) + q1[0] * ks_[u].at(4) + q1[1] * ks_[u].at(5) + q1[2] * ks_[u].at(6) + q1[3] * ks_[u].at(7);
                    dot += dpp_f<DPP_XOR1>(dot);
                    dot += dpp_f<DPP_XOR2>(dot);
                    dot += dpp_f<DPP_HALF_MIRROR>(dot);
                    const float mn = wave_max(fmaxf(mrun, live ? dot : -INFINITY));       // the wave keeps ONE running maximum
                    const float scl = pl_exp_diff(mrun, mn);
                    const float pe = live ? expf(dot - mn) : 0.f;
                    lrun = lrun * scl + pe;
#pragma unroll
                    for (int i = 0; i < 4; ++i) oa[i] *= scl;
#pragma unroll
                    for (int g = 0; g < 8; ++g) oa[g & 3] = fmaf(readlane_f(pe, 8 * g), KvElem<WT>::f(vs_[u][g]), oa[g & 3]);
                    mrun = mn;
                }
            }
            PL_AMARK(5);
            const float lw = wave_sum(lrun) * 0.125f;         // (the 8 lanes of a key group hold the same weights)
            mo[wave * 64 + lane] = (oa[0] + oa[1]) + (oa[2] + oa[3]);
            if (lane == 0) { mm[wave] = mrun; ml[wave] = lw; }
            PL_AMARK(6);
            __syncthreads();                                  // B2
            if (l + 1 < NL) { PL_DMA_KV(l + 1); PL_LOAD_KV(l + 1); }      // the next layer's cached rows: a whole layer ahead of its query (B2: every LDS read of this layer is done)
            __builtin_amdgcn_sched_barrier(0);
        }
    } else if (wave == 8 || (PAIR && wave == 9)) {
        if (__builtin_amdgcn_readfirstlane(done_v | err_v)) return;
        const unsigned tag0 = (unsigned)ep_v * 32u;
        __syncthreads();                                      // S0
        for (int l = 0; l < NL; ++l) {
            const unsigned tag = tag0 + (unsigned)l;
            float v[3];
            const bool got = sweep<3>(a.g_qkv, (unsigned)(rh * 192 + lane), [](int k) { return 64 * k; }, tag, v, a.error, 2, abort_s, a.nap_qkv);
            (void)got;
            qs[lane] = v[0] * 0.125f;                         // 1 / sqrt(64) (llama.py:653-661)
            ks[lane] = v[1];
            vs[lane] = v[2];
            if (l + 1 == NL && wave == 8) PL_MARK(1);
            __syncthreads();                                  // B1
            // this step's own key (slot `m.slot`, the causal end of the row: llama.py:1073-1087): its score, on all 64 lanes (split 0 merges it)
            const float dnew = wave_sum(qs[lane] * ks[lane]);
            __syncthreads();                                  // B2
            if (a.ts != nullptr && l + 1 == NL && lane == 0 && wave == 8) a.ts[(size_t)b * 10 + 7] = wall_clock64();
            // lane = output dim: combine the item's NWI waves' partials.  The NWI + 1 rescale factors (the waves + this step's own key) are computed side by side in lanes 0..NWI
            const float mmv = mm[NWI * isub + (lane & (NWI - 1))], mlv = ml[NWI * isub + (lane & (NWI - 1))];
            const float mj = (lane < NWI) ? mmv : ((lane == NWI && sp == 0) ? dnew : -INFINITY);
            const float lj = (lane < NWI) ? mlv : ((lane == NWI) ? 1.f : 0.f);
            float M = wave_max(mj);
            const float sj = pl_exp_diff(mj, M);
            float L = wave_sum(sj * lj);
            float O = readlane_f(sj, NWI) * vs[lane];
#pragma unroll
            for (int w = 0; w < NWI; ++w) O = fmaf(readlane_f(sj, w), mo[(NWI * isub + w) * 64 + lane], O);
            if (S > 1 && sp != 0) {
                // a share's partial: [max, sum, o[64]] (an empty share publishes max = -inf, sum = 0)
                u64* const gp = a.g_part + (size_t)(rh * S + sp) * 66;
                store_granule(gp + 2 + lane, tag, O);
                if (lane == 0) store_granule(gp, tag, M);
                if (lane == 1) store_granule(gp + 1, tag, L);
            } else {
                if (S > 1) {
                    // split 0: the other shares' partials (3 granules per lane and share: o[lane], max, sum -- the last two at wave-uniform addresses)
                    const u64* p = a.g_part + (size_t)(rh * S) * 66;
                    asm volatile("" : "+v"(p));
                    float po[4], pm[4], pl[4];
#pragma unroll 1
                    for (unsigned spins = 0;; ++spins) {
                        bool okk = true;
#pragma unroll
                        for (int t = 0; t < 4; ++t) {
                            if (t + 1 < S) {
                                const u64 xo = __hip_atomic_load(p + (t + 1) * 66 + 2 + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                                const u64 xm = __hip_atomic_load(p + (t + 1) * 66, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                                const u64 xl = __hip_atomic_load(p + (t + 1) * 66 + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                                po[t] = __builtin_bit_cast(float, (unsigned)xo); pm[t] = __builtin_bit_cast(float, (unsigned)xm); pl[t] = __builtin_bit_cast(float, (unsigned)xl);
                                okk = okk && ((unsigned)(xo >> 32) == tag) && ((unsigned)(xm >> 32) == tag) && ((unsigned)(xl >> 32) == tag);
                            }
                        }
                        if (__all(okk)) break;
                        typedef __attribute__((address_space(3))) int lds_i;
                        bool giveup = spins >= PL_SPIN_LIMIT || __hip_atomic_load((lds_i*)abort_s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) != 0;
                        if (!giveup && (spins & 1023u) == 1023u) giveup = __hip_atomic_load(a.error, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0;
                        if (giveup) { if (lane == 0) { atomicCAS(a.error, 0, 6); __hip_atomic_store((lds_i*)abort_s, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); } break; }
                        __builtin_amdgcn_s_sleep(2);
                    }
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        if (t + 1 < S) {
                            const float mn = fmaxf(M, pm[t]);
                            const float s1 = pl_exp_diff(M, mn), s2 = pl_exp_diff(pm[t], mn);
                            L = L * s1 + pl[t] * s2;
                            O = O * s1 + po[t] * s2;
                            M = mn;
                        }
                    }
                }
                store_granule(a.g_att + (size_t)r * PL_H + hh * CTTS_HEAD_DIM + lane, tag, O / L);
            }
            if (l + 1 == NL && wave == 8) PL_MARK(2);
        }
        if (a.ts != nullptr && lane == 0 && wave == 8) {
            a.ts[(size_t)b * 10 + 0] = t_mark[0]; a.ts[(size_t)b * 10 + 1] = t_mark[1]; a.ts[(size_t)b * 10 + 2] = t_mark[2];
        }
    } else {
        if (__builtin_amdgcn_readfirstlane(done_v | err_v)) return;
        __syncthreads();                                      // S0
        for (int l = 0; l < NL; ++l) {
            __syncthreads();                                  // B1
            __syncthreads();                                  // B2
        }
    }
}

// ---- packed MFMA-A tile images (gpt_engine.hip pack_tiles: [row tile][k tile][lane][16 B]) -> the per-workgroup register images above.  A destination
// fragment = 4 consecutive k of one weight row (16 bytes fp32, 8 bytes fp16); in the fp16 tile image a lane's 16 bytes hold 8 consecutive k of a 32-wide k-tile
template <typename WT>
__global__ __launch_bounds__(256) void persist_repack_kernel(const typename PlW<WT>::frag* qkv, const typename PlW<WT>::frag* o, const typename PlW<WT>::frag* gu,
                                                             const typename PlW<WT>::frag* d, typename PlW<WT>::frag* dst) {
    const int idx = blockIdx.x * 256 + threadIdx.x;               // destination fragment of this layer: [192 workgroups][12288]
    if (idx >= PL_GEMV_BLOCKS * (PL_BLOCK_BYTES / 16)) return;
    const int g = idx / (PL_BLOCK_BYTES / 16), oo = idx % (PL_BLOCK_BYTES / 16);
    const typename PlW<WT>::frag* src;
    int row_tile, i, k4, ktiles;                                   // k4 = index of the 4-wide k group; ktiles = 16-wide k-tiles of the matrix
    if (oo < PL_QKV_BYTES / 16) {
        const int pw = oo / 384, rem = oo % 384, row = rem / 192, j = (rem % 192) / 64, ln = rem % 64;
        const int hh = g >> 4, jj = g & 15;
        int which, dd;
        if (pw < 4) { which = pw >> 1; dd = 2 * jj + (pw & 1) + 32 * row; }
        else { which = 2; dd = 4 * jj + 2 * (pw - 4) + row; }
        row_tile = which * 48 + hh * 4 + (dd & 31) / 8;            // tile rows = dims [8t..8t+7 | 8t+32..8t+39] of one head (finalize_t qkv_row)
        i = (dd & 7) + (dd >= 32 ? 8 : 0);
        k4 = 64 * j + ln; ktiles = 48; src = qkv;
    } else if (oo < (PL_QKV_BYTES + PL_O_BYTES) / 16) {
        const int o2 = oo - PL_QKV_BYTES / 16, w = o2 / 192, rem = o2 % 192, j = rem / 64, ln = rem % 64;
        const int row = 4 * g + w;
        row_tile = row >> 4; i = row & 15; k4 = 64 * j + ln; ktiles = 48; src = o;
    } else if (oo < (PL_QKV_BYTES + PL_O_BYTES + PL_GU_BYTES) / 16) {
        const int o2 = oo - (PL_QKV_BYTES + PL_O_BYTES) / 16, w = o2 / 768, rem = o2 % 768, s = rem / 192, j = (rem % 192) / 64, ln = rem % 64;
        const int dim = 16 * g + 2 * w + (s >> 1), isup = s & 1;
        row_tile = dim >> 3; i = (dim & 7) + 8 * isup;             // tile rows = [8 gate rows | the matching 8 up rows]
        k4 = 64 * j + ln; ktiles = 48; src = gu;
    } else {
        const int o2 = oo - (PL_QKV_BYTES + PL_O_BYTES + PL_GU_BYTES) / 16, w = o2 / 384, rem = o2 % 384, j = rem / 64, ln = rem % 64;
        const int row = 4 * g + (w >> 1);
        row_tile = row >> 4; i = row & 15; k4 = 384 * (w & 1) + 64 * j + ln; ktiles = 192; src = d;
    }
    if constexpr (sizeof(WT) == 4) dst[idx] = src[((size_t)row_tile * ktiles + (k4 >> 2)) * 64 + i + 16 * (k4 & 3)];
    else dst[idx] = src[(((size_t)row_tile * (ktiles / 2) + (k4 >> 3)) * 64 + i + 16 * ((k4 >> 1) & 3)) * 2 + (k4 & 1)];      // 32-wide k-tiles, 8 halfs per lane: two 4-groups each
}

// the folded heads' MFMA tile image [n_tiles][48 k-tiles][lane][16 B] -> [192 workgroups][7 waves][2 rows][3 j][64 lanes] fragments; head row = 14 g + 2 wave + row
template <typename WT>
__global__ __launch_bounds__(256) void persist_repack_heads_kernel(const typename PlW<WT>::frag* whead, int n_tiles, typename PlW<WT>::frag* dst) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= PL_GEMV_BLOCKS * PL_HEAD_FRAGS) return;
    const int g = idx / PL_HEAD_FRAGS, o2 = idx % PL_HEAD_FRAGS, w = o2 / 384, rem = o2 % 384, row = rem / 192, j = (rem % 192) / 64, ln = rem % 64;
    const int hrow = PL_HEAD_ROWS * g + 2 * w + row, row_tile = hrow >> 4, i = hrow & 15, k4 = 64 * j + ln;
    typename PlW<WT>::frag v = {};
    if (row_tile < n_tiles) {
        if constexpr (sizeof(WT) == 4) v = whead[((size_t)row_tile * 48 + (k4 >> 2)) * 64 + i + 16 * (k4 & 3)];
        else v = whead[(((size_t)row_tile * 24 + (k4 >> 3)) * 64 + i + 16 * ((k4 >> 1) & 3)) * 2 + (k4 & 1)];
    }
    dst[idx] = v;
}
int launch_persist_repack_heads(int half_w, const void* whead, int n_tiles, void* dst, hipStream_t s) {
    const int n = PL_GEMV_BLOCKS * PL_HEAD_FRAGS;
    if (half_w) hipLaunchKernelGGL(persist_repack_heads_kernel<half_t>, dim3((n + 255) / 256), dim3(256), 0, s, (const half4*)whead, n_tiles, (half4*)dst);
    else hipLaunchKernelGGL(persist_repack_heads_kernel<float>, dim3((n + 255) / 256), dim3(256), 0, s, (const f32x4*)whead, n_tiles, (f32x4*)dst);
    CTTS_HIP_CHECK(hipGetLastError());
    return 0;
}

int launch_persist_repack(int half_w, const void* qkv, const void* o, const void* gu, const void* d, void* dst, hipStream_t s) {
    const int n = PL_GEMV_BLOCKS * (PL_BLOCK_BYTES / 16);
    if (half_w) hipLaunchKernelGGL(persist_repack_kernel<half_t>, dim3((n + 255) / 256), dim3(256), 0, s, (const half4*)qkv, (const half4*)o, (const half4*)gu, (const half4*)d, (half4*)dst);
    else hipLaunchKernelGGL(persist_repack_kernel<float>, dim3((n + 255) / 256), dim3(256), 0, s, (const f32x4*)qkv, (const f32x4*)o, (const f32x4*)gu, (const f32x4*)d, (f32x4*)dst);
    CTTS_HIP_CHECK(hipGetLastError());
    return 0;
}

static size_t persist_lds_bytes(int R) {
    const size_t gemv = (size_t)(R * PL_I + PL_RED_FLOATS + 4 * R + 4 * R + 256 * R) * 4, tail = (R > PL_MAXR_ONE || PL_TAIL_ALL) ? (size_t)8 * PL_TAIL_IT * 4096 : 0;
    return PL_FRONT_BYTES + (gemv > tail ? gemv : tail);
}

template <int R, int SCHED, typename WT, bool LORA = false>
static int persist_launch_t(const PersistArgs& a, hipStream_t s, bool configure_only) {
    auto kern = persist_layer_kernel<R, SCHED, WT, LORA>;
    if (configure_only) { CTTS_HIP_CHECK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)persist_lds_bytes(R))); return 0; }
    hipLaunchKernelGGL(kern, dim3(PL_BLOCKS), dim3(PL_THREADS(R)), persist_lds_bytes(R), s, a);
    CTTS_HIP_CHECK(hipGetLastError());
    return 0;
}
template <int SCHED, typename WT>
static int persist_launch_r(int R, const PersistArgs& a, hipStream_t s, bool cfg) {
    // exact row counts: a spare row would append stale K / V rows to a live cache lane
    if constexpr (SCHED == 3) if (!cfg && a.lora) {      // per-utterance adapters: the LORA kernels
        switch (R) {
            case 1: return persist_launch_t<1, 3, WT, true>(a, s, false);
            case 2: return persist_launch_t<2, 3, WT, true>(a, s, false);
            case 3: return persist_launch_t<3, 3, WT, true>(a, s, false);
            case 4: return persist_launch_t<4, 3, WT, true>(a, s, false);
            case 5: return persist_launch_t<5, 3, WT, true>(a, s, false);
            case 6: return persist_launch_t<6, 3, WT, true>(a, s, false);
            case 7: return persist_launch_t<7, 3, WT, true>(a, s, false);
            case 8: return persist_launch_t<8, 3, WT, true>(a, s, false);
        }
    }
    if (cfg) {
        if constexpr (SCHED == 3) {
            const int rl = persist_launch_t<1, 3, WT, true>(a, s, true) | persist_launch_t<2, 3, WT, true>(a, s, true) | persist_launch_t<3, 3, WT, true>(a, s, true) | persist_launch_t<4, 3, WT, true>(a, s, true) |
                           persist_launch_t<5, 3, WT, true>(a, s, true) | persist_launch_t<6, 3, WT, true>(a, s, true) | persist_launch_t<7, 3, WT, true>(a, s, true) | persist_launch_t<8, 3, WT, true>(a, s, true);
            if (rl) return rl;
        }
        int rc = persist_launch_t<1, SCHED, WT>(a, s, true) | persist_launch_t<2, SCHED, WT>(a, s, true) | persist_launch_t<3, SCHED, WT>(a, s, true) | persist_launch_t<4, SCHED, WT>(a, s, true) |
                 persist_launch_t<5, SCHED, WT>(a, s, true);
        if constexpr (SCHED == 3) rc |= persist_launch_t<6, 3, WT>(a, s, true) | persist_launch_t<7, 3, WT>(a, s, true) | persist_launch_t<8, 3, WT>(a, s, true);
        return rc;
    }
    switch (R) {
        case 1: return persist_launch_t<1, SCHED, WT>(a, s, false);
        case 2: return persist_launch_t<2, SCHED, WT>(a, s, false);
        case 3: return persist_launch_t<3, SCHED, WT>(a, s, false);
        case 4: return persist_launch_t<4, SCHED, WT>(a, s, false);
        case 5: return persist_launch_t<5, SCHED, WT>(a, s, false);
    }
    if constexpr (SCHED == 3) {      // 6..8 rows (two attention items per workgroup): the paced schedule only
        switch (R) {
            case 6: return persist_launch_t<6, 3, WT>(a, s, false);
            case 7: return persist_launch_t<7, 3, WT>(a, s, false);
            case 8: return persist_launch_t<8, 3, WT>(a, s, false);
        }
    }
    ctts_set_error("persistent layer: %d rows (max %d; 6+ rows need the paced schedule)", R, PL_MAXR);
    return 1;
}

int persist_configure() {
    PersistArgs a = {};
    return persist_launch_r<1, float>(1, a, nullptr, true) | persist_launch_r<2, float>(1, a, nullptr, true) | persist_launch_r<3, float>(1, a, nullptr, true) |
           persist_launch_r<3, half_t>(1, a, nullptr, true);
}

int launch_persist_layer(int R, const PersistArgs& a, hipStream_t s) {
    if (a.lora && a.sched != 3) { ctts_set_error("persistent layer: per-utterance adapters need the paced schedule"); return 1; }
    if (a.half_w) return persist_launch_r<3, half_t>(R, a, s, false);          // fp16 engines: the paced schedule only
    return (a.sched == 1) ? persist_launch_r<1, float>(R, a, s, false) : (a.sched == 2) ? persist_launch_r<2, float>(R, a, s, false) : persist_launch_r<3, float>(R, a, s, false);
}
