// The 20-layer decoder stack of a 5..32-row decode step (fp32 engine) as ONE persistent launch with MFMA projections (gfx950).
//
// Reference arithmetic: LlamaDecoderLayer.forward, chattts_plus/models/llama.py:719-749 (RMSNorm :82-87, q/k/v + RoPE + cache append :619-633,
// SDPA :653-661, o_proj + residual :666,731, SwiGLU MLP + residual :214,737-739) -- the loop it serves is gpt.py:389-546.
//
// Why: at these batch sizes the launch chain is 100 dependent launches of ~5 us each whatever they move (a projection launch streams its 2-19 MB at
// 0.5-2.3 TB/s), and a second 16-row chunk re-reads every weight tile from L2.  Here the stack is one launch of 256 resident workgroups:
//   * every workgroup owns THREE of a layer's 768 weight tiles (144 q|k|v + 48 o_proj + 384 gate|up + 192 down tile-slices; 147 KB per layer) and keeps
//     them in the registers of its 8 GEMM waves; a tile is re-requested for the next layer right after its use, so the weight stream runs a layer
//     ahead of the dependency edges and mostly while the HBM has nothing else to do (everything but the attention phase); both 16-row chunks multiply
//     against the SAME registers (no second pull of the tile);
//   * the products are the launch chain's: exact-f32 MFMA (v_mfma_f32_16x16x4_f32), 8 waves x 6 k-tiles per tile, partial C tiles added in wave order,
//     the down projection in four K slices added in slice order by the last arriver -- the same sums in the same order as skinny_gemm.hip's
//     PRO_NORM / PRO_XH / PRO_PACKED kernels with the EPI_QKV / EPI_RESID_XH / EPI_SWIGLU / EPI_RESID_XH_SK epilogues (the power-of-two row scale of the
//     packed residual copy is an exact no-op in fp32 and is dropped);
//   * 4 attention waves per workgroup serve the (row, head) items -- attn_decode_kernel<float, 4>'s loop, K / V through write-through-coherent loads;
//   * hand-offs follow cdna_hip_programming.md Guideline 16 R1: payload with write-through (sc1) stores, every storing wave drains, ONE lane stores the
//     item's flag word (tag = launch counter, layer, phase); ONE wave per workgroup polls the flag words of the producing phase (a 16-byte sc1 load per lane
//     covers 256 flags), then a workgroup barrier releases the GEMM waves, which read the payload with sc1 loads.  The poller holds no weight loads: its
//     poll never queues behind the weight stream (vmcnt retires in order).
// Every spin is bounded; a give-up sets the engine's device error word, which ends this launch and turns every later one into a no-op
// (ctts_gpt_progress reports it).
#include "kernels.h"
#include "persist_mfma.h"

typedef __amdgpu_buffer_rsrc_t pm_rsrc_t;
typedef unsigned pm_u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) int pm_lds_int;

#define PM_SPIN_LIMIT (1u << 18)      // ~0.3 s of polling before a wave gives up
#define PM_UN 6                       // attention: keys per lane group and loop iteration (12 x 16 B in flight per lane)

__device__ inline pm_rsrc_t pm_rsrc(const void* p) { return __builtin_amdgcn_make_buffer_rsrc((void*)p, 0, 0xFFFFFFFFu, 0x00020000); }
// write-through-coherent accesses (sc1): served by L2, never by this CU's L1; see MI355X_MICROARCH.md "inter-workgroup visibility"
__device__ inline f32x4 pm_ld16(pm_rsrc_t r, unsigned byte_off) { return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, (int)byte_off, 0, 16)); }
__device__ inline float pm_ld4(const float* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ inline void pm_st4(float* p, float v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ inline void pm_drain() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

__device__ inline f32x4 pm_mma(const f32x4 a, const f32x4 b, f32x4 c) {
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(a[0], b[0], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(a[1], b[1], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(a[2], b[2], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(a[3], b[3], c, 0, 0, 0);
    return c;
}
__device__ inline float pm_exp_diff(float m, float mn) { return (m == -INFINITY) ? 0.f : expf(m - mn); }

// ONE wave waits until the first n flag words all carry `tag` (lane: words 4 lane .. 4 lane + 3).  false after PM_SPIN_LIMIT passes or once the engine has given up.
__device__ inline bool pm_wait(const unsigned* flags, int n, unsigned tag, int* err, int code, pm_lds_int* abort_s, int lane, int nap, int delay) {
    for (int z = 0; z < delay; ++z) __builtin_amdgcn_s_sleep(2);
    const pm_rsrc_t r = pm_rsrc(flags);
#pragma unroll 1
    for (unsigned spins = 0;; ++spins) {
        const pm_u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, lane * 16, 0, 16);
        bool ok = true;
#pragma unroll
        for (int j = 0; j < 4; ++j) ok = ok && (4 * lane + j >= n || v[j] == tag);
        if (__all(ok)) return true;
        bool giveup = spins >= PM_SPIN_LIMIT;
        if (!giveup && (spins & 255u) == 255u) giveup = __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0;
        if (giveup) {
            if (lane == 0) { atomicCAS(err, 0, code); __hip_atomic_store(abort_s, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
            return false;
        }
        for (int z = 0; z < nap; ++z) __builtin_amdgcn_s_sleep(2);
    }
}

// C element (weight row i of the tile, activation row n of chunk g) = the 8 waves' partials added in wave order (skinny_gemm.hip c_elem)
template <int NCH>
__device__ inline float pm_c(const float* red, int i, int n, int g) {
    const float* q = red + (g * 64 + ((i >> 2) << 4) + n) * 4 + (i & 3);
    float sum = 0.f;
#pragma unroll
    for (int w = 0; w < PM_GEMM_WAVES; ++w) sum += q[w * NCH * 256];
    return sum;
}

// One (row, head) attention item on a group of 4 waves: attn_decode_kernel<float, 4> (attention.hip) -- wave aw, lane group grp handle keys kv0 + 8 (4 i + aw) + grp,
// online softmax per lane group, the wave's 8 groups merged by shuffles, the wave's partial (max, sum, o[8] per sub) parked in LDS (mg: [4 waves][8][10]).
// UN = keys per lane group and loop iteration (loads in flight only: the arithmetic and its order do not depend on it).
template <int UN>
__device__ __forceinline__ void pm_attn_item(const PmArgs& a, const pm_rsrc_t rk, const pm_rsrc_t rv, const pm_rsrc_t rs_q, const int r, const int h,
                                             const int aw, const int lane, float* const mg) {
    const int grp = lane >> 3, sub = lane & 7;
    const RowMeta m = a.meta[r];
    const int kv0 = m.kv_start, kv1 = m.slot + 1;
    float q[8];
    {
        const unsigned qo = (unsigned)((((size_t)r * PM_NH + h) * CTTS_HEAD_DIM + 8 * sub) * 4);
        const f32x4 q0 = pm_ld16(rs_q, qo), q1 = pm_ld16(rs_q, qo + 16);
        q[0] = q0[0] * 0.125f; q[1] = q0[1] * 0.125f; q[2] = q0[2] * 0.125f; q[3] = q0[3] * 0.125f;
        q[4] = q1[0] * 0.125f; q[5] = q1[1] * 0.125f; q[6] = q1[2] * 0.125f; q[7] = q1[3] * 0.125f;
    }
    const unsigned head_off = (unsigned)(((((size_t)m.seq * PM_NH + h) * a.Lmax) * CTTS_HEAD_DIM + 8 * sub) * 4);     // bytes
    float mrun = -INFINITY, lrun = 0.f, o[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = 0.f;
    for (int wb = kv0 + 8 * aw; wb < kv1; wb += 8 * PM_ATT_WAVES * UN) {
        const int base = wb + grp;
        float kf[UN][8], vf[UN][8];
        bool ok[UN];
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            const int p = base + 8 * PM_ATT_WAVES * u;
            ok[u] = p < kv1;
            const unsigned off = head_off + (unsigned)(ok[u] ? p : kv0) * (CTTS_HEAD_DIM * 4);      // clamp: always a valid address
            const f32x4 k0 = pm_ld16(rk, off), k1 = pm_ld16(rk, off + 16), v0 = pm_ld16(rv, off), v1 = pm_ld16(rv, off + 16);
            kf[u][0] = k0[0]; kf[u][1] = k0[1]; kf[u][2] = k0[2]; kf[u][3] = k0[3]; kf[u][4] = k1[0]; kf[u][5] = k1[1]; kf[u][6] = k1[2]; kf[u][7] = k1[3];
            vf[u][0] = v0[0]; vf[u][1] = v0[1]; vf[u][2] = v0[2]; vf[u][3] = v0[3]; vf[u][4] = v1[0]; vf[u][5] = v1[1]; vf[u][6] = v1[2]; vf[u][7] = v1[3];
        }
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            float dot = 0.f;
#pragma unroll
            for (int j = 0; j < 8; ++j) dot += q[j] * kf[u][j];
            dot += dpp_f<DPP_XOR1>(dot);                    // 8-lane group sum on DPP (quad xor1, xor2, half-mirror)
            dot += dpp_f<DPP_XOR2>(dot);
            dot += dpp_f<DPP_HALF_MIRROR>(dot);
            if (ok[u]) {
                const float mn = fmaxf(mrun, dot);
                const float sc = pm_exp_diff(mrun, mn);
                const float pe = expf(dot - mn);
                lrun = lrun * sc + pe;
#pragma unroll
                for (int j = 0; j < 8; ++j) o[j] = o[j] * sc + pe * vf[u][j];
                mrun = mn;
            }
        }
    }
    // merge the 8 key groups of this wave (lanes with equal `sub`)
#pragma unroll
    for (int off = 8; off < 64; off <<= 1) {
        const float m2 = __shfl_xor(mrun, off), l2 = __shfl_xor(lrun, off);
        const float mn = fmaxf(mrun, m2);
        const float s1 = pm_exp_diff(mrun, mn), s2 = pm_exp_diff(m2, mn);
        lrun = lrun * s1 + l2 * s2;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float o2 = __shfl_xor(o[j], off);
            o[j] = o[j] * s1 + o2 * s2;
        }
        mrun = mn;
    }
    if (grp == 0) {
        float* mp = mg + (aw * 8 + sub) * 10;
        mp[0] = mrun; mp[1] = lrun;
#pragma unroll
        for (int j = 0; j < 8; ++j) mp[2 + j] = o[j];
    }
}
// ... and its end on the group's first wave, lanes 0..7: the 4 waves' partials in wave order, the softmax finished, the row written into o_proj's fragment-major
// B operand (attention.hip packed_out)
__device__ __forceinline__ void pm_attn_finish(const PmArgs& a, const int r, const int h, const int lane, const float* const mg) {
    float M = mg[lane * 10], L = mg[lane * 10 + 1], O[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) O[j] = mg[lane * 10 + 2 + j];
#pragma unroll
    for (int w = 1; w < PM_ATT_WAVES; ++w) {
        const float* mw = mg + (w * 8 + lane) * 10;
        const float m2 = mw[0], l2 = mw[1];
        const float mn = fmaxf(M, m2);
        const float s1 = pm_exp_diff(M, mn), s2 = pm_exp_diff(m2, mn);
        L = L * s1 + l2 * s2;
#pragma unroll
        for (int j = 0; j < 8; ++j) O[j] = O[j] * s1 + mw[2 + j] * s2;
        M = mn;
    }
    const float inv = 1.0f / L;
    float* dst = a.attn_packed + (size_t)(r >> 4) * 48 * 256;
    const int k = h * CTTS_HEAD_DIM + 8 * lane;
#pragma unroll
    for (int j = 0; j < 8; ++j) pm_st4(dst + xfrag_index<float>(r & 15, k + j, 48), O[j] * inv);
}

#define PM_BAR() __syncthreads()
#define PM_PUBLISH(flagp_, tag_) __hip_atomic_store((flagp_), (tag_), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
// after the barrier that follows a poll: has the poller given up?  (uniform: the flag was written before the barrier)
#define PM_ABORT_CHECK() do { if (__hip_atomic_load(ctl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) != 0) { \
        if (b == 0 && tid == 0) __hip_atomic_fetch_add(a.epoch, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); return; } } while (0)

// Which of a layer's 768 weight tiles this workgroup owns (three each: balanced bytes and MFMA work):
//   b < 128       : w0 = q|k|v tile b,        w1 = gate|up tile b, w2 = down item b
//   128 .. 143    : w0 = q|k|v tile b,        w1 = gate|up tile b, w2 = gate|up tile b + 128
//   144 .. 191    : w0 = o_proj tile b - 144, w1 = gate|up tile b, w2 = gate|up tile b + 128
//   192 .. 255    : w0 = down item b - 64,    w1 = gate|up tile b, w2 = gate|up tile b + 128
// (down item = slice * 48 + tile: K slice `slice` of the 16-row tile `tile`)
struct PmRole {
    bool has_qkv, has_o, two_gu, has_d;
    int d_tile, d_slice;
    __device__ PmRole(int b) {
        has_qkv = b < PM_QKV_TILES; has_o = b >= 144 && b < 192; two_gu = b >= 128; has_d = b < 128 || b >= 192;
        const int d_item = (b < 128) ? b : b - 64;
        d_tile = d_item % 48; d_slice = d_item / 48;
    }
};

// The workgroup's two roles run the SAME sequence of workgroup barriers per layer:
//   phase 1 (owners of a q|k|v tile)      : ready | [layer 0: rows normalised] | partial tiles parked | stores drained
//   phase 2 (owners of an attention item) : ready | one per item | stores drained
//   phase 3 (owners of an o_proj tile)    : ready | parked | drained
//   phase 4 (everybody)                   : ready | parked | drained
//   phase 5 (owners of a down item)       : ready | parked | slabs drained | ticket known | [last arriver: combine drained] | end
// "ready" = the poller (wave 8) has seen every flag of the producing phase; the give-up flag is tested right behind it by everybody.

// ---------------------------------------------------------------------------------------------------- waves 0..7: weight tiles in registers, MFMA, epilogues
// (phase 2: GEMM waves 0..3 serve the workgroup's second attention item)
template <int NCH>
__device__ __forceinline__ void pm_gemm_role(const PmArgs& a, float* const bx, float* const red, float* const fac, float* const merge, pm_lds_int* const ctl,
                                             const int tid0, const int lane0, const int wave, const int b) {
    int tid = tid0, lane = lane0;
    const PmRole ro(b);
    const int R = a.R, NL = a.n_layers;
    f32x4 w0[6], w1[6], w2[6];
    const size_t wl = (size_t)wave * 6 * 64 + lane;              // this wave's first fragment of a tile's k-tile row
#define PM_LOAD6(arr_, base_, tile_, ktall_, koff_) do { const f32x4* p_ = (const f32x4*)(base_) + ((size_t)(tile_) * (ktall_) + (koff_)) * 64 + wl; \
        _Pragma("unroll") for (int i_ = 0; i_ < 6; ++i_) arr_[i_] = __builtin_nontemporal_load(p_ + i_ * 64); } while (0)
#define PM_LOAD_W0(l_) do { const size_t lo_ = (size_t)(l_) * a.w_stride; \
        if (ro.has_qkv) PM_LOAD6(w0, a.wqkv + lo_, b, 48, 0); else if (ro.has_o) PM_LOAD6(w0, a.wo + lo_, b - 144, 48, 0); else PM_LOAD6(w0, a.wd + lo_, ro.d_tile, 192, ro.d_slice * 48); } while (0)
#define PM_LOAD_W1(l_) PM_LOAD6(w1, a.wgu + (size_t)(l_) * a.w_stride, b, 48, 0)
#define PM_LOAD_W2(l_) do { const size_t lo_ = (size_t)(l_) * a.w_stride; \
        if (ro.two_gu) PM_LOAD6(w2, a.wgu + lo_, b + 128, 48, 0); else PM_LOAD6(w2, a.wd + lo_, ro.d_tile, 192, ro.d_slice * 48); } while (0)
    // layer 0's tiles, in the order of first use (vmcnt retires in order)
    if (b < 192) { PM_LOAD_W0(0); PM_LOAD_W1(0); PM_LOAD_W2(0); }
    else { PM_LOAD_W1(0); PM_LOAD_W2(0); PM_LOAD_W0(0); }
    __builtin_amdgcn_sched_barrier(0);
    PM_BAR();                                                    // S0: ctl is initialised

    // RMSNorm factor of the rows of chunk g from the producer's 48 per-tile sums of squares, in skinny_gemm.hip's PRO_XH order (4 lanes per row, 12 partials each)
    auto fac_rows = [&](int g) {
        const pm_rsrc_t rs_ssq = pm_rsrc(a.ssq);
        const int n_f = lane >> 2, part = lane & 3, r = g * 16 + n_f;
        const bool live = r < R;
        f32x4 sq[3];
#pragma unroll
        for (int i = 0; i < 3; ++i) sq[i] = live ? pm_ld16(rs_ssq, (unsigned)((r * 48 + part * 12 + 4 * i) * 4)) : (f32x4){0.f, 0.f, 0.f, 0.f};
        float ss = 0.f;
        if (live) {
#pragma unroll
            for (int i = 0; i < 3; ++i) ss += (sq[i][0] + sq[i][1]) + (sq[i][2] + sq[i][3]);
        }
        ss += dpp_f<DPP_XOR1>(ss);
        ss += dpp_f<DPP_XOR2>(ss);
        if (live && part == 0) fac[r] = 1.0f / sqrtf(ss / (float)PM_H + a.eps);
    };
    // this wave's B fragments of both chunks from a fragment-major operand image [chunk][ktall k-tiles][64 lanes][16 B]
#define PM_LOAD_B(bf_, buf_, ktall_, koff_) do { const pm_rsrc_t rs_ = pm_rsrc(buf_); \
        _Pragma("unroll") for (int g_ = 0; g_ < NCH; ++g_) _Pragma("unroll") for (int i_ = 0; i_ < 6; ++i_) \
            bf_[g_][i_] = pm_ld16(rs_, (unsigned)(((g_ * (ktall_) + (koff_) + wave * 6 + i_) * 64 + lane) * 16)); } while (0)
#define PM_MMA(acc_, w_, bf_) do { _Pragma("unroll") for (int i_ = 0; i_ < 6; ++i_) _Pragma("unroll") for (int g_ = 0; g_ < NCH; ++g_) acc_[g_] = pm_mma(w_[i_], bf_[g_][i_], acc_[g_]); } while (0)
#define PM_PARK(slot_, acc_) do { _Pragma("unroll") for (int g_ = 0; g_ < NCH; ++g_) \
        *(f32x4*)(red + (slot_) * (PM_GEMM_WAVES * NCH * 256) + ((wave * NCH + g_) * 64 + lane) * 4) = acc_[g_]; } while (0)

    for (int l = 0; l < NL; ++l) {
        const bool more = l + 1 < NL;
        // every per-thread address below is formed from these two inside the layer: made opaque per layer, or hipcc hoists ~25 loop-invariant 64-bit
        // addresses out of the layer loop and spills the weight tiles to make room for them (the hipcc 7.2 pitfall met in persist_layer.hip)
        asm volatile("" : "+v"(tid), "+v"(lane));
        // ================================================================ phase 1: RMSNorm + q | k | v + RoPE + cache append (llama.py:82-87,619-633)
        if (ro.has_qkv) {
            PM_BAR();
            PM_ABORT_CHECK();
            if (l == 0) {
                // the sampler's fp32 rows: y = x * rsqrt(mean(x^2) + eps) into LDS, fragment-major (skinny_gemm.hip PRO_NORM: a wave owns rows wave, wave + 8 of a chunk)
#pragma unroll
                for (int g = 0; g < NCH; ++g) {
                    f32x4 v[2][3];
#pragma unroll
                    for (int u = 0; u < 2; ++u) {
                        const int r = g * 16 + wave + u * 8;
#pragma unroll
                        for (int i = 0; i < 3; ++i) v[u][i] = (r < R) ? ((const f32x4*)(a.x + (size_t)r * PM_H))[lane + 64 * i] : (f32x4){0.f, 0.f, 0.f, 0.f};
                    }
#pragma unroll
                    for (int u = 0; u < 2; ++u) {
                        const int n = wave + u * 8;
                        if (g * 16 + n >= R) break;
                        float ss = 0.f;
#pragma unroll
                        for (int i = 0; i < 3; ++i) ss += v[u][i][0] * v[u][i][0] + v[u][i][1] * v[u][i][1] + v[u][i][2] * v[u][i][2] + v[u][i][3] * v[u][i][3];
                        ss = wave_sum(ss);
                        const float rs = 1.0f / sqrtf(ss / (float)PM_H + a.eps);
#pragma unroll
                        for (int i = 0; i < 3; ++i) {
                            const int k = 4 * (lane + 64 * i);
                            *(f32x4*)(bx + g * 48 * 256 + xfrag_index<float>(n, k, 48)) = (f32x4){v[u][i][0] * rs, v[u][i][1] * rs, v[u][i][2] * rs, v[u][i][3] * rs};
                        }
                    }
                }
                PM_BAR();
            }
            {
                f32x4 bf[NCH][6];
                // epilogue operands that do not depend on the products: requested with the B fragments
                const int eg = tid >> 7, en = (tid & 127) >> 3, ep = tid & 7, er = eg * 16 + en;
                const bool elive = tid < 128 * NCH && er < R;
                const int which = b / 48, within = b % 48, hh = within >> 2, dd = ((within & 3) << 3) + ep;
                RowMeta em = {0, 0, 0, 0};
                float rc = 1.f, rsn = 0.f;
                if (elive) {
                    em = a.meta[er];
                    if (which < 2) { rc = a.rope_rows[(size_t)er * 64 + dd]; rsn = a.rope_rows[(size_t)er * 64 + 32 + dd]; }
                }
                if (l == 0) {
#pragma unroll
                    for (int g = 0; g < NCH; ++g)
#pragma unroll
                        for (int i = 0; i < 6; ++i) bf[g][i] = ((const f32x4*)bx)[(g * 48 + wave * 6 + i) * 64 + lane];
                } else {
                    PM_LOAD_B(bf, a.xh, 48, 0);
                    if (wave == 7) fac_rows(0);
                    if (NCH == 2 && wave == 6) fac_rows(1);
                }
                f32x4 acc[NCH];
#pragma unroll
                for (int g = 0; g < NCH; ++g) acc[g] = (f32x4){0.f, 0.f, 0.f, 0.f};
                PM_MMA(acc, w0, bf);
                PM_PARK(0, acc);
                PM_BAR();
                if (elive) {
                    float va = pm_c<NCH>(red, ep, en, eg), vb = pm_c<NCH>(red, ep + 8, en, eg);
                    if (l > 0) { const float f = fac[er]; va *= f; vb *= f; }
                    float ya = va, yb = vb;
                    if (which < 2) {      // q * cos + rotate_half(q) * sin, products rounded separately like the reference (llama.py:180-181)
                        ya = __fadd_rn(__fmul_rn(va, rc), __fmul_rn(-vb, rsn));
                        yb = __fadd_rn(__fmul_rn(vb, rc), __fmul_rn(va, rsn));
                    }
                    if (which == 0) {
                        float* q = a.q_buf + ((size_t)er * PM_NH + hh) * CTTS_HEAD_DIM;
                        pm_st4(q + dd, ya); pm_st4(q + dd + 32, yb);
                    } else {              // KV append (llama.py:633)
                        float* c = (float*)a.kv + (size_t)l * 2 * a.kv_per + (which == 2 ? a.kv_per : 0) + (((size_t)em.seq * PM_NH + hh) * a.Lmax + em.slot) * CTTS_HEAD_DIM;
                        pm_st4(c + dd, ya); pm_st4(c + dd + 32, yb);
                    }
                }
                pm_drain();
            }
            PM_BAR();
            if (more) PM_LOAD_W0(l + 1);
        }
        // ================================================================ phase 2: attention.  The workgroup's first item runs on the attention waves; its second one
        // (17..32 rows: items b + 256) on GEMM waves 0..3, which idle here anyway -- both at once, so the phase lasts ONE item whatever the workgroup's share
        if (b < R * PM_NH) {
            PM_BAR();
            PM_ABORT_CHECK();
            const int item2 = b + PM_BLOCKS;
            const bool mine = item2 < R * PM_NH && wave < PM_ATT_WAVES;
            if (mine) {
                float* const kbase = (float*)a.kv + (size_t)l * 2 * a.kv_per;
                pm_attn_item<3>(a, pm_rsrc(kbase), pm_rsrc(kbase + a.kv_per), pm_rsrc(a.q_buf), item2 / PM_NH, item2 % PM_NH, wave, lane, merge + PM_ATT_WAVES * 80);
            }
            PM_BAR();
            if (mine && wave == 0) {
                if (lane < 8) pm_attn_finish(a, item2 / PM_NH, item2 % PM_NH, lane, merge + PM_ATT_WAVES * 80);
                pm_drain();
            }
            PM_BAR();
        }
        // ================================================================ phase 3: o_proj + residual (llama.py:666,731) -> x, per-tile sums of squares, packed copy
        if (ro.has_o) {
            const int rt = b - 144;
            PM_BAR();
            PM_ABORT_CHECK();
            {
                f32x4 bf[NCH][6];
                const int eg = tid >> 8, en = (tid & 255) >> 4, ei = tid & 15, er = eg * 16 + en, col = rt * 16 + ei;
                const bool elive = tid < 256 * NCH && er < R;
                float resid = 0.f;
                if (elive) resid = pm_ld4(a.x + (size_t)er * PM_H + col);
                PM_LOAD_B(bf, a.attn_packed, 48, 0);
                f32x4 acc[NCH];
#pragma unroll
                for (int g = 0; g < NCH; ++g) acc[g] = (f32x4){0.f, 0.f, 0.f, 0.f};
                PM_MMA(acc, w0, bf);
                PM_PARK(0, acc);
                PM_BAR();
                if (elive) {
                    const float xn = resid + pm_c<NCH>(red, ei, en, eg);
                    pm_st4(a.x + (size_t)er * PM_H + col, xn);
                    float sq = xn * xn;
                    sq += dpp_f<DPP_XOR1>(sq); sq += dpp_f<DPP_XOR2>(sq); sq += dpp_f<DPP_HALF_MIRROR>(sq); sq += dpp_f<DPP_MIRROR>(sq);
                    if (ei == 0) pm_st4(a.ssq + (size_t)er * 48 + rt, sq);
                    pm_st4(a.xh + (size_t)eg * 48 * 256 + xfrag_index<float>(en, col, 48), xn);
                }
                pm_drain();
            }
            PM_BAR();
            if (more) PM_LOAD_W0(l + 1);
        }
        // ================================================================ phase 4: RMSNorm + gate | up + SiLU * up (llama.py:214)
        {
            PM_BAR();
            PM_ABORT_CHECK();
            {
                f32x4 bf[NCH][6];
                PM_LOAD_B(bf, a.xh, 48, 0);
                if (wave == 7) fac_rows(0);
                if (NCH == 2 && wave == 6) fac_rows(1);
                f32x4 acc[NCH];
#pragma unroll
                for (int g = 0; g < NCH; ++g) acc[g] = (f32x4){0.f, 0.f, 0.f, 0.f};
                PM_MMA(acc, w1, bf);
                PM_PARK(0, acc);
                if (ro.two_gu) {
#pragma unroll
                    for (int g = 0; g < NCH; ++g) acc[g] = (f32x4){0.f, 0.f, 0.f, 0.f};
                    PM_MMA(acc, w2, bf);
                    PM_PARK(1, acc);
                }
            }
            PM_BAR();
            {
                constexpr int PER = 128 * NCH;                  // epilogue threads per tile
                const int ts = tid / PER, t = tid % PER;
                if (ts < (ro.two_gu ? 2 : 1)) {
                    const int eg = t >> 7, en = (t & 127) >> 3, ep = t & 7, er = eg * 16 + en;
                    const int rt = ts ? b + 128 : b;
                    const float* rd = red + ts * (PM_GEMM_WAVES * NCH * 256);
                    float y = 0.f;
                    if (er < R) {
                        const float f = fac[er];
                        const float va = pm_c<NCH>(rd, ep, en, eg) * f, vb = pm_c<NCH>(rd, ep + 8, en, eg) * f;
                        y = (va / (1.0f + expf(-va))) * vb;
                    }
                    pm_st4(a.act + (size_t)eg * 192 * 256 + xfrag_index<float>(en, rt * 8 + ep, 192), y);
                }
                pm_drain();
            }
            PM_BAR();
            if (more) { PM_LOAD_W1(l + 1); if (ro.two_gu) PM_LOAD_W2(l + 1); }
        }
        // ================================================================ phase 5: down projection, K in four slices, + residual (llama.py:737-739)
        if (ro.has_d) {
            PM_BAR();
            PM_ABORT_CHECK();
            const int eg = tid >> 8, et = tid & 255, en = et >> 4, ei = et & 15, er = eg * 16 + en, col = ro.d_tile * 16 + ei;
            const bool ethr = tid < 256 * NCH, elive = ethr && er < R;
            float resid = 0.f;
            {
                f32x4 bf[NCH][6];
                if (elive) resid = pm_ld4(a.x + (size_t)er * PM_H + col);
                PM_LOAD_B(bf, a.act, 192, ro.d_slice * 48);
                f32x4 acc[NCH];
#pragma unroll
                for (int g = 0; g < NCH; ++g) acc[g] = (f32x4){0.f, 0.f, 0.f, 0.f};
                if (b < 128) PM_MMA(acc, w2, bf);
                else PM_MMA(acc, w0, bf);
                PM_PARK(0, acc);
            }
            PM_BAR();
            // every slice parks its partial tile (write-through); the LAST arriver adds the four slices in slice order (skinny_gemm.hip EPI_RESID_XH_SK)
            float sk = 0.f;
            float* const slab_t = a.slab + (size_t)ro.d_tile * 4 * 2 * 256;
            if (ethr) {
                sk = pm_c<NCH>(red, ei, en, eg);
                pm_st4(slab_t + (ro.d_slice * 2 + eg) * 256 + et, sk);
            }
            pm_drain();
            PM_BAR();
            if (tid == 0) {
                const int ticket = __hip_atomic_fetch_add(a.cnt + ro.d_tile, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (ticket == 3) __hip_atomic_store(a.cnt + ro.d_tile, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // ready for the next layer
                __hip_atomic_store(ctl + 1, (ticket == 3) ? 1 : 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
            PM_BAR();
            const bool last = __hip_atomic_load(ctl + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) != 0;
            if (last) {
                if (ethr) {
                    float pq[4];
#pragma unroll
                    for (int q = 0; q < 4; ++q) pq[q] = (q == ro.d_slice) ? sk : pm_ld4(slab_t + (q * 2 + eg) * 256 + et);
                    const float sum = ((pq[0] + pq[1]) + pq[2]) + pq[3];
                    if (elive) {
                        const float xn = resid + sum;
                        pm_st4(a.x + (size_t)er * PM_H + col, xn);
                        float sq = xn * xn;
                        sq += dpp_f<DPP_XOR1>(sq); sq += dpp_f<DPP_XOR2>(sq); sq += dpp_f<DPP_HALF_MIRROR>(sq); sq += dpp_f<DPP_MIRROR>(sq);
                        if (ei == 0) pm_st4(a.ssq + (size_t)er * 48 + ro.d_tile, sq);
                        pm_st4(a.xh + (size_t)eg * 48 * 256 + xfrag_index<float>(en, col, 48), xn);
                    }
                }
                pm_drain();
                PM_BAR();
            }
            PM_BAR();                                             // (ctl[1] is rewritten by the next layer's ticket only after everybody has read it)
            if (more) { if (b < 128) PM_LOAD_W2(l + 1); else PM_LOAD_W0(l + 1); }
        }
    }
}

// ---------------------------------------------------------------------------------------------------- waves 8..11: attention items; wave 8 also polls and publishes
template <int NCH>
__device__ __forceinline__ void pm_att_role(const PmArgs& a, float* const merge, pm_lds_int* const ctl, const int tid, const int lane, const int wave, const int b, const unsigned tag0) {
    const PmRole ro(b);
    const int R = a.R, NL = a.n_layers;
    const bool poller = wave == PM_GEMM_WAVES;
    const int aw = wave - PM_GEMM_WAVES;
    unsigned* const f_qkv = a.flags, * const f_att = a.flags + 256, * const f_o = a.flags + 512, * const f_gu = a.flags + 768, * const f_d = a.flags + 1024;
    const int n_items = R * PM_NH, n_att_wg = n_items < PM_BLOCKS ? n_items : PM_BLOCKS;
    // diagnostics: the poller's lane 0 stamps the last layer's phase boundaries straight into a.ts (no registers held)
#define PM_MARK(i_) do { if (a.ts != nullptr && l + 1 == NL && poller && lane == 0) a.ts[(size_t)b * PM_NTS + (i_)] = wall_clock64(); } while (0)
    PM_BAR();                                                    // S0
    for (int l = 0; l < NL; ++l) {
        const unsigned tag = tag0 + (unsigned)l * 8u;
        // ---- phase 1
        if (ro.has_qkv) {
            if (poller && l > 0) (void)pm_wait(f_d, PM_O_TILES, tag - 8u + 5u, a.error, 8, ctl, lane, a.nap, a.delay[0]);
            PM_MARK(0);
            PM_BAR();
            PM_ABORT_CHECK();
            if (l == 0) PM_BAR();
            PM_BAR();
            PM_BAR();
            if (poller && lane == 0 && !(a.fault > 0 && b == 5 && l + 1 == a.fault)) PM_PUBLISH(f_qkv + b, tag + 1u);
            PM_MARK(1);
        }
        // ---- phase 2: attention, one (row, head) item at a time (llama.py:653-661)
        if (b < n_items) {
            if (poller) (void)pm_wait(f_qkv, PM_QKV_TILES, tag + 1u, a.error, 9, ctl, lane, a.nap, a.delay[1]);
            PM_MARK(2);
            PM_BAR();
            PM_ABORT_CHECK();
            float* const kbase = (float*)a.kv + (size_t)l * 2 * a.kv_per;          // this layer's K block; V at + kv_per
            pm_attn_item<PM_UN>(a, pm_rsrc(kbase), pm_rsrc(kbase + a.kv_per), pm_rsrc(a.q_buf), b / PM_NH, b % PM_NH, aw, lane, merge);
            PM_BAR();
            if (poller && lane < 8) pm_attn_finish(a, b / PM_NH, b % PM_NH, lane, merge);
            if (poller) pm_drain();
            PM_BAR();
            if (poller && lane == 0) PM_PUBLISH(f_att + b, tag + 2u);
            PM_MARK(3);
        }
        // ---- phase 3
        if (ro.has_o) {
            if (poller) (void)pm_wait(f_att, n_att_wg, tag + 2u, a.error, 10, ctl, lane, a.nap, a.delay[2]);
            PM_MARK(4);
            PM_BAR();
            PM_ABORT_CHECK();
            PM_BAR();
            PM_BAR();
            if (poller && lane == 0) PM_PUBLISH(f_o + (b - 144), tag + 3u);
            PM_MARK(5);
        }
        // ---- phase 4
        {
            if (poller) (void)pm_wait(f_o, PM_O_TILES, tag + 3u, a.error, 11, ctl, lane, a.nap, a.delay[3]);
            PM_MARK(6);
            PM_BAR();
            PM_ABORT_CHECK();
            PM_BAR();
            PM_BAR();
            if (poller && lane == 0) PM_PUBLISH(f_gu + b, tag + 4u);
            PM_MARK(7);
        }
        // ---- phase 5
        if (ro.has_d) {
            if (poller) (void)pm_wait(f_gu, PM_BLOCKS, tag + 4u, a.error, 12, ctl, lane, a.nap, a.delay[4]);
            PM_MARK(8);
            PM_BAR();
            PM_ABORT_CHECK();
            PM_BAR();
            PM_BAR();
            PM_BAR();
            const bool last = __hip_atomic_load(ctl + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) != 0;
            PM_MARK(9);
            if (last) {
                PM_BAR();
                if (poller && lane == 0) PM_PUBLISH(f_d + ro.d_tile, tag + 5u);
            }
            PM_BAR();
            PM_MARK(10);
        }
    }
}

template <int NCH>
__global__ __launch_bounds__(PM_THREADS) void persist_mfma_kernel(const PmArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* const bx = (float*)smem;                              // layer 0: the normalised rows as B operand [NCH][48 k-tiles][64 lanes][4]
    float* const red = bx + NCH * 48 * 256;                      // partial C tiles [2 tile slots][8 waves][NCH][64 lanes][4]
    float* const fac = red + 2 * PM_GEMM_WAVES * NCH * 256;      // [32] RMSNorm factor of every row
    float* const merge = fac + 32;                               // attention: [2][4 waves][8][10]
    pm_lds_int* const ctl = (pm_lds_int*)(merge + 2 * PM_ATT_WAVES * 80);      // [0] give-up flag, [1] "this workgroup is the last K slice of its down tile"
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int b = blockIdx.x;
    const int done_v = vload_flag(a.done), err_v = vload_flag(a.error), ep_v = vload_flag((const int*)a.epoch);
    if (__builtin_amdgcn_readfirstlane(done_v | err_v)) return;      // every sequence finished (gpt.py:545) / an earlier launch gave up: the same for every workgroup
    if (tid == 0) { ctl[0] = 0; ctl[1] = 0; }
    if (wave < PM_GEMM_WAVES) pm_gemm_role<NCH>(a, bx, red, fac, merge, ctl, tid, lane, wave, b);
    else pm_att_role<NCH>(a, merge, ctl, tid, lane, wave, b, (unsigned)ep_v << 8);      // tag = launch counter << 8 | layer * 8 + phase + 1
    // Advance the launch counter.  Safe although other workgroups may still be running: workgroup 0 owns a down item, i.e. it has seen every
    // workgroup's gate|up flag of the last layer, i.e. every workgroup has long read its copy at entry.
    if (b == 0 && tid == 0) __hip_atomic_fetch_add(a.epoch, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

static size_t pm_lds_bytes(int nch) { return (size_t)(nch * 48 * 256 + 2 * PM_GEMM_WAVES * nch * 256 + 32 + 2 * PM_ATT_WAVES * 80) * 4 + 64; }
size_t persist_mfma_slab_floats() { return (size_t)48 * 4 * 2 * 256; }

int persist_mfma_configure() {
    CTTS_HIP_CHECK(hipFuncSetAttribute((const void*)persist_mfma_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)pm_lds_bytes(1)));
    CTTS_HIP_CHECK(hipFuncSetAttribute((const void*)persist_mfma_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)pm_lds_bytes(2)));
    return 0;
}

int launch_persist_mfma(const PmArgs& a, hipStream_t s) {
    if (a.R < 1 || a.R > PM_MAXR || a.n_layers > 31) { ctts_set_error("persistent MFMA stack: %d rows / %d layers", a.R, a.n_layers); return 1; }
    if (a.R <= 16) hipLaunchKernelGGL(persist_mfma_kernel<1>, dim3(PM_BLOCKS), dim3(PM_THREADS), pm_lds_bytes(1), s, a);
    else hipLaunchKernelGGL(persist_mfma_kernel<2>, dim3(PM_BLOCKS), dim3(PM_THREADS), pm_lds_bytes(2), s, a);
    CTTS_HIP_CHECK(hipGetLastError());
    return 0;
}
