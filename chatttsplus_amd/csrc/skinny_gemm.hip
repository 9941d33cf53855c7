// Decode-time projections of the Llama-style decoder as MFMA "skinny GEMMs" (gfx950).
//
//   out[R][N] = prologue(x)[R][K] . W[N][K]^T  + fused epilogue,   R <= 16*NBG rows per chunk
//
// The weight matrix is the MFMA *A* operand (M dimension = output features) and the activations are
// the *B* operand (N dimension = batch rows), so one 1-KiB weight tile = one coalesced 16-byte load per
// lane = one v_mfma_f32_16x16x32_f16 (fp16) or four v_mfma_f32_16x16x4_f32 (fp32 parity mode).  Weights
// are streamed from HBM exactly once per step with non-temporal loads issued *before* the prologue; a
// block owns one 16-row tile, its waves split K and combine through LDS in a fixed order (deterministic).
//
// Reference arithmetic restated by the fused pieces:
//   PRO_NORM   LlamaRMSNorm.forward                       chattts_plus/models/llama.py:82-87
//   PRO_ATTN   softmax normalisation of the split-K attention partials (attention.hip)
//   EPI_QKV    q/k/v proj + apply_rotary_pos_emb + cache append   llama.py:619-633,151-182
//   EPI_RESID  o_proj / down_proj + residual add                  llama.py:666,731,737-739
//   EPI_SWIGLU act_fn(gate_proj(x)) * up_proj(x)                  llama.py:214
//   EPI_LOGITS head_code[i](hidden) for the 4 folded heads        gpt.py:437-447
#include "kernels.h"
#include "lora_worker.h"

#define ATT_SMAX 8     // max key splits combined by PRO_ATTN (gpt_engine.hip decode_splits)

template <typename WT> struct Mma;
template <> struct Mma<half_t> {
    __device__ static inline f32x4 run(half8 a, half8 b, f32x4 c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
    }
};
template <> struct Mma<float> {
    __device__ static inline f32x4 run(f32x4 a, f32x4 b, f32x4 c) {
        c = __builtin_amdgcn_mfma_f32_16x16x4f32(a[0], b[0], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x4f32(a[1], b[1], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x4f32(a[2], b[2], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x4f32(a[3], b[3], c, 0, 0, 0);
        return c;
    }
};

struct SplitFrag { half8 hi, lo; };
template <> struct Mma<split_t> {
    __device__ static inline f32x4 run(const SplitFrag& a, const SplitFrag& b, f32x4 c) {
        // tails first, head product last (prefill_split.hip): the small terms meet while the accumulator's low bits still see them
        c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a.lo, b.hi, c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a.hi, b.lo, c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a.hi, b.hi, c, 0, 0, 0);
        return c;
    }
};

template <typename WT> struct FragOf;
template <> struct FragOf<half_t> { typedef half8 type; };
template <> struct FragOf<float> { typedef f32x4 type; };
template <> struct FragOf<split_t> { typedef SplitFrag type; };

// fragment `tile` (= 64 lanes x 16 B; split: head | tail) of an operand image, this lane's part.  nt: streamed once (weights)
template <typename WT> struct FragIO {
    typedef typename FragOf<WT>::type frag;
    __device__ static inline frag load(const void* base, size_t tile, int lane) { return ((const frag*)base)[tile * 64 + lane]; }
    __device__ static inline frag load_nt(const void* base, size_t tile, int lane) { return __builtin_nontemporal_load((const frag*)base + tile * 64 + lane); }
};
template <> struct FragIO<split_t> {
    typedef SplitFrag frag;
    __device__ static inline frag load(const void* base, size_t tile, int lane) {
        const half8* p = (const half8*)base + tile * 128 + lane;
        return SplitFrag{p[0], p[64]};
    }
    __device__ static inline frag load_nt(const void* base, size_t tile, int lane) {
        const half8* p = (const half8*)base + tile * 128 + lane;
        return SplitFrag{__builtin_nontemporal_load(p), __builtin_nontemporal_load(p + 64)};
    }
};

// store 4 consecutive-k activations of row n into the fragment-major LDS/global image
template <typename WT>
__device__ inline void store_x4(void* base, int n, int k, int ktiles, float y0, float y1, float y2, float y3);
template <>
__device__ inline void store_x4<half_t>(void* base, int n, int k, int ktiles, float y0, float y1, float y2, float y3) {
    half4 h = {(half_t)y0, (half_t)y1, (half_t)y2, (half_t)y3};
    *(half4*)((half_t*)base + xfrag_index<half_t>(n, k, ktiles)) = h;
}
template <>
__device__ inline void store_x4<float>(void* base, int n, int k, int ktiles, float y0, float y1, float y2, float y3) {
    f32x4 v = {y0, y1, y2, y3};
    *(f32x4*)((float*)base + xfrag_index<float>(n, k, ktiles)) = v;
}
template <>
__device__ inline void store_x4<split_t>(void* base, int n, int k, int ktiles, float y0, float y1, float y2, float y3) {
    half_t hh[4], ll[4];
    split_half(y0, hh[0], ll[0], nullptr); split_half(y1, hh[1], ll[1], nullptr); split_half(y2, hh[2], ll[2], nullptr); split_half(y3, hh[3], ll[3], nullptr);
    const half4 h = {hh[0], hh[1], hh[2], hh[3]}, l = {ll[0], ll[1], ll[2], ll[3]};
    char* p = (char*)base + xfrag_split_bytes(n, k, ktiles);
    *(half4*)p = h;
    *(half4*)(p + 1024) = l;
}
// one element (n, k) of the fragment-major image whose first tile is `tile0` (epilogues: the packed residual copy, the SwiGLU output)
template <typename WT> __device__ inline void store_x1(void* base, size_t tile0, int n, int k, int ktiles, float v, int* sat) {
    WT* dst = (WT*)base + tile0 * 64 * WTraits<WT>::EPL;
    dst[xfrag_index<WT>(n, k, ktiles)] = sat_store<WT>(v, sat);
}
template <> __device__ inline void store_x1<split_t>(void* base, size_t tile0, int n, int k, int ktiles, float v, int* sat) {
    half_t hi, lo;
    split_half(v, hi, lo, sat);
    char* p = (char*)base + tile0 * 2048 + xfrag_split_bytes(n, k, ktiles);
    *(half_t*)p = hi;
    *(half_t*)(p + 1024) = lo;
}

// A prefetch block (kernels.h WPrefetch): block j of npf (a multiple of 8), sitting on XCD `xcd` (its linear block index mod 8), pulls the units u = xcd (mod 8) -- as
// share j / 8 of the npf / 8 blocks of that XCD.
// LDS-DMA: no destination registers to protect (cdna_hip_programming.md 5.7 item 1), nothing waits; s_endpgm drains the queue.
template <int WAVES>
__device__ inline void weight_prefetch(const WPrefetch pf, const int j, const int npf, const int xcd, char* smem, const int tid) {
    typedef __attribute__((address_space(1))) const void* gptr_t;
    typedef __attribute__((address_space(3))) void* lptr_t;
    if (pf.ptr == nullptr || npf < 8) return;
    const int sub = j >> 3, nsub = npf >> 3;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    char* dst = smem + wave * 1024;                        // (every DMA of a wave lands on the same 1 KiB: the bytes are not read here)
    for (unsigned u = xcd + 8 * sub; u < pf.n_units; u += 8 * nsub) {
        const char* p = (const char*)pf.ptr + (size_t)u * pf.unit_bytes + (unsigned)(tid * 16);
        for (unsigned off = 0; off < pf.unit_bytes; off += WAVES * 1024)
            __builtin_amdgcn_global_load_lds((gptr_t)(p + off), (lptr_t)dst, 16, 0, 0);
    }
}

// RT = weight row tiles per block: the activation prologue (replicated in every block) is paid once per RT tiles --
// at 17-32 rows the 384 blocks of gate|up otherwise pull 37 MB of residual stream through L2 per launch.
// VR > 0 (fp32 engines, decode batches of <= VR rows): the products run on the VALU.  An exact-f32 MFMA (v_mfma_f32_16x16x4_f32) issues in 32 cycles
// whatever the number of live B columns -- 24 of them per wave and launch (48 per SIMD with two waves on it: 0.64 us of a ~3.5 us kernel at batch 1, twice
// that for the 128 CUs that hold two gate|up blocks) -- while the same weight fragment times VR activation rows is 4 * VR v_fma per lane.  A lane keeps its
// own weight row (lane & 15) and k-group (lane >> 4) of the MFMA-A image, so the packed weights are shared with the MFMA kernels; the four k-groups
// are summed through the LDS crossbar and the waves' partials through `red`, in wave order (deterministic).
template <typename WT, int NBG, int WAVES, int KPW, int PRO, int EPI, int RT = 1, int VR = 0, bool LORA = false>
__global__ __launch_bounds__(WAVES * 64) void skinny_gemm_kernel(const int* done_p, const void* Wq, const void* in0, const void* in1,
                                                                  const float* resid_in, const int R, const int misc, const GemmArgs a) {
    // misc = np | S << 3 | ktiles_total << 7 | lora workers << 15 | first prefetch block << 22: the struct fields that address the first loads of some variants (partial
    // sums, attention splits, split-K weight offset) -- kept out of the by-value struct for the same reason as the pointers
    // The six leading scalars are what the first loads of the kernel need (flag, weights, the prologue's operands, the
    // residual rows, the row count).  As plain kernel arguments they are preloaded into SGPRs at wave launch
    // (-mllvm -amdgpu-kernarg-preload-count, build.py); the by-value struct behind them costs a scalar-cache miss that now
    // overlaps the operand loads instead of preceding them.
    //   in0/in1: PRO_NORM[_P] x / opart   PRO_ATTN part_ml / part_o   PRO_PACKED xpacked / -
    static_assert(RT == 1 || EPI == EPI_QKV || EPI == EPI_SWIGLU, "multi-tile blocks: QKV / SwiGLU epilogues only");
    static_assert(VR == 0 || (sizeof(WT) == 4 && NBG == 1 && RT == 1 && PRO != PRO_XH && VR <= 4), "VALU products: fp32, one 16-row chunk, <= 4 rows");
    constexpr bool VALU = VR > 0;
    constexpr bool SPLIT = (WTraits<WT>::TILE_BYTES == 2048);
    typedef typename FragOf<WT>::type frag;
    constexpr int KT = WTraits<WT>::KT;
    constexpr int KTILES = WAVES * KPW;
    constexpr int K = KTILES * KT;
    constexpr int NB = 16 * NBG;
    constexpr int XS_BYTES = (PRO == PRO_PACKED || PRO == PRO_XH) ? 0 : NBG * KTILES * WTraits<WT>::TILE_BYTES;   // LDS image of the B operand
    constexpr int PER = K / 256;                      // float4 per lane per row in the prologues
    extern __shared__ __attribute__((aligned(16))) char smem[];

    // every sequence finished (gpt.py:545): skip on device -- the flag is requested here, tested after the operand loads
    int done_v = 0;
    if (done_p != nullptr) done_v = vload_flag(done_p);
#define CTTS_EXIT_IF_DONE() if (__builtin_amdgcn_readfirstlane(done_v)) return

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // prefetch blocks (kernels.h WPrefetch): the grid's last x columns.  Compiled into the ONE launch that carries them: o_proj on the packed-residual path.  (Compiled into
    // every instantiation, the role cost each launch that never used it 0.3 us -- the <= 8-row chain 6 %, profiles/r06_ab_chain_vs_r05.jsonl -- and as carriers the launches
    // that stream large matrices themselves lose more than their consumers gain, profiles/r06_ab_weight_prefetch.jsonl.)
    constexpr bool PFCAP = (EPI == EPI_RESID_XH) && (PRO == PRO_PACKED) && (K == 768) && VR == 0 && !LORA;
    if constexpr (PFCAP) {
        const int xreal = (int)((unsigned)misc >> 22);
        if (__builtin_expect(xreal != 0 && (int)blockIdx.x >= xreal, 0)) {      // (unlikely: the block belongs at the END of the code -- laid out in front of the main path it cost every
                                                                                 //  launch of the <= 8-row chain 0.3 us: a taken branch over 1 KB of cold code at every wave's start)
            if (blockIdx.y == 0 && blockIdx.z == 0) {
                weight_prefetch<WAVES>(a.pf, (int)blockIdx.x - xreal, (int)gridDim.x - xreal, (int)blockIdx.x & 7, smem, tid);
            }
            return;
        }
    }
    const int chunk = blockIdx.y;
    const int row0 = chunk * NB;
    // per-utterance LoRA (lora_worker.h): the first lw blocks of every chunk evaluate the rows' low-rank terms, the tiles behind them pick the terms up in
    // their epilogues.  (lw rides on the leading scalar `misc` like the other fields the first instructions need.)
    // (LORA is a template argument: compiled into every capable launch, the worker branch and the look-ahead cost the adapter-less step 0.5 % in fp32 and 2 % in fp16)
    constexpr bool LORA_QKV = LORA && (EPI == EPI_QKV) && (K == 768) && (VR == 0) && (PRO == PRO_NORM || PRO == PRO_XH);
    constexpr bool LORA_O = LORA && (EPI == EPI_RESID || EPI == EPI_RESID_XH) && (K == 768) && (VR == 0) && (PRO == PRO_PACKED);
    static_assert(!LORA || LORA_QKV || LORA_O, "LoRA workers: the q/k/v and o_proj launches of a decode step");
    const int lw = (LORA_QKV || LORA_O) ? ((misc >> 15) & 0x7F) : 0;
    int lora_draw = 0;
    // (the layer index is read where the tag is formed: an unconditional read of the argument struct at entry cost the adapter-less launches 0.2 us)
#define CTTS_LORA_TAG_LO ((unsigned)a.lf.layer * 2u + (LORA_O ? 1u : 0u))
    if constexpr (LORA_QKV || LORA_O) {
        if (lw != 0) {
            lora_draw = vload_flag(done_p - 1);                // DevState: draw sits in front of all_done.  Requested now, used where the tag is needed
            if ((int)blockIdx.x < lw) {                         // (no early exit on `done`: the worker's tiles exit themselves)
                const int w = blockIdx.x;
                if constexpr (LORA_QKV) {
                    const int r = row0 + w / 3;
                    if (r < R) lora_worker_qkv<WAVES>(a.lf, a.x, a.eps, r, w % 3, lora_draw, CTTS_LORA_TAG_LO, (float*)smem, tid);
                } else {
                    const int r = row0 + w;
                    if (r < R) lora_worker_o<WT, WAVES>(a.lf, a.xpacked, NBG, r, lora_draw, CTTS_LORA_TAG_LO, (float*)smem, tid);
                }
                return;
            }
        }
    }
    const int rt0 = ((int)blockIdx.x - lw) * RT;

    // LOAD ORDER MATTERS: vmcnt retires in order, so a wait on any load issued after the weight stream is a wait on
    // the whole stream.  Everything the prologue needs is therefore requested first, the (non-temporal) weight
    // fragments last; nothing issued after them is consumed before the MFMAs.
    // split-K launches (EPI_PART): this block owns k-tiles [blockIdx.z*KTILES, +KTILES) of a matrix with ktiles_total k-tiles
    const int np_ = misc & 0x7, S_ = (misc >> 3) & 0xF;
    constexpr bool SLICED = (EPI == EPI_PART || EPI == EPI_RESID_XH_SK);
    const int kt_all = SLICED ? ((misc >> 7) & 0xFF) : KTILES;
    const int kt_off = SLICED ? (int)blockIdx.z * KTILES : 0;
    const size_t wt0 = (size_t)rt0 * kt_all + kt_off + (size_t)wave * KPW;       // this wave's first weight fragment
    frag wf[RT][KPW];
    // (sched_barrier: hipcc otherwise hoists the weight loads above the prologue loads again)
#define CTTS_ISSUE_WEIGHT_LOADS()                                                            \
    __builtin_amdgcn_sched_barrier(0);                                                       \
    _Pragma("unroll") for (int t = 0; t < RT; ++t)                                           \
    _Pragma("unroll") for (int i = 0; i < KPW; ++i) wf[t][i] = FragIO<WT>::load_nt(Wq, wt0 + (size_t)t * kt_all + i, lane); \
    __builtin_amdgcn_sched_barrier(0)

    // 1b. epilogue operands that do not depend on the GEMM are requested now and consumed at the very end, so
    //     their L2/HBM round trips overlap the weight stream instead of forming a dependent tail
    constexpr int RITEMS = (16 * NB + WAVES * 64 - 1) / (WAVES * 64);
    float resid_pf[RITEMS];
    float xh_scale_pf[RITEMS];
    if (EPI == EPI_RESID || EPI == EPI_RESID_P || EPI == EPI_RESID_XH || EPI == EPI_RESID_XH_SK) {
#pragma unroll
        for (int u = 0; u < RITEMS; ++u) {
            const int t = tid + u * WAVES * 64;
            const int r = row0 + (t >> 4);
            // scale_in rides on the leading scalar in1 (free for PRO_PACKED); the rare PRO_ATTN variant reads it from the struct
            xh_scale_pf[u] = ((EPI == EPI_RESID_XH || EPI == EPI_RESID_XH_SK) && t < 16 * NB && r < R) ? ((PRO == PRO_ATTN) ? a.scale_in[r] : ((const float*)in1)[r])
                           : 1.f;
            const int N = a.n_row_tiles * 16, col = rt0 * 16 + (t & 15);
            float v = 0.f;
            if (t < 16 * NB && r < R) {
                v = resid_in[(size_t)r * N + col];
                if (EPI == EPI_RESID_P) {            // x = ((x + p0) + p1) + ... : partial sums in index order
                    float pp[CTTS_NPART];
#pragma unroll
                    for (int q = 0; q < CTTS_NPART; ++q) pp[q] = (q < np_) ? a.opart[((size_t)r * np_ + q) * N + col] : 0.f;
#pragma unroll
                    for (int q = 0; q < CTTS_NPART; ++q) v += pp[q];
                }
            }
            resid_pf[u] = v;
        }
    }
    RowMeta meta_pf = {0, 0, 0, 0};
    float rope_c[RT], rope_s[RT];
#pragma unroll
    for (int t = 0; t < RT; ++t) { rope_c[t] = 1.f; rope_s[t] = 0.f; }
    if (EPI == EPI_QKV) {
        const int r = row0 + (tid >> 3);
        if (tid < 8 * NB && r < R) {
            meta_pf = a.meta[r];
#pragma unroll
            for (int t = 0; t < RT; ++t) {
                const int d = ((((rt0 + t) % (K / 16)) & 3) << 3) + (tid & 7);
                rope_c[t] = a.rope_rows[(size_t)r * 64 + d];          // per-row copy of the table row: independent of meta
                rope_s[t] = a.rope_rows[(size_t)r * 64 + 32 + d];
            }
        }
    }

    // 2. prologue: build the B operand (activations) in LDS, fragment-major
    if (PRO == PRO_NORM || PRO == PRO_NORM_P) {
        // y = x * rsqrt(mean(x^2)+eps); the RMSNorm *weight* is folded into W's columns at pack time (gpt_engine.hip).
        // rows beyond R are left unwritten: an MFMA output column depends only on its own B column, and the
        // epilogues never read columns >= R.  A wave owns rows wave, wave+WAVES, ...; the loads of RB rows are
        // issued together (one L2 round trip per batch instead of one per row: 8 serial trips at batch 32).
        constexpr int RB = (NB + WAVES - 1) / WAVES;          // rows per wave: one batch (<= 4 for every tiling used)
        static_assert(PRO != PRO_NORM_P || RB <= 4, "partials path is for <= 16 rows");
        const int rows = min(NB, R - row0);
        f32x4 v[RB][PER];
#pragma unroll
        for (int u = 0; u < RB; ++u) {
            const int n = wave + u * WAVES;
            if (n < rows) {
                const f32x4* xr = (const f32x4*)((const float*)in0 + (size_t)(row0 + n) * K);
#pragma unroll
                for (int i = 0; i < PER; ++i) v[u][i] = xr[lane + 64 * i];
            } else {
#pragma unroll
                for (int i = 0; i < PER; ++i) v[u][i] = (f32x4){0.f, 0.f, 0.f, 0.f};
            }
        }
        if (PRO == PRO_NORM_P) {     // residual stream = x + per-head o_proj partials (fused attention path), head order
#pragma unroll
            for (int u = 0; u < RB; ++u) {
                const int n = wave + u * WAVES;
                if (n >= rows) break;
                const size_t rr = (size_t)(row0 + n);
                f32x4 pp[CTTS_NPART][PER];
#pragma unroll
                for (int q = 0; q < CTTS_NPART; ++q)
#pragma unroll
                    for (int i = 0; i < PER; ++i)
                        pp[q][i] = (q < np_) ? ((const f32x4*)((const float*)in1 + (rr * np_ + q) * K))[lane + 64 * i] : (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int q = 0; q < CTTS_NPART; ++q)
#pragma unroll
                    for (int i = 0; i < PER; ++i) v[u][i] += pp[q][i];
            }
        }
        CTTS_ISSUE_WEIGHT_LOADS();
        CTTS_EXIT_IF_DONE();
#pragma unroll
        for (int u = 0; u < RB; ++u) {
            const int n = wave + u * WAVES;
            if (n >= rows) break;
            const int r = row0 + n;
            float ss = 0.f;
#pragma unroll
            for (int i = 0; i < PER; ++i) ss += v[u][i][0] * v[u][i][0] + v[u][i][1] * v[u][i][1] + v[u][i][2] * v[u][i][2] + v[u][i][3] * v[u][i][3];
            ss = wave_sum(ss);
            const float rs = 1.0f / sqrtf(ss / (float)K + a.eps);          // torch.rsqrt(mean(x^2) + eps)
            if ((PRO == PRO_NORM || PRO == PRO_NORM_P) && EPI == EPI_QKV && rt0 == 0 && a.scale_out != nullptr && lane == 0)
                a.scale_out[r] = __builtin_bit_cast(float, __builtin_bit_cast(unsigned, rs) & 0x7F800000u);   // see PRO_XH
#pragma unroll
            for (int i = 0; i < PER; ++i) {
                const int k = 4 * (lane + 64 * i);
                store_x4<WT>(smem, n, k, KTILES, v[u][i][0] * rs, v[u][i][1] * rs, v[u][i][2] * rs, v[u][i][3] * rs);
            }
            float* const hid_out = (a.dyn != nullptr) ? ((SamplerDynPtr)a.dyn)->hidden_out : nullptr;
            const int4 rstate = (hid_out != nullptr && rt0 == 0) ? *(const int4*)(a.rows + r) : make_int4(1, 0, 0, 0);      // {fin, end, ..}
            if (hid_out != nullptr && rt0 == 0 && rstate.x == 0) {         // heads only: hidden = weight * (x * rs) (llama.py:87) -> hiddens[utterance][its own step]
                float* hrow = hid_out + (size_t)a.rows[r].out * ((SamplerDynPtr)a.dyn)->hidden_stride + (size_t)rstate.y * K;
#pragma unroll
                for (int i = 0; i < PER; ++i) {
                    const int k = 4 * (lane + 64 * i);
                    const f32x4 w = *(const f32x4*)(a.lnw + k);
                    *(f32x4*)(hrow + k) = (f32x4){w[0] * (v[u][i][0] * rs), w[1] * (v[u][i][1] * rs), w[2] * (v[u][i][2] * rs), w[3] * (v[u][i][3] * rs)};
                }
            }
        }
        __syncthreads();
    } else if (PRO == PRO_ATTN) {
        // softmax-normalise the flash-decoding partials of attention.hip.  Items (row, 4 dims) are spread over the
        // whole block and all loads of an item batch are issued together (they were a chain of dependent L2 round
        // trips: 21 us per launch at batch 1 with 16 splits, 22 us at batch 32 with 24 serial items per thread).
        constexpr int NH = K / CTTS_HEAD_DIM;
        constexpr int K4 = K / 4;
        const int S = S_;
        const int rows = min(NB, R - row0);
        if (S == 1) {
            CTTS_ISSUE_WEIGHT_LOADS();
            CTTS_EXIT_IF_DONE();
            constexpr int IB = 8;
            for (int it0 = tid; it0 < rows * K4; it0 += WAVES * 64 * IB) {
                float ls[IB];
                f32x4 os[IB];
#pragma unroll
                for (int u = 0; u < IB; ++u) {
                    const int it = it0 + u * WAVES * 64;
                    const int itc = (it < rows * K4) ? it : it0;
                    const int n = itc / K4, k = 4 * (itc % K4);
                    const size_t ph = (size_t)(row0 + n) * NH + (k >> 6);
                    ls[u] = ((const float*)in0)[ph * 2 + 1];
                    os[u] = *(const f32x4*)((const float*)in1 + ph * CTTS_HEAD_DIM + (k & 63));
                }
#pragma unroll
                for (int u = 0; u < IB; ++u) {
                    const int it = it0 + u * WAVES * 64;
                    if (it >= rows * K4) break;
                    const float inv = 1.0f / ls[u];
                    store_x4<WT>(smem, it / K4, 4 * (it % K4), KTILES, os[u][0] * inv, os[u][1] * inv, os[u][2] * inv, os[u][3] * inv);
                }
            }
        } else {
            bool issued = false;
            for (int it = tid; it < rows * K4 || !issued; it += WAVES * 64) {
                if (it >= rows * K4) { CTTS_ISSUE_WEIGHT_LOADS(); issued = true; CTTS_EXIT_IF_DONE(); break; }
                const int n = it / K4, k = 4 * (it % K4);
                const int r = row0 + n, h = k >> 6, d = k & 63;
                const float* ml = (const float*)in0 + ((size_t)(r * NH + h) * S) * 2;
                const float* po = (const float*)in1 + ((size_t)(r * NH + h) * S) * CTTS_HEAD_DIM + d;
                float ms[ATT_SMAX], ls[ATT_SMAX];
                f32x4 os[ATT_SMAX];
#pragma unroll
                for (int s = 0; s < ATT_SMAX; ++s) {
                    const int sc = (s < S) ? s : 0;
                    const float2 t = *(const float2*)(ml + 2 * sc);
                    ms[s] = (s < S) ? t.x : -INFINITY;
                    ls[s] = t.y;
                    os[s] = *(const f32x4*)(po + (size_t)sc * CTTS_HEAD_DIM);
                }
                if (!issued) { CTTS_ISSUE_WEIGHT_LOADS(); issued = true; CTTS_EXIT_IF_DONE(); }     // after this thread's first batch of partial loads
                float mx = -INFINITY;
#pragma unroll
                for (int s = 0; s < ATT_SMAX; ++s) mx = fmaxf(mx, ms[s]);
                float L = 0.f, y0 = 0.f, y1 = 0.f, y2 = 0.f, y3 = 0.f;
#pragma unroll
                for (int s = 0; s < ATT_SMAX; ++s) {
                    const float w = (ms[s] == -INFINITY) ? 0.f : expf(ms[s] - mx);
                    L += ls[s] * w;
                    y0 += os[s][0] * w; y1 += os[s][1] * w; y2 += os[s][2] * w; y3 += os[s][3] * w;
                }
                const float inv = 1.0f / L;
                store_x4<WT>(smem, n, k, KTILES, y0 * inv, y1 * inv, y2 * inv, y3 * inv);
            }
        }
        __syncthreads();
    }

    // PRO_PACKED variants with one weight tile per block and <= 12 B fragments per wave: this wave's B fragments are requested in ONE batch ahead of the weights.  Left to
    // the compiler they were loaded two at a time inside the MFMA loop, each pair behind an s_waitcnt: KPW / 2 dependent L2 round trips in a row
    // (6 at the fp32 down projection, the longest GEMM launch of a batch-32 step).
    constexpr bool PREB = !VALU && (PRO == PRO_PACKED) && ((RT == 1 && NBG * KPW * (SPLIT ? 2 : 1) <= 12) || NBG * KPW * (SPLIT ? 2 : 1) <= 6);       // <= 48 VGPRs of B fragments (the 1024-thread variants have 128)
    frag bpre[NBG][KPW];                                                 // PRO_XH / PREB: this wave's B fragments, requested ahead of the weights
    f32x4 bv[(VALU && PRO == PRO_PACKED) ? VR : 1][KPW];                 // VALU: activation row n, this lane's k-group (the 16 lanes of a group read the same 16 bytes)
    if (PRO == PRO_PACKED) {
        if (VALU) {
            const f32x4* xq = (const f32x4*)in0 + (size_t)chunk * NBG * kt_all * 64 + (lane & 48);
#pragma unroll
            for (int n = 0; n < VR; ++n)
#pragma unroll
                for (int i = 0; i < KPW; ++i) bv[n][i] = xq[(size_t)(kt_off + wave * KPW + i) * 64 + n];
        }
        if (PREB) {
#pragma unroll
            for (int g = 0; g < NBG; ++g)
#pragma unroll
                for (int i = 0; i < KPW; ++i) bpre[g][i] = FragIO<WT>::load(in0, (size_t)chunk * NBG * kt_all + (size_t)(g * kt_all + kt_off + wave * KPW + i), lane);
        }
        CTTS_ISSUE_WEIGHT_LOADS();
        CTTS_EXIT_IF_DONE();
    }
    float* fac_s = (float*)(smem + XS_BYTES + WAVES * NBG * 1024);      // PRO_XH: [NB] rs / scale of each row of the chunk
    if (PRO == PRO_XH) {
        // the RMSNorm factor of every row from the producer's 48 per-tile sums of squares (fixed order: deterministic); one wave,
        // 64 / NB lanes per row; its loads are issued before the weight stream and consumed after this wave's MFMAs are queued.
        // in0 = xh, in1 = ssq, resid_in = scale_in (leading, preloaded scalars: no wait on the argument struct)
        constexpr int LPR = 64 / NB, PPL = 48 / LPR;               // lanes per row, partials per lane (12 or 24)
        f32x4 sq[PPL / 4];
        float sc_in = 1.f;
        const int n_f = lane / LPR, part = lane % LPR;
        const bool frow = (wave == WAVES - 1) && (row0 + n_f < R);
        if (frow) {
            const f32x4* sp = (const f32x4*)((const float*)in1 + (size_t)(row0 + n_f) * 48 + part * PPL);
#pragma unroll
            for (int i = 0; i < PPL / 4; ++i) sq[i] = sp[i];
            sc_in = resid_in[row0 + n_f];
        }
        {
#pragma unroll
            for (int g = 0; g < NBG; ++g)
#pragma unroll
                for (int i = 0; i < KPW; ++i) bpre[g][i] = FragIO<WT>::load(in0, (size_t)chunk * NBG * KTILES + (size_t)(g * KTILES + wave * KPW + i), lane);
        }
        CTTS_ISSUE_WEIGHT_LOADS();
        CTTS_EXIT_IF_DONE();
        if (wave == WAVES - 1) {
            float ss = 0.f;
            if (frow) {
#pragma unroll
                for (int i = 0; i < PPL / 4; ++i) ss += (sq[i][0] + sq[i][1]) + (sq[i][2] + sq[i][3]);
            }
            ss += dpp_f<DPP_XOR1>(ss);
            if (LPR == 4) ss += dpp_f<DPP_XOR2>(ss);
            if (frow && part == 0) {
                const float rs = 1.0f / sqrtf(ss / (float)K + a.eps);
                fac_s[n_f] = SPLIT ? rs / (sc_in * CTTS_SPLIT_WSCALE) : rs / sc_in;      // sc_in (and the split images' weight scale) are powers of two: exact
                if (EPI == EPI_LOGITS) fac_s[NB + n_f] = rs;        // heads: the hidden rows below need the plain factor
                if (rt0 == 0 && a.scale_out != nullptr)             // scale of the rows the next EPI_RESID_XH writes
                    a.scale_out[row0 + n_f] = __builtin_bit_cast(float, __builtin_bit_cast(unsigned, rs) & 0x7F800000u);
            }
        }
    }

    // 3a. VALU products over this wave's K slice (see the template comment)
    float accv[VALU ? VR : 1];
    if constexpr (VALU) {
#pragma unroll
        for (int n = 0; n < VR; ++n) accv[n] = 0.f;
#pragma unroll
        for (int i = 0; i < KPW; ++i) {
            const int kt = wave * KPW + i;
            const f32x4 w = __builtin_bit_cast(f32x4, wf[0][i]);
#pragma unroll
            for (int n = 0; n < VR; ++n) {
                f32x4 b;
                if (PRO == PRO_PACKED) b = bv[n][i];
                else b = ((const f32x4*)smem)[kt * 64 + n + (lane & 48)];
                accv[n] = fmaf(w[3], b[3], fmaf(w[2], b[2], fmaf(w[1], b[1], fmaf(w[0], b[0], accv[n]))));
            }
        }
#pragma unroll
        for (int n = 0; n < VR; ++n) {          // the four k-groups of a weight row sit 16 lanes apart
            accv[n] += __shfl_xor(accv[n], 16);
            accv[n] += __shfl_xor(accv[n], 32);
        }
    }
    // 3. MFMA over this wave's K slice
    f32x4 acc[RT][NBG];
#pragma unroll
    for (int t = 0; t < RT; ++t)
#pragma unroll
        for (int g = 0; g < NBG; ++g) acc[t][g] = (f32x4){0.f, 0.f, 0.f, 0.f};
    if constexpr (!VALU)
#pragma unroll
    for (int i = 0; i < KPW; ++i) {
        const int kt = wave * KPW + i;
#pragma unroll
        for (int g = 0; g < NBG; ++g) {
            frag b;
            if (PRO == PRO_XH || PREB) b = bpre[g][i];
            else if (PRO == PRO_PACKED) b = FragIO<WT>::load(in0, (size_t)chunk * NBG * kt_all + (size_t)(g * kt_all + kt_off + kt), lane);
            else b = FragIO<WT>::load(smem, (size_t)(g * KTILES + kt), lane);
#pragma unroll
            for (int ti = 0; ti < RT; ++ti) acc[ti][g] = Mma<WT>::run(wf[ti][i], b, acc[ti][g]);
        }
    }

    // per-utterance LoRA: a first look at the low-rank terms of this thread's epilogue elements -- by now the workers in front of this chunk have normally
    // published, and the round trip hides behind the cross-wave reduction below
    constexpr int LPEEK = LORA_QKV ? 2 * RT : (LORA_O ? RITEMS : 1);
    lora_u64 lpeek[LPEEK];
#pragma unroll
    for (int i = 0; i < LPEEK; ++i) lpeek[i] = 0;
    const bool ltake = (LORA_QKV || LORA_O) && lw != 0 && a.lf.diag != 2;
    if constexpr (LORA_QKV) {
        if (ltake && tid < 8 * NB && row0 + (tid >> 3) < R) {
#pragma unroll
            for (int ti = 0; ti < RT; ++ti) {
                const int rt = rt0 + ti, which = rt / (K / 16), within = rt % (K / 16);
                const lora_u64* gp = a.lf.g + ((size_t)(row0 + (tid >> 3)) * 3 + which) * K + (within >> 2) * CTTS_HEAD_DIM + ((within & 3) << 3) + (tid & 7);
                lpeek[2 * ti] = lora_peek(gp); lpeek[2 * ti + 1] = lora_peek(gp + 32);
            }
        }
    }
    if constexpr (LORA_O) {
        if (ltake) {
#pragma unroll
            for (int u = 0; u < RITEMS; ++u) {
                const int t = tid + u * WAVES * 64, r = row0 + (t >> 4);
                if (t < 16 * NB && r < R) lpeek[u] = lora_peek(a.lf.g_o + (size_t)r * 768 + rt0 * 16 + (t & 15));
            }
        }
    }

    float* red = (float*)(smem + XS_BYTES);                 // [WAVES][NBG][64][4]
#pragma unroll
    for (int ti = 0; ti < RT; ++ti) {
    const int rt = rt0 + ti;
    if (ti > 0) __syncthreads();                             // the previous tile's epilogue is done with red
    // 4. deterministic cross-wave reduction through LDS: every wave parks its partial C tile, the epilogue threads add the WAVES
    //    partials of their own element in wave order (one barrier; the former second LDS staging round is gone)
    if constexpr (VALU) {
        if (lane < 16) {
#pragma unroll
            for (int n = 0; n < VR; ++n) red[(wave * VR + n) * 16 + lane] = accv[n];
        }
    } else {
#pragma unroll
        for (int g = 0; g < NBG; ++g) *(f32x4*)(red + ((wave * NBG + g) * 64 + lane) * 4) = acc[ti][g];
    }
    __syncthreads();
    // C element (weight row i of the tile, activation row n): lane = (i / 4) * 16 + (n % 16), register i % 4, group n / 16
    auto c_elem = [&](int i, int n) -> float {
        if constexpr (VALU) {
            if (n >= VR) return 0.f;
            float sum = 0.f;
#pragma unroll
            for (int w = 0; w < WAVES; ++w) sum += red[(w * VR + n) * 16 + i];
            return sum;
        }
        const float* q = red + (((n >> 4) * 64) + ((i >> 2) << 4) + (n & 15)) * 4 + (i & 3);
        float sum = 0.f;
#pragma unroll
        for (int w = 0; w < WAVES; ++w) sum += q[w * NBG * 256];
        if (PRO == PRO_XH) sum *= fac_s[n];                       // RMSNorm factor (and the xh row scale) applied to the C tile
        else if (SPLIT) sum *= ((K == 3072 || SLICED) ? CTTS_SPLIT_ACT_SCALE : 1.0f) / CTTS_SPLIT_WSCALE;       // split images: 64 W (and the SwiGLU output / 16 in front of the down projection)
        return sum;
    };

    // 4b. EPI_RESID_XH_SK: in-launch combine of the K slices (kernels.h).  Write-through (sc1) slab stores + a drained ticket -- no fences; the slabs
    //     are read back with sc1 loads (cdna_hip_programming.md Guideline 16 R1).  The sum runs in slice order, so the result does not depend on who is last.
    float sk_v[RITEMS];
    if constexpr (EPI == EPI_RESID_XH_SK) {
        static_assert(RT == 1, "split-K combine: one weight row tile per block");
        constexpr int SLAB = 256 * NBG;                              // one (row tile, chunk) partial: 16 columns x 16 NBG rows
        const int nz = gridDim.z, slice = blockIdx.z;
        float* const slab0 = a.sk_slab + ((size_t)(rt * gridDim.y + chunk) * nz) * SLAB;
        int* const lastp = (int*)(red + WAVES * NBG * 256);          // one LDS word behind the partials (the 16 x 16 x NBG epilogue scratch)
#pragma unroll
        for (int u = 0; u < RITEMS; ++u) {
            const int t = tid + u * WAVES * 64;
            sk_v[u] = 0.f;
            if (t < SLAB) {
                sk_v[u] = c_elem(t & 15, t >> 4);
                __hip_atomic_store(slab0 + (size_t)slice * SLAB + t, sk_v[u], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) {
            int* const cnt = a.sk_cnt + rt * gridDim.y + chunk;
            const int ticket = __hip_atomic_fetch_add(cnt, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (ticket == nz - 1) __hip_atomic_store(cnt, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // ready for the next launch
            *lastp = (ticket == nz - 1) ? 1 : 0;
        }
        __syncthreads();
        if (*lastp == 0) return;
#pragma unroll
        for (int u = 0; u < RITEMS; ++u) {
            const int t = tid + u * WAVES * 64;
            if (t < SLAB) {
                float pq[4];
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    pq[q] = (q >= nz) ? 0.f : (q == slice) ? sk_v[u] : __hip_atomic_load(slab0 + (size_t)q * SLAB + t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                sk_v[u] = ((pq[0] + pq[1]) + pq[2]) + pq[3];
            }
        }
    }
    // 5. fused epilogue
    if (EPI == EPI_RESID || EPI == EPI_RESID_P || EPI == EPI_LOGITS || EPI == EPI_PART || EPI == EPI_RESID_XH || EPI == EPI_RESID_XH_SK) {
#pragma unroll
        for (int u = 0; u < RITEMS; ++u) {
            const int t = tid + u * WAVES * 64;
            if (t >= 16 * NB) continue;
            const int n = t >> 4, i = t & 15;
            const int r = row0 + n;
            if (r >= R) continue;
            const int col = rt * 16 + i;
            float v = (EPI == EPI_RESID_XH_SK) ? sk_v[u] : c_elem(i, n);
            if ((EPI == EPI_RESID || EPI == EPI_RESID_P || EPI == EPI_RESID_XH || EPI == EPI_RESID_XH_SK) && a.lora_delta != nullptr)
                v += a.lora_delta[(size_t)r * (a.n_row_tiles * 16) + col];          // per-utterance LoRA term of o_proj (lora.hip)
            if constexpr (LORA_O) {
                if (ltake) v += lora_take_peeked(lpeek[u], a.lf.g_o + (size_t)r * 768 + col, lora_tag_of(lora_draw, CTTS_LORA_TAG_LO), a.lf.err);      // the same term from this launch's workers
            }
            if (EPI == EPI_PART) {
                a.part_out[((size_t)r * gridDim.z + blockIdx.z) * (a.n_row_tiles * 16) + col] = v;
            } else if (EPI == EPI_RESID || EPI == EPI_RESID_P) {
                const float xn = resid_pf[u] + v;                                    // residual + proj (llama.py:731,739)
                a.x_out[(size_t)r * (a.n_row_tiles * 16) + col] = xn;
            } else if (EPI == EPI_RESID_XH || EPI == EPI_RESID_XH_SK) {
                // the 16 lanes of a DPP row hold the 16 columns of (row r, tile rt): fp32 residual as before, plus what the next
                // PRO_XH kernel reads -- the row's sum of squares over this tile and the fp16 (power-of-two scaled) packed copy
                const float xn = resid_pf[u] + v;
                a.x_out[(size_t)r * (a.n_row_tiles * 16) + col] = xn;
                float sq = xn * xn;
                sq += dpp_f<DPP_XOR1>(sq); sq += dpp_f<DPP_XOR2>(sq); sq += dpp_f<DPP_HALF_MIRROR>(sq); sq += dpp_f<DPP_MIRROR>(sq);
                if (i == 0) a.ssq[(size_t)r * a.n_row_tiles + rt] = sq;
                constexpr int KT_OUT = 768 / KT;                  // the stream is H = 768 wide: 24 fp16 / 48 fp32 k-tiles
                store_x1<WT>(a.xh, (size_t)chunk * NBG * KT_OUT, n, col, KT_OUT, xn * xh_scale_pf[u], a.sat);
            } else if (col < a.n_valid) {
                a.logits[(size_t)r * a.n_valid + col] = v;
            }
        }
    } else if (tid < 8 * NB) {
        const int n = tid >> 3, p = tid & 7;
        const int r = row0 + n;
        const float va = c_elem(p, n), vb = c_elem(p + 8, n);
        if (EPI == EPI_SWIGLU) {
            // packed rows: [8 gate | 8 up] per tile -> act[rt*8+p] = silu(g) * u
            float y = 0.f;
            if (r < R) y = (va / (1.0f + expf(-va))) * vb;
            const int ktiles_out = (a.n_row_tiles * 8) / KT;
            if (SPLIT) y *= 1.0f / CTTS_SPLIT_ACT_SCALE;
            store_x1<WT>(a.act_out, (size_t)chunk * NBG * ktiles_out, n, rt * 8 + p, ktiles_out, y, a.sat);      // silu(g) * u is unbounded: saturate + report (common.h)
        } else if (r < R) {  // EPI_QKV: packed rows per tile = dims [8t..8t+7 | 8t+32..8t+39] of one head
            // keep hipcc from scheduling the cache-address arithmetic (and with it a wait on the meta load) at kernel entry
            asm volatile("" : "+v"(meta_pf.seq), "+v"(meta_pf.slot));
            constexpr int HT = K / 16;                   // tiles per projection (H == K for q/k/v)
            constexpr int NH = K / CTTS_HEAD_DIM;
            const int which = rt / HT, within = rt % HT;
            const int h = within >> 2, d = ((within & 3) << 3) + p;
            float ya = va, yb = vb;
            if (a.lora_delta != nullptr) {        // per-utterance LoRA term of q/k/v (lora.hip): part of the projection, so before RoPE
                const float* dl = a.lora_delta + ((size_t)r * 3 + which) * K + h * CTTS_HEAD_DIM + d;
                ya += dl[0]; yb += dl[32];
            }
            if constexpr (LORA_QKV) {
                if (ltake) {                      // the same term from this launch's workers
                    const lora_u64* gp = a.lf.g + ((size_t)r * 3 + which) * K + h * CTTS_HEAD_DIM + d;
                    const unsigned tg = lora_tag_of(lora_draw, CTTS_LORA_TAG_LO);
                    float d0, d1;
                    if ((unsigned)(lpeek[2 * ti] >> 32) == tg && (unsigned)(lpeek[2 * ti + 1] >> 32) == tg) {
                        d0 = __builtin_bit_cast(float, (unsigned)lpeek[2 * ti]); d1 = __builtin_bit_cast(float, (unsigned)lpeek[2 * ti + 1]);
                    } else lora_take2(gp, gp + 32, tg, a.lf.err, d0, d1);
                    ya += d0; yb += d1;
                }
            }
            const float va2 = ya, vb2 = yb;
            if (which < 2) {
                // q*cos + rotate_half(q)*sin, products rounded separately like the reference (llama.py:180-181)
                ya = __fadd_rn(__fmul_rn(va2, rope_c[ti]), __fmul_rn(-vb2, rope_s[ti]));
                yb = __fadd_rn(__fmul_rn(vb2, rope_c[ti]), __fmul_rn(va2, rope_s[ti]));
            }
            if (which == 0) {
                float* q = a.q_out + ((size_t)r * NH + h) * CTTS_HEAD_DIM;
                q[d] = ya; q[d + 32] = yb;
            } else {
                typedef typename WTraits<WT>::cache_t CT;
                CT* c = (CT*)(which == 1 ? a.k_cache : a.v_cache) +
                        (((size_t)meta_pf.seq * NH + h) * a.Lmax + meta_pf.slot) * CTTS_HEAD_DIM;
                // K / V are projections of RMS-normalised rows (|k| <= ||w_row|| * sqrt(768)): plain conversion.  (A saturating store here made
                // the fp16 prompt pass non-reproducible run to run on gfx950 -- same values, different schedule; profiles/README.md round 3.)
                c[d] = (CT)ya; c[d + 32] = (CT)yb;
            }
        }
    }
    }   // row tiles of this block
    if (PRO == PRO_XH && EPI == EPI_LOGITS) {
        // heads on the packed-fp16 path: block 0 of every chunk also writes hidden = weight * (x * rs) (llama.py:87, gpt.py:422-423) from the fp32 rows
        float* const hid_out = (a.dyn != nullptr) ? ((SamplerDynPtr)a.dyn)->hidden_out : nullptr;
        if (hid_out != nullptr && rt0 == 0) {
            const int rows = min(NB, R - row0);
            for (int n = wave; n < rows; n += WAVES) {
                const int r = row0 + n;
                const float rs = fac_s[NB + n];
                const f32x4* xr = (const f32x4*)(a.x + (size_t)r * K);
                const int4 rstate = *(const int4*)(a.rows + r);             // {fin, end, ..}: a live row's hidden goes to hiddens[utterance][its own step]
                if (rstate.x != 0) continue;
                float* hrow = hid_out + (size_t)a.rows[r].out * ((SamplerDynPtr)a.dyn)->hidden_stride + (size_t)rstate.y * K;
#pragma unroll
                for (int i = 0; i < PER; ++i) {
                    const int k = 4 * (lane + 64 * i);
                    const f32x4 v = xr[lane + 64 * i], w = *(const f32x4*)(a.lnw + k);
                    *(f32x4*)(hrow + k) = (f32x4){w[0] * (v[0] * rs), w[1] * (v[1] * rs), w[2] * (v[2] * rs), w[3] * (v[3] * rs)};
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
template <typename WT, int NBG, int WAVES, int KPW, int PRO, int EPI, int RT = 1, int VR = 0, bool LORA = false>
static int launch_one(const GemmArgs& a, int chunks, hipStream_t s, bool configure_only) {
    constexpr int KTILES = WAVES * KPW;
    constexpr int XS = (PRO == PRO_PACKED || PRO == PRO_XH) ? 0 : NBG * KTILES * WTraits<WT>::TILE_BYTES;
    constexpr bool LORA_OK = LORA && (KTILES * WTraits<WT>::KT == 768) && (VR == 0) &&
                             ((EPI == EPI_QKV && (PRO == PRO_NORM || PRO == PRO_XH)) || ((EPI == EPI_RESID || EPI == EPI_RESID_XH) && PRO == PRO_PACKED));
    constexpr int LDS0 = XS + WAVES * NBG * 1024 + 16 * 16 * NBG * 4 + 16;
    constexpr int LDS = (LORA_OK && LDS0 < 6272) ? 6272 : LDS0;          // a LoRA worker block keeps the row, the RMSNorm weight and u[16] in LDS (lora_worker.h)
    auto kern = skinny_gemm_kernel<WT, NBG, WAVES, KPW, PRO, EPI, RT, VR, LORA>;
    if (configure_only) {
        CTTS_HIP_CHECK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
        return 0;
    }
    int nz = 1;
    if (EPI == EPI_PART || EPI == EPI_RESID_XH_SK) {
        if (a.ktiles_total % KTILES || a.K != a.ktiles_total * WTraits<WT>::KT) { ctts_set_error("skinny_gemm: split-K tiling mismatch"); return 1; }
        nz = a.ktiles_total / KTILES;
    } else if (a.K != KTILES * WTraits<WT>::KT) {
        ctts_set_error("skinny_gemm: K=%d does not match the compiled tiling %d", a.K, KTILES * WTraits<WT>::KT);
        return 1;
    }
    if (a.n_row_tiles % RT) { ctts_set_error("skinny_gemm: %d row tiles not a multiple of %d", a.n_row_tiles, RT); return 1; }
    const int* done_p = a.st ? &a.st->all_done : nullptr;
    const void* in0 = (PRO == PRO_ATTN) ? (const void*)a.part_ml : (PRO == PRO_PACKED) ? (const void*)a.xpacked : (PRO == PRO_XH) ? (const void*)a.xh : (const void*)a.x;
    const void* in1 = (PRO == PRO_ATTN) ? (const void*)a.part_o : (PRO == PRO_NORM_P) ? (const void*)a.opart : (PRO == PRO_XH) ? (const void*)a.ssq :
                      (EPI == EPI_RESID_XH || EPI == EPI_RESID_XH_SK) ? (const void*)a.scale_in : nullptr;
    if (EPI == EPI_RESID_XH_SK && (nz > 4 || a.sk_slab == nullptr || a.sk_cnt == nullptr)) { ctts_set_error("skinny_gemm: split-K combine needs <= 4 slices and its slabs"); return 1; }
    const float* resid_arg = (PRO == PRO_XH) ? a.scale_in : (const float*)a.x_out;
    if (a.lora_w != 0 && (!LORA_OK || a.lora_w != (EPI == EPI_QKV ? 3 : 1) * 16 * NBG || a.st == nullptr || a.ktiles_total > 255)) {
        ctts_set_error("skinny_gemm: this launch cannot carry LoRA workers (pro %d epi %d workers %d)", PRO, EPI, a.lora_w);
        return 1;
    }
    const int xreal = a.n_row_tiles / RT + a.lora_w;
    // prefetch blocks: a multiple of 8 (the real blocks keep their XCDs), whole DMA sweeps only, and the first-prefetch-block field of misc has 10 bits
    constexpr bool PFCAP = (EPI == EPI_RESID_XH) && (PRO == PRO_PACKED) && (KTILES * WTraits<WT>::KT == 768) && VR == 0 && !LORA;      // == the kernel's
    const bool pf_on = PFCAP && a.pf_blocks >= 8 && a.pf.ptr != nullptr && (a.pf.unit_bytes % (WAVES * 1024)) == 0 && xreal < 1024;
    const int npf = pf_on ? (a.pf_blocks & ~7) : 0;
    const int misc = (a.np & 0x7) | ((a.S & 0xF) << 3) | ((a.ktiles_total & 0xFF) << 7) | ((a.lora_w & 0x7F) << 15) | (int)((unsigned)(npf ? xreal : 0) << 22);
    hipLaunchKernelGGL(kern, dim3(xreal + npf, chunks, nz), dim3(WAVES * 64), LDS, s, done_p, a.W, in0, in1, resid_arg, a.R, misc, a);
    CTTS_HIP_CHECK(hipGetLastError());
    return 0;
}

// Tilings (real ChatTTS shapes: H=768, I=3072):
//   K=768  fp16: 24 k-tiles = 4 waves x 6      fp32: 48 = 8 x 6
//   K=3072 fp16: 96 k-tiles = 16 waves x 6     fp32: 192 = 16 x 12
#ifndef CTTS_RT_NORM
#define CTTS_RT_NORM 2
#endif
#ifndef CTTS_RT_XH
#define CTTS_RT_XH 1       // weight row tiles per block of the PRO_XH kernels
#endif
template <typename WT, int NBG>
static int dispatch(int pro, int epi, const GemmArgs& a, int chunks, hipStream_t s, bool cfg) {
    constexpr bool F16 = sizeof(WT) == 2;
    // K=768 : fp16 24 k-tiles, fp32 48.  17-32 rows (NBG=2): twice the waves so the (replicated) prologue has twice the threads
    constexpr int W768 = F16 ? (NBG == 1 ? 4 : 8) : (NBG == 1 ? 8 : 16), P768 = (NBG == 1) ? 6 : 3;
    // K=3072: fp16 96 k-tiles = 16 x 6, fp32 192 = 16 x 12
    constexpr int W3072 = 16, P3072 = F16 ? 6 : 12;
    constexpr int RT_XH = CTTS_RT_XH;
    if (cfg) {
        int rc = 0;
        constexpr int RT_NORM_C = (NBG == 2) ? CTTS_RT_NORM : 1;
        rc |= launch_one<WT, NBG, W768, P768, PRO_NORM, EPI_QKV, RT_NORM_C>(a, chunks, s, true);
        rc |= launch_one<WT, NBG, W768, P768, PRO_ATTN, EPI_RESID>(a, chunks, s, true);
        rc |= launch_one<WT, NBG, W768, P768, PRO_PACKED, EPI_RESID>(a, chunks, s, true);
        rc |= launch_one<WT, NBG, W768, P768, PRO_NORM, EPI_SWIGLU, RT_NORM_C>(a, chunks, s, true);
        if constexpr (NBG == 2) {
            rc |= launch_one<WT, 2, W768, P768, PRO_PACKED, EPI_QKV, 4>(a, chunks, s, true);
            rc |= launch_one<WT, 2, W768, P768, PRO_PACKED, EPI_SWIGLU, 4>(a, chunks, s, true);
        }
        rc |= launch_one<WT, NBG, W3072, P3072, PRO_PACKED, EPI_RESID>(a, chunks, s, true);
        rc |= launch_one<WT, NBG, W768, P768, PRO_NORM, EPI_LOGITS>(a, chunks, s, true);
        {
            rc |= launch_one<WT, NBG, W768, P768, PRO_XH, EPI_QKV, RT_XH>(a, chunks, s, true);
            rc |= launch_one<WT, NBG, W768, P768, PRO_XH, EPI_SWIGLU, RT_XH>(a, chunks, s, true);
            rc |= launch_one<WT, NBG, W768, P768, PRO_XH, EPI_LOGITS>(a, chunks, s, true);
            rc |= launch_one<WT, NBG, W768, P768, PRO_ATTN, EPI_RESID_XH>(a, chunks, s, true);
            rc |= launch_one<WT, NBG, W768, P768, PRO_PACKED, EPI_RESID_XH>(a, chunks, s, true);
            rc |= launch_one<WT, NBG, W3072, P3072, PRO_PACKED, EPI_RESID_XH>(a, chunks, s, true);
            if constexpr (NBG == 1) rc |= launch_one<WT, 1, W768, P768, PRO_PACKED, EPI_RESID_XH_SK>(a, chunks, s, true);
        }
        // the same four launches with LoRA workers (lora_worker.h)
        rc |= launch_one<WT, NBG, W768, P768, PRO_NORM, EPI_QKV, RT_NORM_C, 0, true>(a, chunks, s, true);
        rc |= launch_one<WT, NBG, W768, P768, PRO_XH, EPI_QKV, RT_XH, 0, true>(a, chunks, s, true);
        rc |= launch_one<WT, NBG, W768, P768, PRO_PACKED, EPI_RESID, 1, 0, true>(a, chunks, s, true);
        rc |= launch_one<WT, NBG, W768, P768, PRO_PACKED, EPI_RESID_XH, 1, 0, true>(a, chunks, s, true);
        if constexpr (!F16 && NBG == 1) {
#define CTTS_VALU_CFG(VRN) rc |= launch_one<float, 1, W768, P768, PRO_NORM_P, EPI_QKV, 1, VRN>(a, chunks, s, true); rc |= launch_one<float, 1, W768, P768, PRO_NORM, EPI_QKV, 1, VRN>(a, chunks, s, true); \
            rc |= launch_one<float, 1, W768, P768, PRO_ATTN, EPI_RESID_P, 1, VRN>(a, chunks, s, true); rc |= launch_one<float, 1, W768, P768, PRO_NORM, EPI_SWIGLU, 1, VRN>(a, chunks, s, true); \
            rc |= launch_one<float, 1, W768, P768, PRO_NORM_P, EPI_LOGITS, 1, VRN>(a, chunks, s, true); rc |= launch_one<float, 1, W768, P768, PRO_PACKED, EPI_PART, 1, VRN>(a, chunks, s, true); \
            rc |= launch_one<float, 1, W768, P768, PRO_PACKED, EPI_RESID_P, 1, VRN>(a, chunks, s, true)
            CTTS_VALU_CFG(1); CTTS_VALU_CFG(2); CTTS_VALU_CFG(4);
#undef CTTS_VALU_CFG
        }
        if constexpr (NBG == 1) {
            rc |= launch_one<WT, 1, W768, P768, PRO_NORM_P, EPI_SWIGLU>(a, chunks, s, true);
            rc |= launch_one<WT, 1, W3072, P3072, PRO_PACKED, EPI_RESID_P>(a, chunks, s, true);
            rc |= launch_one<WT, 1, W768, P768, PRO_PACKED, EPI_RESID_P>(a, chunks, s, true);
            rc |= launch_one<WT, 1, W768, P768, PRO_NORM_P, EPI_QKV>(a, chunks, s, true);
            rc |= launch_one<WT, 1, W768, P768, PRO_NORM_P, EPI_LOGITS>(a, chunks, s, true);
            rc |= launch_one<WT, 1, W768, P768, PRO_ATTN, EPI_RESID_P>(a, chunks, s, true);
            rc |= launch_one<WT, 1, W768, P768, PRO_PACKED, EPI_PART>(a, chunks, s, true);
        }
        return rc;
    }
    if constexpr (!F16 && NBG == 1) {
        // decode batches of <= 4 rows on an fp32 engine (GemmArgs.valu: the engine's valu_rows option): VALU products, VR = 1 / 2 / 4 compiled
        if (a.valu && a.st != nullptr && a.R <= 4 && chunks == 1) {
#define CTTS_VALU_CASE(P, E, VRN) if (pro == P && epi == E) return launch_one<float, 1, W768, P768, P, E, 1, VRN>(a, chunks, s, false)
#define CTTS_VALU_ALL(VRN) do { CTTS_VALU_CASE(PRO_NORM_P, EPI_QKV, VRN); CTTS_VALU_CASE(PRO_NORM, EPI_QKV, VRN); CTTS_VALU_CASE(PRO_ATTN, EPI_RESID_P, VRN); \
            CTTS_VALU_CASE(PRO_NORM, EPI_SWIGLU, VRN); CTTS_VALU_CASE(PRO_NORM_P, EPI_LOGITS, VRN); CTTS_VALU_CASE(PRO_PACKED, EPI_PART, VRN); \
            if (a.K == 768) CTTS_VALU_CASE(PRO_PACKED, EPI_RESID_P, VRN); } while (0)
            if (a.R == 1) CTTS_VALU_ALL(1);
            else if (a.R == 2) CTTS_VALU_ALL(2);
            else CTTS_VALU_ALL(4);
#undef CTTS_VALU_ALL
#undef CTTS_VALU_CASE
        }
    }
    // 17-32 rows: RT_NORM weight row tiles per block for the two kernels whose prologue re-normalises every row in every block
    constexpr int RT_NORM = (NBG == 2) ? CTTS_RT_NORM : 1;
    if (a.lora_w != 0) {          // per-utterance LoRA at decode: the launches that carry the workers
        if (pro == PRO_NORM && epi == EPI_QKV) return launch_one<WT, NBG, W768, P768, PRO_NORM, EPI_QKV, RT_NORM, 0, true>(a, chunks, s, false);
        if (pro == PRO_XH && epi == EPI_QKV) return launch_one<WT, NBG, W768, P768, PRO_XH, EPI_QKV, RT_XH, 0, true>(a, chunks, s, false);
        if (pro == PRO_PACKED && epi == EPI_RESID && a.K == 768) return launch_one<WT, NBG, W768, P768, PRO_PACKED, EPI_RESID, 1, 0, true>(a, chunks, s, false);
        if (pro == PRO_PACKED && epi == EPI_RESID_XH && a.K == 768) return launch_one<WT, NBG, W768, P768, PRO_PACKED, EPI_RESID_XH, 1, 0, true>(a, chunks, s, false);
        ctts_set_error("skinny_gemm: prologue/epilogue %d/%d cannot carry LoRA workers", pro, epi);
        return 1;
    }
    if (pro == PRO_NORM && epi == EPI_QKV) return launch_one<WT, NBG, W768, P768, PRO_NORM, EPI_QKV, RT_NORM>(a, chunks, s, false);
    if (pro == PRO_ATTN && epi == EPI_RESID) return launch_one<WT, NBG, W768, P768, PRO_ATTN, EPI_RESID>(a, chunks, s, false);
    if (pro == PRO_NORM && epi == EPI_SWIGLU) return launch_one<WT, NBG, W768, P768, PRO_NORM, EPI_SWIGLU, RT_NORM>(a, chunks, s, false);
    if constexpr (NBG == 2) {      // prompt pass on pre-normalised rows (norm_pack_kernel): 4 weight row tiles per block
        if (pro == PRO_PACKED && epi == EPI_QKV) return launch_one<WT, 2, W768, P768, PRO_PACKED, EPI_QKV, 4>(a, chunks, s, false);
        if (pro == PRO_PACKED && epi == EPI_SWIGLU) return launch_one<WT, 2, W768, P768, PRO_PACKED, EPI_SWIGLU, 4>(a, chunks, s, false);
    }
    if (pro == PRO_PACKED && epi == EPI_RESID && a.K == 768) return launch_one<WT, NBG, W768, P768, PRO_PACKED, EPI_RESID>(a, chunks, s, false);
    if (pro == PRO_PACKED && epi == EPI_RESID) return launch_one<WT, NBG, W3072, P3072, PRO_PACKED, EPI_RESID>(a, chunks, s, false);
    if (pro == PRO_NORM && epi == EPI_LOGITS) return launch_one<WT, NBG, W768, P768, PRO_NORM, EPI_LOGITS>(a, chunks, s, false);
    {
        if (pro == PRO_XH && epi == EPI_QKV) return launch_one<WT, NBG, W768, P768, PRO_XH, EPI_QKV, RT_XH>(a, chunks, s, false);
        if (pro == PRO_XH && epi == EPI_SWIGLU) return launch_one<WT, NBG, W768, P768, PRO_XH, EPI_SWIGLU, RT_XH>(a, chunks, s, false);
        if (pro == PRO_XH && epi == EPI_LOGITS) return launch_one<WT, NBG, W768, P768, PRO_XH, EPI_LOGITS>(a, chunks, s, false);
        if (pro == PRO_ATTN && epi == EPI_RESID_XH) return launch_one<WT, NBG, W768, P768, PRO_ATTN, EPI_RESID_XH>(a, chunks, s, false);
        if (pro == PRO_PACKED && epi == EPI_RESID_XH && a.K == 768) return launch_one<WT, NBG, W768, P768, PRO_PACKED, EPI_RESID_XH>(a, chunks, s, false);
        if (pro == PRO_PACKED && epi == EPI_RESID_XH) return launch_one<WT, NBG, W3072, P3072, PRO_PACKED, EPI_RESID_XH>(a, chunks, s, false);
        if constexpr (NBG == 1) { if (pro == PRO_PACKED && epi == EPI_RESID_XH_SK) return launch_one<WT, 1, W768, P768, PRO_PACKED, EPI_RESID_XH_SK>(a, chunks, s, false); }
    }
    if constexpr (NBG == 1) {
        if (pro == PRO_NORM_P && epi == EPI_SWIGLU) return launch_one<WT, 1, W768, P768, PRO_NORM_P, EPI_SWIGLU>(a, chunks, s, false);
        if (pro == PRO_NORM_P && epi == EPI_QKV) return launch_one<WT, 1, W768, P768, PRO_NORM_P, EPI_QKV>(a, chunks, s, false);
        if (pro == PRO_NORM_P && epi == EPI_LOGITS) return launch_one<WT, 1, W768, P768, PRO_NORM_P, EPI_LOGITS>(a, chunks, s, false);
        if (pro == PRO_ATTN && epi == EPI_RESID_P) return launch_one<WT, 1, W768, P768, PRO_ATTN, EPI_RESID_P>(a, chunks, s, false);
        if (pro == PRO_PACKED && epi == EPI_PART) return launch_one<WT, 1, W768, P768, PRO_PACKED, EPI_PART>(a, chunks, s, false);
        if (pro == PRO_PACKED && epi == EPI_RESID_P && a.K == 768) return launch_one<WT, 1, W768, P768, PRO_PACKED, EPI_RESID_P>(a, chunks, s, false);
        if (pro == PRO_PACKED && epi == EPI_RESID_P) return launch_one<WT, 1, W3072, P3072, PRO_PACKED, EPI_RESID_P>(a, chunks, s, false);
    }
    ctts_set_error("skinny_gemm: unsupported prologue/epilogue %d/%d", pro, epi);
    return 1;
}

// fp32 engines, decode batches of >= split_decode_rows rows on the packed-residual path (gpt_engine.hip run_layers): head / tail fp16 operands, 3 fp16 MFMAs per product
// (common.h split_t).  K = 768: 24 k-tiles of 2 KiB = 8 waves x 3 (the fp32 tiling's bytes per wave: 6 KiB of weights in flight per lane group);
// K = 3072: 96 = 16 x 6, or four in-launch K slices of 8 x 3.
template <int NBG>
static int dispatch_split(int pro, int epi, const GemmArgs& a, int chunks, hipStream_t s, bool cfg) {
    constexpr int W768 = 8, P768 = 3, W3072 = 16, P3072 = 6;
    if (cfg) {
        int rc = 0;
        rc |= launch_one<split_t, NBG, W768, P768, PRO_XH, EPI_QKV>(a, chunks, s, true);
        rc |= launch_one<split_t, NBG, W768, P768, PRO_XH, EPI_SWIGLU>(a, chunks, s, true);
        rc |= launch_one<split_t, NBG, W768, P768, PRO_XH, EPI_LOGITS>(a, chunks, s, true);
        rc |= launch_one<split_t, NBG, W768, P768, PRO_PACKED, EPI_RESID_XH>(a, chunks, s, true);
        rc |= launch_one<split_t, NBG, W3072, P3072, PRO_PACKED, EPI_RESID_XH>(a, chunks, s, true);
        rc |= launch_one<split_t, NBG, W768, P768, PRO_PACKED, EPI_RESID_XH_SK>(a, chunks, s, true);
        return rc;
    }
    if (a.lora_w != 0 || a.lora_delta != nullptr) { ctts_set_error("skinny_gemm: the split decode kernels carry no per-utterance adapters"); return 1; }
    if (pro == PRO_XH && epi == EPI_QKV) return launch_one<split_t, NBG, W768, P768, PRO_XH, EPI_QKV>(a, chunks, s, false);
    if (pro == PRO_XH && epi == EPI_SWIGLU) return launch_one<split_t, NBG, W768, P768, PRO_XH, EPI_SWIGLU>(a, chunks, s, false);
    if (pro == PRO_XH && epi == EPI_LOGITS) return launch_one<split_t, NBG, W768, P768, PRO_XH, EPI_LOGITS>(a, chunks, s, false);
    if (pro == PRO_PACKED && epi == EPI_RESID_XH && a.K == 768) return launch_one<split_t, NBG, W768, P768, PRO_PACKED, EPI_RESID_XH>(a, chunks, s, false);
    if (pro == PRO_PACKED && epi == EPI_RESID_XH) return launch_one<split_t, NBG, W3072, P3072, PRO_PACKED, EPI_RESID_XH>(a, chunks, s, false);
    if (pro == PRO_PACKED && epi == EPI_RESID_XH_SK) return launch_one<split_t, NBG, W768, P768, PRO_PACKED, EPI_RESID_XH_SK>(a, chunks, s, false);
    ctts_set_error("skinny_gemm (split operands): unsupported prologue/epilogue %d/%d", pro, epi);
    return 1;
}

// Prompt pass: RMSNorm once per row into the fragment-major B-operand image the GEMMs read with PRO_PACKED.  In the decode
// step every block normalises its <= 32 rows itself (cheaper than a launch); over a whole prompt that replication is
// tiles x rows x 3 KB of L2 reads per layer (13 GB per pass at 1536 rows), so the prompt pass normalises once instead.
// Same arithmetic as the PRO_NORM prologue (x * rs, then the store_x4 conversion): bit-identical GEMM inputs.
template <typename WT>
__global__ __launch_bounds__(256) void norm_pack_kernel(const float* x, void* out, int R, int nbg, float eps) {
    constexpr int K = 768, KT = WTraits<WT>::KT, KTILES = K / KT, PER = K / 256;
    const int lane = threadIdx.x & 63, r = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= R) return;
    const int NB = 16 * nbg, chunk = r / NB, n = r % NB;
    const f32x4* xr = (const f32x4*)(x + (size_t)r * K);
    f32x4 v[PER];
#pragma unroll
    for (int i = 0; i < PER; ++i) v[i] = xr[lane + 64 * i];
    float ss = 0.f;
#pragma unroll
    for (int i = 0; i < PER; ++i) ss += v[i][0] * v[i][0] + v[i][1] * v[i][1] + v[i][2] * v[i][2] + v[i][3] * v[i][3];
    ss = wave_sum(ss);
    const float rs = 1.0f / sqrtf(ss / (float)K + eps);
    char* base = (char*)out + (size_t)chunk * nbg * KTILES * 1024;
#pragma unroll
    for (int i = 0; i < PER; ++i) {
        const int k = 4 * (lane + 64 * i);
        store_x4<WT>(base, n, k, KTILES, v[i][0] * rs, v[i][1] * rs, v[i][2] * rs, v[i][3] * rs);
    }
}
int launch_norm_pack(int dtype, const float* x, void* out, int R, int nbg, float eps, hipStream_t s) {
    if (dtype == 1) hipLaunchKernelGGL(norm_pack_kernel<half_t>, dim3((R + 3) / 4), dim3(256), 0, s, x, out, R, nbg, eps);
    else hipLaunchKernelGGL(norm_pack_kernel<float>, dim3((R + 3) / 4), dim3(256), 0, s, x, out, R, nbg, eps);
    CTTS_HIP_CHECK(hipGetLastError());
    return 0;
}

int launch_gemm(int dtype, int nbg, int pro, int epi, const GemmArgs& a, int chunks, hipStream_t s) {
    if (dtype == 2) {         // fp32 engine, head / tail fp16 operands (split_t)
        if (nbg == 1) return dispatch_split<1>(pro, epi, a, chunks, s, false);
        if (nbg == 2) return dispatch_split<2>(pro, epi, a, chunks, s, false);
    } else if (dtype == 1) {
        if (nbg == 1) return dispatch<half_t, 1>(pro, epi, a, chunks, s, false);
        if (nbg == 2) return dispatch<half_t, 2>(pro, epi, a, chunks, s, false);
    } else {
        if (nbg == 1) return dispatch<float, 1>(pro, epi, a, chunks, s, false);
        if (nbg == 2) return dispatch<float, 2>(pro, epi, a, chunks, s, false);
    }
    ctts_set_error("skinny_gemm: unsupported dtype/nbg %d/%d", dtype, nbg);
    return 1;
}

int gemm_configure() {
    GemmArgs a = {};
    int rc = 0;
    rc |= dispatch<half_t, 1>(0, 0, a, 1, nullptr, true);
    rc |= dispatch<half_t, 2>(0, 0, a, 1, nullptr, true);
    rc |= dispatch<float, 1>(0, 0, a, 1, nullptr, true);
    rc |= dispatch<float, 2>(0, 0, a, 1, nullptr, true);
    rc |= dispatch_split<1>(0, 0, a, 1, nullptr, true);
    rc |= dispatch_split<2>(0, 0, a, 1, nullptr, true);
    return rc;
}
