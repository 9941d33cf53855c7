// Kernel argument blocks and host launchers of libctts_hip's GPT path.
#pragma once
#include "common.h"

enum { PRO_NORM = 0, PRO_ATTN = 1, PRO_PACKED = 2, PRO_NORM_P = 3, PRO_XH = 4 };     // _P: residual stream = x + the down projection's split-K partial sums
// PRO_XH (decode above the split-K batch sizes): the B operand is the residual stream as the PREVIOUS kernel's epilogue (EPI_RESID_XH) left
// it -- engine dtype, fragment-major, scaled by a per-row power of two -- and the RMSNorm factor is applied to the C tile after the MFMAs from the
// producer's per-tile sums of squares: no block re-reads and re-normalises fp32 rows (16 rows x 3 KB per block, replicated in up to
// 768 blocks per launch, was the prologue of every QKV / gate|up launch).
enum { EPI_QKV = 0, EPI_RESID = 1, EPI_SWIGLU = 2, EPI_LOGITS = 3, EPI_RESID_P = 4, EPI_PART = 5, EPI_RESID_XH = 6, EPI_RESID_XH_SK = 7 };   // PART: write a split-K partial, no residual; RESID_XH: see PRO_XH
// EPI_RESID_XH_SK: EPI_RESID_XH for a launch whose grid.z slices K (the 17-32-row down projection: 96 blocks that each pull 393 KB through one CU become
// 384 blocks of 98 KB): every slice block parks its 16 x 16 partial tile in a 1 KB slab with write-through stores and takes a ticket on the tile's
// counter; the LAST arriver adds the slices IN SLICE ORDER (deterministic whatever the arrival order) and runs the residual / packed-copy epilogue.
#define CTTS_NPART 4       // split-K partial sums of the down projection per row (decode batches <= 4)

// per-utterance LoRA folded into the QKV / o_proj launches of a decode step (lora_worker.h)
typedef unsigned long long lora_u64;
struct LoraFold {
    const float* A;             // this layer's adapters [slot][target 0..3][16][768]
    const float* B;             //                       [slot][target][16][768] (rank-major, B transposed)
    const float* scale;         // [slot][4]
    const float* lnw;           // this layer's input_layernorm weight [768] (q/k/v see the normalised rows, llama.py:726-731)
    signed char slots[CTTS_MAX_B];   // adapter slot of every decode ROW, -1 = none.  In the kernel arguments on purpose: a worker that first loads its slot from
                                //   memory issues its A / B loads one round trip late -- behind the whole weight stream of the launch it rides in (measured: +1.4 us
                                //   per launch).  A captured graph is keyed by these bytes (ensure_graph).
    signed char ranks[32];      // [slot][target] rank of this layer's adapters (CTTS_MAX_ADAPTERS * 4): the workers skip the zero padding up to 16
    lora_u64* g;                // granules: q/k/v [rows][3][768], then o_proj [max rows][768] at g_o
    lora_u64* g_o;
    int* err;                   // give-up word (shared with the persistent launch's; ctts_gpt_progress reports it)
    int layer;
    int diag;                   // timing experiments (option "lora_fold" 2 / 3): 2 = the tiles do not take the terms, 3 = the workers publish zeros at once
};

// Weight prefetch across a launch boundary (round 6).  A decode step is a chain of dependent launches, each of which requests its whole weight matrix at entry and
// then waits: HBM idles at every boundary.  The weights do not depend on the activations, so a launch can carry a few extra workgroups ("prefetch blocks", the
// last ones of its grid) that pull the NEXT launch's weight image into the L2 of the XCD whose workgroups will consume it (block b runs on XCD b % 8: observed
// dispatch order, used for speed only -- a wrong guess costs an L2 miss, never a wrong result).  They issue LDS-DMA loads (no destination registers, up to 63 in
// flight per wave) and end; the consumer's non-temporal loads then hit in L2 instead of paying the HBM round trip at the head of its critical path.
struct WPrefetch {
    const void* ptr;            // the next launch's packed weight image (null = nothing to prefetch)
    unsigned unit_bytes;        // contiguous bytes ONE consumer workgroup column reads (its RT row tiles x all k-tiles): unit u is consumed on XCD u % 8
    unsigned n_units;
};

// out[R rows][N] = prologue(x)[R][K] . W[N][K]^T  followed by a fused epilogue.
// W is pre-packed in 16-row x KT-col MFMA-A tiles, [row tile][k tile][lane][16 B] (gpt_engine.cpp).
struct SamplerDyn;
struct GemmArgs {
    const void* W;
    int n_row_tiles;        // N / 16 (grid.x)
    int K;
    int R;                  // valid rows
    const DevState* st;     // may be null (prefill / tests)
    // prologues
    const float* x;         // PRO_NORM: residual stream [R][K] fp32
    const float* lnw;       // PRO_NORM: RMSNorm weight [K] -- only for hidden_out (the weight is folded into W's columns)
    const float* opart;     // PRO_NORM_P / EPI_RESID_P: partial sums added to the residual stream in index order, [R][np][K] fp32
    int np;                 //   how many (<= CTTS_NPART): the 4 split-K partials of the previous down projection, or 0
    int ktiles_total;       // EPI_PART: k-tiles of the whole matrix (the block's slice is blockIdx.z * KTILES); 0 = KTILES
    float* part_out;        // EPI_PART: [R][gridDim.z][N] fp32
    float eps;
    const SamplerDyn* dyn;  // PRO_NORM (heads only): dyn->hidden_out != null -> normalised rows -> hiddens[seq][step][K]
    const float* part_ml;   // PRO_ATTN: [R][NH][S][2]  (running max, running sum)
    const float* part_o;    // PRO_ATTN: [R][NH][S][64] unnormalised
    int S;
    const void* xpacked;    // PRO_PACKED: fragment-major activations [chunk][g][kt][lane][16 B]
    // epilogues
    float* x_out;           // EPI_RESID: residual stream [R][N] (+=)
    float* q_out;           // EPI_QKV : [R][H] fp32 after RoPE
    void* k_cache;          // EPI_QKV : this layer's K  [maxB][NH][Lmax][64]
    void* v_cache;
    int Lmax;
    const RowMeta* meta;
    const float* rope_rows; // [rows][64] = cos[32] | sin[32] of each row's position (copied from the table by fill_meta / the sampler)
    void* act_out;          // EPI_SWIGLU: fragment-major [chunk][g][kt'][lane][16 B], K' = N/2
    float* logits;          // EPI_LOGITS: [R][n_valid]
    int n_valid;            // valid output rows (2504)
    // PRO_XH / EPI_RESID_XH (see the enum comment)
    void* xh;               // fragment-major residual stream [chunk][g][24 / 48 k-tiles][lane][16 B] (engine dtype): written by EPI_RESID_XH, read by PRO_XH (as in0)
    float* ssq;             // [rows][48] sum of squares of the fp32 residual row over each 16-column tile (written by EPI_RESID_XH)
    const float* scale_in;  // [rows] power-of-two scale of the xh rows: EPI_RESID_XH multiplies by it, PRO_XH divides the C tile by it
    float* scale_out;       // [rows] PRO_XH / PRO_NORM (block 0): 2^floor(log2(rs)) of this normalisation = the scale of the NEXT xh rows
    // per-utterance LoRA (lora.hip): low-rank term of every row, added in the epilogue.  EPI_QKV: [rows][3][768]; EPI_RESID*: [rows][768]
    const float* lora_delta;
    LoraFold lf;            // decode steps: the same term from worker workgroups inside the launch, as tagged granules (lora_worker.h); lora_w workers per chunk
    int lora_w;             //   EPI_QKV: 3 * 16 * NBG, EPI_RESID*: 16 * NBG, 0 = off (lora_delta then holds the term, or null)
    int* sat;               // fp16 engines: counter of saturated / NaN fp16 stores (common.h sat_half); null = do not count
    const RowState* rows;   // heads only: hidden row goes to hiddens[rows[r].out][rows[r].end] while the row is live
    float* sk_slab;         // EPI_RESID_XH_SK: partial tiles [row tile][chunk][slice][256] fp32
    int* sk_cnt;            //                  arrival tickets [row tile][chunk], zero between launches (the last arriver resets its counter)
    WPrefetch pf;           // o_proj launches of the packed-residual path: the gate|up launch's weights, pulled into L2 by pf_blocks extra workgroups (a multiple of 8; 0 = none)
    int pf_blocks;
    int valu;               // fp32 decode, <= 4 rows: products on the VALU instead of exact-f32 MFMA (skinny_gemm.hip, VR; ctts_gpt_set_option "valu_rows")
};

struct AttnArgs {
    const float* q;         // [R][NH][64]
    const void* k_cache;    // this layer
    const void* v_cache;
    int Lmax;
    int NH;
    int R;
    int S;                  // key splits (grid.y)
    const RowMeta* meta;
    const DevState* st;
    float* part_ml;         // [R][NH][S][2]
    float* part_o;          // [R][NH][S][64]
    void* packed_out;       // S == 1 only: normalised output written straight into the o_proj kernel's fragment-major B operand
    int nbg;                //   rows per chunk = 16*nbg
    int packed_split;       //   fp32 engine: packed_out is the head / tail fp16 image pair of the split decode kernels (common.h split_t), not fp32 fragments
    int T, row0;            // prompt pass (MFMA flash kernel): prompt length and the flattened index b*T + t of the pass's first row
    int wide_blocks;        // decode, unsplit: 8-wave blocks while rows x heads < this (0 = 256: fewer blocks than CUs; option "attn_wide_blocks")
};

struct SamplerCfgDev {      // mirrors ctts_sampler_cfg
    float temperature[CTTS_NUM_VQ];
    float top_p_threshold;
    int top_k;
    int min_keep;
    int use_penalty;
    float penalty_table[17];
    int past_window;
    int max_input_ids;
    int eos;
    int min_new;
    int max_new;
};

// Everything one generate()/sampler call may change without changing the launch geometry.  It lives in DEVICE memory
// (ctts_gpt_begin rewrites it) and the kernels read it through the constant address space (scalar loads), so a captured
// decode graph never bakes in a caller's buffers or sampling parameters and is replayed across calls.
struct SamplerDyn {
    SamplerCfgDev cfg;
    int n_draws;
    int* ids;               // [B][max_new][4]
    int* finish;
    int* end_idx;
    const float* noise;     // [n_draws][rows][V] or null -> Philox
    unsigned long long seed;
    float* hidden_out;      // hiddens[seq][step][H] or null
    int hidden_stride;      //   = max_new_token * H
    int rows0;              // multinomial rows of the batch the call STARTED with (B0 * 4, text mode B0): row stride of `noise`, which stays
                            //   indexed by utterance when finished rows are compacted away
};
#define CTTS_CONST_AS __attribute__((address_space(4)))
typedef const CTTS_CONST_AS SamplerDyn* SamplerDynPtr;

struct SamplerArgs {
    const SamplerDyn* dyn;  // device memory
    const float* logits;    // [rows][V]
    int V;
    int B;
    DevState* st;           // null in stand-alone mode
    // generate mode
    int text_mode;          // refine-text pass: one V_text-wide row per sequence, emb_code points at emb_text [V_text][H]
    const float* emb_code;  // [4][V][H] fp32
    int H;
    float* x_next;          // [B][H]
    RowMeta* meta;          // decode rows [B]
    const float* rope;      // table [max_seq][64]
    float* rope_rows;       // decode rows' table rows [B][64], refreshed for the next step's positions
    int* hist_ring;         // [B][4][16] the last 16 sampled ids per (sequence, codebook), slot = step % 16, -1 = none yet
    RowState* finend;       // [B] per-row state (common.h): mirror of {finish, end_idx}, noise key, token limit
    // stand-alone mode (ctts_sampler_run)
    const int* history;     // [rows][hist_len]
    int hist_len;
    int step_override;
    int* idx_out;           // [rows]
};

int launch_gemm(int dtype, int nbg, int pro, int epi, const GemmArgs& a, int chunks, hipStream_t s);
int launch_lora_delta_qkv(const float* x, const float* lnw, float eps, const RowMeta* meta, const int* slot_of_seq, const float* A_l, const float* B_l,
                          const float* scale_l, float* delta, int rows, int H, hipStream_t s);
int launch_lora_delta_o(int dtype, const void* attn_packed, int nbg, const RowMeta* meta, const int* slot_of_seq, const float* A_l, const float* B_l,
                        const float* scale_l, float* delta, int rows, int H, hipStream_t s);
int launch_lora_delta_o_split(const void* hi, const void* lo, const RowMeta* meta, const int* slot_of_seq, const float* A_l, const float* B_l,
                              const float* scale_l, float* delta, int rows, int H, hipStream_t s);
int launch_prefill_gemm(int epi, const GemmArgs& a, hipStream_t s);      // prefill_gemm.hip: fp16 prompt-pass GEMM (QKV / RESID / SWIGLU epilogues)
int launch_norm_pack_split(const float* x, void* hi, void* lo, int R, float eps, hipStream_t s);                    // prefill_split.hip (fp32 engine, >= 1536 prompt rows)
int launch_split_pack(const float* src, void* hi, void* lo, int R, hipStream_t s);
int launch_attention_split(const AttnArgs& a, void* out_hi, void* out_lo, hipStream_t s);
// how a split GEMM of the prompt pass is laid on the chip (prefill_split.hip sp_launch)
struct SplitGemmPolicy {
    int pp_min_blocks;          // "prefill_pp_blocks": 256-row counter-phased blocks when there are at least this many and the round count favours them (0 = never, -4 / -3 = always)
    int sk_rows;                // "prefill_splitk_rows": passes of <= this many rows slice the down projection's K four ways (0 = never)
    float* sk_scratch;          //   the slices' partial outputs [4][rows padded to 128][N] (an engine buffer for <= 2048 rows)
    size_t sk_cap_floats;
    int small_blocks;           // "prefill_small_blocks": a launch of at most this many 128 x 128 blocks (K slices counted) runs on 64 x 64 blocks instead (0 = never)
    int ring4_blocks;           // "prefill_ring4_blocks": a 128 x 128 launch of at most this many blocks (one per CU anyway) keeps three k-tile stages in flight in a 4-stage LDS ring (0 = never)
};
int launch_prefill_split_gemm(int epi, const GemmArgs& a, const void* Wsplit, const void* Xhi, const void* Xlo, void* act_hi, void* act_lo,
                              float scale, const SplitGemmPolicy& pol, hipStream_t s);      // Wsplit: [n tile][k tile][head | tail][lane][16 B] (common.h split_t)
int launch_norm_pack(int dtype, const float* x, void* out_packed, int R, int nbg, float eps, hipStream_t s);
int launch_attention(int dtype, const AttnArgs& a, hipStream_t s);
int launch_sampler(const SamplerArgs& a, int blocks, hipStream_t s);
int launch_gather_last_rows(const float* src, float* dst, int B, int T, int r0, int n, int H, hipStream_t s);
int launch_embed_ids(const int* ids, const float* emb_code, float* x, int B, int V, int H, hipStream_t s);
int launch_fill_meta(RowMeta* prefill_meta, RowMeta* decode_meta, DevState* st, const int* mask, int B, int T, const float* rope, float* rope_pre, hipStream_t s);
int launch_embed_prompt(const int* ids, const int* text_mask, const float* emb_text, const float* emb_code, const float* spk, int spk_id,
                        float* out, int rows, int T, int V, int H, hipStream_t s);
int launch_restart_rows(RowState* rows, int B, hipStream_t s);
// ctts_gpt_admit: n new utterances take over decode rows `rows[i]` (KV lanes `seqs[i]`); prompt rows (all but the last token) -> pm / rope_pre
struct AdmitArgs {
    const int* mask;        // [n][T] device
    const float* emb;       // [n][T][H] device
    const int* rows;        // [n] device: decode rows to overwrite
    const int* seqs;        // [n] device: their KV lanes
    const RowState* fresh;  // [n] device: the new rows' state
    int n, T, H;
    RowMeta* pm; float* rope_pre;                 // [n][T-1]
    RowMeta* dm; float* rope_dec; float* x_dec; int* ring; RowState* finend;      // decode-row arrays
    const float* rope; DevState* st; int* finish; int* end_idx;
};
int launch_admit_rows(const AdmitArgs& a, hipStream_t s);
int launch_compact_rows(const int* keep, int n_keep, int H, float* x, float* rope_rows, RowMeta* meta, int* ring, RowState* fin,
                        float* cx, float* crope, RowMeta* cmeta, int* cring, RowState* cfin, DevState* st, hipStream_t s);
int gemm_configure();
