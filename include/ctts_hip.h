/*
 * ctts_hip.h -- C ABI of libctts_hip.so: the MI355X (gfx950) backend for the ChatTTS hot path.
 *
 * This is the drop-in boundary.  The reference selects a backend per model
 * through the YAML field `infer_type` (chattts_plus/pipelines/chattts_plus_pipeline.py:113-129);
 * its accelerated backend (chattts_plus/trt_models/) drives an opaque engine through
 *   create_kv_cache / predict / get_cur_kv_caches   (trt_models/llama_trt_model.py:25-35,77-81)
 *   TensorRTPredictor.predict(feed_dict, stream)     (trt_models/predictor.py:141-169)
 * with torch owning every user-visible buffer and the engine seeing raw data_ptr()s and a stream.
 * The functions below are what an `infer_type: "hip"` binding needs for the same path; each entry
 * cites the reference interface it replaces.  Python binding: chatttsplus_amd/_lib.py (ctypes);
 * see INTEGRATION.md for the reference-side stub.
 *
 * Conventions
 *   - every function returns 0 on success, non-zero on failure; ctts_last_error() gives the text
 *     (reference: predictor.py:165-167 raises ValueError("ERROR: inference failed.")).
 *   - no C++ exceptions, no abort(), no torch types cross the ABI: plain pointers and sizes.
 *   - `stream` is a hipStream_t passed as void* (torch.cuda.current_stream().cuda_stream).
 *   - device pointers are borrowed; the caller keeps them alive (predictor.py:91-115,162).
 *   - a handle is bound to the device that was current at create() and is not thread-safe.
 */
#ifndef CTTS_HIP_H
#define CTTS_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CTTS_DTYPE_F32 0 /* parity mode: fp32 weights / KV / MFMA (v_mfma_f32_16x16x4_f32) */
#define CTTS_DTYPE_F16 1 /* performance mode: fp16 weights / KV, fp32 accumulate (reference GPU dtype, pipeline:37-41) */

#define CTTS_MAX_BATCH 128  /* sequences decoded together (reference caps at 4: pipeline:391-397) */
#define CTTS_NUM_VQ 4

const char* ctts_last_error(void);
int ctts_version(void);

/* ------------------------------------------------------------------------------------------ */
/* GPT: Llama-style code-token decoder + sampler (models/gpt.py, models/llama.py, processors.py) */
/* ------------------------------------------------------------------------------------------ */

typedef struct ctts_gpt ctts_gpt;

typedef struct {
    int32_t hidden;        /* 768   gpt_config.hidden_size        (configs/infer/chattts_plus.yaml:73) */
    int32_t inter;         /* 3072  gpt_config.intermediate_size  (:74) */
    int32_t heads;         /* 12    num_attention_heads           (:75)  head_dim must be 64 */
    int32_t layers;        /* 20    num_hidden_layers             (:76) */
    int32_t vocab_code;    /* 626   num_audio_tokens              (:81) */
    int32_t num_vq;        /* 4                                   (:82) */
    int32_t max_batch;     /* <= CTTS_MAX_BATCH sequences decoded together */
    int32_t max_seq;       /* KV capacity per sequence: prompt + max_new_token */
    int32_t dtype;         /* CTTS_DTYPE_* */
} ctts_gpt_cfg;

/* replaces models.GPT.__init__ / trt_models.GPT.__init__ (gpt.py:25-82) */
int ctts_gpt_create(const ctts_gpt_cfg* cfg, ctts_gpt** out);
void ctts_gpt_destroy(ctts_gpt* h);

/* Named engine options (the YAML's `kwargs.options` of the hip GPT; no counterpart in the reference -- the TensorRT path fixes such choices when
 * the engine is built, trt_models/llama_trt_model.py:25-81).  Explicit calls only: the product library reads no behaviour from the environment.
 *   "prefill_split_rows"  fp32 engines: prompt passes of >= this many rows use the 3-term fp16 split GEMMs (default 65: every pass of more than 64 rows; 384 until round 6; 0 = never; before finalize)
 *   "split_decode_rows"   fp32 engines: decode batches of >= this many rows (packed-residual path, >= 9 rows, no per-utterance adapters) run their projections on the fp16 matrix
 *                         pipes with head / tail fp16 operands -- 3 MFMAs per product at fp32-level accuracy, the prompt pass's arithmetic -- instead of exact-f32 MFMA
 *                         (default 9; 0 = never.  Set to 0 BEFORE finalize and the engine builds no head / tail weight images unless the prompt pass needs them)
 *   "split_nbg2_rows"     ... and from this many rows on (default 17) they take 32-row blocks instead of 16-row chunks
 *   "weight_prefetch_kb"  launch chain from 9 rows on (fp16 engines: 17): the o_proj launch carries extra workgroups that pull the gate|up launch's weights into L2, one per this
 *                         many KiB (default 96; 0 = none).  The other launches were tried as carriers and lose (profiles/r06_ab_weight_prefetch.jsonl)
 *   "valu_rows"           fp32 engines: decode batches of <= this many rows run their projections on the VALU instead of exact-f32 MFMA (default 2; 0..4)
 *   "persistent_rows"     decode batches of <= this many rows (<= 8; default 8 on fp32 engines, 5 on fp16 engines; up to 5 rows one (row, head) per attention workgroup, 6..8 rows
 *                         two, with contexts up to "persistent_pair_keys" = 704 / 576 / 448 keys at 6 / 7 / 8 rows) run the whole decoder stack of a step as ONE persistent
 *                         launch of 256 resident workgroups (persist_layer.hip; up to 1400 keys per key share -- longer contexts go back to the launch chain; the final
 *                         norm + code heads run inside the launch at <= 2 rows; the two-item workgroups of 6..8 rows keep 192 keys of an item in registers and 128 more in LDS).
 *                         0 = off.  The first
 *                         process that loads an engine (either dtype) on a device holds the mode (advisory lock /tmp/ctts_persist_<pci>.lock); others stay on launches.
 *                         A persistent launch needs all 256 workgroups resident: run ONE decode at a time per device (two engines of one process decoding
 *                         concurrently on different streams would have to share the CUs; every wait is bounded and ctts_gpt_progress reports a give-up)
 *   "persistent_share_keys"  1..5 rows: one key share per (row, head) serves up to this + 128 cached keys (384 in registers, the rest in LDS), longer contexts open more shares (default 384)
 *   "persistent_lora"     1 (default): rows that carry a per-utterance adapter stay on the persistent launch -- its otherwise idle compute waves evaluate A h, the edge lanes add
 *                         B (A h) to their q / k / v / o_proj rows (paced schedule; 0 = such batches take the launch chain's worker workgroups, "lora_fold")
 *   "persistent_delay_lora"  poll delay of that hand-off (-1 = 14 + 2 rows, the default)
 *   "persistent_layers_per_launch"  0 = the whole stack in one launch (default), n = n layers per launch
 *   "persistent_schedule" weight request schedule of the persistent launch (1 / 2 / 3, default 3 = paced requests)    "persistent_pace"  its pacing interval (-1 = by row count, the default)
 *   "persistent_delay", "persistent_delay_act", "persistent_delay_x", "persistent_delay_att", "persistent_nap", "persistent_nap_qkv"  when and how often the edge waves poll
 *   "persistent_poll"     0 / 1: sentinel granules before the full sweeps (default 0)
 *   "persistent_timestamps" diagnostics: the persistent launches record per-workgroup phase marks
 *   "prefill_pp_blocks"   fp32 engines, prompt pass: a split GEMM may run on 256-row blocks with two counter-phased wave groups (prefill_split.hip) when it has at
 *                         least this many such blocks and the round count on 256 CUs favours it (default 1; 0 = 128 x 128 blocks only; -4 / -3 = always, with
 *                         that many n tiles per wave: tests).  All shapes give bit-identical results
 *   "prefill_splitk_rows" fp32 engines, prompt pass: passes of <= this many rows slice the down projection's K = 3072 four ways and add the slices in order (default
 *                         2048; 0 = never): 6 blocks per 128 rows otherwise walk 96 k-tiles each -- an 8 x 56-token pass 3.4 -> 2.4 ms.  Another summation order
 *                         than the unsliced kernel's (hidden rows move by ~1e-6; token ids unchanged on every golden)
 *   "prefill_small_blocks" / "prefill_ring4_blocks"   fp32 engines, short prompt passes: a split GEMM of at most this many 128 x 128 blocks (K slices counted) runs on 64 x 64
 *                         blocks (default 192: four times the blocks for grids that leave most CUs idle) / on a 4-stage LDS ring, one block per CU (default 256); 0 = never.
 *                         Bit-identical to the other block shapes
 *   "attn_wide_blocks"    unsplit decode attention takes 8-wave blocks while rows x heads < this (default 512 on fp32 engines, 4096 on fp16 engines; 0 = 256, the limit until round 6)
 *   "decode_splits"       key splits of the decode attention (0 = policy)       "split_rows"  split-K down projection as launch slices up to this batch size (default 8)
 *   "down_splitk_rows"    packed-residual decode batches of >= this many rows slice the down projection's K inside the launch, last arriver combines (default 9; 0 = never)
 *   "nbg2_rows"           decode batches of >= this many rows use 32-row blocks instead of 16-row chunks (default 81 fp32 / 57 fp16)
 *   "graph_steps"         decode steps captured per hipGraph (default 4)
 *   "graph_steps_persistent"  ... per hipGraph of the persistent paths, whose step is 2 launches (default 16; shorter remainders use "graph_steps")
 *   "lora_fold"           per-utterance adapters at decode: 1 (default) = the rows' low-rank terms come from worker workgroups inside the QKV / o_proj launches
 *                         (lora_worker.h), 0 = two more launches per layer (lora.hip; the prompt pass always uses those)
 *   "persistent_fault"    test hook: one workgroup withholds a hand-off in layer value - 1 (the bounded waits must end the step with an error)
 * Unknown names are an error. */
int ctts_gpt_set_option(ctts_gpt* h, const char* name, int value);
int ctts_gpt_get_option(ctts_gpt* h, const char* name, int* value);      /* the EFFECTIVE value ("persistent_rows" reads 0 where the mode is unavailable) */

/* Diagnostics (tools/persist_probe.py): copies a named internal buffer to HOST memory -- "x_dec", "q_buf", "logits", "pl_g" (the persistent
 * layer's granule buffers), "pl_ts" (its per-workgroup phase marks, option "persistent_timestamps"), "pl_state" ({epoch, error}), "xh" / "ssq" (the packed residual copy and its per-tile sums of squares).  Synchronises. */
int ctts_gpt_debug_read(ctts_gpt* h, const char* name, void* out, size_t max_bytes, size_t* bytes, void* stream);

/* replaces GPT.from_pretrained -> load_state_dict (gpt.py:84-85).  `name` is the reference
 * state-dict key; `data` is HOST fp32, row-major, `numel` elements.
 * Accepted keys: gpt.layers.N.{self_attn.{q,k,v,o}_proj,mlp.{gate,up,down}_proj,input_layernorm,
 * post_attention_layernorm}.weight, gpt.norm.weight, emb_code.N.weight,
 * emb_text.weight, head_code.N.parametrizations.weight.original{0,1}, head_text.parametrizations.weight.original{0,1}
 * (the last two + emb_text enable the refine-text pass, infer_text=1). */
int ctts_gpt_set_weight(ctts_gpt* h, const char* name, const float* data, size_t numel);

/* LoRA merge rule of peft merge_and_unload (pipeline:420-432): W += scale * B @ A for one target
 * ("q_proj","k_proj","v_proj","o_proj") of one layer; A [r, in], B [out, r], host fp32.  Call before finalize. */
int ctts_gpt_merge_lora(ctts_gpt* h, int layer, const char* target, const float* A, const float* B, int r, float scale);

/* Per-utterance LoRA (no counterpart in the reference, which merges ONE adapter for a whole batch, pipeline:420-432):
 * up to CTTS_MAX_ADAPTERS adapters stay resident beside the packed weights; every sequence of a batch selects one slot or none and
 * the projections evaluate  W x + scale * B (A x)  per row.  Call after ctts_gpt_finalize.
 *   set_adapter      one (layer, target in "q_proj","k_proj","v_proj","o_proj") of adapter `slot`: A [r][hidden], B [hidden][r], host fp32, r <= 16
 *   clear_adapter    zeroes a slot (all layers / targets)
 *   set_row_adapters slot (or -1) per sequence for the following ctts_gpt_begin calls; slots == NULL or B == 0 switches the path off */
#define CTTS_MAX_ADAPTERS 8
int ctts_gpt_set_adapter(ctts_gpt* h, int slot, int layer, const char* target, const float* A, const float* B, int r, float scale);
int ctts_gpt_clear_adapter(ctts_gpt* h, int slot);
int ctts_gpt_set_row_adapters(ctts_gpt* h, const int32_t* slots, int B);

/* folds weight-norm heads (W = g*v/||v||_row, gpt.py:57-77), packs QKV / gate|up, converts to the
 * engine dtype, uploads.  After this the host copies are released. */
int ctts_gpt_finalize(ctts_gpt* h);

/* bytes of KV cache the caller must provide: [layers][2][max_batch][heads][max_seq][64] of dtype.
 * replaces LlamaTRTModel.create_kv_cache (llama_trt_model.py:25-29). */
size_t ctts_gpt_kv_bytes(const ctts_gpt* h);
int ctts_gpt_bind_kv(ctts_gpt* h, void* kv_dev, size_t bytes);

/* RoPE table [max_seq][64] fp32 = (cos[32], sin[32]) of pos*inv_freq, computed by the caller with the
 * reference's fp32 arithmetic (llama.py:100,106-119) so that the values are bit-identical to the CPU path. */
int ctts_gpt_set_rope(ctts_gpt* h, const float* rope_host, int n_pos);

typedef struct {
    float temperature[CTTS_NUM_VQ];   /* per-codebook temperature (gpt.py:346-351) */
    float top_p_threshold;            /* (float)(1 - top_P) as torch compares it (TopPLogitsWarper); <0 disables top-p */
    int32_t top_k;                    /* max(top_K, min_tokens_to_keep) (TopKLogitsWarper); <=0 disables */
    int32_t min_tokens_to_keep;       /* 3 (processors.py:45,47) */
    int32_t use_penalty;              /* repetition_penalty != 1 (processors.py:50) */
    float penalty_table[17];          /* penalty**n, n=0..16, as torch.pow(float, int64) gives (processors.py:28) */
    int32_t past_window;              /* 16 (processors.py:53) */
    int32_t max_input_ids;            /* row-count quirk threshold (processors.py:23-27) */
    int32_t eos_token;                /* num_embeddings-1 = 625 (pipeline:209) */
    int32_t min_new_token;            /* (gpt.py:477-478) */
    int32_t max_new_token;
    int32_t infer_text;               /* 1: refine-text pass (gpt.py infer_text=True: 21178-way head_text, temperature[0] only,
                                         next input = emb_text[id], use_penalty must be 0); noise is then [n_draws][B][vocab_text] */
} ctts_sampler_cfg;

/* Outputs of one generate() call, all device memory provided by the caller:
 *   ids      int32 [B][max_new_token][4]     (gpt.py:368-376 inputs_ids_buf, generated part)
 *   hiddens  fp32  [B][max_new_token][hidden] (gpt.py:422-423, post final norm)
 *   finish   int32 [B], end_idx int32 [B]    (gpt.py:339-342,486-487,530-531)
 * noise: fp32 [n_draws][B*4][vocab_code] Exp(1) draws consumed one per sample step (argmax(p/q) ==
 * torch.multinomial), or NULL for the on-device Philox generator seeded with `seed`.
 * utt_ids (device noise only): HOST array [B] of caller-chosen global utterance ids, or NULL for 0..B-1.  The Philox stream of a
 * sequence is keyed by (seed, its utterance id, codebook, its own step, its own regenerate attempt) -- not by its row in the batch and
 * not by the batch's draw counter -- so a request gives every utterance the same noise whatever slice, batch position or rank it is
 * served in (the reference's independence argument for slices: pipeline:391-397). */
typedef struct {
    int32_t* ids;
    float* hiddens;
    int32_t* finish;
    int32_t* end_idx;
    const float* noise;
    int32_t n_draws;
    uint64_t seed;
    const uint64_t* utt_ids;
    const int32_t* row_limits;      /* HOST array [B] or NULL: per-utterance token limit (<= max_new_token): the sequence counts as finished once it
                                       has produced that many tokens -- the per-row form of the loop bound gpt.py:389 */
} ctts_gen_io;

/* GPT.forward / get_emb (gpt.py:125-149) fused with Tokenizer.apply_spk_emb (tokenizer.py:150-178):
 * input_ids int32 [B][T][4] device, text_mask int32 [B][T] device (1 = text row -> emb_text, 0 = code row -> sum of the
 * 4 code embeddings); rows whose first id == spk_id receive spk fp32 [B][hidden] (already L2-normalised; NULL = none).
 * emb_out fp32 [B][T][hidden] device.  Ids must be < the table sizes (no bounds check on the device). */
int ctts_gpt_embed(ctts_gpt* h, const int32_t* input_ids_dev, const int32_t* text_mask_dev, int B, int T, const float* spk_dev, int spk_id,
                   float* emb_out_dev, void* stream);

/* Start a generate() call for B sequences with a T-token (left padded) prompt.
 * attention_mask int32 [B][T] device (1 = token, 0 = pad; tokenizer.py:96-115).
 * replaces the set-up part of GPT.generate (gpt.py:335-387). */
int ctts_gpt_begin(ctts_gpt* h, int B, int T, const int32_t* attention_mask_dev, const ctts_sampler_cfg* sc,
                   const ctts_gen_io* io, void* stream);

/* Prompt pass: emb fp32 [B][T][hidden] device (GPT.forward output after apply_spk_emb, gpt.py:125-149,
 * tokenizer.py:150-178).  Fills the KV cache and leaves the last position's residual row per sequence.
 * Asynchronous with respect to the host (enqueues on `stream`, never synchronises); prompts of more than 16384 rows (B * T) run in
 * several passes.  fp16 engines use the MFMA flash-attention kernel from 64 rows and the LDS-staged prompt GEMM from 1536 rows; fp32
 * (parity) engines run the same tiling on head / tail fp16 operand images from 384 rows (three fp16 MFMAs per product, fp32-accurate, fp32 KV
 * cache; prefill_split.hip); smaller prompts go through the decode kernels in 32-row chunks.
 * replaces the i == 0 iteration's LlamaModel.forward (gpt.py:410-418; llama.py:905-1019). */
int ctts_gpt_prefill(ctts_gpt* h, const float* emb_dev, void* stream);

/* Sample phase for the current hidden rows: final RMSNorm + 4 folded heads + sampler chain +
 * EOS/finish bookkeeping + next-token embedding (gpt.py:422-494,527-532).  Used once after prefill
 * (step 0) and again after an ensure_non_empty restart (gpt.py:496-525). */
int ctts_gpt_sample(ctts_gpt* h, void* stream);

/* Reset step/finish state and re-gather the prompt's last hidden rows (ensure_non_empty regenerate);
 * the noise draw counter keeps running like torch's generator does in the reference. */
int ctts_gpt_restart(ctts_gpt* h, void* stream);

/* n_steps x { 20 decoder layers on the last sampled token ; sample phase } (gpt.py:389-546, i > 0).
 * use_graph != 0: replays captured hipGraphs of 4 steps each (cached per batch size / mode / KV binding / attention split
 * count; they contain no per-call state -- ctts_gpt_begin writes the output buffers, noise buffer, seed and sampling
 * parameters into a device block the kernels read -- so consecutive generate() calls reuse them); a remainder of < 4 steps
 * is launched kernel by kernel.  Steps after every sequence has finished exit early on the device.  Asynchronous w.r.t. the host. */
int ctts_gpt_decode(ctts_gpt* h, int n_steps, int use_graph, void* stream);

/* Host-visible progress: number of sample steps executed and whether every row has finished.
 * Synchronises the stream. */
int ctts_gpt_progress(ctts_gpt* h, int32_t* steps_done, int32_t* all_finished, void* stream);

/* fp16 engines store SwiGLU outputs and the packed residual copy as fp16: values beyond the fp16 range are SATURATED at +-65504 and
 * counted (the reference's .half() path would produce inf -> NaN silently on such a checkpoint, pipeline:37-41); `count` = saturated or NaN
 * stores since ctts_gpt_begin.  Synchronises the stream.  fp32 engines store fp32 everywhere EXCEPT in the prompt pass over >= 65 rows ("prefill_split_rows"), whose
 * split GEMMs keep silu(g) * u / 16 as fp16 head / tail images (prefill_split.hip): a value beyond +-65504 * 16 there is clamped and counted too
 * (the decode steps of an fp32 engine never count). */
int ctts_gpt_saturations(ctts_gpt* h, int32_t* count, void* stream);

/* Finished-row compaction (no counterpart in the reference, whose finished rows keep computing until the slowest sequence ends,
 * gpt.py:527-546): between two ctts_gpt_decode calls the caller may drop rows of the decode batch.
 *   rows_enqueue  copies {finish, end_idx} of the CURRENT rows (2 int32 each, row order) into pinned host memory, asynchronously
 *   compact       keep_rows = HOST array of n_keep current row indices, ascending; the engine re-packs its per-row state (residual
 *                 rows, positions, repetition-penalty windows, noise keys) so that they become rows 0..n_keep-1 and continues with a
 *                 batch of n_keep.  The KV cache and the output arrays are indexed by utterance and do not move.  Code mode only. */
int ctts_gpt_rows_enqueue(ctts_gpt* h, int32_t* host_pinned_2B, void* stream);
int ctts_gpt_compact(ctts_gpt* h, const int32_t* keep_rows, int n_keep, void* stream);

/* Continuous batching (no counterpart in the reference: its slices of 4 run one after the other, each to its slowest row,
 * pipeline:391-397): between two ctts_gpt_decode calls `n` NEW utterances take over decode rows whose utterance has finished (the
 * caller has seen their finish flag through rows_enqueue).
 *   rows        HOST [n] current row indices, distinct
 *   T, mask, emb   the new utterances' left-padded prompts: DEVICE mask [n,T] int32, emb [n,T,H] fp32 (ctts_gpt_embed); T + max_new_token
 *               must fit max_seq and n*(T-1) one prompt pass
 *   utt_ids, row_limits (may be null), out_index, attempts (may be null)   HOST [n]: noise key, token limit, place in the ids / hiddens /
 *               finish / end_idx arrays handed to ctts_gpt_begin (which must be large enough: they are indexed by out_index, not by
 *               row), regenerate attempt (ensure_non_empty: an utterance whose first token was EOS is admitted again with attempt + 1)
 * The prompt but its last token goes through a prompt pass into the rows' KV lanes; the last token becomes the rows' next decode input, so
 * the next decode step samples the utterance's first token.  Step counter, noise stream, limit and outputs are per row, so an utterance's
 * result does not depend on when or where it is admitted.  Device noise only.  Per-utterance adapters: name the new utterances' slots with
 * ctts_gpt_admit_adapters first.  Asynchronous. */
int ctts_gpt_admit(ctts_gpt* h, int n, const int32_t* rows, int T, const int32_t* mask, const float* emb, const uint64_t* utt_ids,
                   const int32_t* row_limits, const int32_t* out_index, const int32_t* attempts, void* stream);

/* Adapter slots (ctts_gpt_set_adapter; -1 = none) of the utterances the NEXT ctts_gpt_admit call seats in `rows`; rows not named keep theirs.  When no
 * live row carries an adapter any more the engine drops back to the plain launches. */
int ctts_gpt_admit_adapters(ctts_gpt* h, int n, const int32_t* rows, const int32_t* slots, void* stream);

/* Non-blocking variant: enqueues a copy of {steps_done, draws, all_finished, -} into 4 int32 of PINNED host memory; the
 * caller records an event after it and reads the words once the event has completed -- lets the host keep one chunk of
 * decode steps in flight while it inspects the previous one. */
int ctts_gpt_progress_enqueue(ctts_gpt* h, int32_t* host_pinned4, void* stream);

/* Test hooks (parity tests call the stages one by one through the same ABI):
 * logits of the current hidden rows fp32 [B][4][vocab_code] into `logits_dev` without sampling. */
int ctts_gpt_logits(ctts_gpt* h, float* logits_dev, void* stream);
/* force the next input token ids (teacher forcing): int32 [B][4] device; replaces the sampled ids and
 * re-embeds them (gpt.py:403-407). */
int ctts_gpt_force_ids(ctts_gpt* h, const int32_t* ids_dev, void* stream);
/* the Exp(1) noise the generate-mode sampler draws on the device (ctts_gen_io.noise == NULL) for one multinomial row:
 * out[j] = -log(u_j), u_j from Philox4x32-10 keyed by `seed`, counter (j | stream << 24, utterance id, step | attempt << 20); stream = codebook
 * 0..3, or 4 for the refine-text row.  fp32 [n] device.  Pinned by oracle/device_noise.py (tests/test_gpu_sampler.py). */
int ctts_sampler_noise(uint64_t seed, uint64_t utt_id, int stream_id, int step, int attempt, int n, float* out_dev, void* stream);
/* stand-alone sampler on caller-provided logits (fp32 [rows][vocab], rows = B*4; history int32
 * [rows][hist_len]; q fp32 [rows][vocab]) -> idx int32 [rows]; A15-A19 of SURVEY 8(a). */
int ctts_sampler_run(const ctts_sampler_cfg* sc, const float* logits_dev, const int32_t* history_dev, int hist_len,
                     const float* q_dev, int rows, int vocab, int step, int32_t* idx_dev, void* stream);

/* last measured average duration (ms) of one captured decode step, measured with hipEvents on the launch
 * stream around `n` graph replays; used by bench.py for the roofline object. */
int ctts_gpt_time_decode(ctts_gpt* h, int n_steps, float* ms_per_step, void* stream);

/* algorithmic bytes of one decode step at mean context L: s*(W + B*(L+1)*KV_tok). */
double ctts_gpt_step_bytes(const ctts_gpt* h, int B, double mean_ctx);

/* ------------------------------------------------------------------------------------------ */
/* DVAE decoder + Vocos (models/dvae.py decode branch; third-party vocos)                     */
/* ------------------------------------------------------------------------------------------ */

typedef struct ctts_voc ctts_voc;

typedef struct {
    int32_t dvae_idim;      /* 384 */
    int32_t dvae_hidden;    /* 512 */
    int32_t dvae_bn;        /* 128 */
    int32_t dvae_layers;    /* 12 */
    int32_t n_mels;         /* 100 */
    int32_t vocos_dim;      /* 512 */
    int32_t vocos_inter;    /* 1536 */
    int32_t vocos_layers;   /* 8 */
    int32_t n_fft;          /* 1024 */
    int32_t hop;            /* 256 */
    int32_t max_frames;     /* capacity in mel frames (2 per generated token) per utterance */
    int32_t max_batch;      /* utterances synthesised by one ctts_synth_batch call (<= 64) */
    /* optional quantiser of the DVAE_full model (configs/infer/chattts_plus.yaml dvae_encode.vq_config: G=2, R=2, levels 5,5,5,5):
     * vq_groups > 0 makes the handle a "decode codes" model (use_decoder=False, pipeline:292) -- dvae_idim = vq dim / G (512),
     * dvae_hidden 256; vq_groups == 0: the plain decoder (Decoder.pt) */
    int32_t vq_groups;
    int32_t vq_residuals;
    int32_t vq_levels[4];
} ctts_voc_cfg;

/* replaces DVAE.__init__ (dvae.py:203-239) + vocos.Vocos construction (pipeline:93-111) */
int ctts_voc_create(const ctts_voc_cfg* cfg, ctts_voc** out);
void ctts_voc_destroy(ctts_voc* h);
/* `name` = "dvae." + DVAE state-dict key  or  "vocos." + Vocos state-dict key; host fp32.  With vq_groups > 0 also
 * "dvae.vq_layer.quantizer.rvqs.{g}.project_out.{weight,bias}" (GroupedResidualFSQ, vector_quantize_pytorch). */
int ctts_voc_set_weight(ctts_voc* h, const char* name, const float* data, size_t numel);
int ctts_voc_finalize(ctts_voc* h);

/* DVAE.forward(mode="decode") (dvae.py:272-291): hidden fp32 [n][768] device (one utterance) ->
 * mel fp32 [100][2n] device. */
int ctts_dvae_decode(ctts_voc* h, const float* hidden_dev, int n_tokens, float* mel_dev, void* stream);
/* vocos.Vocos.decode (pipeline:303): mel fp32 [100][F] device -> wav fp32 [hop*(F-1)] device */
int ctts_vocos_decode(ctts_voc* h, const float* mel_dev, int frames, float* wav_dev, void* stream);
/* the whole of ChatTTSPlusPipeline._decode_to_wavs (pipeline:286-305) for B utterances in one launch sequence:
 * hidden_ptrs[u] fp32 [n_tokens[u]][768] device -> wav_ptrs[u] fp32 [hop*(2*n_tokens[u]-1)] device; the two pointer
 * arrays and n_tokens are HOST arrays of length B (<= max_batch).  Same arithmetic as dvae_decode + vocos_decode. */
int ctts_synth_batch(ctts_voc* h, const float* const* hidden_ptrs, const int32_t* n_tokens, int B, float* const* wav_ptrs, void* stream);

/* use_decoder=False branch (pipeline:292,435-439: the DVAE_full model decodes the generated CODE IDS instead of the hidden states):
 * DVAE.forward decode with a quantiser (dvae.py:272-291) = GFSQ._embed (dvae.py:85-96: ids -> implicit FSQ codes, summed over the R
 * residual levels with scale (levels-1)^-r -> project_out, groups concatenated) + the same decoder stack.  Needs a handle created with
 * vq_groups > 0.  ids int32 [n][G*R] device (GPT.generate's ids rows, gpt.py:295-297) -> mel fp32 [100][2n]. */
int ctts_dvae_decode_codes(ctts_voc* h, const int32_t* ids_dev, int n_tokens, float* mel_dev, void* stream);
/* ... and _decode_to_wavs(result.ids, use_decoder=False) for B utterances in one launch sequence; ids_ptrs[u] int32 [n_tokens[u]][G*R]. */
int ctts_synth_batch_codes(ctts_voc* h, const int32_t* const* ids_ptrs, const int32_t* n_tokens, int B, float* const* wav_ptrs, void* stream);

/* ---------------------------------------------------------------------------------------------------------------
 * Zero-shot speaker prompt: waveform -> audio-prompt codes.
 * Replaces ChatTTSPlusPipeline.sample_audio_speaker (pipelines/chattts_plus_pipeline.py:279-284, called from :486-499) =
 * DVAE.forward(mode="encode") (models/dvae.py:263-270): MelSpectrogramFeatures (:171-199), / coef, downsample_conv
 * (:224-229), encoder = DVAEDecoder (:130-168), GFSQ (:66-126).  Weight names are the DVAE_full.pt state-dict keys
 * ("downsample_conv.*", "encoder.*", "vq_layer.quantizer.rvqs.{g}.project_in.*", "coef") plus two host-computed tables:
 * "mel.window" [n_fft] (periodic hann) and "mel.fb" [n_fft/2+1][n_mels] (torchaudio melscale_fbanks, htk, norm=None).
 * --------------------------------------------------------------------------------------------------------------- */
typedef struct ctts_enc ctts_enc;
typedef struct {
    int32_t n_mels;        /* 100 */
    int32_t dim;           /* 512: downsample_conv width = encoder idim */
    int32_t enc_hidden;    /* 256 */
    int32_t enc_bn;        /* 128 */
    int32_t enc_layers;    /* 12  */
    int32_t enc_odim;      /* 1024 = vq dim */
    int32_t vq_groups;     /* G = 2 */
    int32_t vq_residuals;  /* R = 2 */
    int32_t n_fft;         /* 1024 */
    int32_t hop;           /* 256 */
    int32_t max_samples;   /* longest reference clip, in 24 kHz samples */
    int32_t pre_bound;     /* GroupedResidualFSQ release detail: 1 = residual starts from bound(project_in(x)) (1.17.8, see oracle) */
} ctts_enc_cfg;

int ctts_enc_create(const ctts_enc_cfg* cfg, ctts_enc** out);
void ctts_enc_destroy(ctts_enc* h);
int ctts_enc_set_weight(ctts_enc* h, const char* name, const float* host_data, size_t numel);
int ctts_enc_finalize(ctts_enc* h);
/* wav: n_samples fp32 on the device (mono, 24 kHz).  ids: int32 [G*R][T] on the device, T = ((1 + n_samples/hop) - 2)/2 + 1,
 * row g*R + r like GFSQ.forward's `ind` (dvae.py:108-113,126).  Optional test hooks (device, may be null):
 * mel_out [n_mels][F] = log-mel / coef, feat_out [T][enc_odim] = encoder output. */
int ctts_dvae_encode(ctts_enc* h, const float* wav, int n_samples, int32_t* ids, float* mel_out, float* feat_out, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* CTTS_HIP_H */
