// Host side of the GPT path: C ABI (include/ctts_hip.h), weight packing, launch sequencing, hipGraph.
//
// Layout in HBM (all owned by the handle unless noted):
//   packed weights   per layer  Wqkv [144 tiles][K=768]  Wo [48][768]  Wgu [384][768]  Wd [48][3072]
//                    + heads [157 tiles][768], each tile = 16 rows x KT cols stored [k-tile][lane][16 B]
//   KV cache         [layers][2][max_batch][heads][max_seq][64]   (caller-owned, bound with ctts_gpt_bind_kv)
//   residual stream  x_dec [B][768] fp32 (decode), x_pre [<= 16384 rows][768] (one prompt pass; sized min(16384, max_batch * max_seq))
//   act              fragment-major SwiGLU output for the down projection
//   xh / ssq / scale fp16 decode above the split-K batch sizes: the residual stream as packed fp16 B operand + per-tile sums of squares + per-row
//                    power-of-two scales, handed from the o_proj / down epilogues to the next QKV / gate|up / heads kernels (kernels.h PRO_XH)
//   lora_*           resident per-utterance adapters and the low-rank terms of the rows being processed (lora.hip)
//   dpart            [rows <= 32][4][768] ordered split-K partial sums of the down projection (decode batches <= split_rows)
//   st / dyn         DevState (per-step counters) and SamplerDyn (per-call buffers and sampling parameters): everything a captured
//                    decode graph would otherwise bake in is read from these two device blocks
#include <fcntl.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/file.h>
#include <sys/stat.h>
#include <unistd.h>

#include <map>
#include <mutex>
#include <string>
#include <atomic>
#include <thread>
#include <vector>

#include "../../include/ctts_hip.h"
#include "kernels.h"
#include "persist.h"
#include "roctx_range.h"

static thread_local char g_err[512] = "";
void ctts_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
extern "C" const char* ctts_last_error(void) { return g_err; }
extern "C" int ctts_version(void) { return 1; }

#define PASS_ROWS_MAX 16384   // prompt rows per pass (32 x 512 tokens in one pass); an engine's workspaces are sized for min(this, max_batch * max_seq)
#define PASS_PAD 256          // + PASS_PAD rows so that whole GEMM blocks stay in bounds
#define SMAX 8        // == ATT_SMAX in skinny_gemm.hip
#define CTTS_PERSIST_MAX_ROWS PL_MAXR

struct LayerW {
    void *qkv, *o, *gu, *d;      // RMSNorm weights are folded into qkv / gu columns
    // fp32 engines: head / tail fp16 images of 64 * W, [tile][k tile][head | tail][lane][16 B] in the fp16 tile order (common.h split_t): the prompt pass's split
    // GEMMs (prefill_split.hip) and the decode projections from split_decode_rows rows on (skinny_gemm.hip) read them
    void *qkv_sp = nullptr, *o_sp = nullptr, *gu_sp = nullptr, *d_sp = nullptr;
};

struct ctts_gpt {
    ctts_gpt_cfg cfg;
    int H, I, NH, L, V, NVQ;
    int esz;                                     // element size of weights / KV
    std::map<std::string, std::vector<float>> host;   // staged fp32 weights until finalize
    bool finalized = false;
    // device
    char* wblob = nullptr;
    char* wsplit = nullptr;                      // fp32 engines: the split images of every layer matrix (2 x 2 bytes per weight), or null
    void* whead_sp = nullptr;                    //   ... and of the folded code heads
    bool split_ok = false;                       //   every weight x 64 is inside the fp16 range (otherwise the engine stays on the exact fp32 kernels)
    int split_nbg2_rows = 17;                    // ... and from this many rows on they take 32-row blocks (one workgroup per weight tile streams it once for all 32 rows; the down projection's K is
                                                 // sliced four ways inside the launch as at 16-row chunks).  ms/step 16-row chunks / 32-row blocks: 17 rows 0.723 / 0.700, 24: 0.822 / 0.800, 32: 0.846 / 0.824
                                                 // (profiles/r06_ab_split_shapes.jsonl).  "split_nbg2_rows"
    int prefetch_kb = 96;                        // decode launch chain, packed-residual path (>= 9 rows; fp16 engines: >= 17): the o_proj launch -- 2.4 MB of weights of its own -- carries extra
                                                 // workgroups that pull the gate|up launch's 18.9 MB weight image into L2 (kernels.h WPrefetch), one per this many KiB of it (8..256 of
                                                 // them); 0 = off.  "weight_prefetch_kb".  ms/step without / with (profiles/r06_ab_weight_prefetch.jsonl): fp32 batch 9 0.610 / 0.588, 16 0.635 / 0.605,
                                                 // 32 0.826 / 0.804; fp16 batch 32 0.603 / 0.588.  The other carriers stream large matrices themselves and lose what their consumers gain
                                                 // (down carried by gate|up +4.8 %, the next q|k|v carried by down +2.7 %), and at 6-8 rows (split-K launch slices) the o_proj carrier loses
                                                 // too: off there.  The role is compiled into that one launch only: present in a kernel it costs 0.3 us per launch even when unused
    int split_dec_rows = 9;                      // fp32 engines: decode batches of >= this many rows (packed-residual path, no per-utterance adapters) run their projections on the
                                                 // head / tail images: 3 fp16 MFMAs per product instead of 8 exact-f32 ones (skinny_gemm.hip dispatch_split); 0 = never.  "split_decode_rows"
    void *sp_x_hi = nullptr, *sp_x_lo = nullptr, *sp_act_hi = nullptr, *sp_act_lo = nullptr;   //   ... and of the prompt rows' operands
    int split_rows_min = 65;                     //   prompt passes of at least this many rows use them.  384 until the end of round 6 (prompt pass ms row kernels / split GEMMs then:
                                                 //   192 rows 2.0 / 3.0, 384 rows 3.1 / 3.0, 768 rows 5.3 / 3.6); with 64 x 64 blocks and the sliced down projection of short
                                                 //   passes (prefill_split.hip sp_launch) the split GEMMs win from the first pass that has more than one 64-row block:
                                                 //   66 rows 1.48 / 1.22, 96: 1.52 / 1.23, 192: 2.13 / 1.31, 288: 2.58 / 1.39, 360: 3.24 / 1.60 (diagnostic builds: CTTS_PREFILL_SPLIT, 0 = never)
    std::vector<LayerW> lw;
    void* whead = nullptr;
    void* whead_text = nullptr;                  // refine-text head (21178 x H), packed like whead; optional
    int text_mode = 0;                           // current generate() call: infer_text=True
    float* lnf = nullptr;
    float* emb_code = nullptr;
    float* emb_text = nullptr;                   // [V_text][H] prompt embedding table (optional)
    int vocab_text = 0;
    int vocab_text_head = 0;
    float* rope = nullptr;
    float *rope_pre = nullptr, *rope_dec = nullptr;   // per-row copies of the table rows (prefill rows / decode rows)
    int rope_n = 0;
    char* kv = nullptr;
    size_t kv_bytes = 0;
    float* sk_scratch = nullptr; size_t sk_cap_floats = 0;      // split-K partial outputs of short prompt passes (finalize)
    float *x_dec = nullptr, *x_last = nullptr, *x_pre = nullptr, *q_buf = nullptr, *part_ml = nullptr, *part_o = nullptr, *logits = nullptr;
    void* act = nullptr;
    void* attn_packed = nullptr;
    void* norm_packed = nullptr;                 // prompt pass: RMSNorm'ed rows in the GEMMs' fragment-major operand layout (norm_pack_kernel)
    int split_rows = 4;                          // decode batches up to this size run the down projection as 4 split-K launch slices whose
                                                 // partial sums the next consumers add (0 = off): 48 x 1024-thread blocks -> 192 x 256/512.
                                                 // fp16: 8 (us/step with the threshold at 4 / 16: batch 5 474 / 458, 8 477 / 470, 16 515 / 585 -- above 8 the
                                                 // packed-fp16 residual hand-off wins); fp32: 16 = every single-chunk batch (batch 5 633 -> 572, 8 674 -> 612,
                                                 // 16 766 -> 712: 48 blocks pulling 9.4 MB of fp32 weights + as many bytes of fp32 activations through 48 CUs were
                                                 // the slowest launch of the layer; with two 16-row chunks the partial sums cost the next QKV prologue what the
                                                 // split saves: batch 20 / 24 / 28 / 32 835 / 892 / 914 / 941 either way)
                                                 // Round 4: fp32 8 as well -- with the in-launch split-K combine (down_sk_rows) the packed-residual path wins from 9 rows on:
                                                 // ms/step launch slices / packed + combine: batch 8 0.599 / 0.613, 9 0.632 / 0.623, 10 0.641 / 0.627, 12 0.664 / 0.637,
                                                 // 14 0.687 / 0.648, 16 0.709 / 0.661; launch slices beyond 16 rows lose (17 0.765 -> 0.812, 32 0.860 -> 0.939)
    float* dpart = nullptr;                      // [rows<=32][4][768]
    int cur_splits = 1;                          // key splits of the decode attention for the steps being launched (decode_splits)
    int launched = 0;                            // decode steps enqueued since begin / restart: host-side bound on the context length
    int nbg2_rows = 33;                          // decode batches of at least this many rows use 32-row blocks instead of 16-row chunks.
                                                 // Measured: two 16-row chunks beat one 32-row block at batch 24 / 32 (598 vs 624,
                                                 // 640 vs 660 us/step) -- per-block prologue latency, not L2 traffic, is what these launches pay for;
                                                 // the prompt pass keeps 32-row blocks
                                                 // Round 4 (with the in-launch split-K combine of the 16-row chunks): ms/step 32-row blocks / 16-row chunks, fp32: batch 40 1.176 / 1.012,
                                                 // 48 1.266 / 1.100, 64 1.343 / 1.259, 96 1.717 / 1.754, 128 2.081 / 2.159; fp16: 40 0.676 / 0.652, 64 0.767 / 0.768, 128 1.099 / 1.169
                                                 // -> 32-row blocks from 81 (fp32) / 57 (fp16) rows ("nbg2_rows")
    int force_splits = 0;                        // key splits of the decode attention (0 = decode_splits policy); ctts_gpt_set_option("decode_splits")
    int down_sk_rows = 9;                        // xh-mode decode batches of >= this many rows (one 16-row chunk per block) slice the down projection's K four ways inside the
                                                 // launch (EPI_RESID_XH_SK); 0 = never.  "down_splitk_rows".  us/step without -> with (profiles/r04_ab_down_splitk.jsonl):
                                                 // fp32 batch 17 795 -> 767, 24 862 -> 833, 32 894 -> 868; fp16 batch 9 475 -> 465, 16 507 -> 496, 24 / 32 unchanged
    float* sk_slab = nullptr; int* sk_cnt = nullptr;
    int opt_gen = 0;                             // bumped by ctts_gpt_set_option: part of the decode-graph key
    int graph_gen = 0;                           //   the opt_gen the cached graphs were captured under: graphs of an older generation can never be selected again and are dropped
    int valu_rows = 2;                           // fp32 engines: decode batches of <= this many rows multiply on the VALU (skinny_gemm.hip, VR template argument):
                                                 // an exact-f32 MFMA costs 32 cycles whatever the number of live columns, 48 of them per SIMD and launch.
                                                 // Measured (us/step, MFMA -> VALU, profiles/r04_ab_valu_rows.jsonl): batch 1 481.9 -> 450.8, 2 491.7 -> 477.5,
                                                 // 3 532.6 -> 555.9, 4 536.7 -> 560.3 (the 4-row variant re-reads four LDS operand rows per weight fragment)
    int persist_rows = 0;                        // default 8 (fp32) / 5 (fp16), set at create: decode batches of <= this many rows run the decoder stack as ONE persistent
                                                 // launch (persist_layer.hip).  us/step, launch chain -> persistent (profiles/r04_ab_persist_options.jsonl):
                                                 // batch 1 452 -> 285, batch 2 480 -> 347, batch 4 540 -> 467
    bool persist_ok = false;                     //   the mode's preconditions hold and this process holds the device's lock (ensure_persist)
    char* pimg = nullptr;                        //   per-workgroup register images of the layer weights [L][192][192 KB], built on the device from the packed tiles -- on the FIRST
                                                 //   decode call of <= persistent_rows rows (persist_images): an engine that only ever decodes larger batches (a LoRA-merged
                                                 //   sibling serving batch 32) never pays the second 755 MB weight copy
    char* pimg_head = nullptr;                   //   the folded heads as 14 register-fragment rows per GEMV workgroup (persist.h PL_HEAD_FRAGS): the launch that ends the stack also runs
    int persist_delay_u = -1;                    //   "persistent_delay_lora": poll delay of the u granules, -1 = 14 + 2 rows (ms/step with an adapter on every row, delay 0 / 8 / 16 / 24:
                                                 //   batch 1 0.362 / 0.350 / 0.301 / 0.315, 2: 0.417 / 0.410 / 0.349 / 0.353, 4: 0.486 / 0.513 / 0.441 / 0.430; profiles/r06_ab_lora_persistent.jsonl)
    int prefill_pp_blocks = 1;                   // "prefill_pp_blocks": a split GEMM of the prompt pass may run on prefill_split_gemm_pp_kernel (256-row blocks, counter-phased wave groups) when it has
                                                 //   at least this many such blocks and the round count favours it (prefill_split.hip sp_launch); 0 = never
    int prefill_small_blocks = 192;              // "prefill_small_blocks": prompt-pass split GEMMs of at most this many 128 x 128 blocks (K slices counted) run on 64 x 64 blocks (0 = never)
    int prefill_ring4_blocks = 256;              // "prefill_ring4_blocks": prompt-pass split GEMMs of at most this many 128 x 128 blocks run with a 4-stage LDS ring, one block per CU (0 = never)
    int prefill_sk_rows = 2048;                  // "prefill_splitk_rows": prompt passes of <= this many rows slice the down projection's K four ways (prefill_split.hip sp_launch; 0 = never)
    int attn_wide_blocks = 0;                    // "attn_wide_blocks": decode attention takes 8-wave blocks while rows x heads < this (0 = 256).  Set at create: 512 (fp32) / 4096 (fp16).
                                                 //   Until round 6 the limit was one block per CU (256), tuned on the round-4 attention layout; on the V-one-dim-per-lane layout
                                                 //   8-wave blocks win further up (ms/step 4-wave / 8-wave, fp32: 22 rows 0.749 / 0.724, 28: 0.773 / 0.749, 32: 0.785 / 0.768, 40: 0.946 / 0.939,
                                                 //   48: 1.010 / 1.032; fp16: 32 rows 0.574 / 0.567, 48: 0.683 / 0.668, 64: 0.740 / 0.713; profiles/r06_ab_attn_wide_blocks.jsonl) --
                                                 //   this was round 5's unexplained 20 -> 22-row step (+10 %)
    int persist_share_keys = 384;                //   "persistent_share_keys": keys per key share at 1..5 rows; one share serves up to this + 128 keys (384 in registers, the rest -- up to 256 -- waits in LDS
                                                 //   since round 6 instead of streaming behind the query), more keys open a second share.  Re-swept with the LDS tail: 384 stays the best or ties
                                                 //   from 400 to 1000 keys (one share of 640: context 630 0.279 vs 0.263 ms/step with two shares; profiles/r06_ab_pair_lds_tail.jsonl)
    int persist_lora = 1;                        //   "persistent_lora": rows with per-utterance adapters stay on the persistent launch (round 6; 0 = they take the launch chain, as until round 5)
    int persist_heads = 1;                       //   the final RMSNorm + heads ("persistent_heads"; code mode, paced schedule): one launch fewer per step
    unsigned long long* pl_g = nullptr;          //   granule buffers g_qkv | g_att | g_x1 | g_act
    unsigned* pl_epoch = nullptr;                //   launch counter = granule tag
    int* pl_error = nullptr;                     //   first give-up code (0 = none); reported by ctts_gpt_progress
    unsigned long long* pl_ts = nullptr;         //   diagnostics: per-workgroup phase marks of the last launch ("persistent_timestamps")
    int cur_persist = 0;                         //   the steps being launched use the persistent layer
    int pl_ts_on = 0;
    int persist_fault = 0;                       //   test hook ("persistent_fault"): see PersistArgs.fault
    int persist_pair_keys = 704;                 //   6..8 rows: contexts beyond this many keys (minus 128 per row above 6) go back to the launch chain; "persistent_pair_keys"
    int persist_splits = 0;                      //   cap on the attention's key splits per (row, head) (0 = PL_SMAX); "persistent_splits"
    int persist_max_keys = 0;                    //   contexts beyond this many keys go back to the launch chain (0 = no limit); "persistent_max_keys"
    int persist_lpl = 0;                         //   decoder layers per persistent launch (0 = all of them in one launch)
    int persist_sched = 3;                       //   weight request schedule (PersistArgs.sched): 1 and 2 measure the same (389.3 / 389.5 us at batch 1, 436.0 / 436.4 at 2);
                                                 //   3 = paced requests: batch 1 373.0 -> 337.4, 2 425.7 -> 404.0, 3 490.1 -> 467.6 (profiles/r04_ab_persist_options.jsonl)
    int persist_pace = -1;                       //   -1 = by row count: 3 at 1-2 rows, 2 from 3 rows on (round 5, after the joint row sums: ms/step at pace 2 / 3, batch 1: 0.2377 / 0.2342,
                                                 //   2: 0.2747 / 0.2722, 3: 0.3190 / 0.3221, 4: 0.3489 / 0.3574, 5: 0.4066 / 0.4136).  Round 4 on PersistArgs.pace (us/step at batch 1 before the poll delays: 0 -> 356, 2 -> 342, 3 -> 337.5, 4 -> 339, 6 -> 341, 8 -> 354.6; with them 2 / 3 / 4 / 6: 285.0 / 286.6 / 293.0 / 297.7)
    // ~128-cycle units an edge wave sleeps before its first poll of the (x + attention) / act / layer-output edge: a pass that starts before the producers'
    // stores are visible fails and costs a whole extra pass, and its loads sit in the queues of the very stores it waits for.  us/step at batch 1 with all three at
    // 0 / 6 / 10 / 14 / 18 / 24 / 32: 334.9 / 307.6 / 288.8 / 281.6 / 286.5 / 295.9 / 316.3; batch 2 at 0 / 12: 402.3 / 346.9; batch 4: 521.2 / 467.4.
    // One at a time around 14 each curve is flat from 11 to 17 (profiles/r04_ab_persist_options.jsonl).  The attention edge needs none (its wait is long).
    // Round 5, after the attention workgroups' phase went from 1.9 to 1.25 us (tools/sweep_persist_delays.sh, ms/step at batch 1 | 4): attention edge 0 / 4 / 8:
    // 0.2457 / 0.2456 / 0.2456 | 0.4088 / 0.4044 / 0.4033; act edge 12 / 14 / 16 / 18 / 20: 0.2409 / 0.2402 / 0.2425 / 0.2430 / 0.2436 | 0.4165 / 0.4136 / 0.4125 /
    // 0.4117 / 0.4113 -> 14 + 2 per further row; the other two stayed where they were (10-12 and 13-15).
    int persist_delay_att = 8, persist_delay = 12, persist_delay_act = 14, persist_delay_x = 15, persist_nap = 1, persist_nap_qkv = 1;
    int persist_poll = -1;                       //   PersistArgs.poll; -1 = by row count: the sentinel pass costs one serial poll at 1-2 rows (batch 1 379.2 -> 387.1 us,
                                                 //   batch 2 430.7 -> 437.8) and pays from 3 rows on, where a full sweep re-reads up to 96 granules per lane (batch 4 547.5 -> 537.6)
    int no_prepack = 0, prefill_gemm_rows = 1536, xh_heads = 1;   // diagnostic builds only: see run_layers / run_decode_step
    RowMeta *meta_pre = nullptr, *meta_dec = nullptr, *meta_dec0 = nullptr;
    DevState* st = nullptr;
    int* last_rows = nullptr;
    int* host_pin = nullptr;
    // per-utterance LoRA (lora.hip): resident adapters A [layer][slot][target][16][768] and B^T in the same shape (zero padded to r = 16),
    // per-sequence slot table, low-rank terms of the rows being processed
    float *lora_A = nullptr, *lora_B = nullptr, *lora_scale = nullptr, *ln1 = nullptr;
    float* lora_Af = nullptr;                    //   A of q | k | v with the input RMSNorm weight folded into the columns [layer][slot][3][16][768]: the persistent launch's waves read it (persist_layer.hip LORA)
    std::vector<float> ln1_host;                 //   host copy of ln1 (fetched at the first ctts_gpt_set_adapter)
    int* lora_slot_of_seq = nullptr;
    signed char lora_row_slots[CTTS_MAX_B];      //   the same per decode ROW (rows move when finished rows are compacted away): travels in the kernel arguments of the folded launches
    std::vector<signed char> lora_rank;          //   [layer][slot][target] rank as loaded (0 = empty)
    std::vector<int> lora_slot_host;             //   host copy of lora_slot_of_seq
    std::vector<int> lora_req_host;              //   what ctts_gpt_set_row_adapters asked for ("the following generate() calls"): every begin() starts from it again, whatever
                                                 //   compaction (lora_refresh) and ctts_gpt_admit_adapters did to the live tables of the previous call
    float *lora_dqkv = nullptr, *lora_do = nullptr;
    unsigned long long* lora_g = nullptr;        //   decode steps: the same terms as tagged granules from worker workgroups inside the QKV / o_proj launches (lora_worker.h)
    int lora_fold = 1;                           //   "lora_fold" option: 0 = the two extra launches per layer at decode too
    int lora_rows = 0;                           // 1: the current / next generate() calls carry per-sequence adapters
    int pass_rows = PASS_ROWS_MAX;               // prompt rows per pass of this engine (env CTTS_PASS_ROWS, read once at finalize, lowers it: the
                                                 // one capacity knob of the product library; the multi-pass tests use it)
    void* xh = nullptr;                          // fp16 decode, > split_rows rows: residual stream as packed fp16 B operand (EPI_RESID_XH -> PRO_XH)
    float *ssq = nullptr, *scale_o = nullptr, *scale_d = nullptr;   //   per-tile sums of squares [rows][48]; per-row power-of-two scales of the xh rows
    int xh_mode = 1;                             //   0 (diagnostic builds) switches the path off (every block re-normalises fp32 rows: PRO_NORM)
    int* sat = nullptr;                          // fp16 engines: saturated / NaN fp16 stores since begin (common.h sat_half; ctts_gpt_saturations)
    int* hist_ring = nullptr;                    // sampler: repetition-penalty window ring [max_B][4][16] (sampler.hip)
    RowState* finend = nullptr;                  // sampler: per-row state [max_B] (common.h): mirror of {finish, end_idx}, noise key, token limit
    std::vector<RowState> rows_host;             //   its initial image for the current generate() (uploaded by begin)
    // finished-row compaction (ctts_gpt_compact): gather targets + the kept row indices
    float *cx = nullptr, *crope = nullptr; RowMeta* cmeta = nullptr; int* cring = nullptr; RowState* cfin = nullptr; int* keep_dev = nullptr;
    std::vector<int> keep_host;
    // host mirrors of the decode rows: KV lane, context length (upper bound) and its cap (prompt + limit); see advance_rows / ctts_gpt_admit
    std::vector<int> row_seq, row_ctx, row_cap;
    std::vector<RowState> fresh_host;
    std::vector<int> seq_host;
    int pre_T = 0;                               // tokens per sequence of the prompt rows being passed (begin: T; admit: T - 1)
    int B0 = 0;                                  // sequences the current generate() started with (h->B = rows still in the decode batch)
    bool admitted = false;                       // ctts_gpt_admit handed a row of this generate() to another utterance: x_last / meta_dec0 no longer describe the rows
    // per generate()
    int B = 0, T = 0;
    SamplerCfgDev sc;
    ctts_gen_io io = {};
    hipStream_t cap_stream = nullptr;
    SamplerDyn* dyn = nullptr;                   // device copy of the per-call sampler state (rewritten by begin; read by the heads + sampler nodes)
    // Captured decode graphs, keyed by what shapes the launches (batch, mode, KV binding): caller buffers and sampling
    // parameters are read from `dyn` at run time, so consecutive generate() calls replay the same executable graph.
    struct GraphEntry { hipGraph_t graph; hipGraphExec_t exec; };
    std::map<std::string, GraphEntry> graphs;
    hipGraphExec_t gexec = nullptr;              // entry selected by the last ensure_graph
    int graph_steps = 4;                         // decode steps captured per graph: a replay costs ~8 us of
                                                 // GPU-side gap, amortised over 4 x 102 kernel nodes
    int graph_steps_persist = 16;                // ... and per graph of the persistent paths (2 nodes per step: 32 nodes): the replay gap is 8.1 us whatever the graph
                                                 // holds (profiles/r05_trace_gaps_b1.json), 2.0 us per step at 4 steps, 0.5 us at 16
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
};

static int dev_alloc(void** p, size_t bytes) {
    CTTS_HIP_CHECK(hipMalloc(p, bytes));
    CTTS_HIP_CHECK(hipMemset(*p, 0, bytes));
    return 0;
}

extern "C" int ctts_gpt_create(const ctts_gpt_cfg* c, ctts_gpt** out) {
    if (!c || !out) { ctts_set_error("null argument"); return 1; }
    if (c->hidden != 768 || c->inter != 3072 || c->heads * CTTS_HEAD_DIM != c->hidden) {
        ctts_set_error("this build is specialised to hidden=768, inter=3072, head_dim=64 (got %d/%d/%d heads)", c->hidden, c->inter, c->heads);
        return 1;
    }
    if (c->num_vq != CTTS_NUM_VQ || c->vocab_code > 640 || c->max_batch < 1 || c->max_batch > CTTS_MAX_B || c->max_seq < 2 ||
        (c->dtype != CTTS_DTYPE_F32 && c->dtype != CTTS_DTYPE_F16)) {
        ctts_set_error("unsupported configuration (num_vq=%d vocab=%d max_batch=%d dtype=%d)", c->num_vq, c->vocab_code, c->max_batch, c->dtype);
        return 1;
    }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) { ctts_set_error("no HIP device visible"); return 1; }
    ctts_gpt* h = new ctts_gpt();
    h->cfg = *c;
    h->H = c->hidden; h->I = c->inter; h->NH = c->heads; h->L = c->layers; h->V = c->vocab_code; h->NVQ = c->num_vq;
    h->esz = (c->dtype == CTTS_DTYPE_F16) ? 2 : 4;
    h->split_rows = 8;                                           // both dtypes (the comment at split_rows)
    // both dtypes since round 5 (fp16 engines: half weights + half K / V in the image, fp32 activations).  fp16: up to 5 rows -- ms/step launch chain / persistent launch on the
    // final round-5 kernel (four edge waves, joint row sums; profiles/r05_step_time_vs_batch_fp16.jsonl): batch 1 0.370 / 0.235, 2: 0.393 / 0.276, 3: 0.423 / 0.312, 4: 0.425 / 0.346,
    // 5: 0.431 / 0.389 (round 6 with the 16-byte act granules: 0.229 / 0.268 / - / 0.342 / 0.380; the first version, two edge waves, had lost at 4 rows: 0.427 / 0.442)
    // fp32: 8 since round 6 (6..8 rows: two attention items per workgroup; ms/step launch chain / persistent launch at mean context 310: 6 rows 0.586 / 0.484, 7: 0.605 / 0.522,
    // 8: 0.617 / 0.557; at context 560: 6 rows 0.690 / 0.660, 8: 0.704 / 0.721 -> persist_pair_keys).  fp16 stays at 5: its chain is faster there (6 rows 0.467 / 0.489, 8: 0.472 / 0.550)
    h->persist_rows = (c->dtype == CTTS_DTYPE_F16) ? PL_MAXR_ONE : PL_MAXR;      // (ms/step launch chain / persistent launch at 5 rows, fp32: 0.540 / 0.389; fp16 at 4 rows: 0.425 / 0.345)
    h->attn_wide_blocks = (c->dtype == CTTS_DTYPE_F16) ? 4096 : 512;
    h->nbg2_rows = (c->dtype == CTTS_DTYPE_F16) ? 57 : 81;
    h->down_sk_rows = 9;                                         // = the first batch size of the packed-residual path (split_rows + 1)
    // Diagnostic switches exist only in builds with -DCTTS_DIAG (python -m chatttsplus_amd.build --diag) and are read HERE, once: the
    // product library takes no behaviour from the environment on its launch paths (diag_env() is a constant null there).
    if (const char* sr = diag_env("CTTS_SPLIT_ROWS")) { h->split_rows = atoi(sr); if (h->split_rows > 32) h->split_rows = 32; }
    if (const char* nr = diag_env("CTTS_NBG2_ROWS")) h->nbg2_rows = atoi(nr);
    if (const char* xm = diag_env("CTTS_XH")) h->xh_mode = atoi(xm) ? 1 : 0;
    if (const char* gs = diag_env("CTTS_GRAPH_STEPS")) { h->graph_steps = atoi(gs); if (h->graph_steps < 1) h->graph_steps = 1; }
    if (const char* e = diag_env("CTTS_SPLITS")) { const int v = atoi(e); if (v >= 1 && v <= SMAX) h->force_splits = v; }
    if (diag_env("CTTS_NO_PREPACK")) h->no_prepack = 1;
    if (const char* e = diag_env("CTTS_PREFILL_GEMM")) { const int v = atoi(e); h->prefill_gemm_rows = v > 1 ? v : (v == 1 ? 1536 : 0); }
    if (diag_env("CTTS_NO_XH_HEADS")) h->xh_heads = 0;
    if (const char* e = diag_env("CTTS_PREFILL_SPLIT")) h->split_rows_min = atoi(e);
    if (gemm_configure()) { delete h; return 1; }
    if (hipStreamCreateWithFlags(&h->cap_stream, hipStreamNonBlocking) != hipSuccess ||
        hipEventCreate(&h->ev0) != hipSuccess || hipEventCreate(&h->ev1) != hipSuccess) {
        ctts_set_error("stream/event creation failed"); delete h; return 1;
    }
    *out = h;
    return 0;
}

// Persistent launches need all 256 workgroups resident at once: two processes that both run them on ONE device can starve each other until the
// spin limit (the engine then reports an error).  One advisory file lock per device and process keeps the mode to the first process that asks for
// it on that device; later processes (the 2-rank-on-one-GPU test, a second service on a shared box) quietly stay on the launch path.
static bool persist_device_lock(int dev) {
    static std::mutex mu;
    static std::map<int, int> fds;
    std::lock_guard<std::mutex> lk(mu);
    auto it = fds.find(dev);
    if (it != fds.end()) return it->second >= 0;
    char bus[64] = {0};
    if (hipDeviceGetPCIBusId(bus, sizeof(bus), dev) != hipSuccess) snprintf(bus, sizeof(bus), "dev%d", dev);
    for (char* c = bus; *c; ++c) if (*c == ':' || *c == '.' || *c == '/') *c = '_';
    char path[160];
    snprintf(path, sizeof(path), "/tmp/ctts_persist_%s.lock", bus);
    int fd = open(path, O_CREAT | O_RDWR | O_CLOEXEC | O_NOFOLLOW, 0666);      // (never follows a planted symlink; not inherited by child processes)
    if (fd >= 0) (void)fchmod(fd, 0666);             // (the umask must not make the file another user's obstacle)
    else fd = open(path, O_RDONLY | O_CLOEXEC | O_NOFOLLOW);      // a file left by another user: flock works on a read-only descriptor too
    if (fd >= 0 && flock(fd, LOCK_EX | LOCK_NB) != 0) { close(fd); fd = -1; }
    fds[dev] = fd;                                 // (kept for the life of the process)
    return fd >= 0;
}

// The persistent decode layer's device state: weight images (repacked on the device from the MFMA tile images: no host copy of the weights is
// needed after finalize), granule buffers, launch counter, error word.  `required`: the caller asked for the mode explicitly (an unmet precondition is
// an error); otherwise the mode is simply left off.
static int ensure_persist(ctts_gpt* h, bool required) {
    if (h->persist_ok || !h->finalized) return 0;               // (before finalize: checked by finalize)
    int dev = 0, cus = 0;
    CTTS_HIP_CHECK(hipGetDevice(&dev));
    CTTS_HIP_CHECK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
    const char* why = nullptr;
    if (h->L > 31) why = "at most 31 decoder layers";
    else if (cus < PL_BLOCKS) why = "the device has fewer than 256 compute units";
    else if (!persist_device_lock(dev)) why = "another process already runs persistent launches on this device";
    if (why) {
        h->persist_rows = 0;
        if (required) { ctts_set_error("persistent layers unavailable: %s", why); return 1; }
        return 0;
    }
    if (persist_configure()) return 1;
    if (dev_alloc((void**)&h->pl_g, (size_t)PL_G_TOTAL * 8) || dev_alloc((void**)&h->pl_epoch, 4)) return 1;
    if (!h->pl_error && dev_alloc((void**)&h->pl_error, 4)) return 1;
    const unsigned one = 1;
    CTTS_HIP_CHECK(hipMemcpy(h->pl_epoch, &one, 4, hipMemcpyHostToDevice));
    h->persist_ok = true;
    return 0;
}
// the weight images, on first need (called outside stream capture: allocates, launches on the null stream, synchronises)
static int persist_images(ctts_gpt* h) {
    if (h->pimg != nullptr || !h->persist_ok) return 0;
    const size_t layer_bytes = PL_LAYER_BYTES / 4 * h->esz;
    // built into locals and published only when complete: a failed allocation or repack (an engine beside others on a full device) leaves the engine on the launch
    // chain -- the mode is an optimisation -- instead of failing the decode call that happened to need the images first, or leaving half-built ones behind
    char* img = nullptr; char* img_head = nullptr;
    bool ok = dev_alloc((void**)&img, layer_bytes * h->L) == 0;
    for (int l = 0; ok && l < h->L; ++l)
        ok = launch_persist_repack(h->esz == 2, h->lw[l].qkv, h->lw[l].o, h->lw[l].gu, h->lw[l].d, img + layer_bytes * l, nullptr) == 0;
    ok = ok && dev_alloc((void**)&img_head, (size_t)PL_GEMV_BLOCKS * PL_HEAD_FRAGS * 4 * h->esz) == 0;
    ok = ok && launch_persist_repack_heads(h->esz == 2, h->whead, (h->NVQ * h->V + 15) / 16, img_head, nullptr) == 0;
    ok = ok && hipDeviceSynchronize() == hipSuccess;
    if (!ok) {
        (void)hipGetLastError();
        if (img) (void)hipFree(img);
        if (img_head) (void)hipFree(img_head);
        h->persist_rows = 0;                                     // (ctts_gpt_get_option reports it; "persistent_rows" can be set again)
        return 0;
    }
    h->pimg = img; h->pimg_head = img_head;
    return 0;
}

extern "C" int ctts_gpt_get_option(ctts_gpt* h, const char* name, int* value) {
    if (!h || !name || !value) { ctts_set_error("get_option: null argument"); return 1; }
    const std::string n(name);
    if (n == "persistent_rows") *value = (h->persist_ok || !h->finalized) ? h->persist_rows : 0;      // the EFFECTIVE value (0 when the mode is unavailable)
    else if (n == "persistent_heads") *value = h->persist_heads;
    else if (n == "persistent_lora") *value = h->persist_lora;
    else if (n == "persistent_share_keys") *value = h->persist_share_keys;
    else if (n == "attn_wide_blocks") *value = h->attn_wide_blocks;
    else if (n == "prefill_pp_blocks") *value = h->prefill_pp_blocks;
    else if (n == "prefill_splitk_rows") *value = h->prefill_sk_rows;
    else if (n == "prefill_ring4_blocks") *value = h->prefill_ring4_blocks;
    else if (n == "prefill_small_blocks") *value = h->prefill_small_blocks;
    else if (n == "persistent_delay_lora") *value = h->persist_delay_u;
    else if (n == "valu_rows") *value = h->valu_rows;
    else if (n == "prefill_split_rows") *value = h->split_rows_min;
    else if (n == "split_decode_rows") *value = h->split_dec_rows;
    else if (n == "weight_prefetch_kb") *value = h->prefetch_kb;
    else if (n == "nbg2_rows") *value = h->nbg2_rows;
    else if (n == "split_nbg2_rows") *value = h->split_nbg2_rows;
    else if (n == "split_rows") *value = h->split_rows;
    else if (n == "graph_steps") *value = h->graph_steps;
    else if (n == "graph_steps_persistent") *value = h->graph_steps_persist;
    else if (n == "decode_splits") *value = h->force_splits;
    else if (n == "lora_fold") *value = h->lora_fold;
    else if (n == "down_splitk_rows") *value = h->down_sk_rows;
    else { ctts_set_error("get_option: unknown option '%s'", name); return 1; }
    return 0;
}

// Named engine options: explicit calls of the host side (hip_models.GPT(options=...)), never the environment.  Options that shape the launches bump
// `opt_gen`, which is part of the decode-graph key, so graphs captured under other settings are not replayed.
extern "C" int ctts_gpt_set_option(ctts_gpt* h, const char* name, int value) {
    if (!h || !name) { ctts_set_error("set_option: null argument"); return 1; }
    const std::string n(name);
    if (n == "prefill_split_rows") {             // prompt passes of >= this many rows use the head / tail fp16 split GEMMs (fp32 engines); 0 = never.  Before finalize.
        // before finalize: also decides whether the images are built; afterwards the pass can be switched off (0: a checkpoint that leaves the fp16 range, see
        // ctts_gpt_saturations) or moved while the images exist
        if (h->finalized && value > 0 && !(h->split_ok && h->wsplit)) { ctts_set_error("set_option(prefill_split_rows): this engine holds no head / tail weight images (fp16 engine, a weight beyond the fp16 range, or the option was 0 at finalize)"); return 1; }
        h->split_rows_min = value < 0 ? 0 : value;
    } else if (n == "split_decode_rows") {       // fp32 engines: decode batches of >= this many rows multiply on the fp16 pipes with head / tail operands (0 = never; see split_dec_rows)
        if (!h->finalized) h->split_dec_rows = value < 0 ? 0 : value;      // before finalize: also decides whether the images are built
        else if (value > 0 && !(h->split_ok && h->wsplit)) { ctts_set_error("set_option(split_decode_rows): this engine holds no head / tail weight images (fp16 engine, a weight beyond the fp16 range, or the option was 0 at finalize)"); return 1; }
        else h->split_dec_rows = value < 0 ? 0 : value;
    } else if (n == "split_nbg2_rows") {
        h->split_nbg2_rows = value < 17 ? 17 : value;
    } else if (n == "weight_prefetch_kb") {      // launch chain: KiB of the next launch's weights per prefetch workgroup (0 = no prefetch workgroups; see prefetch_kb)
        h->prefetch_kb = value < 0 ? 0 : (value > 4096 ? 4096 : value);
    } else if (n == "valu_rows") {               // fp32 engines: decode batches of <= this many rows run their projections on the VALU instead of exact-f32 MFMA (0..4)
        h->valu_rows = value < 0 ? 0 : (value > 4 ? 4 : value);
    } else if (n == "persistent_rows") {         // fp32 engines: decode batches of <= this many rows run each layer as ONE persistent launch (0 = off)
        h->persist_rows = value < 0 ? 0 : (value > CTTS_PERSIST_MAX_ROWS ? CTTS_PERSIST_MAX_ROWS : value);
        if (h->persist_rows > 0 && ensure_persist(h, true)) { h->persist_rows = 0; return 1; }
    } else if (n == "persistent_delay_lora") {
        h->persist_delay_u = value < 0 ? -1 : (value > 256 ? 256 : value);
    } else if (n == "prefill_small_blocks") {
        h->prefill_small_blocks = value < 0 ? 0 : value;
    } else if (n == "prefill_ring4_blocks") {
        h->prefill_ring4_blocks = value < 0 ? 0 : value;
    } else if (n == "prefill_splitk_rows") {
        h->prefill_sk_rows = value < 0 ? 0 : value;
    } else if (n == "prefill_pp_blocks") {
        h->prefill_pp_blocks = value < -4 ? 0 : value;      // -3 / -4: always the counter-phased kernel with that many n tiles per wave (tests, A/B)
    } else if (n == "attn_wide_blocks") {
        h->attn_wide_blocks = value < 0 ? 0 : value;
    } else if (n == "persistent_share_keys") {
        h->persist_share_keys = value < 64 ? 384 : value;
    } else if (n == "persistent_lora") {         // 1 (default): rows with per-utterance adapters stay on the persistent launch; 0 = they take the launch chain
        h->persist_lora = value ? 1 : 0;
    } else if (n == "persistent_heads") {        // 1 (default): the persistent launch that ends the stack also runs the final norm + heads; 0 = the separate heads launch
        h->persist_heads = value ? 1 : 0;
    } else if (n == "persistent_layers_per_launch") {      // 0 = the whole stack in one launch (default); 1 = one launch per layer
        h->persist_lpl = value < 0 ? 0 : value;
    } else if (n == "persistent_schedule") {               // 1 / 2: see persist_layer.hip
        h->persist_sched = (value >= 1 && value <= 3) ? value : 1;
    } else if (n == "lora_fold") {                         // per-utterance adapters at decode: 1 = workers inside the QKV / o_proj launches, 0 = two more launches per layer
        h->lora_fold = (value < 0 || value > 3) ? 1 : value;
    } else if (n == "persistent_fault") {                  // test hook: a withheld hand-off; every wait is bounded, ctts_gpt_progress reports the edge
        h->persist_fault = value < 0 ? 0 : value;
    } else if (n == "persistent_pair_keys") {
        h->persist_pair_keys = value < 0 ? 0 : value;
    } else if (n == "persistent_max_keys") {
        h->persist_max_keys = value < 0 ? 0 : value;
    } else if (n == "persistent_splits") {
        h->persist_splits = value < 0 ? 0 : (value > PL_SMAX ? PL_SMAX : value);
    } else if (n == "persistent_pace") {                   // SCHED 3: ~128-cycle units between two paced weight requests of a wave
        h->persist_pace = value < 0 ? -1 : (value > 64 ? 64 : value);          // (< 0: by row count)
    } else if (n == "persistent_delay_att") {
        h->persist_delay_att = value < 0 ? 0 : (value > 256 ? 256 : value);
    } else if (n == "persistent_delay") {
        h->persist_delay = value < 0 ? 0 : (value > 256 ? 256 : value);
    } else if (n == "persistent_delay_act") {
        h->persist_delay_act = value < 0 ? 0 : (value > 256 ? 256 : value);
    } else if (n == "persistent_delay_x") {
        h->persist_delay_x = value < 0 ? 0 : (value > 256 ? 256 : value);
    } else if (n == "persistent_nap_qkv") {
        h->persist_nap_qkv = value < 0 ? 0 : (value > 256 ? 256 : value);
    } else if (n == "persistent_nap") {
        h->persist_nap = value < 0 ? 0 : (value > 256 ? 256 : value);
    } else if (n == "persistent_poll") {                   // bit 0: sentinel granules before the full sweeps
        h->persist_poll = value < 0 ? -1 : (value & 1);
    } else if (n == "persistent_timestamps") {   // diagnostics: every workgroup of a persistent launch records wall_clock64 marks (ctts_gpt_debug_read "pl_ts")
        if (value && !h->pl_ts && dev_alloc((void**)&h->pl_ts, (size_t)PL_BLOCKS * 10 * 8)) return 1;
        h->pl_ts_on = value ? 1 : 0;
    } else if (n == "decode_splits") {           // key splits of the decode attention (0 = the decode_splits policy)
        if (value < 0 || value > SMAX) { ctts_set_error("set_option(decode_splits): 0..%d", SMAX); return 1; }
        h->force_splits = value;
    } else if (n == "split_rows") {              // decode batches up to this size run the down projection as split-K launch slices
        h->split_rows = value < 0 ? 0 : (value > 32 ? 32 : value);
    } else if (n == "nbg2_rows") {               // decode batches of >= this many rows use 32-row blocks (see nbg2_rows)
        h->nbg2_rows = value < 17 ? 17 : value;
    } else if (n == "down_splitk_rows") {        // see down_sk_rows
        h->down_sk_rows = value < 0 ? 0 : value;
    } else if (n == "graph_steps") {
        h->graph_steps = value < 1 ? 1 : (value > 64 ? 64 : value);
    } else if (n == "graph_steps_persistent") {
        h->graph_steps_persist = value < 1 ? 1 : (value > 64 ? 64 : value);
    } else {
        ctts_set_error("set_option: unknown option '%s'", name);
        return 1;
    }
    h->opt_gen++;
    return 0;
}

extern "C" void ctts_gpt_destroy(ctts_gpt* h) {
    if (!h) return;
    for (auto& kv : h->graphs) { (void)hipGraphExecDestroy(kv.second.exec); (void)hipGraphDestroy(kv.second.graph); }
    void* bufs[] = {h->sk_scratch, h->dyn, h->wblob, h->wsplit, h->whead_sp, h->sp_x_hi, h->sp_x_lo, h->sp_act_hi, h->sp_act_lo, h->whead_text, h->lnf, h->emb_code, h->emb_text, h->rope, h->x_dec, h->x_last, h->x_pre, h->q_buf, h->part_ml, h->part_o, h->logits,
                    h->act, h->attn_packed, h->norm_packed, h->dpart, h->rope_pre, h->rope_dec, h->meta_pre, h->meta_dec, h->meta_dec0, h->st, h->last_rows,
                    h->hist_ring, h->sat, h->finend, h->xh, h->ssq, h->scale_o, h->scale_d, h->cx, h->crope, h->cmeta, h->cring, h->cfin, h->keep_dev,
                    h->lora_A, h->lora_B, h->lora_Af, h->lora_scale, h->ln1, h->lora_slot_of_seq, h->lora_dqkv, h->lora_do, h->lora_g, h->pimg, h->pimg_head, h->pl_g, h->pl_epoch, h->pl_error, h->pl_ts, h->sk_slab, h->sk_cnt};
    for (void* b : bufs) if (b) (void)hipFree(b);
    if (h->host_pin) (void)hipHostFree(h->host_pin);
    if (h->cap_stream) (void)hipStreamDestroy(h->cap_stream);
    if (h->ev0) (void)hipEventDestroy(h->ev0);
    if (h->ev1) (void)hipEventDestroy(h->ev1);
    delete h;
}

// The 196 keys of the reference's GPT state dict: 9 per decoder layer, gpt.norm, 4 emb_code, emb_text, 2 + 8 weight-norm
// parametrizations of the heads.
static bool known_weight_key(const ctts_gpt* h, const std::string& n) {
    int l = -1, consumed = 0;
    if (sscanf(n.c_str(), "gpt.layers.%d.%n", &l, &consumed) == 1 && consumed > 0) {
        if (l < 0 || l >= h->L) return false;
        const std::string rest = n.substr(consumed);
        static const char* per_layer[] = {"self_attn.q_proj.weight", "self_attn.k_proj.weight", "self_attn.v_proj.weight", "self_attn.o_proj.weight",
                                          "mlp.gate_proj.weight", "mlp.up_proj.weight", "mlp.down_proj.weight", "input_layernorm.weight",
                                          "post_attention_layernorm.weight"};
        for (const char* k : per_layer) if (rest == k) return true;
        return false;
    }
    if (n == "gpt.norm.weight" || n == "emb_text.weight" || n == "head_text.parametrizations.weight.original0" ||
        n == "head_text.parametrizations.weight.original1") return true;
    int i = -1; consumed = 0;
    if (sscanf(n.c_str(), "emb_code.%d.weigh%n", &i, &consumed) == 1 && consumed > 0) return i >= 0 && i < h->NVQ && n.substr(consumed) == "t";
    if (sscanf(n.c_str(), "head_code.%d.parametrizations.weight.origina%n", &i, &consumed) == 1 && consumed > 0) {
        const std::string rest = n.substr(consumed);
        return i >= 0 && i < h->NVQ && (rest == "l0" || rest == "l1");
    }
    return false;
}

extern "C" int ctts_gpt_set_weight(ctts_gpt* h, const char* name, const float* data, size_t numel) {
    if (!h || !name || !data) { ctts_set_error("null argument"); return 1; }
    if (h->finalized) { ctts_set_error("weights already finalized"); return 1; }
    std::string n(name);
    if (!known_weight_key(h, n)) {          // strict load, like the reference's load_state_dict(strict=True) (gpt.py:84-85)
        ctts_set_error("set_weight: unexpected key '%s' (not a tensor of the ChatTTS GPT state dict)", name);
        return 1;
    }
    h->host[n].assign(data, data + numel);
    return 0;
}

extern "C" int ctts_gpt_merge_lora(ctts_gpt* h, int layer, const char* target, const float* A, const float* B, int r, float scale) {
    if (!h || h->finalized) { ctts_set_error("merge_lora must precede finalize"); return 1; }
    char key[128];
    snprintf(key, sizeof(key), "gpt.layers.%d.self_attn.%s.weight", layer, target);
    auto it = h->host.find(key);
    if (it == h->host.end()) { ctts_set_error("merge_lora: %s not loaded", key); return 1; }
    const int out = h->H, in = h->H;
    std::vector<float>& W = it->second;
    // peft merge_and_unload: W' = W + scale * B @ A   (pipeline:420-432; scale = lora_alpha / r)
    for (int o = 0; o < out; ++o)
        for (int i = 0; i < in; ++i) {
            float acc = 0.f;
            for (int k = 0; k < r; ++k) acc += B[(size_t)o * r + k] * A[(size_t)k * in + i];
            W[(size_t)o * in + i] += scale * acc;
        }
    return 0;
}

// ---- per-utterance LoRA (lora.hip) ---------------------------------------------------------------
static int lora_target_index(const char* t) {
    static const char* names[4] = {"q_proj", "k_proj", "v_proj", "o_proj"};
    for (int i = 0; i < 4; ++i) if (t && !strcmp(t, names[i])) return i;
    return -1;
}
static int lora_ensure_storage(ctts_gpt* h) {
    if (h->lora_A) return 0;
    const size_t per = (size_t)h->L * CTTS_MAX_ADAPTERS * 4 * 16 * h->H;
    if (dev_alloc((void**)&h->lora_Af, per / 4 * 3 * 4)) return 1;
    CTTS_HIP_CHECK(hipMemset(h->lora_Af, 0, per / 4 * 3 * 4));
    if (dev_alloc((void**)&h->lora_A, per * 4) || dev_alloc((void**)&h->lora_B, per * 4) ||
        dev_alloc((void**)&h->lora_scale, (size_t)h->L * CTTS_MAX_ADAPTERS * 4 * 4) || dev_alloc((void**)&h->lora_slot_of_seq, CTTS_MAX_B * 4) ||
        dev_alloc((void**)&h->lora_dqkv, (size_t)h->pass_rows * 3 * h->H * 4) || dev_alloc((void**)&h->lora_do, (size_t)h->pass_rows * h->H * 4))
        return 1;
    CTTS_HIP_CHECK(hipMemset(h->lora_slot_of_seq, 0xFF, CTTS_MAX_B * 4));
    if (dev_alloc((void**)&h->lora_g, (size_t)CTTS_MAX_B * 4 * h->H * 8)) return 1;           // [rows][3][768] q/k/v | [rows][768] o_proj
    CTTS_HIP_CHECK(hipMemset(h->lora_g, 0, (size_t)CTTS_MAX_B * 4 * h->H * 8));                 // tag 0 never matches (tags start at 64)
    if (!h->pl_error) { if (dev_alloc((void**)&h->pl_error, 4)) return 1; CTTS_HIP_CHECK(hipMemset(h->pl_error, 0, 4)); }
    return 0;
}
extern "C" int ctts_gpt_set_adapter(ctts_gpt* h, int slot, int layer, const char* target, const float* A, const float* B, int r, float scale) {
    if (!h || !h->finalized || !A || !B) { ctts_set_error("set_adapter: handle not finalized or null argument"); return 1; }
    const int t = lora_target_index(target);
    if (slot < 0 || slot >= CTTS_MAX_ADAPTERS || layer < 0 || layer >= h->L || t < 0 || r < 1 || r > 16) {
        ctts_set_error("set_adapter: slot %d / layer %d / target %s / r %d out of range", slot, layer, target ? target : "(null)", r);
        return 1;
    }
    if (lora_ensure_storage(h)) return 1;
    const int H = h->H;
    std::vector<float> a16((size_t)16 * H, 0.f), b16((size_t)H * 16, 0.f);
    for (int k = 0; k < r; ++k) memcpy(&a16[(size_t)k * H], A + (size_t)k * H, (size_t)H * 4);
    for (int n = 0; n < H; ++n) for (int k = 0; k < r; ++k) b16[(size_t)k * H + n] = B[(size_t)n * r + k];       // rank-major like A: rank r reads r rows
    const size_t off = (((size_t)layer * CTTS_MAX_ADAPTERS + slot) * 4 + t) * 16 * H;
    if (t < 3) {                                    // the persistent launch's copy: columns times the layer's input RMSNorm weight (its waves see x * rs, not w * (x * rs))
        if (h->ln1_host.empty()) { h->ln1_host.resize((size_t)h->L * H); CTTS_HIP_CHECK(hipMemcpy(h->ln1_host.data(), h->ln1, h->ln1_host.size() * 4, hipMemcpyDeviceToHost)); }
        std::vector<float> af(a16);
        const float* w = &h->ln1_host[(size_t)layer * H];
        for (int k = 0; k < r; ++k) for (int c = 0; c < H; ++c) af[(size_t)k * H + c] *= w[c];
        CTTS_HIP_CHECK(hipMemcpy(h->lora_Af + (((size_t)layer * CTTS_MAX_ADAPTERS + slot) * 3 + t) * 16 * H, af.data(), af.size() * 4, hipMemcpyHostToDevice));
    }
    CTTS_HIP_CHECK(hipMemcpy(h->lora_A + off, a16.data(), a16.size() * 4, hipMemcpyHostToDevice));
    CTTS_HIP_CHECK(hipMemcpy(h->lora_B + off, b16.data(), b16.size() * 4, hipMemcpyHostToDevice));
    CTTS_HIP_CHECK(hipMemcpy(h->lora_scale + ((size_t)layer * CTTS_MAX_ADAPTERS + slot) * 4 + t, &scale, 4, hipMemcpyHostToDevice));
    if (h->lora_rank.empty()) h->lora_rank.assign((size_t)h->L * CTTS_MAX_ADAPTERS * 4, 0);
    h->lora_rank[((size_t)layer * CTTS_MAX_ADAPTERS + slot) * 4 + t] = (signed char)r;
    h->opt_gen++;                                   // the ranks are kernel arguments of the folded launches: captured graphs are stale
    return 0;
}
extern "C" int ctts_gpt_clear_adapter(ctts_gpt* h, int slot) {
    if (!h || slot < 0 || slot >= CTTS_MAX_ADAPTERS) { ctts_set_error("clear_adapter: bad slot"); return 1; }
    if (!h->lora_A) return 0;
    const size_t per = (size_t)4 * 16 * h->H;
    for (int l = 0; l < h->L; ++l) {
        const size_t off = ((size_t)l * CTTS_MAX_ADAPTERS + slot) * per;
        CTTS_HIP_CHECK(hipMemset(h->lora_Af + off / 4 * 3, 0, per / 4 * 3 * 4));
        CTTS_HIP_CHECK(hipMemset(h->lora_A + off, 0, per * 4));
        CTTS_HIP_CHECK(hipMemset(h->lora_B + off, 0, per * 4));
        CTTS_HIP_CHECK(hipMemset(h->lora_scale + ((size_t)l * CTTS_MAX_ADAPTERS + slot) * 4, 0, 16));
        if (!h->lora_rank.empty()) for (int t = 0; t < 4; ++t) h->lora_rank[((size_t)l * CTTS_MAX_ADAPTERS + slot) * 4 + t] = 0;
    }
    h->opt_gen++;
    return 0;
}
extern "C" int ctts_gpt_set_row_adapters(ctts_gpt* h, const int32_t* slots, int B) {
    if (!h || !h->finalized) { ctts_set_error("set_row_adapters: handle not finalized"); return 1; }
    auto clear_rows = [h]() -> int {              // no row carries an adapter: also forget what an earlier request's rows carried (ctts_gpt_admit_adapters builds on these)
        h->lora_rows = 0;
        h->lora_req_host.clear();
        if (!h->lora_slot_host.empty()) h->lora_slot_host.assign(CTTS_MAX_B, -1);
        for (int b = 0; b < CTTS_MAX_B; ++b) h->lora_row_slots[b] = -1;
        if (h->lora_slot_of_seq) CTTS_HIP_CHECK(hipMemset(h->lora_slot_of_seq, 0xFF, CTTS_MAX_B * 4));
        return 0;
    };
    if (!slots || B <= 0) return clear_rows();
    if (B > CTTS_MAX_B) { ctts_set_error("set_row_adapters: B=%d > %d", B, CTTS_MAX_B); return 1; }
    bool any = false;
    for (int b = 0; b < B; ++b) {
        if (slots[b] >= CTTS_MAX_ADAPTERS) { ctts_set_error("set_row_adapters: slot %d out of range", slots[b]); return 1; }
        any = any || slots[b] >= 0;
    }
    if (!any) return clear_rows();
    if (lora_ensure_storage(h)) return 1;
    std::vector<int> tab(CTTS_MAX_B, -1);
    for (int b = 0; b < B; ++b) tab[b] = slots[b] < 0 ? -1 : slots[b];
    CTTS_HIP_CHECK(hipMemcpy(h->lora_slot_of_seq, tab.data(), CTTS_MAX_B * 4, hipMemcpyHostToDevice));
    h->lora_slot_host = tab;
    h->lora_req_host = tab;
    for (int b = 0; b < CTTS_MAX_B; ++b) h->lora_row_slots[b] = (signed char)tab[b];      // rows == sequences until a compaction
    h->lora_rows = 1;
    return 0;
}

// ---- packing ---------------------------------------------------------------------------------
template <typename WT> static inline WT cvt(float v);
template <> inline float cvt<float>(float v) { return v; }
template <> inline half_t cvt<half_t>(float v) { return (half_t)v; }

// rows(pr) -> pointer to the source row (K floats) or nullptr for zero padding
template <typename WT, typename RowFn>
static void pack_tiles(WT* dst, int n_row_tiles, int K, RowFn rows, const float* colscale = nullptr) {
    constexpr int KT = WTraits<WT>::KT, EPL = WTraits<WT>::EPL;
    const int ktiles = K / KT;
    for (int rt = 0; rt < n_row_tiles; ++rt)
        for (int i = 0; i < 16; ++i) {
            const float* src = rows(rt * 16 + i);
            for (int kt = 0; kt < ktiles; ++kt)
                for (int kq = 0; kq < 4; ++kq) {
                    WT* d = dst + (((size_t)rt * ktiles + kt) * 64 + i + 16 * kq) * EPL;
                    const int k0 = kt * KT + kq * EPL;
                    for (int j = 0; j < EPL; ++j) d[j] = src ? cvt<WT>(colscale ? src[k0 + j] * colscale[k0 + j] : src[k0 + j]) : cvt<WT>(0.f);
                }
        }
}

// head / tail fp16 images of 64 * W in the fp16 tile layout (prefill_split.hip): hi = fp16(v), lo = fp16(v - hi)
// Returns false when a (norm-folded) weight times 64 leaves the fp16 range (|w| >= 1023.5): the images would hold inf; the caller then keeps the
// engine on the exact fp32 prompt kernels.
template <typename RowFn>
static bool pack_tiles_split(half_t* dst, int n_row_tiles, int K, RowFn rows, const float* colscale = nullptr) {
    constexpr int KT = 32, EPL = 8;
    const int ktiles = K / KT;
    bool in_range = true;
    for (int rt = 0; rt < n_row_tiles; ++rt)
        for (int i = 0; i < 16; ++i) {
            const float* src = rows(rt * 16 + i);
            for (int kt = 0; kt < ktiles; ++kt)
                for (int kq = 0; kq < 4; ++kq) {
                    half_t* hi = dst + ((((size_t)rt * ktiles + kt) * 2) * 64 + i + 16 * kq) * EPL;      // head fragment of the (tile, k-tile) pair; the tail 64 lanes behind it
                    half_t* lo = hi + 64 * EPL;
                    const int k0 = kt * KT + kq * EPL;
                    for (int j = 0; j < EPL; ++j) {
                        const float v = src ? CTTS_SPLIT_WSCALE * (colscale ? src[k0 + j] * colscale[k0 + j] : src[k0 + j]) : 0.f;      // the product rounds exactly like pack_tiles'
                        const half_t h = (half_t)v;
                        if (!(fabsf(v) <= 65504.0f)) in_range = false;
                        hi[j] = h;
                        lo[j] = (half_t)(v - (float)h);
                    }
                }
        }
    return in_range;
}

static const std::vector<float>* need(ctts_gpt* h, const std::string& k, size_t numel) {
    auto it = h->host.find(k);
    if (it == h->host.end()) { ctts_set_error("missing weight %s", k.c_str()); return nullptr; }
    if (it->second.size() != numel) { ctts_set_error("weight %s has %zu elements, expected %zu", k.c_str(), it->second.size(), numel); return nullptr; }
    return &it->second;
}

template <typename WT>
static int finalize_t(ctts_gpt* h) {
    const int H = h->H, I = h->I, L = h->L, V = h->V;
    const size_t n_qkv = (size_t)3 * H * H, n_o = (size_t)H * H, n_gu = (size_t)2 * I * H, n_d = (size_t)H * I;
    const int head_tiles = (h->NVQ * V + 15) / 16;
    const size_t n_head = (size_t)head_tiles * 16 * H;
    const size_t per_layer = n_qkv + n_o + n_gu + n_d;
    const size_t total = per_layer * L + n_head;
    std::vector<WT> blob(total);
    h->lw.resize(L);
    if (dev_alloc((void**)&h->wblob, total * sizeof(WT))) return 1;
    // fp32 engine whose prompt passes can reach the split-GEMM threshold, or whose decode batches can reach split_decode_rows: head / tail fp16 images of the layer matrices too
    const bool want_split = (sizeof(WT) == 4) && ((h->split_rows_min > 0 && (long)h->cfg.max_batch * h->cfg.max_seq >= h->split_rows_min) ||
                                                   (h->split_dec_rows > 0 && h->cfg.max_batch >= h->split_dec_rows));
    std::vector<half_t> sblob;
    if (want_split) {
        sblob.resize(per_layer * 2 * L);
        if (dev_alloc((void**)&h->wsplit, per_layer * L * 2 * sizeof(half_t))) return 1;
    }
    if (dev_alloc((void**)&h->ln1, (size_t)L * H * 4)) return 1;       // input_layernorm weights, unfolded: the LoRA path needs w * x_hat itself
    struct LayerSrc { const std::vector<float>*q, *k, *v, *o, *g, *u, *d, *l1, *l2; };
    std::vector<LayerSrc> ls(L);
    for (int l = 0; l < L; ++l) {
        const std::string p = "gpt.layers." + std::to_string(l) + ".";
        LayerSrc& t = ls[l];
        t.q = need(h, p + "self_attn.q_proj.weight", n_o); t.k = need(h, p + "self_attn.k_proj.weight", n_o);
        t.v = need(h, p + "self_attn.v_proj.weight", n_o); t.o = need(h, p + "self_attn.o_proj.weight", n_o);
        t.g = need(h, p + "mlp.gate_proj.weight", (size_t)I * H); t.u = need(h, p + "mlp.up_proj.weight", (size_t)I * H);
        t.d = need(h, p + "mlp.down_proj.weight", n_d); t.l1 = need(h, p + "input_layernorm.weight", H);
        t.l2 = need(h, p + "post_attention_layernorm.weight", H);
        if (!t.q || !t.k || !t.v || !t.o || !t.g || !t.u || !t.d || !t.l1 || !t.l2) return 1;
        CTTS_HIP_CHECK(hipMemcpy(h->ln1 + (size_t)l * H, t.l1->data(), (size_t)H * 4, hipMemcpyHostToDevice));
    }
    // host-side packing of the 20 layers on a few threads (pure CPU work on disjoint slices of the blobs)
    std::atomic<bool> split_in_range(true);
    auto pack_layer = [&](int l) {
        const LayerSrc& t = ls[l];
        const int HT = H / 16;
        auto qkv_row = [&](int pr) -> const float* {      // QKV: tile rows = dims [8t..8t+7 | 8t+32..8t+39] of one head, so RoPE's (d, d+32) pair sits in one tile
            const int rt = pr / 16, i = pr % 16;
            const int which = rt / HT, within = rt % HT, hh = within / 4, tq = within % 4;
            const int dd = (i < 8) ? 8 * tq + i : 8 * tq + (i - 8) + 32;
            const std::vector<float>* src = which == 0 ? t.q : (which == 1 ? t.k : t.v);
            return src->data() + (size_t)(hh * CTTS_HEAD_DIM + dd) * H;
        };
        auto o_row = [&](int pr) { return t.o->data() + (size_t)pr * H; };
        auto gu_row = [&](int pr) -> const float* {       // gate|up: tile rows = [8 gate rows | the matching 8 up rows]
            const int rt = pr / 16, i = pr % 16;
            return (i < 8) ? t.g->data() + (size_t)(rt * 8 + i) * H : t.u->data() + (size_t)(rt * 8 + i - 8) * H;
        };
        auto d_row = [&](int pr) { return t.d->data() + (size_t)pr * I; };
        WT* base = blob.data() + per_layer * l;
        // RMSNorm weights folded into the columns: W (w * xn) == (W diag(w)) xn  (llama.py:87,619-621)
        pack_tiles<WT>(base, 3 * HT, H, qkv_row, t.l1->data());
        pack_tiles<WT>(base + n_qkv, HT, H, o_row);
        pack_tiles<WT>(base + n_qkv + n_o, 2 * I / 16, H, gu_row, t.l2->data());
        pack_tiles<WT>(base + n_qkv + n_o + n_gu, HT, I, d_row);
        if (want_split) {
            half_t* sp = sblob.data() + per_layer * 2 * l;
            bool ok = pack_tiles_split(sp, 3 * HT, H, qkv_row, t.l1->data());
            ok = pack_tiles_split(sp + 2 * n_qkv, HT, H, o_row) && ok;
            ok = pack_tiles_split(sp + 2 * (n_qkv + n_o), 2 * I / 16, H, gu_row, t.l2->data()) && ok;
            ok = pack_tiles_split(sp + 2 * (n_qkv + n_o + n_gu), HT, I, d_row) && ok;
            if (!ok) split_in_range.store(false);
        }
    };
    {
        const int nthreads = L < 8 ? L : 8;
        std::vector<std::thread> pool;
        for (int w = 0; w < nthreads; ++w)
            pool.emplace_back([&, w]() { for (int l = w; l < L; l += nthreads) pack_layer(l); });
        for (auto& th : pool) th.join();
    }
    for (int l = 0; l < L; ++l) {
        char* dv = h->wblob + per_layer * l * sizeof(WT);
        h->lw[l].qkv = dv;
        h->lw[l].o = dv + n_qkv * sizeof(WT);
        h->lw[l].gu = dv + (n_qkv + n_o) * sizeof(WT);
        h->lw[l].d = dv + (n_qkv + n_o + n_gu) * sizeof(WT);
        if (want_split) {
            char* sv = h->wsplit + per_layer * l * 2 * sizeof(half_t);
            h->lw[l].qkv_sp = sv;
            h->lw[l].o_sp = sv + n_qkv * 4;
            h->lw[l].gu_sp = sv + (n_qkv + n_o) * 4;
            h->lw[l].d_sp = sv + (n_qkv + n_o + n_gu) * 4;
        }
    }
    if (want_split) CTTS_HIP_CHECK(hipMemcpy(h->wsplit, sblob.data(), sblob.size() * sizeof(half_t), hipMemcpyHostToDevice));
    h->split_ok = want_split && split_in_range.load();
    if (want_split && !h->split_ok) { h->split_rows_min = 0; h->split_dec_rows = 0; }      // a weight beyond +-1023: the head / tail images would hold inf -- this engine keeps the exact fp32 kernels
    // heads: fold weight norm, W = v * (g / ||v||_row)  (gpt.py:57-77; torch._weight_norm dim=0)
    std::vector<float> folded((size_t)h->NVQ * V * H);
    for (int i = 0; i < h->NVQ; ++i) {
        const std::string p = "head_code." + std::to_string(i) + ".parametrizations.weight.original";
        const std::vector<float>*g0 = need(h, p + "0", V), *v1 = need(h, p + "1", (size_t)V * H);
        if (!g0 || !v1) return 1;
        for (int r = 0; r < V; ++r) {
            double ss = 0.0;
            const float* vr = v1->data() + (size_t)r * H;
            for (int c = 0; c < H; ++c) ss += (double)vr[c] * vr[c];
            const float a = (*g0)[r] / (float)sqrt(ss);
            float* dst = folded.data() + ((size_t)i * V + r) * H;
            for (int c = 0; c < H; ++c) dst[c] = vr[c] * a;
        }
    }
    const int nvalid = h->NVQ * V;
    const std::vector<float>* nf = need(h, "gpt.norm.weight", H);
    if (!nf) return 1;
    pack_tiles<WT>(blob.data() + per_layer * L, head_tiles, H, [&](int pr) -> const float* {
        return pr < nvalid ? folded.data() + (size_t)pr * H : nullptr;
    }, nf->data());
    h->whead = h->wblob + per_layer * L * sizeof(WT);
    if (h->split_ok && h->split_dec_rows > 0) {      // the code heads' split images (the decode path's last projection)
        std::vector<half_t> hs((size_t)head_tiles * 16 * H * 2);
        const bool ok = pack_tiles_split(hs.data(), head_tiles, H, [&](int pr) -> const float* { return pr < nvalid ? folded.data() + (size_t)pr * H : nullptr; }, nf->data());
        if (ok) {
            if (dev_alloc(&h->whead_sp, hs.size() * sizeof(half_t))) return 1;
            CTTS_HIP_CHECK(hipMemcpy(h->whead_sp, hs.data(), hs.size() * sizeof(half_t), hipMemcpyHostToDevice));
        }
    }
    {   // refine-text head (gpt.py:57-64): same weight-norm fold + final-norm fold, its own allocation
        auto g0 = h->host.find("head_text.parametrizations.weight.original0");
        auto v1 = h->host.find("head_text.parametrizations.weight.original1");
        if (g0 != h->host.end() && v1 != h->host.end() && v1->second.size() == g0->second.size() * (size_t)H) {
            const int Vt = (int)g0->second.size(), tiles = (Vt + 15) / 16;
            std::vector<float> ft((size_t)Vt * H);
            for (int r = 0; r < Vt; ++r) {
                double ss = 0.0;
                const float* vr = v1->second.data() + (size_t)r * H;
                for (int c = 0; c < H; ++c) ss += (double)vr[c] * vr[c];
                const float a = g0->second[r] / (float)sqrt(ss);
                for (int c = 0; c < H; ++c) ft[(size_t)r * H + c] = vr[c] * a;
            }
            std::vector<WT> packed((size_t)tiles * 16 * H);
            pack_tiles<WT>(packed.data(), tiles, H, [&](int pr) -> const float* { return pr < Vt ? ft.data() + (size_t)pr * H : nullptr; }, nf->data());
            if (dev_alloc(&h->whead_text, packed.size() * sizeof(WT))) return 1;
            CTTS_HIP_CHECK(hipMemcpy(h->whead_text, packed.data(), packed.size() * sizeof(WT), hipMemcpyHostToDevice));
            h->vocab_text_head = Vt;
        }
    }
    CTTS_HIP_CHECK(hipMemcpy(h->wblob, blob.data(), total * sizeof(WT), hipMemcpyHostToDevice));
    if (dev_alloc((void**)&h->lnf, H * 4)) return 1;
    CTTS_HIP_CHECK(hipMemcpy(h->lnf, nf->data(), H * 4, hipMemcpyHostToDevice));
    if (dev_alloc((void**)&h->emb_code, (size_t)h->NVQ * V * H * 4)) return 1;
    for (int i = 0; i < h->NVQ; ++i) {
        const std::vector<float>* e = need(h, "emb_code." + std::to_string(i) + ".weight", (size_t)V * H);
        if (!e) return 1;
        CTTS_HIP_CHECK(hipMemcpy(h->emb_code + (size_t)i * V * H, e->data(), (size_t)V * H * 4, hipMemcpyHostToDevice));
    }
    return 0;
}

extern "C" int ctts_gpt_finalize(ctts_gpt* h) {
    if (!h) { ctts_set_error("null handle"); return 1; }
    if (h->finalized) return 0;
    int rc = (h->cfg.dtype == CTTS_DTYPE_F16) ? finalize_t<half_t>(h) : finalize_t<float>(h);
    if (rc) return rc;
    const int H = h->H, NH = h->NH, MB = h->cfg.max_batch;
    {   // prompt rows per pass: never more than the engine can hold at all
        long cap = (long)MB * h->cfg.max_seq;
        cap = (cap + 255) / 256 * 256;
        h->pass_rows = (int)(cap < PASS_ROWS_MAX ? cap : PASS_ROWS_MAX);
        if (const char* pr = getenv("CTTS_PASS_ROWS")) { const int v = atoi(pr); if (v >= 256 && v < h->pass_rows) h->pass_rows = v / 256 * 256; }
    }
    const int PASS_ROWS = h->pass_rows;
    const size_t act_bytes = (size_t)((PASS_ROWS + PASS_PAD) / 16) * (h->I / (h->esz == 2 ? 32 : 16)) * 1024;
    if (dev_alloc((void**)&h->x_dec, (size_t)CTTS_MAX_B * H * 4) || dev_alloc((void**)&h->x_last, (size_t)CTTS_MAX_B * H * 4) ||
        dev_alloc((void**)&h->x_pre, (size_t)PASS_ROWS * H * 4) ||
        dev_alloc((void**)&h->q_buf, (size_t)PASS_ROWS * H * 4) ||
        dev_alloc((void**)&h->part_ml, (size_t)(CTTS_MAX_B + 32) * NH * SMAX * 2 * 4) ||          // flash-decoding partials: decode rows only (the prompt pass never splits keys)
        dev_alloc((void**)&h->part_o, (size_t)(CTTS_MAX_B + 32) * NH * SMAX * CTTS_HEAD_DIM * 4) ||
        dev_alloc((void**)&h->logits, (size_t)CTTS_MAX_B * (h->NVQ * h->V > h->vocab_text_head ? h->NVQ * h->V : h->vocab_text_head) * 4) || dev_alloc(&h->act, act_bytes) || dev_alloc((void**)&h->rope_pre, (size_t)MB * h->cfg.max_seq * 64 * 4) ||
        dev_alloc((void**)&h->rope_dec, (size_t)CTTS_MAX_B * 64 * 4) || dev_alloc((void**)&h->dpart, (size_t)32 * 4 * H * 4) || dev_alloc(&h->attn_packed, (size_t)((PASS_ROWS + PASS_PAD) / 16) * (H / (h->esz == 2 ? 32 : 16)) * 1024) ||
        dev_alloc(&h->norm_packed, (size_t)((PASS_ROWS + PASS_PAD) / 16) * (H / (h->esz == 2 ? 32 : 16)) * 1024) ||
        dev_alloc((void**)&h->meta_pre, (size_t)MB * h->cfg.max_seq * sizeof(RowMeta)) ||
        dev_alloc((void**)&h->meta_dec, CTTS_MAX_B * sizeof(RowMeta)) || dev_alloc((void**)&h->meta_dec0, CTTS_MAX_B * sizeof(RowMeta)) ||
        dev_alloc((void**)&h->st, sizeof(DevState)) || dev_alloc((void**)&h->last_rows, CTTS_MAX_B * 4) ||
        dev_alloc((void**)&h->dyn, sizeof(SamplerDyn)) || dev_alloc((void**)&h->hist_ring, (size_t)CTTS_MAX_B * CTTS_NUM_VQ * 16 * 4) ||
        dev_alloc((void**)&h->finend, (size_t)CTTS_MAX_B * sizeof(RowState)) || dev_alloc((void**)&h->sat, 4) ||
        dev_alloc((void**)&h->sk_slab, (size_t)(H / 16) * (CTTS_MAX_B / 16) * 4 * 256 * 4) || dev_alloc((void**)&h->sk_cnt, (size_t)(H / 16) * (CTTS_MAX_B / 16) * 4) ||
        dev_alloc((void**)&h->cx, (size_t)CTTS_MAX_B * H * 4) || dev_alloc((void**)&h->crope, (size_t)CTTS_MAX_B * 64 * 4) ||
        dev_alloc((void**)&h->cmeta, CTTS_MAX_B * sizeof(RowMeta)) || dev_alloc((void**)&h->cring, (size_t)CTTS_MAX_B * CTTS_NUM_VQ * 16 * 4) ||
        dev_alloc((void**)&h->cfin, (size_t)CTTS_MAX_B * sizeof(RowState)) || dev_alloc((void**)&h->keep_dev, CTTS_MAX_B * 4) ||
        dev_alloc(&h->xh, (size_t)((CTTS_MAX_B + 32) / 16) * (H / (h->esz == 2 ? 32 : 16)) * 1024) || dev_alloc((void**)&h->ssq, (size_t)(CTTS_MAX_B + 32) * (H / 16) * 4) ||
        dev_alloc((void**)&h->scale_o, (size_t)(CTTS_MAX_B + 32) * 4) || dev_alloc((void**)&h->scale_d, (size_t)(CTTS_MAX_B + 32) * 4))
        return 1;
    if (h->wsplit) {     // operand images of the prompt rows for the split GEMMs: heads / tails of the normalised rows | attention outputs (K = 768) and of the SwiGLU outputs (K = 3072)
        const size_t rows = (size_t)PASS_ROWS + PASS_PAD;
        if (dev_alloc(&h->sp_x_hi, rows * H * 2) || dev_alloc(&h->sp_x_lo, rows * H * 2) || dev_alloc(&h->sp_act_hi, rows * h->I * 2) || dev_alloc(&h->sp_act_lo, rows * h->I * 2)) return 1;
        // short passes slice the down projection's K four ways (prefill_split.hip sp_launch): the slices' partial outputs [4][rows <= 2048, padded to 128][H] fp32 (<= 25 MB)
        h->sk_cap_floats = (size_t)4 * (size_t)(((PASS_ROWS < 2048 ? PASS_ROWS : 2048) + 127) / 128 * 128) * H;
        if (dev_alloc((void**)&h->sk_scratch, h->sk_cap_floats * 4)) return 1;
    }
    CTTS_HIP_CHECK(hipHostMalloc((void**)&h->host_pin, 64));
    {
        auto it = h->host.find("emb_text.weight");
        if (it != h->host.end() && it->second.size() % H == 0) {
            h->vocab_text = (int)(it->second.size() / H);
            if (dev_alloc((void**)&h->emb_text, it->second.size() * 4)) return 1;
            CTTS_HIP_CHECK(hipMemcpy(h->emb_text, it->second.data(), it->second.size() * 4, hipMemcpyHostToDevice));
        }
    }
    if (h->whead_text && h->emb_text && h->vocab_text_head > h->vocab_text) {
        // the text sampler re-embeds any sampled id < vocab_text_head through emb_text: a larger head would read past the table
        ctts_set_error("finalize: head_text has %d rows but emb_text only %d", h->vocab_text_head, h->vocab_text);
        return 1;
    }
    h->host.clear();
    h->finalized = true;
    if (h->persist_rows > 0 && ensure_persist(h, false)) return 1;
    return 0;
}

extern "C" int ctts_gpt_embed(ctts_gpt* h, const int32_t* ids, const int32_t* text_mask, int B, int T, const float* spk, int spk_id, float* emb_out, void* stream) {
    if (!h || !h->finalized || !ids || !text_mask || !emb_out || B < 1 || T < 1) { ctts_set_error("embed: bad argument"); return 1; }
    if (!h->emb_text) { ctts_set_error("embed: emb_text.weight was not loaded"); return 1; }
    return launch_embed_prompt(ids, text_mask, h->emb_text, h->emb_code, spk, spk_id, emb_out, B * T, T, h->V, h->H, (hipStream_t)stream);
}

extern "C" size_t ctts_gpt_kv_bytes(const ctts_gpt* h) {
    return (size_t)h->L * 2 * h->cfg.max_batch * h->NH * h->cfg.max_seq * CTTS_HEAD_DIM * h->esz;
}
extern "C" int ctts_gpt_bind_kv(ctts_gpt* h, void* kv, size_t bytes) {
    if (!h || !kv || bytes < ctts_gpt_kv_bytes(h)) { ctts_set_error("bind_kv: need %zu bytes", h ? ctts_gpt_kv_bytes(h) : 0); return 1; }
    h->kv = (char*)kv; h->kv_bytes = bytes;
    // graphs are keyed by the KV pointer; entries captured against an older binding are simply never selected again
    return 0;
}
extern "C" int ctts_gpt_set_rope(ctts_gpt* h, const float* rope_host, int n_pos) {
    if (!h || !rope_host || n_pos < h->cfg.max_seq) { ctts_set_error("set_rope: need at least max_seq=%d positions", h ? h->cfg.max_seq : 0); return 1; }
    if (h->rope) (void)hipFree(h->rope);
    if (dev_alloc((void**)&h->rope, (size_t)(n_pos + 1) * 64 * 4)) return 1;   // +1: the sampler prefetches the row of the position after the last one
    CTTS_HIP_CHECK(hipMemcpy(h->rope, rope_host, (size_t)n_pos * 64 * 4, hipMemcpyHostToDevice));
    h->rope_n = n_pos;
    return 0;
}

// ---- launch sequencing -----------------------------------------------------------------------
static inline void* kv_layer(ctts_gpt* h, int l, int which) {
    const size_t per = (size_t)h->cfg.max_batch * h->NH * h->cfg.max_seq * CTTS_HEAD_DIM * h->esz;
    return h->kv + ((size_t)l * 2 + which) * per;
}
// Key splits of the decode attention for a batch of B rows whose longest context in the coming steps is L keys.
// Cap: enough (row, head, split) blocks to cover the chip.  Measured (us/step, 256 steps from a 48-token prompt unless noted):
//   B=1: S=1 412, 2 400, 4 392, 8 392;  L~1900: S=1 628, 2 589, 8 438        -> one row: always the cap
//   B=4: S=1 438, 2 456, 4 449;  B=8: S=1 476, 2 535                          -> S > 1 pays a softmax-combine prologue per row
// in o_proj (S = 1 lets the attention write o_proj's packed operand directly): from 4 rows on, split only when one
// 8-wave block would otherwise loop over more than ~768 keys.
static inline int decode_splits(const ctts_gpt* h, int B, int L) {
    if (h->lora_rows) return 1;                            // per-utterance LoRA reads the attention output from o_proj's packed operand (S = 1 path)
    if (h->force_splits) return h->force_splits;
    int cap = 256 / (B * h->NH);
    cap = cap < 1 ? 1 : (cap > SMAX ? SMAX : cap);
    if (B <= 2) return cap;
    if (B >= 8) return 1;                                  // B=8, L 1200..1700: S=1 703, S=2 746
    const int want = (L + 767) / 768;
    return want < 1 ? 1 : (want > cap ? cap : want);
}

// What run_layers decided about the hand-off of the residual stream to whatever reads it next (the heads): one place computes the
// predicates, the consumer uses what was actually launched.
struct StreamForm { bool parts; bool xh; bool logits; bool split; };     // x = x_dec + dpart[0..3] (split-K down projection) / packed copy + sums of squares exist / the logits (and hidden rows) exist already / the packed copy is a head / tail fp16 pair (split_t)

// 20 decoder layers on R rows of residual stream x (llama.py:719-749 per layer)
static int run_layers(ctts_gpt* h, float* x, const RowMeta* meta, const float* rope_rows, int R, int S, const DevState* st, hipStream_t s, StreamForm* form = nullptr) {
    const int dt = h->cfg.dtype;
    const bool lora = h->lora_rows != 0;                   // per-utterance adapters: the residual stream must be materialised in x (no split-K partials)
    // fp32 engines, decode batches of >= split_decode_rows rows: head / tail fp16 operands (the predicate is completed below; it needs the packed-residual path)
    const bool spd_ok = st != nullptr && dt == CTTS_DTYPE_F32 && h->xh_mode && h->split_ok && h->wsplit != nullptr && h->split_dec_rows > 0 && R >= h->split_dec_rows && R > h->split_rows && !lora && S == 1;
    const int nbg = (R <= 16 || (st != nullptr && R < (spd_ok ? h->split_nbg2_rows : h->nbg2_rows))) ? 1 : 2;     // decode rows: see nbg2_rows / split_nbg2_rows
    const int NB = 16 * nbg;
    const int chunks = (R + NB - 1) / NB;
    // small decode batches: the down projection is launched as 4 split-K slices (192 blocks instead of 48 x 1024 threads);
    // its partial sums dp[0..3] are added, in order, by the next consumers of the residual stream (QKV RMSNorm, o_proj
    // residual, final heads) -- deterministic, no atomics.  x itself is re-materialised by every o_proj.
    const bool splitd = (st != nullptr) && (R <= h->split_rows) && (nbg == 1) && !lora;
    // prompt pass over more than a few chunks: normalise every row once (norm_pack_kernel) instead of in every GEMM block
    const bool prepack = (st == nullptr) && (nbg == 2) && (R > 64) && !h->no_prepack;
    // prompt pass over >= 1536 rows, fp16: LDS-staged 256/128 x 128 MFMA GEMM (prefill_gemm.hip) instead of one weight tile per 32-row block
    // (measured with 128 x 128 blocks, prompt pass ms with / without it: 512 rows 2.31 / 1.48, 1024 rows 2.47 / 2.29, 1536 rows 2.70 / 3.12,
    //  2048 rows 2.81 / 3.9, 3072 rows 3.6 / 5.3, 16384 rows 10.3 / 29)
    const bool pfg = prepack && (dt == CTTS_DTYPE_F16) && h->prefill_gemm_rows > 0 && (R >= h->prefill_gemm_rows) && !lora;
    // prompt pass over >= 384 rows, fp32 engine: the same tiling on the fp16 pipes with head / tail operands -- 3 MFMAs per product instead of 16,
    // fp32-level accuracy (prefill_split.hip); the attention stays the fp32 row kernel
    // (round 5: also with per-utterance adapters -- their low-rank terms come from the two lora.hip launches per layer and are added in the split GEMMs' epilogues)
    const bool pfs = prepack && (dt == CTTS_DTYPE_F32) && h->wsplit != nullptr && h->split_rows_min > 0 && (R >= h->split_rows_min);
    const float sp_scale = 1.0f / 64.0f;
    const SplitGemmPolicy sp_pol = {h->prefill_pp_blocks, h->prefill_sk_rows, h->sk_scratch, h->sk_cap_floats, h->prefill_small_blocks, h->prefill_ring4_blocks};
    // decode above the split-K batch sizes: the residual stream travels between kernels as a packed B operand in the engine dtype + per-tile
    // sums of squares (EPI_RESID_XH -> PRO_XH, kernels.h); layer 0 still normalises the sampler's fp32 rows itself.  (Handing gate|up
    // the packed copy at batches <= 4 too was measured slower in fp16: batch 1 393 vs 381 us/step, 2 415 vs 403, 4 450 vs 443.)
    // fp32: the RMSNorm factor then multiplies the C tile instead of the operand -- (sum w x) rs instead of sum w (x rs), one rounding
    // apart; token ids stay bit-exact on every golden (tests/test_gpu_gpt.py)
    const bool xhm = (st != nullptr) && h->xh_mode && !splitd;
    // ... and from split_decode_rows rows on (fp32 engines) the projections read head / tail fp16 images of weights and operands: 3 fp16 MFMAs per product instead of 8
    // exact-f32 ones (common.h split_t; the prompt pass's arithmetic, prefill_split.hip).  Layer 0's q|k|v projection normalises the sampler's fp32 rows and stays exact.
    const bool spd = xhm && spd_ok;
    const int dts = spd ? 2 : dt;                              // launch_gemm's operand format
    if (st != nullptr && h->cur_persist && h->pimg != nullptr && R <= PL_MAXR) {
        // one persistent launch per layer (persist_layer.hip): the residual stream stays in x, nothing is left in partial or packed form
        // the launch that ends the stack also runs the final norm + the 4 code heads (persist_layer.hip phase H): code mode, paced schedule, images built
        // (ms/step separate heads launch / fused, tools/ab_options.py: fp32 batch 1 0.2798 / 0.2786, 2 0.3396 / 0.3384, 4 0.4711 / 0.4721; fp16 batch 3 0.3793 / 0.3800 -> up to 2 rows)
        const bool fuse_heads = h->persist_heads && R <= 2 && !h->text_mode && h->persist_sched == 3 && h->pimg_head != nullptr && h->dyn != nullptr;
        if (form) { form->parts = false; form->xh = false; form->logits = fuse_heads; form->split = false; }
        // (persistent_layers_per_launch, default all: the whole stack is ONE launch; 1 = a launch per layer, the first version of the structure)
        const int per = (h->persist_lpl > 0 && h->persist_lpl < h->L) ? h->persist_lpl : h->L;
        for (int l = 0; l < h->L; l += per) {
            PersistArgs pa = {};
            if (fuse_heads && l + per >= h->L) {
                pa.heads = 1; pa.hw = h->pimg_head; pa.logits = h->logits; pa.n_valid = h->NVQ * h->V; pa.lnf = h->lnf; pa.dyn = h->dyn; pa.rows = h->finend;
            }
            pa.w = h->pimg + PL_LAYER_BYTES / 4 * h->esz * l; pa.half_w = h->esz == 2 ? 1 : 0; pa.n_layers = (h->L - l < per) ? h->L - l : per;
            pa.x = x; pa.meta = meta; pa.rope_rows = rope_rows;
            pa.kv = kv_layer(h, l, 0); pa.kv_per = (size_t)h->cfg.max_batch * h->NH * h->cfg.max_seq * CTTS_HEAD_DIM; pa.Lmax = h->cfg.max_seq;
            pa.g_qkv = h->pl_g; pa.g_att = pa.g_qkv + PL_G_QKV; pa.g_x1 = pa.g_att + PL_G_ATT; pa.g_act = pa.g_x1 + PL_G_X1; pa.g_x = pa.g_act + PL_G_ACT; pa.g_part = pa.g_x + PL_G_X; pa.S = h->cur_persist;
            pa.epoch = h->pl_epoch; pa.error = h->pl_error; pa.done = &st->all_done; pa.ts = h->pl_ts_on ? h->pl_ts : nullptr; pa.eps = 1e-6f; pa.sched = h->persist_sched; pa.pace = h->persist_pace < 0 ? (R <= 2 ? 3 : 2) : h->persist_pace; pa.fault = h->persist_fault; pa.delay_att = h->persist_delay_att; pa.delay = h->persist_delay; pa.delay_act = h->persist_delay_act + 2 * (R - 1); pa.delay_x = h->persist_delay_x; pa.nap = h->persist_nap; pa.nap_qkv = h->persist_nap_qkv; pa.poll = (h->persist_poll < 0) ? 0 : h->persist_poll;
            if (lora) {
                // per-utterance adapters: the rows' slots travel in the arguments (part of the decode-graph key), the operands are read from the resident tables
                pa.lora = 1; pa.lslots = 0; pa.delay_u = h->persist_delay_u < 0 ? 14 + 2 * R : h->persist_delay_u;
                for (int r = 0; r < PL_MAXR; ++r) pa.lslots |= (unsigned long long)(unsigned char)(r < R ? h->lora_row_slots[r] : -1) << (8 * r);
                pa.la_qkv_stride = (size_t)CTTS_MAX_ADAPTERS * 3 * 16 * h->H; pa.la_stride = (size_t)CTTS_MAX_ADAPTERS * 4 * 16 * h->H;
                pa.la_qkv = h->lora_Af + pa.la_qkv_stride * l; pa.la = h->lora_A + pa.la_stride * l; pa.lb = h->lora_B + pa.la_stride * l;
                pa.lscale = h->lora_scale + (size_t)l * CTTS_MAX_ADAPTERS * 4; pa.g_u = pa.g_part + PL_G_PART;
            }
            if (launch_persist_layer(R, pa, s)) return 1;
        }
        return 0;
    }
    if (form) { form->parts = splitd; form->xh = xhm; form->logits = false; form->split = spd; }
    // weight prefetch across a launch boundary of a decode step (kernels.h WPrefetch): the o_proj launch carries workgroups that pull the gate|up launch's weight image into L2
    // (packed-residual path: every consumer workgroup column is one row tile; per-utterance adapters shift the consumers' block indices by their workers: off)
    const int pf_kb = (st != nullptr && xhm && !lora && (spd || (dt == CTTS_DTYPE_F16 && R >= 17))) ? h->prefetch_kb : 0;      // (the exact-f32 kernels lose with it: batch 32 0.845 -> 0.897)
    auto set_pf = [&](GemmArgs& g, const void* w, int n_tiles, int K, int fmt) {          // fmt: 0 fp32 tiles, 1 fp16 tiles, 2 head / tail pairs
        if (pf_kb <= 0 || w == nullptr) return;
        const size_t tile = (size_t)16 * K * (fmt == 1 ? 2 : 4);
        size_t nb = (tile * n_tiles + (size_t)pf_kb * 1024 - 1) / ((size_t)pf_kb * 1024);
        nb = (nb + 7) & ~(size_t)7;
        nb = nb < 8 ? 8 : (nb > 256 ? 256 : nb);
        g.pf.ptr = w; g.pf.unit_bytes = (unsigned)tile; g.pf.n_units = (unsigned)n_tiles; g.pf_blocks = (int)nb;
    };
    for (int l = 0; l < h->L; ++l) {
        GemmArgs a = {};
        a.st = st; a.R = R; a.eps = 1e-6f; a.meta = meta; a.Lmax = h->cfg.max_seq; a.sat = h->sat;
        a.valu = (st != nullptr && dt == CTTS_DTYPE_F32 && R <= h->valu_rows && !lora) ? 1 : 0;
        // RMSNorm + QKV + RoPE + KV append
        GemmArgs g1 = a;
        g1.W = h->lw[l].qkv; g1.n_row_tiles = 3 * h->H / 16; g1.K = h->H; g1.x = x;
        g1.q_out = h->q_buf; g1.k_cache = kv_layer(h, l, 0); g1.v_cache = kv_layer(h, l, 1); g1.rope_rows = rope_rows;
        g1.opart = h->dpart; g1.np = (splitd && l > 0) ? 4 : 0;
        const size_t lora_l = (size_t)l * CTTS_MAX_ADAPTERS * 4 * 16 * h->H;
        // decode steps: the rows' low-rank terms come from worker workgroups inside the QKV / o_proj launches (lora_worker.h) instead of two more launches
        const bool lfold = lora && st != nullptr && h->lora_fold && h->lora_g != nullptr && h->H == 768 && R <= CTTS_MAX_B;
        if (lfold) {
            LoraFold lf = {};
            lf.A = h->lora_A + lora_l; lf.B = h->lora_B + lora_l; lf.scale = h->lora_scale + (size_t)l * CTTS_MAX_ADAPTERS * 4; lf.lnw = h->ln1 + (size_t)l * h->H;
            memcpy(lf.slots, h->lora_row_slots, sizeof(lf.slots)); static_assert(sizeof(lf.ranks) == CTTS_MAX_ADAPTERS * 4, "LoraFold.ranks");
            if (!h->lora_rank.empty()) memcpy(lf.ranks, &h->lora_rank[(size_t)l * CTTS_MAX_ADAPTERS * 4], sizeof(lf.ranks));
            lf.g = h->lora_g; lf.g_o = h->lora_g + (size_t)CTTS_MAX_B * 3 * h->H; lf.err = h->pl_error; lf.layer = l; lf.diag = h->lora_fold;
            g1.lf = lf; g1.lora_w = 3 * NB;
        } else if (lora) {
            if (launch_lora_delta_qkv(x, h->ln1 + (size_t)l * h->H, a.eps, meta, h->lora_slot_of_seq, h->lora_A + lora_l, h->lora_B + lora_l,
                                      h->lora_scale + (size_t)l * CTTS_MAX_ADAPTERS * 4, h->lora_dqkv, R, h->H, s)) return 1;
            g1.lora_delta = h->lora_dqkv;
        }
        if (pfs) {
            if (launch_norm_pack_split(x, h->sp_x_hi, h->sp_x_lo, R, a.eps, s)) return 1;
            if (launch_prefill_split_gemm(EPI_QKV, g1, h->lw[l].qkv_sp, h->sp_x_hi, h->sp_x_lo, nullptr, nullptr, sp_scale, sp_pol, s)) return 1;
        } else if (prepack) {
            g1.xpacked = h->norm_packed;
            if (launch_norm_pack(dt, x, h->norm_packed, R, nbg, a.eps, s)) return 1;
            if (pfg ? launch_prefill_gemm(EPI_QKV, g1, s) : launch_gemm(dt, nbg, PRO_PACKED, EPI_QKV, g1, chunks, s)) return 1;
        } else if (xhm) {
            g1.scale_out = h->scale_o;
            if (l > 0) { g1.xh = h->xh; g1.ssq = h->ssq; g1.scale_in = h->scale_d; }
            if (spd && l > 0) g1.W = h->lw[l].qkv_sp;
            if (launch_gemm(l > 0 ? dts : dt, nbg, l > 0 ? PRO_XH : PRO_NORM, EPI_QKV, g1, chunks, s)) return 1;
        } else if (launch_gemm(dt, nbg, splitd ? PRO_NORM_P : PRO_NORM, EPI_QKV, g1, chunks, s)) return 1;
        AttnArgs at = {};
        at.q = h->q_buf; at.k_cache = g1.k_cache; at.v_cache = g1.v_cache; at.Lmax = h->cfg.max_seq; at.NH = h->NH; at.R = R; at.S = S;
        at.meta = meta; at.st = st; at.part_ml = h->part_ml; at.part_o = h->part_o; at.wide_blocks = h->attn_wide_blocks;
        if (st == nullptr) { at.T = h->pre_T; at.row0 = (int)(meta - h->meta_pre); }       // prompt pass: position of this pass in the flattened [B][T] prompt
        at.packed_out = (S == 1) ? h->attn_packed : nullptr; at.nbg = nbg; at.packed_split = spd ? 1 : 0;
        if (pfs && S == 1) { if (launch_attention_split(at, h->sp_x_hi, h->sp_x_lo, s)) return 1; }      // writes o_proj's head / tail operand images directly
        else if (launch_attention(dt, at, s)) return 1;
        // softmax combine + o_proj + residual (S == 1: attention already wrote the normalised, packed B operand)
        GemmArgs g2 = a;
        g2.W = h->lw[l].o; g2.n_row_tiles = h->H / 16; g2.K = h->H; g2.part_ml = h->part_ml; g2.part_o = h->part_o; g2.S = S; g2.x_out = x;
        g2.xpacked = h->attn_packed;
        g2.opart = h->dpart; g2.np = (splitd && l > 0) ? 4 : 0;      // the down projection's partial sums are folded into x here
        if (xhm) { g2.xh = h->xh; g2.ssq = h->ssq; g2.scale_in = h->scale_o; }
        if (lora && S != 1) { ctts_set_error("per-utterance LoRA needs unsplit attention"); return 1; }
        if (lfold) { g2.lf = g1.lf; g2.lora_w = NB; }
        else if (lora) {
            if (pfs && S == 1) {
                if (launch_lora_delta_o_split(h->sp_x_hi, h->sp_x_lo, meta, h->lora_slot_of_seq, h->lora_A + lora_l, h->lora_B + lora_l,
                                              h->lora_scale + (size_t)l * CTTS_MAX_ADAPTERS * 4, h->lora_do, R, h->H, s)) return 1;
            } else if (launch_lora_delta_o(dt, h->attn_packed, nbg, meta, h->lora_slot_of_seq, h->lora_A + lora_l, h->lora_B + lora_l,
                                           h->lora_scale + (size_t)l * CTTS_MAX_ADAPTERS * 4, h->lora_do, R, h->H, s)) return 1;
            g2.lora_delta = h->lora_do;
        }
        if (pfs && S == 1) {
            if (launch_prefill_split_gemm(EPI_RESID, g2, h->lw[l].o_sp, h->sp_x_hi, h->sp_x_lo, nullptr, nullptr, sp_scale, sp_pol, s)) return 1;
        } else if (pfg && S == 1) { if (launch_prefill_gemm(EPI_RESID, g2, s)) return 1; }
        else {
            if (spd) g2.W = h->lw[l].o_sp;
            set_pf(g2, spd ? h->lw[l].gu_sp : h->lw[l].gu, 2 * h->I / 16, h->H, dts);
            if (launch_gemm(dts, nbg, (S == 1) ? PRO_PACKED : PRO_ATTN, xhm ? EPI_RESID_XH : (splitd ? EPI_RESID_P : EPI_RESID), g2, chunks, s)) return 1;
        }
        // RMSNorm + gate|up + SiLU*up
        GemmArgs g3 = a;
        g3.W = h->lw[l].gu; g3.n_row_tiles = 2 * h->I / 16; g3.K = h->H; g3.x = x; g3.act_out = h->act;
        if (pfs) {
            if (launch_norm_pack_split(x, h->sp_x_hi, h->sp_x_lo, R, a.eps, s)) return 1;
            if (launch_prefill_split_gemm(EPI_SWIGLU, g3, h->lw[l].gu_sp, h->sp_x_hi, h->sp_x_lo, h->sp_act_hi, h->sp_act_lo, sp_scale, sp_pol, s)) return 1;
        } else if (prepack) {
            g3.xpacked = h->norm_packed;
            if (launch_norm_pack(dt, x, h->norm_packed, R, nbg, a.eps, s)) return 1;
            if (pfg ? launch_prefill_gemm(EPI_SWIGLU, g3, s) : launch_gemm(dt, nbg, PRO_PACKED, EPI_SWIGLU, g3, chunks, s)) return 1;
        } else if (xhm) {
            g3.xh = h->xh; g3.ssq = h->ssq; g3.scale_in = h->scale_o; g3.scale_out = h->scale_d;
            if (spd) g3.W = h->lw[l].gu_sp;
            if (launch_gemm(dts, nbg, PRO_XH, EPI_SWIGLU, g3, chunks, s)) return 1;
        } else if (launch_gemm(dt, nbg, PRO_NORM, EPI_SWIGLU, g3, chunks, s)) return 1;
        // down + residual
        GemmArgs g4 = a;
        g4.W = h->lw[l].d; g4.n_row_tiles = h->H / 16; g4.K = h->I; g4.xpacked = h->act; g4.x_out = x;
        if (pfs) {      // the SwiGLU images hold silu(g) * u / 16
            if (launch_prefill_split_gemm(EPI_RESID, g4, h->lw[l].d_sp, h->sp_act_hi, h->sp_act_lo, nullptr, nullptr, sp_scale * 16.0f, sp_pol, s)) return 1;
        } else if (pfg) {
            if (launch_prefill_gemm(EPI_RESID, g4, s)) return 1;
        } else if (splitd) {
            g4.part_out = h->dpart; g4.ktiles_total = h->I / (h->esz == 2 ? 32 : 16);
            if (launch_gemm(dt, nbg, PRO_PACKED, EPI_PART, g4, chunks, s)) return 1;
        } else if (xhm) {                          // (the last layer's copy is for the heads: run_heads)
            g4.xh = h->xh; g4.ssq = h->ssq; g4.scale_in = h->scale_d;
            if (spd) g4.W = h->lw[l].d_sp;
            if ((nbg == 1 || spd) && R >= h->down_sk_rows && h->down_sk_rows > 0) {
                // K sliced 4 ways inside the launch, last arriver combines (EPI_RESID_XH_SK, kernels.h)
                g4.ktiles_total = h->I / ((h->esz == 2 || spd) ? 32 : 16); g4.sk_slab = h->sk_slab; g4.sk_cnt = h->sk_cnt;
                if (launch_gemm(dts, nbg, PRO_PACKED, EPI_RESID_XH_SK, g4, chunks, s)) return 1;
            } else if (launch_gemm(dts, nbg, PRO_PACKED, EPI_RESID_XH, g4, chunks, s)) return 1;
        } else if (launch_gemm(dt, nbg, PRO_PACKED, EPI_RESID, g4, chunks, s)) return 1;
    }
    return 0;
}

// final RMSNorm + heads on the h->B decode rows; `form` = how the last run_layers left them (all false after a prompt pass / restart)
static int run_heads(ctts_gpt* h, bool write_hidden, StreamForm form, hipStream_t s) {
    const int nbg = (h->B <= 16 || h->B < (form.split ? h->split_nbg2_rows : h->nbg2_rows)) ? 1 : 2;      // (the packed rows' layout does not depend on the producer's block height)
    const int chunks = (h->B + 16 * nbg - 1) / (16 * nbg);
    GemmArgs a = {};
    a.st = h->st; a.R = h->B; a.eps = 1e-6f; a.meta = h->meta_dec;
    a.valu = (h->cfg.dtype == CTTS_DTYPE_F32 && h->B <= h->valu_rows) ? 1 : 0;
    const int nv = h->text_mode ? h->vocab_text_head : h->NVQ * h->V;
    a.W = h->text_mode ? h->whead_text : h->whead; a.n_row_tiles = (nv + 15) / 16; a.K = h->H; a.x = h->x_dec; a.lnw = h->lnf;
    a.logits = h->logits; a.n_valid = nv;
    a.dyn = write_hidden ? h->dyn : nullptr;       // the kernel tests dyn->hidden_out itself
    a.rows = h->finend;
    a.opart = h->dpart; a.np = form.parts ? 4 : 0;
    // the last down projection left the rows as packed fp16 + sums of squares (PRO_XH): no fp32 re-normalisation per block.  The text head is a
    // different launch shape (not measured): it keeps the fp32 prologue
    if (form.xh && !h->text_mode && h->xh_heads && (!form.split || h->whead_sp != nullptr)) {
        a.xh = h->xh; a.ssq = h->ssq; a.scale_in = h->scale_d;
        if (form.split) a.W = h->whead_sp;
        return launch_gemm(form.split ? 2 : h->cfg.dtype, nbg, PRO_XH, EPI_LOGITS, a, chunks, s);
    }
    return launch_gemm(h->cfg.dtype, nbg, (nbg == 1) ? PRO_NORM_P : PRO_NORM, EPI_LOGITS, a, chunks, s);
}

static int run_sample_phase(ctts_gpt* h, StreamForm form, hipStream_t s) {
    if (!form.logits && run_heads(h, true, form, s)) return 1;      // (the persistent launch that ended the stack wrote the logits and the hidden rows itself)
    SamplerArgs sa = {};
    sa.dyn = h->dyn; sa.logits = h->logits; sa.V = h->text_mode ? h->vocab_text_head : h->V; sa.B = h->B; sa.st = h->st;
    sa.text_mode = h->text_mode;
    sa.emb_code = h->text_mode ? h->emb_text : h->emb_code; sa.H = h->H; sa.x_next = h->x_dec; sa.meta = h->meta_dec; sa.rope = h->rope; sa.rope_rows = h->rope_dec;
    sa.hist_ring = h->hist_ring; sa.finend = h->finend;
    return launch_sampler(sa, h->B, s);
}

static int reset_state(ctts_gpt* h, bool keep_draw, hipStream_t s) {
    // step/all_done/ticket <- 0; decode rows back to (slot T-1, pos cum-1); finish/end_idx <- 0
    h->launched = 0;
    CTTS_HIP_CHECK(hipMemsetAsync(&h->st->step, 0, 4, s));
    if (!keep_draw) CTTS_HIP_CHECK(hipMemsetAsync(&h->st->draw, 0, 4, s));
    CTTS_HIP_CHECK(hipMemsetAsync(&h->st->all_done, 0, 8, s));          // all_done + ticket
    CTTS_HIP_CHECK(hipMemcpyAsync(h->meta_dec, h->meta_dec0, h->B * sizeof(RowMeta), hipMemcpyDeviceToDevice, s));
    CTTS_HIP_CHECK(hipMemsetAsync(h->io.finish, 0, h->B * 4, s));
    CTTS_HIP_CHECK(hipMemsetAsync(h->io.end_idx, 0, h->B * 4, s));
    if (!keep_draw) {       // begin: {fin 0, end 0, attempt 0, limit, utterance id} per row (pageable source: staged before the call returns)
        CTTS_HIP_CHECK(hipMemcpyAsync(h->finend, h->rows_host.data(), h->B * sizeof(RowState), hipMemcpyHostToDevice, s));
    } else {
        if (launch_restart_rows(h->finend, h->B, s)) return 1;
        h->row_ctx.assign(h->B, h->T);
    }      // ensure_non_empty regenerate: rows that ended at step 0 move on to their next noise attempt
    CTTS_HIP_CHECK(hipMemsetAsync(h->hist_ring, 0xFF, (size_t)h->B * CTTS_NUM_VQ * 16 * 4, s));      // -1: no id sampled yet
    return 0;
}

extern "C" int ctts_gpt_begin(ctts_gpt* h, int B, int T, const int32_t* mask, const ctts_sampler_cfg* sc, const ctts_gen_io* io, void* stream) {
    if (!h || !h->finalized || !h->kv || !h->rope) { ctts_set_error("begin: handle not ready (finalize / bind_kv / set_rope)"); return 1; }
    if (!mask || !sc || !io || !io->ids || !io->finish || !io->end_idx) { ctts_set_error("begin: null argument"); return 1; }
    CTTS_RANGE("ctts_gpt_begin");               // reference: nvtx "adjust_buffer" / "set_tensors" (trt_models/predictor.py:142,159)
    if (B < 1 || B > h->cfg.max_batch || T < 1 || T + sc->max_new_token > h->cfg.max_seq) {
        ctts_set_error("begin: B=%d T=%d max_new=%d exceed max_batch=%d / max_seq=%d", B, T, sc->max_new_token, h->cfg.max_batch, h->cfg.max_seq);
        return 1;
    }
    h->text_mode = sc->infer_text ? 1 : 0;
    if (h->text_mode) {
        if (!h->whead_text || !h->emb_text) { ctts_set_error("begin: infer_text needs head_text.* and emb_text.weight"); return 1; }
        if (sc->use_penalty) { ctts_set_error("begin: infer_text supports repetition_penalty == 1 only (the reference's processor mis-broadcasts the [B,n,1] history in this mode)"); return 1; }
        if (sc->eos_token >= h->vocab_text_head) { ctts_set_error("begin: eos out of range"); return 1; }
    } else if (sc->past_window > 16 || sc->eos_token >= h->V) { ctts_set_error("begin: past_window>16 or eos out of range"); return 1; }
    hipStream_t s = (hipStream_t)stream;
    h->B = B; h->B0 = B; h->T = T; h->io = *io;
    h->admitted = false;
    h->rows_host.assign(B, RowState{});
    for (int b = 0; b < B; ++b) {
        RowState& r = h->rows_host[b];
        const unsigned long long uid = io->utt_ids ? io->utt_ids[b] : (unsigned long long)b;
        int lim = io->row_limits ? io->row_limits[b] : sc->max_new_token;
        r.limit = lim < 1 ? 1 : (lim > sc->max_new_token ? sc->max_new_token : lim);
        r.uid_lo = (unsigned)uid; r.uid_hi = (unsigned)(uid >> 32);
        r.out = b;
    }
    h->row_seq.resize(B); h->row_ctx.assign(B, T); h->row_cap.resize(B);
    for (int b = 0; b < B; ++b) { h->row_seq[b] = b; h->row_cap[b] = T + h->rows_host[b].limit; }
    if (!h->lora_slot_host.empty()) {
        // per-utterance adapters: the call starts from what ctts_gpt_set_row_adapters requested.  (lora_rows is the LIVE state: ctts_gpt_compact clears it once every
        // adapter-carrying row has left the batch and ctts_gpt_admit_adapters rewrites the tables -- a second generate() after ONE set_row_adapters() call used to
        // run without its adapters, silently)
        if (h->lora_req_host.empty()) h->lora_slot_host.assign(CTTS_MAX_B, -1);
        else h->lora_slot_host = h->lora_req_host;
        bool any = false;
        for (int b = 0; b < CTTS_MAX_B; ++b) { h->lora_row_slots[b] = (signed char)h->lora_slot_host[b]; any = any || (b < B && h->lora_slot_host[b] >= 0); }
        h->lora_rows = any ? 1 : 0;
        if (h->lora_slot_of_seq)        // (pageable source: staged before the call returns)
            CTTS_HIP_CHECK(hipMemcpyAsync(h->lora_slot_of_seq, h->lora_slot_host.data(), CTTS_MAX_B * 4, hipMemcpyHostToDevice, (hipStream_t)stream));
        // the folded launches tag their granules with the sampler's draw counter, which restarts at every begin(): forget the previous call's granules, or a tile could
        // accept a stale term whose tag happens to match (a previous call of exactly as many steps)
        if (h->lora_rows && h->lora_g) CTTS_HIP_CHECK(hipMemsetAsync(h->lora_g, 0, (size_t)CTTS_MAX_B * 4 * h->H * 8, (hipStream_t)stream));
    }
    h->pre_T = T;
    h->io.utt_ids = nullptr; h->io.row_limits = nullptr;      // host arrays are consumed here, not kept
    memcpy(h->sc.temperature, sc->temperature, sizeof(sc->temperature));
    h->sc.top_p_threshold = sc->top_p_threshold; h->sc.top_k = sc->top_k; h->sc.min_keep = sc->min_tokens_to_keep;
    h->sc.use_penalty = sc->use_penalty; memcpy(h->sc.penalty_table, sc->penalty_table, sizeof(sc->penalty_table));
    h->sc.past_window = sc->past_window; h->sc.max_input_ids = sc->max_input_ids; h->sc.eos = sc->eos_token;
    h->sc.min_new = sc->min_new_token; h->sc.max_new = sc->max_new_token;
    SamplerDyn d = {};
    d.cfg = h->sc; d.n_draws = io->n_draws; d.ids = io->ids; d.finish = io->finish; d.end_idx = io->end_idx; d.noise = io->noise;
    d.seed = io->seed; d.hidden_out = io->hiddens; d.hidden_stride = sc->max_new_token * h->H;
    d.rows0 = B * (h->text_mode ? 1 : CTTS_NUM_VQ);
    CTTS_HIP_CHECK(hipMemcpyAsync(h->dyn, &d, sizeof(d), hipMemcpyHostToDevice, s));       // pageable source: staged before the call returns
    if (launch_fill_meta(h->meta_pre, h->meta_dec0, h->st, mask, B, T, h->rope, h->rope_pre, s)) return 1;
    CTTS_HIP_CHECK(hipMemsetAsync(h->sat, 0, 4, s));
    if (h->pl_error) CTTS_HIP_CHECK(hipMemsetAsync(h->pl_error, 0, 4, s));
    return reset_state(h, false, s);
}

extern "C" int ctts_gpt_prefill(ctts_gpt* h, const float* emb, void* stream) {
    if (!h || !emb || h->B == 0) { ctts_set_error("prefill: call begin first"); return 1; }
    CTTS_RANGE("ctts_gpt_prefill");             // reference: nvtx "forward" (trt_models/llama_trt_model.py:44,74), q_len > 1
    hipStream_t s = (hipStream_t)stream;
    const int R = h->B * h->T;
    const int PASS_ROWS = h->pass_rows;
    for (int r0 = 0; r0 < R; r0 += PASS_ROWS) {
        const int n = (R - r0 < PASS_ROWS) ? R - r0 : PASS_ROWS;
        CTTS_HIP_CHECK(hipMemcpyAsync(h->x_pre, emb + (size_t)r0 * h->H, (size_t)n * h->H * 4, hipMemcpyDeviceToDevice, s));
        if (run_layers(h, h->x_pre, h->meta_pre + r0, h->rope_pre + (size_t)r0 * 64, n, 1, nullptr, s)) return 1;
        // rows (b, T-1) that live in this pass -> x_dec[b]  (indices computed on the device: the call stays asynchronous)
        if (launch_gather_last_rows(h->x_pre, h->x_dec, h->B, h->T, r0, n, h->H, s)) return 1;
    }
    // keep a copy of the prompt's last residual rows for ensure_non_empty restarts
    CTTS_HIP_CHECK(hipMemcpyAsync(h->x_last, h->x_dec, (size_t)h->B * h->H * 4, hipMemcpyDeviceToDevice, s));
    return 0;
}

extern "C" int ctts_gpt_sample(ctts_gpt* h, void* stream) {
    if (!h || h->B == 0) { ctts_set_error("sample: call begin first"); return 1; }
    CTTS_RANGE("ctts_gpt_sample");
    return run_sample_phase(h, StreamForm{false, false, false, false}, (hipStream_t)stream);
}

extern "C" int ctts_gpt_restart(ctts_gpt* h, void* stream) {
    if (!h || h->B == 0) { ctts_set_error("restart: call begin first"); return 1; }
    if (h->B != h->B0) { ctts_set_error("restart: rows were compacted away (a regenerate restarts the whole batch at step 0, gpt.py:496-525)"); return 1; }
    if (h->admitted) { ctts_set_error("restart: rows of this batch were handed to other utterances (ctts_gpt_admit); re-admit the utterance with attempt + 1 instead"); return 1; }
    hipStream_t s = (hipStream_t)stream;
    CTTS_HIP_CHECK(hipMemcpyAsync(h->x_dec, h->x_last, (size_t)h->B * h->H * 4, hipMemcpyDeviceToDevice, s));
    return reset_state(h, true, s);
}

// host mirror of the decode rows' context lengths (an upper bound: a row stops growing when it finishes, at the latest at prompt + limit):
// the longest one after `n_steps` more steps picks the attention's key-split count
static int advance_rows(ctts_gpt* h, int n_steps) {
    int longest = 1;
    for (int r = 0; r < h->B; ++r) {
        int c = h->row_ctx[r] + n_steps;
        if (c > h->row_cap[r]) c = h->row_cap[r];
        h->row_ctx[r] = c;
        if (c > longest) longest = c;
    }
    return longest;
}

// The steps about to be launched run their layers as persistent launches: small fp32 batches without per-utterance adapters whose longest context stays
// within what one workgroup per (row, head) serves.
// Returns the key splits per (row, head) (0 = launch chain): as many shares as keep every share within what a workgroup prefetches, at most 64 / (12 B) and
// PL_SMAX; a context beyond twice that many prefetchable keys goes back to the launch chain (its attention spreads the keys over up to 96 workgroups).
static inline int decode_persist(const ctts_gpt* h, int B, int L) {
    if (!(h->persist_rows > 0 && h->pimg != nullptr && B <= h->persist_rows && B <= PL_MAXR)) return 0;
    if (h->lora_rows && !(h->persist_lora && h->persist_sched == 3 && h->lora_Af != nullptr && h->H == PL_H)) return 0;      // per-utterance adapters ride inside the launch (round 6, paced schedule)
    int cap = PL_ATT_BLOCKS / (PL_NH * B);
    cap = cap > PL_SMAX ? PL_SMAX : (cap < 1 ? 1 : cap);
    if (h->persist_splits > 0) cap = h->persist_splits < cap ? h->persist_splits : cap;
    // (until round 5 a context beyond 2 * 384 * cap + 256 keys went back to the launch chain: the share's tail streamed one 64-key step per round trip.  Four steps
    //  per round trip now; "persistent_max_keys" restores a limit for A/Bs)
    // Measured (tools/long_ctx_probe.py, ms/step launch chain / persistent): 3 rows 1200 keys 0.702 / 0.644, 1900 keys 0.733 / 0.808; 4 rows 1200 keys 0.717 / 0.708,
    // 1900 keys 0.748 / 0.877; 2 rows (2 shares) 1900 keys 0.568 / 0.535; 1 row 1900 keys 0.504 / 0.364: one workgroup streams a share at ~40 GB/s, so a share
    // beyond ~1400 keys loses to the chain's 144 attention blocks -> the limit is 1400 keys per share
    if (B > PL_MAXR_ONE) {
        // 6..8 rows: two (row, head) items per attention workgroup, no key splits: 192 keys per item are requested before the query exists, the rest streams behind it
        // 128 keys per round trip -- the chain's 72..96 attention blocks win beyond a few such trips ("persistent_max_keys")
        if (h->persist_sched != 3) return 0;
        return (L > (h->persist_max_keys > 0 ? h->persist_max_keys : h->persist_pair_keys - 128 * (B - PL_MAXR_ONE - 1))) ? 0 : 1;      // (704 / 576 / 448 keys at 6 / 7 / 8 rows)
    }
    if (L > (h->persist_max_keys > 0 ? h->persist_max_keys : 1400 * cap)) return 0;
    const int share = h->persist_share_keys > 0 ? h->persist_share_keys : PL_SHARE_KEYS;
    if (L <= share + 128) return 1;                         // (the 128 keys beyond the registers wait in LDS since round 6: cheaper than the extra hop)
    const int want = (L + share - 1) / share;
    return want > cap ? cap : want;
}

static inline int pick_decode_path(ctts_gpt* h, int longest) {
    h->cur_splits = decode_splits(h, h->B, longest);
    if (h->persist_ok && h->pimg == nullptr && h->persist_rows > 0 && h->B <= h->persist_rows && (!h->lora_rows || h->persist_lora) && persist_images(h)) return 1;
    h->cur_persist = decode_persist(h, h->B, longest);
    return 0;
}

// Persistent launches of ONE process on one device take turns.  Each needs all 256 workgroups resident; two of them enqueued on different streams (two engines, two
// threads: a base engine and a LoRA-merged sibling, two pipelines) could each be given a share of the CUs and wait for the other's until both give up.  The
// per-device file lock keeps OTHER processes off the mode; this keeps the process's own streams apart: a decode call that launches persistent kernels first makes its
// stream wait for the previous such call's last launch (an event, no host wait), then records its own.  Calls on one stream are ordered anyway.
struct PersistTurn {
    std::mutex mu;
    hipEvent_t ev = nullptr;
    hipStream_t last = nullptr;
    bool pending = false;
};
static PersistTurn* persist_turn() {
    static std::mutex mu;
    static std::map<int, PersistTurn*> turns;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return nullptr;
    std::lock_guard<std::mutex> lk(mu);
    auto it = turns.find(dev);
    if (it != turns.end()) return it->second;
    PersistTurn* t = new PersistTurn();
    if (hipEventCreateWithFlags(&t->ev, hipEventDisableTiming) != hipSuccess) { delete t; return nullptr; }
    turns[dev] = t;
    return t;
}
struct PersistTurnGuard {
    PersistTurn* t = nullptr;
    hipStream_t s = nullptr;
    int rc = 0;
    PersistTurnGuard(bool persistent, hipStream_t stream) : s(stream) {
        if (!persistent) return;
        t = persist_turn();
        if (!t) return;
        t->mu.lock();
        if (t->pending && t->last != s && hipStreamWaitEvent(s, t->ev, 0) != hipSuccess) rc = 1;
    }
    ~PersistTurnGuard() {
        if (!t) return;
        if (hipEventRecord(t->ev, s) == hipSuccess) { t->last = s; t->pending = true; }
        t->mu.unlock();
    }
};

static int run_decode_step(ctts_gpt* h, hipStream_t s) {
    StreamForm form = {false, false, false, false};
    if (run_layers(h, h->x_dec, h->meta_dec, h->rope_dec, h->B, h->cur_splits, h->st, s, &form)) return 1;
    return run_sample_phase(h, form, s);            // the heads add dpart[0..3] / read the packed copy; the sampler then re-materialises x_dec
}

// steps per replay of the current decode path, given `left` steps to launch: the long graph of the persistent paths while it fits, else the short one
static int graph_span(const ctts_gpt* h, int left) {
    return (h->cur_persist && h->graph_steps_persist > h->graph_steps && left >= h->graph_steps_persist) ? h->graph_steps_persist : h->graph_steps;
}

static int ensure_graph(ctts_gpt* h, int n_steps) {
    char sig[160];
    snprintf(sig, sizeof(sig), "%d|%d|%p|%d|%d|%d|%d|%d", h->B, h->text_mode, (void*)h->kv, h->cur_splits, h->lora_rows, h->opt_gen, h->cur_persist, n_steps);      // (diagnostic switches are fixed at create)
    std::string key(sig);
    if (h->lora_rows) key.append((const char*)h->lora_row_slots, (size_t)h->B);      // the rows' adapter slots are kernel arguments of the folded launches (LoraFold)
    if (h->graph_gen != h->opt_gen || h->graphs.size() >= 96) {
        // graphs captured under other option / adapter settings are dead weight (loading one adapter bumps opt_gen ~80 times), and the cache is bounded: a serving
        // process cycles through few (batch, mode) shapes.  Replays of the dropped graphs may still be in flight (generate() keeps two chunks enqueued): drain first.
        if (!h->graphs.empty()) {
            CTTS_HIP_CHECK(hipDeviceSynchronize());
            for (auto& kv : h->graphs) { (void)hipGraphExecDestroy(kv.second.exec); (void)hipGraphDestroy(kv.second.graph); }
            h->graphs.clear(); h->gexec = nullptr;
        }
        h->graph_gen = h->opt_gen;
    }
    auto it = h->graphs.find(key);
    if (it != h->graphs.end()) { h->gexec = it->second.exec; return 0; }
    ctts_gpt::GraphEntry ge = {nullptr, nullptr};
    CTTS_HIP_CHECK(hipStreamBeginCapture(h->cap_stream, hipStreamCaptureModeThreadLocal));
    int rc = 0;
    for (int i = 0; i < n_steps && !rc; ++i) rc = run_decode_step(h, h->cap_stream);
    hipError_t e = hipStreamEndCapture(h->cap_stream, &ge.graph);
    if (rc) { if (ge.graph) (void)hipGraphDestroy(ge.graph); return 1; }
    if (e != hipSuccess) { ctts_set_error("hipStreamEndCapture: %s", hipGetErrorString(e)); return 1; }
    e = hipGraphInstantiate(&ge.exec, ge.graph, nullptr, nullptr, 0);
    if (e != hipSuccess) { (void)hipGraphDestroy(ge.graph); ctts_set_error("hipGraphInstantiate: %s", hipGetErrorString(e)); return 1; }
    h->graphs[key] = ge;
    h->gexec = ge.exec;
    return 0;
}

extern "C" int ctts_gpt_decode(ctts_gpt* h, int n_steps, int use_graph, void* stream) {
    if (!h || h->B == 0) { ctts_set_error("decode: call begin first"); return 1; }
    CTTS_RANGE("ctts_gpt_decode");              // reference: nvtx "forward" per decode step + "execute" (trt_models/predictor.py:164)
    hipStream_t s = (hipStream_t)stream;
    {
        const int longest = advance_rows(h, n_steps) + 1;
        if (pick_decode_path(h, longest)) return 1;
    }
    h->launched += n_steps;
    PersistTurnGuard turn(h->cur_persist != 0, s);
    if (turn.rc) { ctts_set_error("decode: hipStreamWaitEvent failed"); return 1; }
    if (use_graph) {
        int left = n_steps;
        while (left >= h->graph_steps) {
            const int span = graph_span(h, left);
            if (ensure_graph(h, span)) return 1;
            for (; left >= span; left -= span) CTTS_HIP_CHECK(hipGraphLaunch(h->gexec, s));
        }
        for (; left > 0; --left) if (run_decode_step(h, s)) return 1;
    } else {
        for (int i = 0; i < n_steps; ++i) if (run_decode_step(h, s)) return 1;
    }
    return 0;
}

extern "C" int ctts_gpt_progress(ctts_gpt* h, int32_t* steps_done, int32_t* all_finished, void* stream) {
    if (!h) { ctts_set_error("null handle"); return 1; }
    hipStream_t s = (hipStream_t)stream;
    CTTS_HIP_CHECK(hipMemcpyAsync(h->host_pin, h->st, 16, hipMemcpyDeviceToHost, s));
    h->host_pin[12] = 0;
    if (h->pl_error) CTTS_HIP_CHECK(hipMemcpyAsync(h->host_pin + 12, h->pl_error, 4, hipMemcpyDeviceToHost, s));
    CTTS_HIP_CHECK(hipStreamSynchronize(s));
    if (steps_done) *steps_done = h->host_pin[0];
    if (all_finished) *all_finished = h->host_pin[2];
    if (h->host_pin[12] == 7) {
        ctts_set_error("per-utterance LoRA: a projection tile gave up waiting for its low-rank term (lora_worker.h); use options={'lora_fold': 0}");
        return 1;
    }
    if (h->host_pin[12] != 0) {
        ctts_set_error("persistent decode layer: a workgroup gave up waiting on edge %d (2 = q|k|v -> attention, 3 = attention -> o_proj, 4 = o_proj -> gate|up, "
                       "5 = gate|up -> down); is the GPU shared with another process?  Use options={'persistent_rows': 0}", h->host_pin[12]);
        return 1;
    }
    return 0;
}

extern "C" int ctts_gpt_progress_enqueue(ctts_gpt* h, int32_t* host_pinned4, void* stream) {
    if (!h || !host_pinned4) { ctts_set_error("progress_enqueue: null argument"); return 1; }
    CTTS_HIP_CHECK(hipMemcpyAsync(host_pinned4, h->st, 16, hipMemcpyDeviceToHost, (hipStream_t)stream));
    return 0;
}

extern "C" int ctts_gpt_saturations(ctts_gpt* h, int32_t* count, void* stream) {
    if (!h || !count || !h->sat) { ctts_set_error("saturations: bad argument"); return 1; }
    CTTS_HIP_CHECK(hipMemcpyAsync(h->host_pin + 8, h->sat, 4, hipMemcpyDeviceToHost, (hipStream_t)stream));
    CTTS_HIP_CHECK(hipStreamSynchronize((hipStream_t)stream));
    *count = h->host_pin[8];
    return 0;
}

extern "C" int ctts_gpt_rows_enqueue(ctts_gpt* h, int32_t* host_pinned_2B, void* stream) {
    if (!h || !host_pinned_2B || h->B == 0) { ctts_set_error("rows_enqueue: call begin first"); return 1; }
    CTTS_HIP_CHECK(hipMemcpy2DAsync(host_pinned_2B, 8, h->finend, sizeof(RowState), 8, h->B, hipMemcpyDeviceToHost, (hipStream_t)stream));
    return 0;
}

// per-utterance adapters: the engine keeps the LoRA launches only while a live row carries an adapter
static void lora_refresh(ctts_gpt* h) {
    bool any = false;
    for (int r = 0; r < h->B; ++r) any = any || h->lora_row_slots[r] >= 0;
    h->lora_rows = any ? 1 : 0;
}
extern "C" int ctts_gpt_compact(ctts_gpt* h, const int32_t* keep_rows, int n_keep, void* stream) {
    if (!h || !keep_rows || h->B == 0) { ctts_set_error("compact: call begin first"); return 1; }
    if (n_keep < 1 || n_keep > h->B) { ctts_set_error("compact: n_keep=%d of %d rows", n_keep, h->B); return 1; }
    for (int i = 0; i < n_keep; ++i)
        if (keep_rows[i] < 0 || keep_rows[i] >= h->B || (i > 0 && keep_rows[i] <= keep_rows[i - 1])) { ctts_set_error("compact: keep_rows must be ascending row indices < %d", h->B); return 1; }
    if (n_keep == h->B) return 0;
    hipStream_t s = (hipStream_t)stream;
    h->keep_host.assign(keep_rows, keep_rows + n_keep);
    CTTS_HIP_CHECK(hipMemcpyAsync(h->keep_dev, h->keep_host.data(), (size_t)n_keep * 4, hipMemcpyHostToDevice, s));
    if (launch_compact_rows(h->keep_dev, n_keep, h->H, h->x_dec, h->rope_dec, h->meta_dec, h->hist_ring, h->finend, h->cx, h->crope, h->cmeta, h->cring, h->cfin, h->st, s)) return 1;
    for (int i = 0; i < n_keep; ++i) { h->row_seq[i] = h->row_seq[keep_rows[i]]; h->row_ctx[i] = h->row_ctx[keep_rows[i]]; h->row_cap[i] = h->row_cap[keep_rows[i]]; }
    if (!h->lora_slot_host.empty()) for (int i = 0; i < n_keep; ++i) h->lora_row_slots[i] = (signed char)h->lora_slot_host[h->row_seq[i]];
    h->B = n_keep;
    if (h->lora_rows) lora_refresh(h);
    return 0;
}


// Adapter slots of the utterances the NEXT ctts_gpt_admit call seats in `rows` (-1 = none); rows not named keep theirs.  The slot follows the row's KV
// lane (the prompt pass of the admitted rows looks it up by sequence, lora.hip) and the row itself (the decode launches carry the rows' slots in their
// arguments, lora_worker.h).  When no live row has an adapter any more the engine drops back to the plain launches.
extern "C" int ctts_gpt_admit_adapters(ctts_gpt* h, int n, const int32_t* rows, const int32_t* slots, void* stream) {
    if (!h || h->B == 0 || !rows || !slots) { ctts_set_error("admit_adapters: call begin first / null argument"); return 1; }
    if (h->lora_slot_host.empty()) {
        h->lora_slot_host.assign(CTTS_MAX_B, -1);
        for (int b = 0; b < CTTS_MAX_B; ++b) h->lora_row_slots[b] = -1;
    }
    bool any = false;
    for (int i = 0; i < n; ++i) {
        if (rows[i] < 0 || rows[i] >= h->B || slots[i] >= CTTS_MAX_ADAPTERS) { ctts_set_error("admit_adapters: row %d / slot %d out of range", rows[i], slots[i]); return 1; }
        any = any || slots[i] >= 0;
    }
    if (any && lora_ensure_storage(h)) return 1;
    for (int i = 0; i < n; ++i) {
        const int sl = slots[i] < 0 ? -1 : slots[i];
        h->lora_slot_host[h->row_seq[rows[i]]] = sl;
        h->lora_row_slots[rows[i]] = (signed char)sl;
    }
    if (h->lora_slot_of_seq)        // (pageable source: staged before the call returns)
        CTTS_HIP_CHECK(hipMemcpyAsync(h->lora_slot_of_seq, h->lora_slot_host.data(), CTTS_MAX_B * 4, hipMemcpyHostToDevice, (hipStream_t)stream));
    lora_refresh(h);
    return 0;
}

// Continuous batching (no counterpart in the reference, whose slices run to their slowest row, pipeline:391-397 / gpt.py:527-546): `n` new
// utterances take over decode rows whose utterance has finished.  Their prompts (all tokens but the last) go through an ordinary prompt pass
// into the rows' KV lanes, starting at slot 0 of the lane; the last prompt token becomes the row's next decode input, so the next decode
// step produces the utterance's first token together with everybody else's next one.  The row's step counter, noise stream (utterance id,
// attempt), token limit and output index are its own (RowState), so an utterance's tokens do not depend on when or where it was admitted.
extern "C" int ctts_gpt_admit(ctts_gpt* h, int n, const int32_t* rows, int T, const int32_t* mask, const float* emb, const uint64_t* utt_ids,
                              const int32_t* row_limits, const int32_t* out_index, const int32_t* attempts, void* stream) {
    if (!h || h->B == 0) { ctts_set_error("admit: call begin first"); return 1; }
    if (!rows || !mask || !emb || !utt_ids || !out_index) { ctts_set_error("admit: null argument"); return 1; }
    if (h->io.noise != nullptr) { ctts_set_error("admit: device noise only (caller-supplied noise is indexed by the batch's draw counter)"); return 1; }
    if (n < 1 || n > h->B || T < 1 || T + h->sc.max_new > h->cfg.max_seq || (long long)n * (T - 1) > h->pass_rows) {
        ctts_set_error("admit: n=%d of %d rows, T=%d (max_seq %d, %d prompt rows per pass)", n, h->B, T, h->cfg.max_seq, h->pass_rows);
        return 1;
    }
    CTTS_RANGE("ctts_gpt_admit");
    hipStream_t s = (hipStream_t)stream;
    h->keep_host.assign(rows, rows + n);
    std::vector<int>& seqs = h->seq_host;
    seqs.assign(n, 0);
    h->fresh_host.assign(n, RowState{});
    for (int i = 0; i < n; ++i) {
        if (rows[i] < 0 || rows[i] >= h->B) { ctts_set_error("admit: row %d of %d", rows[i], h->B); return 1; }
        for (int j = 0; j < i; ++j) if (rows[j] == rows[i]) { ctts_set_error("admit: row %d named twice", rows[i]); return 1; }
        seqs[i] = h->row_seq[rows[i]];
        RowState& r = h->fresh_host[i];
        int lim = row_limits ? row_limits[i] : h->sc.max_new;
        r.limit = lim < 1 ? 1 : (lim > h->sc.max_new ? h->sc.max_new : lim);
        r.uid_lo = (unsigned)utt_ids[i]; r.uid_hi = (unsigned)(utt_ids[i] >> 32);
        r.attempt = attempts ? attempts[i] : 0;
        r.out = out_index[i];
        if (r.out < 0) { ctts_set_error("admit: negative output index"); return 1; }
    }
    int* rows_dev = h->keep_dev;
    int* seqs_dev = h->cring;                    // (compaction scratch: consumed in stream order)
    RowState* fresh_dev = h->cfin;
    CTTS_HIP_CHECK(hipMemcpyAsync(rows_dev, h->keep_host.data(), (size_t)n * 4, hipMemcpyHostToDevice, s));
    CTTS_HIP_CHECK(hipMemcpyAsync(seqs_dev, seqs.data(), (size_t)n * 4, hipMemcpyHostToDevice, s));
    CTTS_HIP_CHECK(hipMemcpyAsync(fresh_dev, h->fresh_host.data(), (size_t)n * sizeof(RowState), hipMemcpyHostToDevice, s));
    AdmitArgs a = {};
    a.mask = mask; a.emb = emb; a.rows = rows_dev; a.seqs = seqs_dev; a.fresh = fresh_dev; a.n = n; a.T = T; a.H = h->H;
    a.pm = h->meta_pre; a.rope_pre = h->rope_pre; a.dm = h->meta_dec; a.rope_dec = h->rope_dec; a.x_dec = h->x_dec; a.ring = h->hist_ring; a.finend = h->finend;
    a.rope = h->rope; a.st = h->st; a.finish = h->io.finish; a.end_idx = h->io.end_idx;
    if (launch_admit_rows(a, s)) return 1;
    const int R = n * (T - 1);
    if (R > 0) {
        // prompt rows [n][T-1] <- emb[i][0 .. T-2]
        CTTS_HIP_CHECK(hipMemcpy2DAsync(h->x_pre, (size_t)(T - 1) * h->H * 4, emb, (size_t)T * h->H * 4, (size_t)(T - 1) * h->H * 4, n, hipMemcpyDeviceToDevice, s));
        const int keepT = h->pre_T;
        h->pre_T = T - 1;
        const int rc = run_layers(h, h->x_pre, h->meta_pre, h->rope_pre, R, 1, nullptr, s);
        h->pre_T = keepT;
        if (rc) return 1;
    }
    for (int i = 0; i < n; ++i) { h->row_ctx[rows[i]] = T; h->row_cap[rows[i]] = T + h->fresh_host[i].limit; }
    h->admitted = true;
    return 0;
}

// Diagnostics: copies a named internal buffer to HOST memory (tools/persist_probe.py compares the persistent layer's intermediates with the
// launch path's).  Synchronises the stream.  Returns the number of bytes copied through *bytes.
extern "C" int ctts_gpt_debug_read(ctts_gpt* h, const char* name, void* out, size_t max_bytes, size_t* bytes, void* stream) {
    if (!h || !name || !out) { ctts_set_error("debug_read: null argument"); return 1; }
    const std::string n(name);
    const void* src = nullptr; size_t nb = 0;
    if (n == "x_dec") { src = h->x_dec; nb = (size_t)CTTS_MAX_B * h->H * 4; }
    else if (n == "q_buf") { src = h->q_buf; nb = (size_t)32 * h->H * 4; }
    else if (n == "logits") { src = h->logits; nb = (size_t)CTTS_MAX_B * h->NVQ * h->V * 4; }
    else if (n == "pl_g" && h->pl_g) { src = h->pl_g; nb = (size_t)PL_G_TOTAL * 8; }
    else if (n == "pl_ts" && h->pl_ts) { src = h->pl_ts; nb = (size_t)PL_BLOCKS * 10 * 8; }
    else if (n == "xh") { src = h->xh; nb = (size_t)2 * 48 * 1024; }
    else if (n == "ssq") { src = h->ssq; nb = (size_t)32 * 48 * 4; }
    else if (n == "pl_state" && h->pl_epoch) {
        unsigned* o = (unsigned*)out;
        if (max_bytes < 8) { ctts_set_error("debug_read: buffer too small"); return 1; }
        CTTS_HIP_CHECK(hipStreamSynchronize((hipStream_t)stream));
        CTTS_HIP_CHECK(hipMemcpy(o, h->pl_epoch, 4, hipMemcpyDeviceToHost));
        CTTS_HIP_CHECK(hipMemcpy(o + 1, h->pl_error, 4, hipMemcpyDeviceToHost));
        if (bytes) *bytes = 8;
        return 0;
    }
    if (!src) { ctts_set_error("debug_read: unknown or unallocated buffer '%s'", name); return 1; }
    if (nb > max_bytes) nb = max_bytes;
    CTTS_HIP_CHECK(hipStreamSynchronize((hipStream_t)stream));
    CTTS_HIP_CHECK(hipMemcpy(out, src, nb, hipMemcpyDeviceToHost));
    if (bytes) *bytes = nb;
    return 0;
}

extern "C" int ctts_gpt_logits(ctts_gpt* h, float* out, void* stream) {
    if (!h || !out || h->B == 0) { ctts_set_error("logits: call begin first"); return 1; }
    CTTS_HIP_CHECK(hipMemcpyAsync(out, h->logits, (size_t)h->B * (h->text_mode ? h->vocab_text_head : h->NVQ * h->V) * 4, hipMemcpyDeviceToDevice, (hipStream_t)stream));
    return 0;
}

extern "C" int ctts_gpt_force_ids(ctts_gpt* h, const int32_t* ids, void* stream) {
    if (!h || !ids || h->B == 0) { ctts_set_error("force_ids: call begin first"); return 1; }
    return launch_embed_ids(ids, h->emb_code, h->x_dec, h->B, h->V, h->H, (hipStream_t)stream);
}

extern "C" int ctts_sampler_run(const ctts_sampler_cfg* sc, const float* logits, const int32_t* history, int hist_len, const float* q,
                                int rows, int vocab, int step, int32_t* idx, void* stream) {
    if (!sc || !logits || !q || !idx || (hist_len > 0 && !history)) { ctts_set_error("sampler_run: null argument"); return 1; }
    // the kernels read their configuration from device memory: a small ring of slots, one per call in flight
    static std::mutex mu;
    static SamplerDyn* ring = nullptr;
    static unsigned next = 0;
    const unsigned NSLOT = 64;
    SamplerDyn* slot;
    {
        std::lock_guard<std::mutex> lk(mu);
        if (!ring) CTTS_HIP_CHECK(hipMalloc((void**)&ring, NSLOT * sizeof(SamplerDyn)));
        slot = ring + (next++ % NSLOT);
    }
    SamplerDyn d = {};
    memcpy(d.cfg.temperature, sc->temperature, sizeof(sc->temperature));
    d.cfg.top_p_threshold = sc->top_p_threshold; d.cfg.top_k = sc->top_k; d.cfg.min_keep = sc->min_tokens_to_keep;
    d.cfg.use_penalty = sc->use_penalty; memcpy(d.cfg.penalty_table, sc->penalty_table, sizeof(sc->penalty_table));
    d.cfg.past_window = sc->past_window; d.cfg.max_input_ids = sc->max_input_ids; d.cfg.eos = sc->eos_token;
    d.cfg.min_new = sc->min_new_token; d.cfg.max_new = sc->max_new_token;
    d.noise = q;
    CTTS_HIP_CHECK(hipMemcpyAsync(slot, &d, sizeof(d), hipMemcpyHostToDevice, (hipStream_t)stream));
    SamplerArgs sa = {};
    sa.dyn = slot;
    sa.logits = logits; sa.V = vocab; sa.B = rows; sa.st = nullptr; sa.history = history; sa.hist_len = hist_len;
    sa.step_override = step; sa.idx_out = idx;
    return launch_sampler(sa, (rows + 3) / 4, (hipStream_t)stream);
}

extern "C" int ctts_gpt_time_decode(ctts_gpt* h, int n_steps, float* ms_per_step, void* stream) {
    if (!h || h->B == 0 || n_steps < 1 || !ms_per_step) { ctts_set_error("time_decode: bad argument"); return 1; }
    hipStream_t s = (hipStream_t)stream;
    // same chunking as generate(): 32 steps at a time, each chunk with the key-split count its context length asks for;
    // the graphs of every split count the run will need are captured before the clock starts
    n_steps = (n_steps + h->graph_steps - 1) / h->graph_steps * h->graph_steps;
    const int CH = 32 / h->graph_steps * h->graph_steps > 0 ? 32 / h->graph_steps * h->graph_steps : h->graph_steps;
    for (int pass = 0; pass < 2; ++pass) {
        int launched = h->launched;
        if (pass == 1) CTTS_HIP_CHECK(hipEventRecord(h->ev0, s));
        for (int i = 0; i < n_steps; i += CH) {
            const int n = (n_steps - i < CH) ? n_steps - i : CH;
            int longest = 1;
            for (int r = 0; r < h->B; ++r) { const int c = std::min(h->row_ctx[r] + (launched - h->launched) + n, h->row_cap[r]); if (c > longest) longest = c; }
            if (pick_decode_path(h, longest + 1)) return 1;
            launched += n;
            for (int left = n; left > 0;) {              // (n is a multiple of graph_steps)
                const int span = graph_span(h, left);
                if (ensure_graph(h, span)) return 1;
                for (; left >= span; left -= span) if (pass == 1) CTTS_HIP_CHECK(hipGraphLaunch(h->gexec, s));
            }
        }
        if (pass == 1) { (void)advance_rows(h, launched - h->launched); h->launched = launched; }
    }
    CTTS_HIP_CHECK(hipEventRecord(h->ev1, s));
    CTTS_HIP_CHECK(hipEventSynchronize(h->ev1));
    float ms = 0.f;
    CTTS_HIP_CHECK(hipEventElapsedTime(&ms, h->ev0, h->ev1));
    *ms_per_step = ms / n_steps;
    return 0;
}

extern "C" double ctts_gpt_step_bytes(const ctts_gpt* h, int B, double mean_ctx) {
    // s * [ W + B*(L+1)*KV_tok ],  W = layers*(4H^2 + 3HI) + 4*V*H + (2*layers+1)*H
    const double W = (double)h->L * (4.0 * h->H * h->H + 3.0 * h->H * h->I) + (double)h->NVQ * h->V * h->H + (2.0 * h->L + 1) * h->H;
    const double kv_tok = (double)h->L * 2 * h->H;
    return h->esz * (W + B * (mean_ctx + 1.0) * kv_tok);
}
