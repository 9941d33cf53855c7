// Persistent decode layer (persist_layer.hip): geometry and the argument block shared with gpt_engine.hip.
#pragma once
#include "common.h"

#define PL_GEMV_BLOCKS 192                 // workgroups that own weight slices
#define PL_ATT_BLOCKS 64                   // workgroups that own one (row, head) of the attention
#define PL_BLOCKS (PL_GEMV_BLOCKS + PL_ATT_BLOCKS)
#ifndef PL_EW_ONE_ROW
#define PL_EW_ONE_ROW 4                    // edge waves per workgroup at ONE row (A/B builds: tools/build_variant.sh ew2 -DPL_EW_ONE_ROW=2).  ms/step at batch 1, same box,
#endif                                     //   2 / 4 edge waves: 0.2781 / 0.2685, 0.2784 / 0.2705 (round 5); at 2-4 rows four waves also removed every spill
#define PL_EDGE_WAVES(R) ((R) == 1 ? PL_EW_ONE_ROW : 4)
#define PL_THREADS(R) ((8 + PL_EDGE_WAVES(R)) * 64)        // 8 compute waves + 4 edge waves (round 4: 2)
#define PL_THREADS_MAX 768
#define PL_H 768
#define PL_I 3072
#define PL_NH 12
#define PL_MAXR_ONE 5                      // 12 heads x 5 rows = 60 of the 64 attention workgroups: up to here one (row, head[, key share]) per attention workgroup
#define PL_MAXR 8                          // 6..8 rows: two (row, head) items per attention workgroup, 4 compute waves + 1 edge wave each (round 6)
#define PL_SHARE_KEYS 384                  // cached keys a workgroup requests before the query exists; a longer share streams behind the query
// per-workgroup weight image of one layer: 12 q|k|v rows, 4 o_proj rows, 16 gate|up pairs, 4 down rows (fp32)
#define PL_QKV_BYTES (12 * 768 * 4)
#define PL_O_BYTES (4 * 768 * 4)
#define PL_GU_BYTES (32 * 768 * 4)
#define PL_D_BYTES (4 * 3072 * 4)
#define PL_BLOCK_BYTES (PL_QKV_BYTES + PL_O_BYTES + PL_GU_BYTES + PL_D_BYTES)      // 196608
#define PL_LAYER_BYTES ((size_t)PL_GEMV_BLOCKS * PL_BLOCK_BYTES)                   // 37.75 MB = the layer's weights, once
// granule buffers (8 bytes = {tag, value}), sized for PL_MAXR rows
#define PL_G_QKV (PL_MAXR * PL_NH * 192)
#define PL_G_ATT (PL_MAXR * PL_H)
#define PL_G_X1 (PL_MAXR * PL_H)
#define PL_G_ACT (PL_MAXR * PL_I)
#define PL_G_X (PL_MAXR * PL_H)
#define PL_SMAX 5                           // key splits per (row, head): 12 rows x heads x splits <= 64 attention workgroups
#define PL_G_PART (PL_ATT_BLOCKS * 66)
// final norm + heads inside the launch (round 5): 14 head rows per GEMV workgroup (192 x 14 = 2688 >= 2504 = 4 x 626), 2 per compute wave 0..6
#define PL_HEAD_ROWS 14
#define PL_HEAD_FRAGS (7 * 2 * 3 * 64)     // fragments (4 weights each) of a workgroup's head image: [wave 7][row 2][j 3][lane 64]
#define PL_G_U (PL_MAXR * 64)                // per-utterance adapters: u = A h of a row's q (16) | k (16) | v (16) | o_proj (16) terms
#define PL_G_TOTAL (PL_G_QKV + PL_G_ATT + PL_G_X1 + PL_G_ACT + PL_G_X + PL_G_PART + PL_G_U)

struct PersistArgs {
    const char* w;                  // layer 0's image [192][PL_BLOCK_BYTES]; layer l at + l * PL_LAYER_BYTES (fp16 engines: half of both)
    const char* hw;                 // heads == 1: the folded heads' image [192][PL_HEAD_FRAGS] fragments (rows >= n_valid are zero)
    int heads;                      // 1: this launch ends the stack and also runs the final RMSNorm + the 4 folded code heads (gpt.py:422-447): logits, hidden row
    float* logits;                  //   [R][n_valid]
    int n_valid;                    //   2504
    const float* lnf;               //   final norm weight [768] (the heads' columns carry it already; the hidden row needs it)
    const SamplerDyn* dyn;          //   hidden_out / hidden_stride (device memory, constant address space)
    const RowState* rows;           //   per-row state: a live row's hidden goes to hiddens[out][end]
    int half_w;                     // 1: fp16 engine -- half weights in the image, half K / V cache; activations and granules stay fp32
    int n_layers;                   // decoder layers run by this launch (<= 31: a granule tag is launch counter * 32 + layer)
    float* x;                       // residual stream [R][768], read at entry, rewritten at the end
    const RowMeta* meta;            // decode rows
    const float* rope_rows;         // [R][64] cos | sin of each row's position
    void* kv;                       // KV cache [layer][K | V][maxB][12][Lmax][64] fp32
    size_t kv_per;                  //   floats per [maxB][12][Lmax][64] block
    int Lmax;
    unsigned long long* g_qkv;      // [R][12][q 64 | k 64 | v 64]
    unsigned long long* g_att;      // [R][768]
    unsigned long long* g_x1;       // [R][768]
    unsigned long long* g_act;      // [R][3072]
    unsigned long long* g_x;        // [R][768] a layer's output on its way to the next layer
    unsigned long long* g_part;     // [(row, head) x S][max | sum | o 64] attention partials of the key splits (S > 1)
    int S;                          // key splits per (row, head), 1..PL_SMAX, 12 R S <= 64
    unsigned* epoch;                // launch counter (never 0)
    int* error;                     // 0, or the edge code of the first wave that gave up
    const int* done;                // DevState.all_done
    unsigned long long* ts;         // diagnostics: [256][10] wall_clock64 marks per workgroup (last layer), or null
    float eps;
    int sched;                      // weight request schedule of the compute waves (persist_layer.hip: 1 or 2)
    int pace;                       // SCHED 3: s_sleep(2) units (~128 cycles each) between two paced weight requests of a compute wave
    int delay_act, delay_x, nap_qkv;   // (x1 edge: delay; act edge: delay_act; layer-output edge: delay_x; poll interval of the attention workgroups' q|k|v sweep)
    int delay_att, delay;           // ~128-cycle units an edge wave sleeps before it starts polling the attention edge / the other edges (the data cannot be there yet)
    int nap;                        // ~128-cycle units between two poll passes of a GEMV edge wave
    int fault;                      // test hook: > 0 = one workgroup withholds a hand-off in layer fault - 1 (the give-up path must end the step, not hang it)
    int poll;                       // bit 0: watch one sentinel granule per producer before the full sweep of an edge
    // per-utterance LoRA adapters inside the launch (round 6; the LORA kernels, paced schedule): row r adds scale * B (A h) of adapter slot lslots[r] to q / k / v / o_proj
    int lora;                       // 1: some decode row carries an adapter
    int delay_u;                    //   ~128-cycle units an edge wave sleeps before it polls the u granules (they are published ~0.4 us after the gather completes)
    unsigned long long lslots;      //   byte r = the adapter slot of decode row r (0xFF: none)
    const float* la_qkv;            //   layer 0's A of q | k | v with the input RMSNorm weight folded into the columns [slot][3][16][768]; layer l at + l * la_qkv_stride floats
    const float* la;                //   layer 0's A [slot][4][16][768] (target 3 = o_proj); layer l at + l * la_stride floats
    const float* lb;                //   B^T, rank-major, same shape and stride as la
    const float* lscale;            //   [layer][slot][4]
    size_t la_qkv_stride, la_stride;
    unsigned long long* g_u;        //   [R][64] granules: u = A h, q 16 | k 16 | v 16 | o_proj 16
};

int launch_persist_layer(int R, const PersistArgs& a, hipStream_t s);
int launch_persist_repack_heads(int half_w, const void* whead, int n_tiles, void* dst, hipStream_t s);
int launch_persist_repack(int half_w, const void* qkv, const void* o, const void* gu, const void* d, void* dst, hipStream_t s);
int persist_configure();
