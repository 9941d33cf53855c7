// Persistent MFMA decode stack (persist_mfma.hip): geometry and the argument block shared with gpt_engine.hip.
#pragma once
#include "common.h"

#define PM_BLOCKS 256                      // one workgroup per CU, all resident
#define PM_GEMM_WAVES 8                    // waves 0..7: weight tiles in registers, MFMA, epilogues
#define PM_ATT_WAVES 4                     // waves 8..11: attention items; wave 8 also polls and publishes
#define PM_THREADS ((PM_GEMM_WAVES + PM_ATT_WAVES) * 64)
#define PM_MAXR 32                         // decode rows served: one or two 16-row chunks
#define PM_MINR 5
#define PM_H 768
#define PM_I 3072
#define PM_NH 12
#define PM_QKV_TILES 144                   // 16-row weight tiles of q | k | v
#define PM_O_TILES 48
#define PM_GU_TILES 384                    // [8 gate rows | 8 up rows] tiles
#define PM_D_ITEMS 192                     // 48 down tiles x 4 K slices
#define PM_NPHASE 5                        // flag arrays: q|k|v, attention, o_proj, gate|up, down
#define PM_NTS 16                          // diagnostics: wall_clock64 marks per workgroup

struct PmArgs {
    const char* wqkv;               // layer 0's packed MFMA-A tile images ([row tile][k tile][lane][16 B], gpt_engine.hip pack_tiles); layer l at + l * w_stride
    const char* wo;
    const char* wgu;
    const char* wd;
    size_t w_stride;                // bytes between two layers' images
    int n_layers;                   // <= 31
    int R;                          // decode rows, PM_MINR..PM_MAXR
    float* x;                       // residual stream [R][768]: read at entry (the sampler's rows), rewritten by every o_proj / down projection
    const RowMeta* meta;
    const float* rope_rows;         // [R][64] cos | sin of each row's position
    void* kv;                       // KV cache [layer][K | V][maxB][12][Lmax][64] fp32
    size_t kv_per;                  //   floats per [maxB][12][Lmax][64] block (kv_per * 4 < 4 GB: buffer offsets are 32-bit)
    int Lmax;
    float* q_buf;                   // [R][12][64] queries after RoPE
    float* attn_packed;             // [chunk][48 k-tiles][64 lanes][4]: normalised attention output, fragment-major (common.h xfrag_index)
    float* xh;                      // [chunk][48][64][4]: the residual stream as the next projection's B operand (unscaled: fp32)
    float* ssq;                     // [R][48] sums of squares of the residual rows over each 16-column tile
    float* act;                     // [chunk][192][64][4]: silu(gate) * up
    float* slab;                    // [48 tiles][4 slices][2 chunks][256] split-K partial tiles of the down projection
    int* cnt;                       // [48] arrival tickets of the down projection's slices (zero between launches)
    unsigned* flags;                // [PM_NPHASE][256] "this item / workgroup has published" words, value = tag of (launch, layer, phase)
    unsigned* epoch;                // launch counter (the launch itself advances it: graph replay freezes kernel arguments)
    int* error;                     // 0, or the code of the first wait that gave up (shared with the <= 4-row persistent launch)
    const int* done;                // DevState.all_done
    unsigned long long* ts;         // diagnostics: [256][PM_NTS] wall_clock64 marks (last layer), or null
    float eps;
    int nap;                        // ~128-cycle units between two poll passes
    int delay[PM_NPHASE];           // ~128-cycle units the poller sleeps before its first pass at the wait in front of phase p (the data cannot be there yet)
    int fault;                      // test hook: > 0 = workgroup 5 withholds its q|k|v flag in layer fault - 1
};

int launch_persist_mfma(const PmArgs& a, hipStream_t s);
int persist_mfma_configure();
size_t persist_mfma_slab_floats();
