// Microbenchmark: what does one dependent weight-streaming kernel cost inside a 100-launch hipGraph on MI355X?
// Build: timeout 120 hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/mb/microbench.hip -o tools/mb/microbench
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ void k_empty(const f32x4*, float*, const float*) {}

template <int NL, bool NT>
__global__ void k_stream(const f32x4* W, float* out, const float* x) {
    const int tid = threadIdx.x;
    const f32x4* wp = W + ((size_t)blockIdx.x * blockDim.x + tid) * NL;
    f32x4 w[NL];
#pragma unroll
    for (int i = 0; i < NL; ++i) w[i] = NT ? __builtin_nontemporal_load(wp + i) : wp[i];
    const f32x4 xv = *(const f32x4*)(x + 4 * (tid & 127));            // produced by the previous launch
    float acc = xv[0] + xv[1] + xv[2] + xv[3];
#pragma unroll
    for (int i = 0; i < NL; ++i) acc += w[i][0] * w[i][1] + w[i][2] * w[i][3];
    out[(size_t)blockIdx.x * blockDim.x + tid] = acc;                  // every thread's loads are live
}

template <int NL, bool NT>
__global__ void k_stream_lds(const f32x4* W, float* out, const float* x) {
    __shared__ float lds[1024];
    const int tid = threadIdx.x;
    const f32x4* wp = W + ((size_t)blockIdx.x * blockDim.x + tid) * NL;
    f32x4 w[NL];
#pragma unroll
    for (int i = 0; i < NL; ++i) w[i] = NT ? __builtin_nontemporal_load(wp + i) : wp[i];
    const f32x4 xv = *(const f32x4*)(x + 4 * (tid & 127));
    lds[tid] = xv[0] + xv[1] + xv[2] + xv[3];
    __syncthreads();
    float acc = lds[tid ^ 64];
#pragma unroll
    for (int i = 0; i < NL; ++i) acc += w[i][0] * w[i][1] + w[i][2] * w[i][3];
    __syncthreads();
    lds[tid] = acc;
    __syncthreads();
    out[(size_t)blockIdx.x * blockDim.x + tid] = acc + lds[tid ^ 1];
}

// cache-policy variants through inline asm: POL 0 plain, 1 nt, 2 sc1, 3 sc0 sc1, 4 sc1 nt, 5 sc0
template <int POL>
__device__ inline f32x4 ld16(const f32x4* p) {
    f32x4 v;
    if (POL == 0) asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(v) : "v"(p) : "memory");
    if (POL == 1) asm volatile("global_load_dwordx4 %0, %1, off nt" : "=v"(v) : "v"(p) : "memory");
    if (POL == 2) asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(v) : "v"(p) : "memory");
    if (POL == 3) asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1" : "=v"(v) : "v"(p) : "memory");
    if (POL == 4) asm volatile("global_load_dwordx4 %0, %1, off sc1 nt" : "=v"(v) : "v"(p) : "memory");
    if (POL == 5) asm volatile("global_load_dwordx4 %0, %1, off sc0" : "=v"(v) : "v"(p) : "memory");
    return v;
}
template <int NL, int POL>
__global__ void k_pol(const f32x4* W, float* out, const float* x) {
    const int tid = threadIdx.x;
    const f32x4* wp = W + ((size_t)blockIdx.x * blockDim.x + tid) * NL;
    f32x4 w[NL];
#pragma unroll
    for (int i = 0; i < NL; ++i) w[i] = ld16<POL>(wp + i);
    const f32x4 xv = *(const f32x4*)(x + 4 * (tid & 127));
    float acc = xv[0] + xv[1] + xv[2] + xv[3];
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int i = 0; i < NL; ++i) { asm volatile("" : "+v"(w[i])); acc += w[i][0] * w[i][1] + w[i][2] * w[i][3]; }
    out[(size_t)blockIdx.x * blockDim.x + tid] = acc;
}

typedef void (*kern_t)(const f32x4*, float*, const float*);

static int run(const char* name, kern_t k, int blocks, int threads, const f32x4* W, size_t w_per_launch, int nbuf, float* x, int reps) {
    hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
    for (int i = 0; i < 100; ++i)
        hipLaunchKernelGGL(k, dim3(blocks), dim3(threads), 0, s, W + (size_t)(i % nbuf) * w_per_launch, x + (size_t)((i + 1) & 1) * (1 << 20), x + (size_t)(i & 1) * (1 << 20));
    CK(hipStreamEndCapture(s, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) CK(hipGraphLaunch(ge, s));
    CK(hipStreamSynchronize(s));
    CK(hipEventRecord(e0, s));
    for (int i = 0; i < reps; ++i) CK(hipGraphLaunch(ge, s));
    CK(hipEventRecord(e1, s));
    CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("%-46s grid %4d x %4d : %7.3f us per kernel\n", name, blocks, threads, ms * 1e3 / (reps * 100));
    fflush(stdout);
    return 0;
}

int main() {
    const size_t wbytes = (size_t)400 << 20;           // 400 MB of "weights" > MALL (256 MB)
    f32x4* W; float* x;
    CK(hipMalloc(&W, wbytes));
    {   // random-ish contents (not a constant fill)
        unsigned* hbuf = (unsigned*)malloc(wbytes); unsigned v = 12345u;
        for (size_t i = 0; i < wbytes / 4; ++i) { v = v * 1664525u + 1013904223u; hbuf[i] = (v >> 9) | 0x3F800000u; }
        CK(hipMemcpy(W, hbuf, wbytes, hipMemcpyHostToDevice)); free(hbuf);
    }
    CK(hipMalloc(&x, (size_t)2 * (1 << 20) * 4)); CK(hipMemset(x, 0, (size_t)2 * (1 << 20) * 4));
    const int reps = 30;
    run("empty", k_empty, 144, 256, W, 0, 1, x, reps);
    printf("--- 3.5 MB per launch (QKV-like), 100 different weight blocks (350 MB cycle)\n");
    run("nt    144x256x6", k_stream<6, true>, 144, 256, W, 144 * 256 * 6, 100, x, reps);
    run("plain 144x256x6", k_stream<6, false>, 144, 256, W, 144 * 256 * 6, 100, x, reps);
    run("nt    144x256x6 + lds/3 barriers", k_stream_lds<6, true>, 144, 256, W, 144 * 256 * 6, 100, x, reps);
    run("plain 144x256x6 + lds/3 barriers", k_stream_lds<6, false>, 144, 256, W, 144 * 256 * 6, 100, x, reps);
    run("plain 288x128x6", k_stream<6, false>, 288, 128, W, 288 * 128 * 6, 100, x, reps);
    run("plain 576x64x6", k_stream<6, false>, 576, 64, W, 576 * 64 * 6, 100, x, reps);
    run("plain 288x256x3", k_stream<3, false>, 288, 256, W, 288 * 256 * 3, 100, x, reps);
    run("plain 576x128x3", k_stream<3, false>, 576, 128, W, 576 * 128 * 3, 100, x, reps);
    run("plain 72x256x12", k_stream<12, false>, 72, 256, W, 72 * 256 * 12, 100, x, reps);
    run("nt    72x256x12", k_stream<12, true>, 72, 256, W, 72 * 256 * 12, 100, x, reps);
    run("plain 144x256x6 same weights each launch", k_stream<6, false>, 144, 256, W, 144 * 256 * 6, 1, x, reps);
    run("nt    288x128x6", k_stream<6, true>, 288, 128, W, 288 * 128 * 6, 100, x, reps);
    run("nt    576x64x6", k_stream<6, true>, 576, 64, W, 576 * 64 * 6, 100, x, reps);
    run("nt    288x256x3", k_stream<3, true>, 288, 256, W, 288 * 256 * 3, 100, x, reps);
    run("nt    576x256x... 576x128x3", k_stream<3, true>, 576, 128, W, 576 * 128 * 3, 100, x, reps);
    run("asm plain        144x256x6", k_pol<6, 0>, 144, 256, W, 144 * 256 * 6, 100, x, reps);
    run("asm nt           144x256x6", k_pol<6, 1>, 144, 256, W, 144 * 256 * 6, 100, x, reps);
    run("asm sc1          144x256x6", k_pol<6, 2>, 144, 256, W, 144 * 256 * 6, 100, x, reps);
    run("asm sc0 sc1      144x256x6", k_pol<6, 3>, 144, 256, W, 144 * 256 * 6, 100, x, reps);
    run("asm sc1 nt       144x256x6", k_pol<6, 4>, 144, 256, W, 144 * 256 * 6, 100, x, reps);
    run("asm sc0          144x256x6", k_pol<6, 5>, 144, 256, W, 144 * 256 * 6, 100, x, reps);
    printf("--- 3.5 MB per launch, 110 different blocks of a 385 MB cycle (engine-like footprint), plain vs nt\n");
    run("plain 144x256x6, 385 MB cycle", k_stream<6, false>, 144, 256, W, 144 * 256 * 6, 110, x, reps);
    run("nt    144x256x6, 385 MB cycle", k_stream<6, true>, 144, 256, W, 144 * 256 * 6, 110, x, reps);
    printf("--- 9.4 MB per launch (gate|up-like), 40 blocks\n");
    run("nt    384x256x6", k_stream<6, true>, 384, 256, W, 384 * 256 * 6, 40, x, reps);
    run("plain 384x256x6", k_stream<6, false>, 384, 256, W, 384 * 256 * 6, 40, x, reps);
    run("plain 768x128x6", k_stream<6, false>, 768, 128, W, 768 * 128 * 6, 40, x, reps);
    run("plain 768x256x3", k_stream<3, false>, 768, 256, W, 768 * 256 * 3, 40, x, reps);
    run("plain 1536x64x6", k_stream<6, false>, 1536, 64, W, 1536 * 64 * 6, 40, x, reps);
    run("plain 192x512x6", k_stream<6, false>, 192, 512, W, 192 * 512 * 6, 40, x, reps);
    printf("--- 4.7 MB per launch (down-like), 80 blocks\n");
    run("nt    48x1024x6", k_stream<6, true>, 48, 1024, W, 48 * 1024 * 6, 80, x, reps);
    run("plain 48x1024x6", k_stream<6, false>, 48, 1024, W, 48 * 1024 * 6, 80, x, reps);
    run("plain 48x1024x6 + lds/3 barriers", k_stream_lds<6, false>, 48, 1024, W, 48 * 1024 * 6, 80, x, reps);
    run("plain 192x256x6", k_stream<6, false>, 192, 256, W, 192 * 256 * 6, 80, x, reps);
    run("plain 384x128x6", k_stream<6, false>, 384, 128, W, 384 * 128 * 6, 80, x, reps);
    run("plain 96x512x6", k_stream<6, false>, 96, 512, W, 96 * 512 * 6, 80, x, reps);
    printf("--- 1.2 MB per launch (o_proj-like)\n");
    run("plain 48x256x6", k_stream<6, false>, 48, 256, W, 48 * 256 * 6, 100, x, reps);
    run("plain 192x64x6", k_stream<6, false>, 192, 64, W, 192 * 64 * 6, 100, x, reps);
    run("plain 96x256x3", k_stream<3, false>, 96, 256, W, 96 * 256 * 3, 100, x, reps);
    return 0;
}
