// Microbenchmark: does a cyclic weight stream that fits the 256 MB Infinity Cache (MALL) get served from it?
// Chain of N dependent kernels, each block streams 24 KB (coalesced 16 B per lane) + reads a vector the previous kernel wrote;
// the whole chain (N x 4.7 MB) is replayed many times.  Compare non-temporal vs cacheable loads for a working set of
// 189 MB (fits) and 472 MB (does not).
// Build: timeout 300 hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/mb/mall.hip -o tools/mb/mall
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s (line %d)\n", #x, hipGetErrorString(e), __LINE__); return 1; } } while (0)
constexpr int NL = 6;

template <int POL>
__device__ inline f32x4 ld(const f32x4* p) {
    if (POL == 0) return *p;
    if (POL == 1) return __builtin_nontemporal_load(p);
    f32x4 v;
    if (POL == 2) asm volatile("global_load_dwordx4 %0, %1, off sc1\n s_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    if (POL == 3) asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1\n s_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    return v;
}
template <int POL>
__global__ __launch_bounds__(256) void k_chain(const f32x4* W, const float* vin, float* vout) {
    const int tid = threadIdx.x;
    const f32x4* wp = W + (size_t)blockIdx.x * 256 * NL + tid;
    f32x4 w[NL];
#pragma unroll
    for (int i = 0; i < NL; ++i) w[i] = ld<POL>(wp + i * 256);
    const f32x4 xv = *(const f32x4*)(vin + 4 * (tid & 127));
    float acc = xv[0] + xv[1] + xv[2] + xv[3];
#pragma unroll
    for (int i = 0; i < NL; ++i) acc += w[i][0] * w[i][1] + w[i][2] * w[i][3];
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
    if ((tid & 63) == 0) vout[(blockIdx.x * 4 + (tid >> 6)) % 768] = acc * 1e-6f;
}

template <int POL>
static int run(const f32x4* W, float* v0, float* v1, int N, int G, size_t per, hipStream_t s0, const char* name) {
    hipGraph_t g; hipGraphExec_t ge; hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipStreamBeginCapture(s0, hipStreamCaptureModeThreadLocal));
    for (int i = 0; i < N; ++i) hipLaunchKernelGGL((k_chain<POL>), dim3(G), dim3(256), 0, s0, W + per * i, (i & 1) ? v1 : v0, (i & 1) ? v0 : v1);
    CK(hipStreamEndCapture(s0, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipEventRecord(e0, s0));
        for (int k = 0; k < 30; ++k) CK(hipGraphLaunch(ge, s0));
        CK(hipEventRecord(e1, s0)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (rep == 2) printf("  %-22s %.3f us per dependent launch (%.0f GB/s)\n", name, ms * 1e3f / (30 * N), (double)per * 16 * N * 30 / (ms * 1e-3) / 1e9);
    }
    CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
    return 0;
}

int main() {
    const int G = 192;
    const size_t per = (size_t)G * 256 * NL;
    f32x4* W; float *v0, *v1;
    CK(hipMalloc(&W, per * 100 * sizeof(f32x4))); CK(hipMemset(W, 0, per * 100 * sizeof(f32x4)));
    CK(hipMalloc(&v0, 4096)); CK(hipMalloc(&v1, 4096)); CK(hipMemset(v0, 0, 4096)); CK(hipMemset(v1, 0, 4096));
    hipStream_t s0; CK(hipStreamCreateWithFlags(&s0, hipStreamNonBlocking));
    for (int N : {8, 40, 100}) {
        printf("working set %.0f MB (%d launches x 4.7 MB):\n", (double)per * 16 * N / 1e6, N);
        if (run<1>(W, v0, v1, N, G, per, s0, "non-temporal")) return 1;
        if (run<0>(W, v0, v1, N, G, per, s0, "cacheable")) return 1;
        if (run<2>(W, v0, v1, N, G, per, s0, "sc1")) return 1;
        if (run<3>(W, v0, v1, N, G, per, s0, "sc0 sc1")) return 1;
    }
    return 0;
}
