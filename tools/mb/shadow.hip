// Microbenchmark: can a "shadow" launch on a second graph branch pull the NEXT dependent kernel's weight tile into the
// XCD-local L2 while the current kernel runs, so that the dependent chain sees L2 instead of HBM latency?
// Chain of N dependent kernels (each block streams 24 KB of its own weights + reads a vector the previous kernel wrote).
//   A: chain only, non-temporal weight loads (what the decode step does today)
//   B: chain only, cacheable weight loads
//   C: chain (cacheable) + shadow branch, shadow_i after chain_{i-2}            (no edge shadow_i -> chain_i)
//   D: like C plus the edge shadow_i -> chain_i
// Build: timeout 300 hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/mb/shadow.hip -o tools/mb/shadow
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s (line %d)\n", #x, hipGetErrorString(e), __LINE__); return 1; } } while (0)
constexpr int NL = 6;

template <bool NT>
__global__ __launch_bounds__(256) void k_chain(const f32x4* W, const float* vin, float* vout) {
    const int tid = threadIdx.x;
    const f32x4* wp = W + ((size_t)blockIdx.x * 256 + tid) * NL;
    f32x4 w[NL];
#pragma unroll
    for (int i = 0; i < NL; ++i) w[i] = NT ? __builtin_nontemporal_load(wp + i) : wp[i];
    const f32x4 xv = *(const f32x4*)(vin + 4 * (tid & 127));
    float acc = xv[0] + xv[1] + xv[2] + xv[3];
#pragma unroll
    for (int i = 0; i < NL; ++i) acc += w[i][0] * w[i][1] + w[i][2] * w[i][3];
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
    if ((tid & 63) == 0) vout[(blockIdx.x * 4 + (tid >> 6)) % 768] = acc * 1e-6f;
}
__global__ __launch_bounds__(256) void k_shadow(const f32x4* W, float* sink) {
    const int tid = threadIdx.x;
    const f32x4* wp = W + ((size_t)blockIdx.x * 256 + tid) * NL;
    f32x4 w[NL];
#pragma unroll
    for (int i = 0; i < NL; ++i) w[i] = wp[i];
    float acc = 0.f;
#pragma unroll
    for (int i = 0; i < NL; ++i) acc += w[i][0] + w[i][1] + w[i][2] + w[i][3];
    if (acc == 123.456f) sink[0] = acc;            // keeps the loads alive; never true for zero weights
}

int main() {
    const int N = 100, G = 192;                                   // 100 dependent launches, 192 blocks x 24 KB = 4.7 MB each
    const size_t per = (size_t)G * 256 * NL;                       // f32x4 per launch
    f32x4* W; float *v0, *v1, *sink;
    CK(hipMalloc(&W, per * N * sizeof(f32x4))); CK(hipMemset(W, 0, per * N * sizeof(f32x4)));
    CK(hipMalloc(&v0, 4096)); CK(hipMalloc(&v1, 4096)); CK(hipMalloc(&sink, 64));
    CK(hipMemset(v0, 0, 4096)); CK(hipMemset(v1, 0, 4096));
    hipStream_t s0, s1; CK(hipStreamCreateWithFlags(&s0, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    std::vector<hipEvent_t> evc(N), evs(N);
    for (int i = 0; i < N; ++i) { CK(hipEventCreateWithFlags(&evc[i], hipEventDisableTiming)); CK(hipEventCreateWithFlags(&evs[i], hipEventDisableTiming)); }
    for (int mode = 0; mode < 4; ++mode) {
        hipGraph_t g; hipGraphExec_t ge;
        CK(hipStreamBeginCapture(s0, hipStreamCaptureModeThreadLocal));
        if (mode >= 2) { CK(hipEventRecord(evc[0], s0)); CK(hipStreamWaitEvent(s1, evc[0], 0)); }      // fork
        for (int i = 0; i < N; ++i) {
            const f32x4* w = W + per * i;
            if (mode >= 2) {
                // shadow for launch i+1 runs beside chain launch i (it was allowed to start after chain launch i-1)
                if (i + 1 < N) {
                    hipLaunchKernelGGL(k_shadow, dim3(G), dim3(256), 0, s1, W + per * (i + 1), sink);
                    CK(hipEventRecord(evs[i + 1], s1));
                }
                if (mode == 3 && i > 0) CK(hipStreamWaitEvent(s0, evs[i], 0));
            }
            if (mode == 0) hipLaunchKernelGGL((k_chain<true>), dim3(G), dim3(256), 0, s0, w, (i & 1) ? v1 : v0, (i & 1) ? v0 : v1);
            else hipLaunchKernelGGL((k_chain<false>), dim3(G), dim3(256), 0, s0, w, (i & 1) ? v1 : v0, (i & 1) ? v0 : v1);
            if (mode >= 2 && i + 2 < N) { CK(hipEventRecord(evc[i + 1], s0)); CK(hipStreamWaitEvent(s1, evc[i + 1], 0)); }   // pace the shadow
        }
        if (mode >= 2) { CK(hipEventRecord(evs[0], s1)); CK(hipStreamWaitEvent(s0, evs[0], 0)); }       // join
        CK(hipStreamEndCapture(s0, &g));
        CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        for (int rep = 0; rep < 3; ++rep) {
            CK(hipEventRecord(e0, s0));
            for (int k = 0; k < 20; ++k) CK(hipGraphLaunch(ge, s0));
            CK(hipEventRecord(e1, s0)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (rep == 2) printf("mode %c: %.3f us per dependent launch (%.0f GB/s)\n", 'A' + mode, ms * 1e3f / (20 * N), (double)per * 16 * N * 20 / (ms * 1e-3) / 1e9);
        }
        CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
    }
    return 0;
}
