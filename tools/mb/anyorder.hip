// Microbenchmark: dependent chain WITHOUT the inter-kernel barrier: kernels are launched "any order" (no barrier bit), every block
// prefetches its weights, then polls the per-block completion flags of the previous kernel (plain stores / loads, no same-address
// atomics), computes, publishes its own flag.  Compares with the ordinary barrier-ordered chain.
// Build: timeout 300 hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/mb/anyorder.hip -o tools/mb/anyorder
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s (line %d)\n", #x, hipGetErrorString(e), __LINE__); return 1; } } while (0)
constexpr int NL = 6, G = 192;

// flags[i][b]: number of times block b of kernel i has completed (monotonic across graph replays)
__global__ __launch_bounds__(256) void k_flag(const f32x4* W, const float* vin, float* vout, const unsigned* prev_flags, unsigned* my_flags, int* err) {
    const int tid = threadIdx.x;
    const f32x4* wp = W + (size_t)blockIdx.x * 256 * NL + tid;
    f32x4 w[NL];
#pragma unroll
    for (int i = 0; i < NL; ++i) w[i] = __builtin_nontemporal_load(wp + i * 256);
    __shared__ unsigned target_s;
    if (tid == 0) target_s = __hip_atomic_load(my_flags + blockIdx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u;
    __syncthreads();
    const unsigned target = target_s;
    if (prev_flags != nullptr && tid < 64) {          // one wave polls the G flags of the previous kernel (3 per lane)
        int spins = 0;
        while (true) {
            bool ok = true;
#pragma unroll
            for (int j = 0; j < (G + 63) / 64; ++j) {
                const int b = tid + 64 * j;
                if (b < G) ok = ok && (__hip_atomic_load(prev_flags + b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= target);
            }
            if (__builtin_amdgcn_ballot_w64(!ok) == 0ull) break;
            if (++spins > (1 << 18)) { *err = 1; break; }
        }
    }
    __syncthreads();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    const float xv0 = __hip_atomic_load(vin + 4 * (tid & 127), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    float acc = xv0;
#pragma unroll
    for (int i = 0; i < NL; ++i) acc += w[i][0] * w[i][1] + w[i][2] * w[i][3];
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
    if ((tid & 63) == 0) __hip_atomic_store(vout + (blockIdx.x * 4 + (tid >> 6)) % 768, acc * 1e-6f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    if (tid == 0) __hip_atomic_store(my_flags + blockIdx.x, target, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
}

int main() {
    const int N = 100;
    const size_t per = (size_t)G * 256 * NL;
    f32x4* W; float *v0, *v1; unsigned* flags; int* err;
    CK(hipMalloc(&W, per * 40 * sizeof(f32x4))); CK(hipMemset(W, 0, per * 40 * sizeof(f32x4)));
    CK(hipMalloc(&v0, 4096)); CK(hipMalloc(&v1, 4096)); CK(hipMemset(v0, 0, 4096)); CK(hipMemset(v1, 0, 4096));
    CK(hipMalloc(&flags, (size_t)N * G * 4)); CK(hipMalloc(&err, 4)); CK(hipMemset(err, 0, 4));
    hipStream_t s0; CK(hipStreamCreateWithFlags(&s0, hipStreamNonBlocking));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int mode = 0; mode < 3; ++mode) {          // 0: ordinary launches (flags always ready)  1: any-order launches in a graph  2: any-order eager
        CK(hipMemset(flags, 0, (size_t)N * G * 4));
        hipGraph_t g = nullptr; hipGraphExec_t ge = nullptr;
        auto enqueue = [&]() -> int {
            for (int i = 0; i < N; ++i) {
                const f32x4* w = W + per * (i % 40);
                const unsigned* pf = i ? flags + (size_t)(i - 1) * G : nullptr;
                unsigned* mf = flags + (size_t)i * G;
                const float* vin = (i & 1) ? v1 : v0; float* vout = (i & 1) ? v0 : v1;
                if (mode == 0) hipLaunchKernelGGL(k_flag, dim3(G), dim3(256), 0, s0, w, vin, vout, pf, mf, err);
                else hipExtLaunchKernelGGL(k_flag, dim3(G), dim3(256), 0, s0, nullptr, nullptr, hipExtAnyOrderLaunch, w, vin, vout, pf, mf, err);
                CK(hipGetLastError());
            }
            return 0;
        };
        if (mode < 2) {
            CK(hipStreamBeginCapture(s0, hipStreamCaptureModeThreadLocal));
            if (enqueue()) return 1;
            hipError_t e = hipStreamEndCapture(s0, &g);
            if (e != hipSuccess) { printf("mode %d: capture failed: %s\n", mode, hipGetErrorString(e)); continue; }
            CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        }
        for (int rep = 0; rep < 3; ++rep) {
            CK(hipEventRecord(e0, s0));
            for (int k = 0; k < 20; ++k) { if (ge) CK(hipGraphLaunch(ge, s0)); else if (enqueue()) return 1; }
            CK(hipEventRecord(e1, s0)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (rep == 2) printf("mode %d (%s): %.3f us per dependent launch\n", mode, mode == 0 ? "barrier-ordered graph" : mode == 1 ? "any-order graph" : "any-order eager", ms * 1e3f / (20 * N));
        }
        int herr = 0; CK(hipMemcpy(&herr, err, 4, hipMemcpyDeviceToHost));
        if (herr) { printf("  spin timeout hit (dependency not satisfied in time)\n"); CK(hipMemset(err, 0, 4)); }
        if (ge) { CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g)); }
    }
    return 0;
}
