// Microbenchmark: cost of a device-wide barrier inside one persistent kernel on MI355X, and of a "phase" =
// prefetched 16 KB weight slice per block + barrier + read of a vector other blocks produced.
// Build: timeout 300 hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/mb/gridbar.hip -o tools/mb/gridbar
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__device__ inline void grid_barrier(unsigned* ctr, unsigned target, int* err) {
    __syncthreads();
    if (threadIdx.x == 0) {
        __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        int spins = 0;
        while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
            if (++spins > (1 << 22)) { *err = 1; break; }
        }
        __atomic_thread_fence(__ATOMIC_ACQUIRE);       // agent scope is the default for hip device code fences? use builtin below
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    __syncthreads();
}

__global__ __launch_bounds__(256) void k_bar(unsigned* ctr, int n, int* err) {
    const unsigned G = gridDim.x;
    for (int i = 0; i < n; ++i) grid_barrier(ctr, (unsigned)(i + 1) * G, err);
}

// phase: every block prefetches NL x 16 B per thread of weights BEFORE the barrier, then reads the 768-float vector the
// previous phase produced (all blocks wrote a slice), reduces, writes its slice of the next vector.
template <int NL, bool PREFETCH>
__global__ __launch_bounds__(256) void k_phase(const f32x4* W, float* vec, unsigned* ctr, int n, int* err, size_t wstride) {
    const unsigned G = gridDim.x;
    const int tid = threadIdx.x;
    __shared__ float red[4];
    float carry = 0.f;
    for (int i = 0; i < n; ++i) {
        const f32x4* wp = W + (size_t)(i % 24) * wstride + ((size_t)blockIdx.x * 256 + tid) * NL;
        f32x4 w[NL];
        if (PREFETCH) {
#pragma unroll
            for (int j = 0; j < NL; ++j) w[j] = __builtin_nontemporal_load(wp + j);
        }
        grid_barrier(ctr, (unsigned)(i + 1) * G, err);
        if (!PREFETCH) {
#pragma unroll
            for (int j = 0; j < NL; ++j) w[j] = __builtin_nontemporal_load(wp + j);
        }
        const float* vin = vec + (size_t)(i & 1) * 1024;
        float* vout = vec + (size_t)((i + 1) & 1) * 1024;
        float x = __hip_atomic_load(vin + (tid * 3 % 768), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        float acc = x + carry;
#pragma unroll
        for (int j = 0; j < NL; ++j) acc += w[j][0] * w[j][1] + w[j][2] * w[j][3];
        // block reduce (cheap stand-in for the MFMA + LDS reduction)
        for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
        if ((tid & 63) == 0) red[tid >> 6] = acc;
        __syncthreads();
        const float tot = red[0] + red[1] + red[2] + red[3];
        if (tid < 3) vout[(blockIdx.x * 3 + tid) % 768] = tot * 1e-6f;
        carry = tot * 1e-9f;
        __syncthreads();
    }
}

int main() {
    unsigned* ctr; int* err; float* vec; f32x4* W;
    const size_t wstride = (size_t)512 * 256 * 8;             // f32x4 per "layer-phase" slice (16 MB)
    CK(hipMalloc(&ctr, 4)); CK(hipMalloc(&err, 4)); CK(hipMalloc(&vec, 2 * 1024 * 4));
    CK(hipMalloc(&W, wstride * 24 * sizeof(f32x4)));
    CK(hipMemset(W, 0, wstride * 24 * sizeof(f32x4))); CK(hipMemset(vec, 0, 2 * 1024 * 4)); CK(hipMemset(err, 0, 4));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int N = 2000;
    for (int G : {48, 128, 256, 512}) {
        for (int rep = 0; rep < 2; ++rep) {
            CK(hipMemset(ctr, 0, 4));
            CK(hipEventRecord(e0, 0));
            hipLaunchKernelGGL(k_bar, dim3(G), dim3(256), 0, 0, ctr, N, err);
            CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (rep) printf("barrier only      G=%3d: %.3f us / barrier\n", G, ms * 1e3f / N);
        }
    }
    for (int G : {48, 192, 256, 512}) {
        for (int mode = 0; mode < 4; ++mode) {
            for (int rep = 0; rep < 2; ++rep) {
                CK(hipMemset(ctr, 0, 4));
                CK(hipEventRecord(e0, 0));
                if (mode == 0) hipLaunchKernelGGL((k_phase<4, true>), dim3(G), dim3(256), 0, 0, W, vec, ctr, N, err, wstride);
                if (mode == 1) hipLaunchKernelGGL((k_phase<4, false>), dim3(G), dim3(256), 0, 0, W, vec, ctr, N, err, wstride);
                if (mode == 2) hipLaunchKernelGGL((k_phase<8, true>), dim3(G), dim3(256), 0, 0, W, vec, ctr, N, err, wstride);
                if (mode == 3) hipLaunchKernelGGL((k_phase<8, false>), dim3(G), dim3(256), 0, 0, W, vec, ctr, N, err, wstride);
                CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                if (rep) printf("phase G=%3d %2d KB/block %s: %.3f us / phase  (%.0f GB/s)\n", G, mode < 2 ? 16 : 32, (mode & 1) ? "load after barrier " : "prefetch before bar",
                                ms * 1e3f / N, (double)G * (mode < 2 ? 16384 : 32768) * N / (ms * 1e-3) / 1e9);
            }
        }
    }
    int herr = 0; CK(hipMemcpy(&herr, err, 4, hipMemcpyDeviceToHost));
    printf("spin-timeout flag: %d\n", herr);
    return 0;
}
