// Microbenchmark: is a 16-byte aligned, 16-byte wide store of one lane (buffer_store_dwordx4 sc1) ever OBSERVED TORN by a 16-byte sc1 load of a lane on another
// CU / XCD of a gfx950?  (Nothing in the ISA promises it; the persistent decode launch's 16-byte act granules, persist_layer.hip PL_ACT16, carry a checksum word
// for that reason -- this program says how often the check would have to fire.)
// Writers: W workgroups, every lane owns one 16-byte slot and stores {i, i ^ A, i ^ B, i ^ C} for i = 1 .. ITER with no waits in between (so that several stores
// to one slot are in flight at once).  Readers: all other workgroups re-read every slot until the writers are done and classify each value they see:
//   whole  : the four words belong to one i
//   torn8  : words {0,1} from one store, {2,3} from another (an 8-byte split)
//   torn4  : any other mix
// Also printed: how many DISTINCT i a reader saw per slot on average (> 1 proves the reads overlapped the writes in time).
// Build: timeout 300 hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/mb/tear16.hip -o tools/mb/tear16      Run: tools/mb/tear16 [iters]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s (line %d)\n", #x, hipGetErrorString(e), __LINE__); return 1; } } while (0)
#define KA 0x9E3779B9u
#define KB 0x7F4A7C15u
#define KC 0x85EBCA6Bu
constexpr int W = 32, THREADS = 256, SLOTS = W * THREADS;

struct Counts { unsigned long long whole, torn8, torn4, changes, reads; };

__global__ __launch_bounds__(THREADS) void tear_kernel(u32x4* slots, int iters, int* done, Counts* out, int flat) {
    const int b = blockIdx.x, tid = threadIdx.x;
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)slots, 0, SLOTS * 16, 0x00020000);
    if (b < W) {
        const int slot = b * THREADS + tid;
        for (int i = 1; i <= iters; ++i) {
            const unsigned u = (unsigned)i;
            const u32x4 v = {u, u ^ KA, u ^ KB, u ^ KC};
            if (flat) asm volatile("global_store_dwordx4 %0, %1, off sc1" :: "v"(slots + slot), "v"(v) : "memory");
            else __builtin_amdgcn_raw_buffer_store_b128(v, rsrc, slot * 16, 0, 16);
        }
        __syncthreads();
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        if (tid == 0) atomicAdd(done, 1);
        return;
    }
    unsigned long long whole = 0, torn8 = 0, torn4 = 0, changes = 0, reads = 0;
    const int nr = gridDim.x - W, rb = b - W;
    unsigned last = 0;
    for (unsigned pass = 0;; ++pass) {
        // every reader lane walks the slots with its own stride so that all slots are read from all XCDs
        const int slot = (int)(((unsigned)(rb * THREADS + tid) + pass * (unsigned)(nr * THREADS + 7)) % (unsigned)SLOTS);
        u32x4 x;
        if (flat) asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(x) : "v"(slots + slot) : "memory");
        else x = __builtin_amdgcn_raw_buffer_load_b128(rsrc, slot * 16, 0, (int)(16u | 0x80000000u));
        ++reads;
        const bool lo = (x[1] == (x[0] ^ KA)), hi = ((x[2] ^ KB) == (x[3] ^ KC)), mid = ((x[0] ^ KB) == x[2]);
        if (x[0] == 0 && x[1] == 0 && x[2] == 0 && x[3] == 0) ++whole;       // never written yet
        else if (lo && hi && mid) ++whole;
        else if (lo && hi) ++torn8;
        else ++torn4;
        if (x[0] != last) { ++changes; last = x[0]; }
        if ((pass & 63u) == 63u && __hip_atomic_load(done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= W) break;
        if (pass > (1u << 24)) break;
    }
    atomicAdd(&out->whole, whole); atomicAdd(&out->torn8, torn8); atomicAdd(&out->torn4, torn4); atomicAdd(&out->changes, changes); atomicAdd(&out->reads, reads);
}

int main(int argc, char** argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 20000;
    u32x4* slots; int* done; Counts* out;
    CK(hipMalloc(&slots, SLOTS * 16)); CK(hipMalloc(&done, 4)); CK(hipMalloc(&out, sizeof(Counts)));
    for (int flat = 0; flat < 2; ++flat) {
        Counts tot = {0, 0, 0, 0, 0};
        for (int rep = 0; rep < 5; ++rep) {
            CK(hipMemset(slots, 0, SLOTS * 16)); CK(hipMemset(done, 0, 4)); CK(hipMemset(out, 0, sizeof(Counts)));
            hipLaunchKernelGGL(tear_kernel, dim3(256), dim3(THREADS), 0, 0, slots, iters, done, out, flat);
            CK(hipGetLastError()); CK(hipDeviceSynchronize());
            Counts c; CK(hipMemcpy(&c, out, sizeof(c), hipMemcpyDeviceToHost));
            tot.whole += c.whole; tot.torn8 += c.torn8; tot.torn4 += c.torn4; tot.changes += c.changes; tot.reads += c.reads;
        }
        printf("{\"store\": \"%s\", \"writers\": %d, \"readers\": %d, \"stores_per_slot\": %d, \"reps\": 5, \"reads\": %llu, \"value_changes_seen\": %llu, \"whole\": %llu, \"torn_8_byte_split\": %llu, \"torn_other\": %llu}\n",
               flat ? "global_store_dwordx4 sc1 / global_load_dwordx4 sc1" : "buffer_store_dwordx4 sc1 / buffer_load_dwordx4 sc1", W, 256 - W, iters, tot.reads, tot.changes, tot.whole, tot.torn8, tot.torn4);
    }
    return 0;
}
