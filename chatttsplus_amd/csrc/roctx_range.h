// roctx ranges at the reference's NVTX points (trt_models/llama_trt_model.py:44,74 "forward"; trt_models/predictor.py:92,142,159,164
// "allocate_max_buffers" / "adjust_buffer" / "set_tensors" / "execute"): host-side ranges around the C-ABI entry points that enqueue
// the prompt pass, a decode chunk, the sampler and the vocoder, visible in `rocprofv3 --marker-trace`.  The marker library is
// resolved at first use with dlopen (librocprofiler-sdk-roctx.so, then libroctx64.so); without it the ranges are no-ops, so the
// product library carries no hard dependency on a profiler component.
#pragma once
#include <dlfcn.h>

namespace ctts {
struct RoctxApi {
    int (*push)(const char*) = nullptr;
    int (*pop)() = nullptr;
    RoctxApi() {
        const char* libs[] = {"librocprofiler-sdk-roctx.so", "librocprofiler-sdk-roctx.so.1", "libroctx64.so", "libroctx64.so.4"};
        for (const char* l : libs) {
            void* h = dlopen(l, RTLD_LAZY | RTLD_GLOBAL);
            if (!h) continue;
            push = (int (*)(const char*))dlsym(h, "roctxRangePushA");
            pop = (int (*)())dlsym(h, "roctxRangePop");
            if (push && pop) return;
            push = nullptr; pop = nullptr;
        }
    }
};
inline RoctxApi& roctx_api() { static RoctxApi a; return a; }
struct RoctxRange {
    bool on;
    explicit RoctxRange(const char* name) : on(false) {
        RoctxApi& a = roctx_api();
        if (a.push) { a.push(name); on = true; }
    }
    ~RoctxRange() { if (on) roctx_api().pop(); }
    RoctxRange(const RoctxRange&) = delete;
    RoctxRange& operator=(const RoctxRange&) = delete;
};
}  // namespace ctts
#define CTTS_RANGE(name) ctts::RoctxRange _ctts_range_(name)
