// Launch plumbing shared by the translation units of libsxamd: the sampled kernel timer (sx_profile_*), the dynamic-LDS
// grant and the launch check.  Defined in sx_kernels.hip.
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include <cstdlib>

#include "../../include/sx_amd.h"

namespace sx {

constexpr size_t kMaxLdsBytes = 160 * 1024;

// Takes a (start, stop) event pair for one launch of kernel class `kind`, or returns false (timer off / cap reached / not
// this launch's turn).
bool prof_take(int kind, hipEvent_t* start, hipEvent_t* stop);
int check_launch();
int allow_lds_ptr(const void* kernel, size_t bytes);
int device_cus();

// SX_PROF_RECORD=1: the round-1/2 timer (two hipEventRecord around the launch), kept for A/B runs.
inline bool prof_by_record() {
    static const bool on = [] {
        const char* e = std::getenv("SX_PROF_RECORD");
        return e && e[0] == '1';
    }();
    return on;
}

// Every kernel of the path is launched through here.  With the timer on (and this launch sampled), the launch carries a
// start and a stop event (hipExtLaunchKernelGGL): the two HIP events read the dispatch's own begin / end timestamps -- the
// interval rocprofv3's kernel trace reports -- and no marker packets go into the queue, so the neighbouring launches are
// not held back (two hipEventRecord around the launch cost the solve 2.9 % at every third launch; round 3 measurement).
template <typename F, typename... Args>
inline void launch(int kind, F kernel, dim3 grid, dim3 block, size_t lds, hipStream_t stream, Args... args) {
    hipEvent_t start = nullptr, stop = nullptr;
    const bool timed = prof_take(kind, &start, &stop);
    if (timed && !prof_by_record()) {
        hipExtLaunchKernelGGL(kernel, grid, block, (std::uint32_t)lds, stream, start, stop, 0u, args...);
        return;
    }
    if (timed) (void)hipEventRecord(start, stream);
    hipLaunchKernelGGL(kernel, grid, block, lds, stream, args...);
    if (timed) (void)hipEventRecord(stop, stream);
}

template <typename K>
inline int allow_lds(K kernel, size_t bytes) {
    return allow_lds_ptr(reinterpret_cast<const void*>(kernel), bytes);
}

}  // namespace sx
