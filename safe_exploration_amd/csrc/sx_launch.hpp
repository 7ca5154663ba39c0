// Launch plumbing shared by the translation units of libsxamd: the sampled kernel timer (sx_profile_*), the dynamic-LDS
// grant and the launch check.  Defined in sx_kernels.hip.
#pragma once
#include <hip/hip_runtime.h>

#include "../../include/sx_amd.h"

namespace sx {

constexpr size_t kMaxLdsBytes = 160 * 1024;

// Takes a (start, stop) event pair for one launch of kernel class `kind`, or returns false (timer off / cap reached / not
// this launch's turn).
bool prof_take(int kind, hipEvent_t* start, hipEvent_t* stop);
int check_launch();
int allow_lds_ptr(const void* kernel, size_t bytes);
int device_cus();

// Every kernel of the path is launched through here.  With the timer on (and this launch sampled), the launch is bracketed
// by two hipEventRecord on the same stream.
template <typename F, typename... Args>
inline void launch(int kind, F kernel, dim3 grid, dim3 block, size_t lds, hipStream_t stream, Args... args) {
    hipEvent_t start = nullptr, stop = nullptr;
    const bool timed = prof_take(kind, &start, &stop);
    if (timed) (void)hipEventRecord(start, stream);
    hipLaunchKernelGGL(kernel, grid, block, lds, stream, args...);
    if (timed) (void)hipEventRecord(stop, stream);
}

template <typename K>
inline int allow_lds(K kernel, size_t bytes) {
    return allow_lds_ptr(reinterpret_cast<const void*>(kernel), bytes);
}

}  // namespace sx
