// device_info.h -- per-device facts the launch plans are sized by, and the one place kernels are launched from.
#pragma once
#include <hip/hip_runtime.h>
#include <atomic>

namespace iris {

// A dry run (iris_hifigan_describe_plan: host-only, no device needed) walks the same forward code -- argument checks,
// workspace layout, every launch plan -- and records what WOULD be launched instead of launching it.
struct DryRunLaunch { const char* kernel; unsigned grid[3]; unsigned block; unsigned long long lds_bytes; };
struct DryRun {
    DryRunLaunch* out; int capacity; int n;
    int cu_count;                      // the device the plans are sized for (no device is queried in a dry run)
};
inline DryRun*& dry_run_slot() { static thread_local DryRun* d = nullptr; return d; }
inline DryRun* dry_run() { return dry_run_slot(); }
inline const DryRun* dry_run_peek() { return dry_run_slot(); }

// Compute units of the CURRENT HIP device (launch plans are sized per device: a process may hold engines on
// several GPUs, so the count is cached per device ordinal, not per process).
inline int device_cu_count() {
    if (const DryRun* d = dry_run_peek()) return d->cu_count;
    static std::atomic<int> cache[64];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0) dev = 0;
    const int slot = dev < 64 ? dev : 63;
    int n = dev < 64 ? cache[slot].load(std::memory_order_relaxed) : 0;
    if (n <= 0) {
        n = 256;
        (void)hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev);
        if (n <= 0) n = 256;
        if (dev < 64) cache[slot].store(n, std::memory_order_relaxed);
    }
    return n;
}

// ---- launches --------------------------------------------------------------------------------------------------
// Sets the kernel's dynamic-LDS limit when it needs more than the default 64 KB, launches, returns the launch status.
template <class K, class... A>
inline hipError_t launch_kernel_named(const char* name, K kfn, dim3 grid, dim3 block, size_t lds_bytes, hipStream_t stream, const A&... args) {
    if (DryRun* d = dry_run()) {
        if (d->out && d->n < d->capacity)
            d->out[d->n] = DryRunLaunch{name, {grid.x, grid.y, grid.z}, block.x, (unsigned long long)lds_bytes};
        ++d->n;
        return (grid.x == 0 || grid.y == 0 || grid.z == 0 || grid.y > 65535u || grid.z > 65535u || lds_bytes > 160 * 1024)
                   ? hipErrorInvalidValue : hipSuccess;
    }
    if (lds_bytes > 64 * 1024) {
        const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(kfn, grid, block, lds_bytes, stream, args...);
    return hipGetLastError();
}
#define launch_kernel(kfn, ...) launch_kernel_named(#kfn, kfn, __VA_ARGS__)

}  // namespace iris
