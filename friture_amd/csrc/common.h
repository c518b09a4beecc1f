// common.h — host-side helpers shared by the C-ABI translation units.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/friture_hip.h"

namespace frt {

void set_last_error(const char* fmt, ...);

#define FRT_HIP_CHECK(expr)                                                                  \
    do {                                                                                     \
        hipError_t e__ = (expr);                                                             \
        if (e__ != hipSuccess) {                                                             \
            frt::set_last_error("%s:%d: %s failed: %s", __FILE__, __LINE__, #expr,           \
                                hipGetErrorString(e__));                                     \
            return FRT_ERR_HIP;                                                         \
        }                                                                                    \
    } while (0)

#define FRT_REQUIRE(cond, ...)                     \
    do {                                           \
        if (!(cond)) {                             \
            frt::set_last_error(__VA_ARGS__);      \
            return FRT_ERR_INVALID;           \
        }                                          \
    } while (0)

#define FRT_REQUIRE_CODE(cond, code, ...)          \
    do {                                           \
        if (!(cond)) {                             \
            frt::set_last_error(__VA_ARGS__);      \
            return (code);                         \
        }                                          \
    } while (0)

// ---- path selection ---------------------------------------------------------------------------------------------------
// The library never reads the environment: no variable can change what a call computes.  Product code paths that are
// normally chosen by the shape of a call can be forced through frt_set_option (include/friture_hip.h) — explicit, per
// process, visible in the caller's code; tests use it to reach both sides of a shape rule on one input.
enum Option {
    kOptGccOneWorkgroup,      // GCC-PHAT: 1 = one workgroup per pair whatever the batch, 0 = a pair as launches of its phases
    kOptGccAnyLength,         // GCC-PHAT: 1 = the chirp-z path even for lengths the mixed-radix plan serves (at frt_gcc_create)
    kOptOlaChunkKernels,      // FFT overlap-add bank, one block of <= 1024 host samples: 0 = the per-stage transform launches
    kOptPitchGridTwoPass,     // pitch tracker: 1 = the two-pass log-grid kernel on the widget's grid too
    kOptOlaDefer,             // FFT overlap-add bank, batched, >= 6 bands per octave: 0 = a launch per stage (no deferred band filters)
    kOptIirLookback,          // exact IIR bank, time-parallel energies: 0 = a chunk-scan launch at every stage (no look-back output pass)
    kOptGccResident,          // GCC-PHAT, default window, one workgroup per pair: 0 = the kernel with the scratch slab (gcc_phat_kernel)
    kOptIirLaneColumns,       // exact IIR bank, time-parallel energies: output pass as chunk columns with LDS-staged samples — 1 = wherever it can, 0 = nowhere
    kOptCount
};
int option(Option o);         // -1 = not set: the shape rule decides
// Experiment switches of tools/exp (A/B of kernel generations, ablations) exist only in -DFRT_EXPERIMENTS builds
// (tools/exp/build_variant.sh); in the shipped library the switch and the code behind it fold away at compile time.
#ifdef FRT_EXPERIMENTS
inline const char* exp_env(const char* name) { return getenv(name); }
inline int exp_int(const char* name, int otherwise) { const char* e = exp_env(name); return e ? atoi(e) : otherwise; }
#else
constexpr const char* exp_env(const char*) { return nullptr; }
constexpr int exp_int(const char*, int otherwise) { return otherwise; }
#endif

// true when `p` points to device (or managed) memory usable by kernels directly.
bool is_device_pointer(const void* p);

// Allocations a growing DeviceBuffer has replaced.  Kernels enqueued on a caller's non-blocking stream may still use the old
// block, and hipFree synchronises the whole device (not legal at all while a stream of this thread is capturing): the old
// block is PARKED here instead — no synchronisation on growth.  Parked blocks are released together, behind one device
// synchronisation, once more than kRetiredLimit bytes (or 64 blocks) are parked and the calling thread is not capturing,
// and whenever a handle is destroyed (free_retired_allocations(true)).  Buffers grow geometrically, so the parked total
// stays below twice the live total; when an allocation fails outside a capture the parked blocks are released and the
// allocation is tried once more.
void retire_allocation(void* p, size_t bytes);
void free_retired_allocations(bool force);
constexpr size_t kRetiredLimit = (size_t)1 << 28;      // 256 MB parked at most between handle destructions
struct CaptureScope {                       // marks the calling thread as capturing a stream for its lifetime
    CaptureScope();
    ~CaptureScope();
    static bool active();
};

// A grow-only device buffer used for staging host-pointer calls and for plan-owned scratch.
struct DeviceBuffer {
    void* ptr = nullptr;
    size_t bytes = 0;
    int reserve(size_t n) {
        if (n <= bytes) return FRT_OK;
        if (ptr) {
            retire_allocation(ptr, bytes);
            if (n < bytes + bytes / 2) n = bytes + bytes / 2;      // geometric growth bounds what gets parked
            if (!CaptureScope::active()) free_retired_allocations(false);
        }
        ptr = nullptr;
        bytes = 0;
        hipError_t e = hipMalloc(&ptr, n);
        if (e != hipSuccess && !CaptureScope::active()) {       // out of memory with blocks parked: release them, try once more
            (void)hipGetLastError();
            free_retired_allocations(true);
            e = hipMalloc(&ptr, n);
        }
        FRT_HIP_CHECK(e);
        bytes = n;
        return FRT_OK;
    }
    void release() {
        if (ptr) (void)hipFree(ptr);
        ptr = nullptr;
        bytes = 0;
    }
    template <typename T>
    T* as() const { return reinterpret_cast<T*>(ptr); }
};

template <typename T>
int upload(DeviceBuffer& buf, const std::vector<T>& host) {
    int rc = buf.reserve(host.size() * sizeof(T));
    if (rc) return rc;
    FRT_HIP_CHECK(hipMemcpy(buf.ptr, host.data(), host.size() * sizeof(T), hipMemcpyHostToDevice));
    return FRT_OK;
}

// Re-upload of a table that kernels already enqueued on `stream` may still be reading (torch side streams are
// non-blocking: a copy on the null stream is not ordered behind them): only when the values changed, and then behind
// a synchronisation of that stream.  `cache` keeps the host copy.
template <typename T>
int upload_if_changed(DeviceBuffer& buf, std::vector<T>& cache, const std::vector<T>& host, hipStream_t stream) {
    if (buf.ptr && cache == host) return FRT_OK;
    if (buf.ptr) FRT_HIP_CHECK(hipStreamSynchronize(stream));
    int rc = upload(buf, host);
    if (rc) return rc;
    cache = host;
    return FRT_OK;
}

// calls whose host buffers total at most this are served in place from page-locked memory (kernels read / write it over
// the bus) instead of by copies around the kernels
constexpr size_t kZeroCopyMax = 256 * 1024;
int device_cu_count();

// ---- packed staging of host arrays for the stateless entry points ------------------------------------------------------
// The widget-facing functions of pipeline.hip / spectrum.hip take host arrays (the reference's numpy arguments).  Rounds
// 1-2 gave every argument its own hipMalloc + blocking hipMemcpy + hipFree: 0.3 ms for a chain whose arithmetic is
// microseconds.  A StageCall packs all host inputs of a call into ONE pinned block (one asynchronous upload), hands out
// device addresses inside one device arena, and brings all outputs back with ONE download and ONE stream synchronisation.
// The arena (pinned block, device block, stream) is a process-level object created on first use and never freed: no HIP
// call runs from a static or thread-local destructor at exit.  Calls are serialised by its mutex (the reference calls from
// one GUI thread).  Arguments that already live in device memory pass through untouched; a call with any such argument
// launches on the null stream, which is ordered behind the (blocking) stream that produced them.
class StageCall {
  public:
    StageCall();
    ~StageCall();
    // register arguments first (any order), then begin(); ids index ptr()
    int add_in(const void* host_or_dev, size_t bytes);
    int add_out(void* host_or_dev, size_t bytes);
    int add_scratch(size_t bytes);                       // device-only workspace inside the arena
    int begin();                                         // capacity, host -> pinned, one async upload
    template <typename T>
    T* ptr(int id) const { return reinterpret_cast<T*>(arg_[id].dev); }
    size_t offset(int id) const { return arg_[id].off; }   // of a staged input inside the call's input block (fixed at add_in)
    hipStream_t stream() const { return launch_stream_; }
    int finish();                                        // one async download of the outputs, one synchronisation, pinned -> host
  private:
    struct Arg { const void* src; void* dst; void* dev; size_t bytes, off; int kind; bool staged; };
    std::vector<Arg> arg_;
    size_t in_bytes_ = 0, out_bytes_ = 0, scratch_bytes_ = 0;
    hipStream_t launch_stream_ = nullptr;
    bool locked_ = false, any_device_ = false, zero_copy_ = false;
};

}  // namespace frt
