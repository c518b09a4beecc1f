// lib.hip — library-level entry points of libfriture_hip.so (init, errors, pointer queries).
#include <atomic>
#include <mutex>
#include <vector>

#include "common.h"

namespace frt {

static thread_local char g_last_error[512] = "";
static int g_cu_count = 0;

void set_last_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_last_error, sizeof(g_last_error), fmt, ap);
    va_end(ap);
}

namespace {
struct Options {                                   // every option starts at -1 (the shape rule), however many there are
    std::atomic<int> v[kOptCount];
    Options() { for (auto& o : v) o.store(-1, std::memory_order_relaxed); }
    std::atomic<int>& operator[](int o) { return v[o]; }
} g_options;
const char* const kOptionNames[kOptCount] = {"gcc_one_workgroup", "gcc_any_length", "ola_chunk_kernels", "pitch_grid_two_pass", "ola_defer", "iir_lookback", "gcc_resident", "iir_lane_columns"};
std::mutex g_retired_mutex;
std::vector<void*> g_retired;
size_t g_retired_bytes = 0;
thread_local int g_capture_depth = 0;
}  // namespace

int option(Option o) { return g_options[o].load(std::memory_order_relaxed); }

void retire_allocation(void* p, size_t bytes) {
    std::lock_guard<std::mutex> lock(g_retired_mutex);
    g_retired.push_back(p);
    g_retired_bytes += bytes;
}

void free_retired_allocations(bool force) {
    std::vector<void*> mine;
    {
        std::lock_guard<std::mutex> lock(g_retired_mutex);
        if (!force && g_retired_bytes <= kRetiredLimit && g_retired.size() <= 64) return;
        mine.swap(g_retired);
        g_retired_bytes = 0;
    }
    if (mine.empty()) return;
    (void)hipDeviceSynchronize();              // whatever was enqueued against the parked blocks has run
    for (void* p : mine) (void)hipFree(p);
}

CaptureScope::CaptureScope() { ++g_capture_depth; }
CaptureScope::~CaptureScope() { --g_capture_depth; }
bool CaptureScope::active() { return g_capture_depth > 0; }

bool is_device_pointer(const void* p) {
    if (!p) return false;
    hipPointerAttribute_t attr;
    hipError_t e = hipPointerGetAttributes(&attr, p);
    if (e != hipSuccess) {
        (void)hipGetLastError();  // plain malloc'ed host memory: clear the sticky error
        return false;
    }
    return attr.type == hipMemoryTypeDevice || attr.type == hipMemoryTypeManaged;
}

int device_cu_count() {
    if (g_cu_count > 0) return g_cu_count;
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess)
        g_cu_count = prop.multiProcessorCount;
    if (g_cu_count <= 0) g_cu_count = 256;  // MI355X
    return g_cu_count;
}

// ---- StageCall (common.h) ---------------------------------------------------------------------------------------------
namespace {
struct StageArena {
    std::mutex mu;
    hipStream_t stream = nullptr;
    char* pin = nullptr;
    char* dev = nullptr;
    size_t pin_bytes = 0, dev_bytes = 0;
};
StageArena& arena() {
    static StageArena* a = new StageArena();          // never destroyed: see common.h
    return *a;
}
constexpr size_t kStageAlign = 256;
size_t stage_round(size_t n) { return (n + kStageAlign - 1) / kStageAlign * kStageAlign; }
}  // namespace

StageCall::StageCall() {
    arena().mu.lock();
    locked_ = true;
}

StageCall::~StageCall() {
    if (locked_) arena().mu.unlock();
}

int StageCall::add_in(const void* p, size_t bytes) {
    const bool dev = is_device_pointer(p);
    any_device_ |= dev;
    arg_.push_back({p, nullptr, dev ? const_cast<void*>(p) : nullptr, bytes, in_bytes_, 0, !dev});
    if (!dev) in_bytes_ += stage_round(bytes);
    return (int)arg_.size() - 1;
}

int StageCall::add_out(void* p, size_t bytes) {
    const bool dev = is_device_pointer(p);
    any_device_ |= dev;
    arg_.push_back({nullptr, p, dev ? p : nullptr, bytes, out_bytes_, 1, !dev});
    if (!dev) out_bytes_ += stage_round(bytes);
    return (int)arg_.size() - 1;
}

int StageCall::add_scratch(size_t bytes) {
    arg_.push_back({nullptr, nullptr, nullptr, bytes, scratch_bytes_, 2, true});
    scratch_bytes_ += stage_round(bytes);
    return (int)arg_.size() - 1;
}

int StageCall::begin() {
    StageArena& a = arena();
    if (!a.stream) FRT_HIP_CHECK(hipStreamCreateWithFlags(&a.stream, hipStreamNonBlocking));
    const size_t host_need = in_bytes_ + out_bytes_, dev_need = in_bytes_ + out_bytes_ + scratch_bytes_;
    if (host_need > a.pin_bytes) {
        FRT_HIP_CHECK(hipStreamSynchronize(a.stream));
        if (a.pin) (void)hipHostFree(a.pin);
        a.pin = nullptr;
        a.pin_bytes = 0;
        FRT_HIP_CHECK(hipHostMalloc((void**)&a.pin, 2 * host_need + 4096, hipHostMallocDefault));
        a.pin_bytes = 2 * host_need + 4096;
    }
    if (dev_need > a.dev_bytes) {
        FRT_HIP_CHECK(hipStreamSynchronize(a.stream));
        if (a.dev) (void)hipFree(a.dev);
        a.dev = nullptr;
        a.dev_bytes = 0;
        FRT_HIP_CHECK(hipMalloc((void**)&a.dev, 2 * dev_need + 4096));
        a.dev_bytes = 2 * dev_need + 4096;
    }
    // Small calls (the widgets': a few KB per argument) skip the copy engines altogether: pinned host memory is mapped into
    // the device's address space, the kernel reads its inputs from the pinned block and writes its outputs there — one PCIe
    // round trip inside the kernel instead of two DMA set-ups around it (about half of such a call's wall time).
    zero_copy_ = host_need <= kZeroCopyMax;
    // device arena: [inputs][outputs][scratch]; pinned block: [inputs][outputs]
    for (Arg& g : arg_) {
        if (!g.staged) continue;
        const size_t base = g.kind == 0 ? 0 : g.kind == 1 ? in_bytes_ : in_bytes_ + out_bytes_;
        g.dev = (zero_copy_ && g.kind != 2) ? a.pin + base + g.off : a.dev + base + g.off;
        if (g.kind == 0 && g.bytes) memcpy(a.pin + g.off, g.src, g.bytes);
    }
    // device-resident arguments were produced on a stream this call does not know: the null stream waits for the blocking ones
    launch_stream_ = any_device_ ? nullptr : a.stream;
    if (in_bytes_ && !zero_copy_) FRT_HIP_CHECK(hipMemcpyAsync(a.dev, a.pin, in_bytes_, hipMemcpyHostToDevice, launch_stream_));
    return FRT_OK;
}

int StageCall::finish() {
    StageArena& a = arena();
    FRT_HIP_CHECK(hipGetLastError());
    if (out_bytes_ && !zero_copy_)
        FRT_HIP_CHECK(hipMemcpyAsync(a.pin + in_bytes_, a.dev + in_bytes_, out_bytes_, hipMemcpyDeviceToHost, launch_stream_));
    FRT_HIP_CHECK(hipStreamSynchronize(launch_stream_));
    for (const Arg& g : arg_)
        if (g.staged && g.kind == 1 && g.bytes) memcpy(g.dst, a.pin + in_bytes_ + g.off, g.bytes);
    return FRT_OK;
}

}  // namespace frt

using namespace frt;

extern "C" const char* frt_last_error(void) { return g_last_error; }

extern "C" const char* frt_version(void) { return "friture_hip 0.1 (gfx950)"; }

extern "C" int frt_is_device_pointer(const void* p) { return is_device_pointer(p) ? 1 : 0; }

extern "C" int frt_set_option(const char* name, int value) {
    FRT_REQUIRE(name, "frt_set_option: null name");
    for (int o = 0; o < kOptCount; ++o)
        if (!strcmp(name, kOptionNames[o])) {
            g_options[o].store(value < 0 ? -1 : value, std::memory_order_relaxed);
            return FRT_OK;
        }
    set_last_error("frt_set_option: unknown option '%s'", name);
    return FRT_ERR_INVALID;
}

extern "C" int frt_get_option(const char* name, int* value_out) {
    FRT_REQUIRE(name && value_out, "frt_get_option: null argument");
    for (int o = 0; o < kOptCount; ++o)
        if (!strcmp(name, kOptionNames[o])) {
            *value_out = option((Option)o);
            return FRT_OK;
        }
    set_last_error("frt_get_option: unknown option '%s'", name);
    return FRT_ERR_INVALID;
}

extern "C" int frt_device_properties(int device, int* n_cus_out, int64_t* hbm_bytes_out) {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) {
        set_last_error("frt_device_properties: no HIP device visible (%s)", e == hipSuccess ? "count = 0" : hipGetErrorString(e));
        return FRT_ERR_NO_DEVICE;
    }
    FRT_REQUIRE(device >= 0 && device < n, "frt_device_properties: device %d out of range (have %d)", device, n);
    hipDeviceProp_t prop;
    FRT_HIP_CHECK(hipGetDeviceProperties(&prop, device));
    if (n_cus_out) *n_cus_out = prop.multiProcessorCount;
    if (hbm_bytes_out) *hbm_bytes_out = (int64_t)prop.totalGlobalMem;
    return FRT_OK;
}

extern "C" int frt_init(int device, int* n_cus_out, int64_t* hbm_bytes_out) {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) {
        set_last_error("frt_init: no HIP device visible (%s)", e == hipSuccess ? "count = 0" : hipGetErrorString(e));
        return FRT_ERR_NO_DEVICE;
    }
    FRT_REQUIRE(device >= 0 && device < n, "frt_init: device %d out of range (have %d)", device, n);
    FRT_HIP_CHECK(hipSetDevice(device));
    hipDeviceProp_t prop;
    FRT_HIP_CHECK(hipGetDeviceProperties(&prop, device));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        set_last_error("frt_init: device %d is %s; this library only carries gfx950 code", device, prop.gcnArchName);
        return FRT_ERR_NO_DEVICE;
    }
    g_cu_count = prop.multiProcessorCount;
    if (n_cus_out) *n_cus_out = prop.multiProcessorCount;
    if (hbm_bytes_out) *hbm_bytes_out = (int64_t)prop.totalGlobalMem;
    return FRT_OK;
}
