// lib.hip — library-level entry points of libfriture_hip.so (init, errors, pointer queries).
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

}  // namespace frt

using namespace frt;

extern "C" const char* frt_last_error(void) { return g_last_error; }

extern "C" const char* frt_version(void) { return "friture_hip 0.1 (gfx950)"; }

extern "C" int frt_is_device_pointer(const void* p) { return is_device_pointer(p) ? 1 : 0; }

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
