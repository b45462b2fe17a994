// dig_capi.hip -- library-level C ABI: version, error string, device census.
#include <stdarg.h>
#include <string.h>

#include "dig_common.hpp"

namespace dig {

std::string& last_error_ref()
{
    static thread_local std::string err;
    return err;
}

int set_error(int code, const char* fmt, ...)
{
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    last_error_ref() = buf;
    return code;
}

int cu_count()
{
    static thread_local int cached_dev = -1;
    static thread_local int cached_cus = 256;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return cached_cus;
    if (dev != cached_dev) {
        int cus = 0;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && cus > 0)
            cached_cus = cus;
        cached_dev = dev;
    }
    return cached_cus;
}

static thread_local StageTimer* tl_armed[2] = {nullptr, nullptr};      // [0] the dot stage, [1] the statistics stage
static int timer_slot(int stage) { return stage == DIG_PIPE_DOT ? 0 : stage == DIG_PIPE_STATISTICS ? 1 : -1; }

StageTimer* take_armed_timer(int stage)
{
    const int k = timer_slot(stage);
    if (k < 0) return nullptr;
    StageTimer* t = tl_armed[k];
    tl_armed[k] = nullptr;
    return t;
}

// a timer armed for a stage that the call did not launch through DIG_LAUNCH_STAGE (another form of the kernel, a stage the
// call did not include) must not stay armed and attach to a later, unrelated call: dig_element_pipeline calls this on return
void disarm_stage_timers()
{
    tl_armed[0] = tl_armed[1] = nullptr;
}

}  // namespace dig

extern "C" {

int dig_stage_timer_create(void** timer)
{
    DIG_REQUIRE(timer, "non-null argument");
    auto* t = new dig::StageTimer();
    if (hipEventCreate(&t->start) != hipSuccess || hipEventCreate(&t->stop) != hipSuccess) {
        if (t->start) (void)hipEventDestroy(t->start);
        delete t;
        return dig::set_error(DIG_EHIP, "dig_stage_timer_create: hipEventCreate failed");
    }
    *timer = t;
    return DIG_OK;
}

int dig_stage_timer_arm(void* timer, int stage)
{
    DIG_REQUIRE(timer, "a timer of dig_stage_timer_create");
    const int k = dig::timer_slot(stage);
    DIG_REQUIRE(k >= 0, "stage: DIG_PIPE_DOT or DIG_PIPE_STATISTICS");
    auto* t = static_cast<dig::StageTimer*>(timer);
    t->launched = 0;
    dig::tl_armed[k] = t;
    return DIG_OK;
}

int dig_stage_timer_read(void* timer, double* ms)
{
    DIG_REQUIRE(timer && ms, "non-null arguments");
    auto* t = static_cast<dig::StageTimer*>(timer);
    DIG_REQUIRE(t->launched, "no launch of the armed stage followed dig_stage_timer_arm on the arming thread");
    DIG_HIP_TRY(hipEventSynchronize(t->stop));
    float f = 0.f;
    DIG_HIP_TRY(hipEventElapsedTime(&f, t->start, t->stop));
    *ms = (double)f;
    return DIG_OK;
}

__global__ void stage_timer_empty_kernel(int* sink)
{
    if (sink && threadIdx.x == 4096) *sink = 0;
}

int dig_stage_timer_selftest(void* timer, void* stream)
{
    DIG_REQUIRE(timer, "a timer of dig_stage_timer_create");
    auto* t = static_cast<dig::StageTimer*>(timer);
    hipExtLaunchKernelGGL(stage_timer_empty_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, t->start, t->stop, 0, (int*)nullptr);
    DIG_HIP_TRY(hipGetLastError());
    t->launched = 1;
    return DIG_OK;
}

int dig_stage_timer_destroy(void* timer)
{
    if (!timer) return DIG_OK;
    auto* t = static_cast<dig::StageTimer*>(timer);
    for (int k = 0; k < 2; ++k)
        if (dig::tl_armed[k] == t) dig::tl_armed[k] = nullptr;
    (void)hipEventDestroy(t->start);
    (void)hipEventDestroy(t->stop);
    delete t;
    return DIG_OK;
}

int dig_abi_version(void) { return DIG_ABI_VERSION; }

const char* dig_last_error(void) { return dig::last_error_ref().c_str(); }

int dig_device_count(void)
{
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        return dig::set_error(DIG_ENODEV, "hipGetDeviceCount failed: %s", hipGetErrorString(e));
    }
    int ok = 0;
    for (int d = 0; d < n; ++d) {
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, d) == hipSuccess && strncmp(prop.gcnArchName, "gfx950", 6) == 0) ++ok;
    }
    return ok;
}

}  // extern "C"
