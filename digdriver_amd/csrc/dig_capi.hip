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

}  // namespace dig

extern "C" {

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
