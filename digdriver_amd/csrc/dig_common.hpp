// dig_common.hpp -- error plumbing and launch helpers shared by the .hip translation units.
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdint.h>
#include <stdio.h>

#include <string>

#include "../../include/dig_hip.h"

namespace dig {

std::string& last_error_ref();
int set_error(int code, const char* fmt, ...);

#define DIG_HIP_TRY(expr)                                                                              \
    do {                                                                                               \
        hipError_t _e = (expr);                                                                        \
        if (_e != hipSuccess)                                                                          \
            return ::dig::set_error(DIG_EHIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e),   \
                                    __FILE__, __LINE__);                                               \
    } while (0)

#define DIG_REQUIRE(cond, msg)                                                                  \
    do {                                                                                        \
        if (!(cond)) return ::dig::set_error(DIG_EINVAL, "%s: requirement failed: %s", __func__, msg); \
    } while (0)

// number of CUs of the current device (cached per device)
int cu_count();

// RAII device buffer used only by the *_host twins
struct DevBuf {
    void* p = nullptr;
    ~DevBuf()
    {
        if (p) (void)hipFree(p);
    }
    hipError_t alloc(size_t bytes) { return hipMalloc(&p, bytes ? bytes : 1); }
    template <typename T>
    T* as()
    {
        return static_cast<T*>(p);
    }
};

// i / C with a host-computed magic multiplier: e = hi64(i * ceil(2^64 / C)), exact while
// i * C < 2^64 (the per-pair int64 division the flat [E, C] index would otherwise need costs
// more than a whole recurrence step).
struct FastDiv {
    uint64_t magic;   // ceil(2^64 / d), d >= 2
};
inline FastDiv make_fastdiv(int64_t d)
{
    FastDiv f;
    f.magic = (d >= 2) ? (~(uint64_t)0 / (uint64_t)d) + 1 : 0;
    return f;
}
__device__ __forceinline__ int64_t fastdiv(int64_t i, const FastDiv& f)
{
    return (int64_t)__umul64hi((uint64_t)i, f.magic);
}

// A stage timer (dig_stage_timer_*, include/dig_hip.h): two events that the next launch of a stage's kernel on the arming
// thread fills with the kernel's own begin and end (hipExtLaunchKernelGGL: taken from the dispatch, no packet added to the stream).
struct StageTimer {
    hipEvent_t start = nullptr, stop = nullptr;
    int launched = 0;
};
StageTimer* take_armed_timer(int stage);
void disarm_stage_timers();                    // end of a dig_element_pipeline call: nothing stays armed       // the timer armed for `stage` by this thread, or NULL; disarms it
// launch `kernel` as hipLaunchKernelGGL does, through the armed timer of `stage` if there is one
#define DIG_LAUNCH_STAGE(stage, kernel, grid, block, lds, stream, ...)                                                   \
    do {                                                                                                                 \
        if (::dig::StageTimer* _t = ::dig::take_armed_timer(stage)) {                                                    \
            hipExtLaunchKernelGGL(kernel, grid, block, lds, stream, _t->start, _t->stop, 0, __VA_ARGS__);                \
            _t->launched = 1;                                                                                            \
        } else                                                                                                           \
            hipLaunchKernelGGL(kernel, grid, block, lds, stream, __VA_ARGS__);                                           \
    } while (0)

// dig_tiles_rows.hip: the row walk of the tile probabilities, one launch per cohort pass; regions it does not take are left with
// n_valid = -2 for the general kernel (dig_tiles.hip)
int launch_tile_probs_rows(const uint32_t* words, int64_t n_words, const int64_t* chrom_off, const int64_t* chrom_len,
                           const int32_t* reg_chrom, const int64_t* reg_start, const int64_t* reg_end, int64_t R, const double* s_prob,
                           int64_t C, int n_up, int binsize, int64_t n_tiles, double* pt, int64_t* first_pos, int32_t* n_valid,
                           hipStream_t stream);

inline int grid_for(int64_t n, int block, int max_blocks_per_cu = 8)
{
    int64_t want = (n + block - 1) / block;
    int64_t cap = (int64_t)cu_count() * max_blocks_per_cu;
    if (want < 1) want = 1;
    return (int)(want < cap ? want : cap);
}

}  // namespace dig
