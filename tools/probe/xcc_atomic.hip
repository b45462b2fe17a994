// Probe: is a plain atomicAdd (agent scope: no sc1 on gfx950) on hipMalloc memory coherent ACROSS XCCs inside one kernel?
//   hipcc -O3 --offload-arch=gfx950 xcc_atomic.hip -o xcc_atomic && ./xcc_atomic
// Every workgroup adds 1 to one counter `reps` times and keeps the values it got back; coherent = all values distinct.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
template <int SCOPE>
__global__ __launch_bounds__(64) void hit(unsigned* counter, unsigned* got, int reps)
{
    if (threadIdx.x) return;
    for (int r = 0; r < reps; ++r) {
        unsigned v;
        if (SCOPE == 0) v = atomicAdd(counter, 1u);
        else v = __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        got[blockIdx.x * reps + r] = v;
        __builtin_amdgcn_s_sleep(32);
    }
}
int main()
{
    const int grid = 64, reps = 64;
    unsigned *c, *g;
    (void)hipMalloc(&c, 256); (void)hipMalloc(&g, sizeof(unsigned) * grid * reps);
    std::vector<unsigned> h(grid * reps);
    for (int scope = 0; scope < 2; ++scope) {
        (void)hipMemset(c, 0, 256);
        if (scope == 0) hipLaunchKernelGGL(hit<0>, dim3(grid), dim3(64), 0, 0, c, g, reps);
        else hipLaunchKernelGGL(hit<1>, dim3(grid), dim3(64), 0, 0, c, g, reps);
        (void)hipMemcpy(h.data(), g, sizeof(unsigned) * grid * reps, hipMemcpyDeviceToHost);
        unsigned fin; (void)hipMemcpy(&fin, c, 4, hipMemcpyDeviceToHost);
        std::sort(h.begin(), h.end());
        int dup = 0; for (size_t i = 1; i < h.size(); ++i) dup += h[i] == h[i - 1];
        printf("%s: %d adds, final counter %u, duplicate return values %d, largest %u\n", scope ? "system scope (sc1)" : "atomicAdd (agent scope)", grid * reps, fin, dup, h.back());
    }
    return 0;
}
