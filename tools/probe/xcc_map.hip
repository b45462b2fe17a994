// Probe: which XCC (XCD) does workgroup b of a 256 x 1024-thread launch land on?  (the statistics kernel assumes b % 8)
//   hipcc -O3 --offload-arch=gfx950 xcc_map.hip -o xcc_map && ./xcc_map
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ __launch_bounds__(1024) void where(unsigned* out, int spin)
{
    unsigned x = __builtin_amdgcn_s_getreg((20) | (0 << 6) | ((4 - 1) << 11));      // HW_REG_XCC_ID, bits [3:0]
    unsigned hw = __builtin_amdgcn_s_getreg((4) | (0 << 6) | ((32 - 1) << 11));     // HW_REG_HW_ID
    if (threadIdx.x == 0) { out[2 * blockIdx.x] = x; out[2 * blockIdx.x + 1] = hw; }
    for (int i = 0; i < spin; ++i) __builtin_amdgcn_s_sleep(64);                    // keep every workgroup resident
}
int main()
{
    unsigned* d; unsigned h[2 * 1024];
    (void)hipMalloc(&d, sizeof(h));
    hipDeviceProp_t p; (void)hipGetDeviceProperties(&p, 0);
    for (int grid : {256, 512}) {
        hipLaunchKernelGGL(where, dim3(grid), dim3(1024), 0, 0, d, 2000);
        (void)hipMemcpy(h, d, sizeof(unsigned) * 2 * grid, hipMemcpyDeviceToHost);
        int bad = 0, cnt[16] = {};
        for (int b = 0; b < grid; ++b) { bad += (h[2 * b] != (unsigned)(b % 8)); cnt[h[2 * b] & 15]++; }
        printf("grid %d: %d workgroups NOT on XCC b %% 8; per XCC:", grid, bad);
        for (int x = 0; x < 8; ++x) printf(" %d", cnt[x]);
        printf("\n first 32:");
        for (int b = 0; b < 32; ++b) printf(" %u", h[2 * b]);
        printf("\n");
    }
    printf("CUs %d\n", p.multiProcessorCount);
    return 0;
}
