// Probe (developer tool): which lanes of a wave64 ds_read_b128 are served in the same LDS clock on gfx950, and what row patterns of
// the penta-nucleotide row walk (dig_tiles_rows.hip) cost.  One workgroup of sixteen waves (four per SIMD: one wave issues an LDS instruction every ~20 clocks at most) reads per-lane addresses in
// a loop; reported: LDS clocks per wave-instruction (elapsed / instructions of all waves).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

template <int WIDTH>
__global__ __launch_bounds__(1024) void probe(const unsigned* addr, long long* cyc, double* sink, int iters)
{
    __shared__ __attribute__((aligned(16))) double s[8192];        // 64 KB
    for (int i = threadIdx.x; i < 8192; i += blockDim.x) s[i] = i;
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const unsigned a = addr[lane];
    double acc = 0.0;
    const long long t0 = (long long)__builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
        if (WIDTH == 16) {
            double2 v0, v1, v2, v3, v4, v5, v6, v7;
            asm volatile("ds_read_b128 %0, %8\n ds_read_b128 %1, %8\n ds_read_b128 %2, %8\n ds_read_b128 %3, %8\n"
                         "ds_read_b128 %4, %8\n ds_read_b128 %5, %8\n ds_read_b128 %6, %8\n ds_read_b128 %7, %8\n s_waitcnt lgkmcnt(0)"
                         : "=&v"(v0), "=&v"(v1), "=&v"(v2), "=&v"(v3), "=&v"(v4), "=&v"(v5), "=&v"(v6), "=&v"(v7)
                         : "v"(a));
            acc += v0.x + v7.y;
        } else if (WIDTH == 8) {
            double v0, v1, v2, v3, v4, v5, v6, v7;
            asm volatile("ds_read_b64 %0, %8\n ds_read_b64 %1, %8\n ds_read_b64 %2, %8\n ds_read_b64 %3, %8\n"
                         "ds_read_b64 %4, %8\n ds_read_b64 %5, %8\n ds_read_b64 %6, %8\n ds_read_b64 %7, %8\n s_waitcnt lgkmcnt(0)"
                         : "=&v"(v0), "=&v"(v1), "=&v"(v2), "=&v"(v3), "=&v"(v4), "=&v"(v5), "=&v"(v6), "=&v"(v7)
                         : "v"(a));
            acc += v0 + v7;
        } else {
            float v0, v1, v2, v3, v4, v5, v6, v7;
            asm volatile("ds_read_b32 %0, %8\n ds_read_b32 %1, %8\n ds_read_b32 %2, %8\n ds_read_b32 %3, %8\n"
                         "ds_read_b32 %4, %8\n ds_read_b32 %5, %8\n ds_read_b32 %6, %8\n ds_read_b32 %7, %8\n s_waitcnt lgkmcnt(0)"
                         : "=&v"(v0), "=&v"(v1), "=&v"(v2), "=&v"(v3), "=&v"(v4), "=&v"(v5), "=&v"(v6), "=&v"(v7)
                         : "v"(a));
            acc += v0 + v7;
        }
    }
    const long long t1 = (long long)__builtin_readcyclecounter();
    if (lane == 0) cyc[threadIdx.x >> 6] = t1 - t0;
    sink[threadIdx.x] = acc;
}

static unsigned* d_addr;
static long long* d_cyc;
static double* d_sink;

static int g_width = 16, g_waves = 16;
static double run(const std::vector<unsigned>& a)
{
    const int iters = 2000;
    (void)hipMemcpy(d_addr, a.data(), 64 * sizeof(unsigned), hipMemcpyHostToDevice);
    double best = 1e30;
    for (int rep = 0; rep < 3; ++rep) {
        if (g_width == 16) hipLaunchKernelGGL(probe<16>, dim3(1), dim3(64 * g_waves), 0, 0, d_addr, d_cyc, d_sink, iters);
        else if (g_width == 8) hipLaunchKernelGGL(probe<8>, dim3(1), dim3(64 * g_waves), 0, 0, d_addr, d_cyc, d_sink, iters);
        else hipLaunchKernelGGL(probe<4>, dim3(1), dim3(64 * g_waves), 0, 0, d_addr, d_cyc, d_sink, iters);
        (void)hipDeviceSynchronize();
        long long c[16];
        (void)hipMemcpy(c, d_cyc, sizeof(c), hipMemcpyDeviceToHost);
        long long mx = 0;
        for (int w = 0; w < g_waves; ++w) mx = c[w] > mx ? c[w] : mx;
        const double v = (double)mx / (iters * 8.0 * g_waves);
        best = v < best ? v : best;
    }
    return best;
}

static int slot_of(int lane)        // dig_tiles_rows.hip rw_slot
{
    const int l = lane & 31;
    int q;
    if (l < 4) q = l;
    else if (l < 12) q = l + 12;
    else if (l < 16) q = l - 8;
    else if (l < 20) q = l + 8;
    else if (l < 28) q = l - 12;
    else q = l;
    return q | (lane & 32);
}

int main()
{
    (void)hipMalloc(&d_addr, 64 * sizeof(unsigned));
    (void)hipMalloc(&d_cyc, 16 * sizeof(long long));
    (void)hipMalloc(&d_sink, 1024 * sizeof(double));
    std::vector<unsigned> a(64);
    for (int l = 0; l < 64; ++l) a[l] = 16 * l;
    printf("contiguous 1 KB: %.2f clocks per wave-instruction\n", run(a));
    for (int l = 0; l < 64; ++l) a[l] = 0;
    printf("all lanes one address: %.2f\n", run(a));
    for (int l = 0; l < 64; ++l) a[l] = 16 * (l & 7);
    printf("eight walkers, all row 0 (eight distinct addresses): %.2f\n", run(a));
    for (g_width = 4; g_width <= 16; g_width *= 2)
        for (g_waves = 1; g_waves <= 16; g_waves *= 4)
            for (int st = g_width; st <= 1024; st *= 2) {
                for (int l = 0; l < 64; ++l) a[l] = (st * l) % 65536;
                printf("width %d bytes, %d waves, lane stride %d bytes: %.2f\n", g_width, g_waves, st, run(a));
            }
    g_width = 16, g_waves = 16;
    // which lanes share a clock with lane i: lane j moved onto lane i's banks (another address)
    const int probes[6] = {0, 4, 12, 16, 20, 40};
    for (int pi = 0; pi < 6; ++pi) {
        const int i = probes[pi];
        for (int l = 0; l < 64; ++l) a[l] = 16 * l;
        const double base = run(a);
        printf("lanes sharing a clock with lane %d (alone %.2f):", i, base);
        for (int j = 0; j < 64; ++j) {
            if (j == i) continue;
            for (int l = 0; l < 64; ++l) a[l] = 16 * l;
            a[j] = a[i] + 1024 * (1 + (j & 7));
            const double v = run(a);
            if (v > base + 0.4) printf(" %d", j);
        }
        printf("\n");
    }
    // walker patterns: 8 lanes read a 128-byte row; rows[w] per walker
    srand(7);
    auto walkers = [&](bool permute, int mode) {
        double tot = 0.0;
        const int reps = mode == 3 ? 24 : 1;
        for (int r = 0; r < reps; ++r) {
            int rows[8];
            for (int w = 0; w < 8; ++w) {
                int row = rand() % 400;
                if (mode == 1) row = (row & ~1) | (w & 1);       // neighbours differ in parity
                if (mode == 2) row = row & ~1;                    // all even
                rows[w] = row;
            }
            for (int l = 0; l < 64; ++l) {
                const int s = permute ? slot_of(l) : l;
                a[l] = rows[s >> 3] * 128 + 16 * (s & 7);
            }
            tot += run(a);
        }
        return tot / reps;
    };
    for (int p = 0; p < 2; ++p)
        printf("8-lane walkers, 128-byte rows, %s lanes: neighbours of different parity %.2f | all rows even %.2f | random rows %.2f\n",
               p ? "permuted" : "natural", walkers(p, 1), walkers(p, 2), walkers(p, 3));
    // 16-lane walkers reading 256-byte rows (32 cohorts): conflict-free by construction?
    for (int l = 0; l < 64; ++l) a[l] = (rand() % 200) * 256;
    {
        int rows[4];
        for (int w = 0; w < 4; ++w) rows[w] = rand() % 200;
        for (int l = 0; l < 64; ++l) a[l] = rows[l >> 4] * 256 + 16 * (l & 15);
        printf("16-lane walkers, 256-byte rows, natural lanes: %.2f\n", run(a));
        for (int l = 0; l < 64; ++l) { const int s = slot_of(l); a[l] = rows[s >> 4] * 256 + 16 * (s & 15); }
        printf("16-lane walkers, 256-byte rows, permuted lanes: %.2f\n", run(a));
    }
    // lane = own random row, column skewed by lane (XOR): 256-byte rows -> always conflict-free?
    for (int trial = 0; trial < 2; ++trial) {
        for (int l = 0; l < 64; ++l) a[l] = (rand() % 200) * 256 + 16 * ((l ^ trial) & 15);
        printf("a row of 256 bytes per lane, column = lane mod 16: %.2f\n", run(a));
    }
    for (int l = 0; l < 64; ++l) a[l] = (rand() % 400) * 128 + 16 * (l & 7);
    printf("a row of 128 bytes per lane, column = lane mod 8: %.2f\n", run(a));
    for (int l = 0; l < 64; ++l) a[l] = (rand() % 800) * 64 + 16 * (l & 3);
    printf("a row of 64 bytes per lane, column = lane mod 4: %.2f\n", run(a));
    return 0;
}
