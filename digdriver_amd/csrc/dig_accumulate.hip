// dig_accumulate.hip -- per-element expected-count accumulation for all cohorts at once.
//
// Reference loops replaced (one Python iteration per element, ~6 k elements/s/process):
//   genic_driver_tools.py:300-431 nonc_model, :31-203 genic_model, :599-690 tiled_nonc_model,
//   driver_model/onthefly_tools.py:109-164 (loop body of DIG_onthefly).
//
// Mapping onto CDNA4 (workspace path, the one every caller of the package uses; kernels further down)
//   * acc_region_kernel     HBM-bound bin-table sums: (element, cohort) pairs flattened to 64-pair tiles for the
//                           [N, C] rate tables, 16 lanes x 16 bytes per element for the 256-byte context rows;
//   * acc_dot_mfma_kernel   the [E x 256] x [256 x C] FP64 product on the matrix cores (v_mfma_f64_16x16x4_f64),
//                           parameter table pre-swizzled in LDS, quotient + element sizes in registers.
// Without a workspace (NULL) a single LDS kernel does everything (accumulate_kernel below, ~3.5x slower):
//   * one 64-lane wave per element; a workgroup of W waves (W = 16 / 8 / 4 by LDS budget) walks
//     groups of W consecutive elements, one workgroup per CU, persistent grid;
//   * phase 1, lanes = the 64 trinucleotide contexts: the wave sums the 256-B context rows of the
//     element's overlapped bins and stages the reverse-complement-permuted counts and the 192 L counts of the
//     element as doubles in LDS; lanes = cohorts accumulate MU / VAR / R_OBS / FLAG in the same bin loop;
//   * phase 2, lanes = cohorts: d_pr[C,192] lives in LDS transposed ([192][C], conflict-free 8-byte reads) next
//     to its per-context sums; each lane runs the 64-term and 192-term dot products against LDS-broadcast counts;
//   * blockIdx -> element-group mapping is XCD-aware: workgroups that share an XCD (b % 8) walk one contiguous
//     eighth of the genome-ordered element list, so shared bin rows stay in that XCD's L2.
#include <algorithm>
#include <type_traits>
#include <vector>

#include "dig_common.hpp"
#include "dig_math.hpp"

namespace dig {

constexpr int kAccLdsBudget = 160 * 1024 - 1024;

struct AccArgs {
    const double *bin_mu, *bin_std;
    const int32_t* bin_y;
    const uint8_t* bin_flag;
    const int32_t* bin_ctx;
    const int64_t* ov_ptr;
    const int32_t* ov_idx;
    const int32_t* L;
    const uint8_t* strand_minus;
    const int32_t* gene_length;
    const double* d_pr;
    double *MU, *SIGMA;
    int32_t *R_OBS, *FLAG;
    double* P;
    int32_t *R_SIZE, *ELT_SIZE;
    double* P_INDEL;
    int64_t E, C;
    int c0, Cc;   // cohort chunk handled by this launch: lanes 0..Cc-1 <-> cohorts c0..c0+Cc-1
};

__device__ __forceinline__ int wave_sum_i32(int v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

// reverse complement of a context index (A=0,C=1,G=2,T=3; idx = 16 b0 + 4 b1 + b2)
__device__ __forceinline__ int revcomp_ctx(int c)
{
    const int b0 = c >> 4, b1 = (c >> 2) & 3, b2 = c & 3;
    return ((3 - b2) << 4) | ((3 - b1) << 2) | (3 - b0);
}

// A cohort whose table d_pr itself holds an entry that is not positive -- an exact zero (0 / 0 in t_pi = d_pr / 0), a NaN, a
// negative value (-inf beside +inf) -- gives NaN for a zero denominator whatever L is (genic_driver_tools.py:361-366; round 5:
// rounds 3-4 said inf when L held no zero).  Every kernel notes such cohorts of its chunk in LDS (g_cohort_bad) while it stages
// the table -- one compare per staged entry, an LDS atomic only when one is found -- and the zero-denominator path reads the note.
constexpr int kChunkCohorts = 64;       // (the matrix kernels take 48 cohorts per launch, the no-workspace kernel up to 64)
__shared__ unsigned g_cohort_bad[kChunkCohorts];

template <int NCLASS, int WAVES>
__global__ __launch_bounds__(WAVES * 64) void accumulate_kernel(AccArgs a)
{
    extern __shared__ double smem[];
    const int Cc = a.Cc;
    double* dprT = smem;                 // [192][Cc]
    double* d64T = dprT + 192 * Cc;      // [64][Cc]
    double* stage = d64T + 64 * Cc;      // [WAVES][NCLASS*192 + 64]
    constexpr int kStage = NCLASS * 192 + 64;

    const int tid = threadIdx.x;
    const int wave = tid >> 6, lane = tid & 63;

    // ---- stage the per-cohort trinucleotide parameters (once per workgroup) ----
    if (tid < kChunkCohorts) g_cohort_bad[tid] = 0u;
    __syncthreads();
    for (int idx = tid; idx < Cc * 192; idx += WAVES * 64) {
        const int c = idx / 192, j = idx - c * 192;
        const double v = a.d_pr[(int64_t)(a.c0 + c) * 192 + j];
        dprT[j * Cc + c] = v;
        if (!(v > 0.0)) atomicOr(&g_cohort_bad[c], 1u);
    }
    __syncthreads();
    for (int idx = tid; idx < Cc * 64; idx += WAVES * 64) {
        const int ctx = idx / Cc, c = idx - ctx * Cc;
        d64T[idx] = (dprT[(3 * ctx) * Cc + c] + dprT[(3 * ctx + 1) * Cc + c]) + dprT[(3 * ctx + 2) * Cc + c];
    }
    __syncthreads();

    double* st = stage + wave * kStage;
    const int64_t n_groups = (a.E + WAVES - 1) / WAVES;
    // XCD-aware walk: workgroups with equal blockIdx % 8 share one contiguous range of groups
    const int nx = (gridDim.x >= 8) ? 8 : 1;
    const int xcd = blockIdx.x % nx, slot = blockIdx.x / nx;
    const int nslots = (gridDim.x - xcd + nx - 1) / nx;
    const int64_t per_x = (n_groups + nx - 1) / nx;
    const int64_t g_begin = (int64_t)xcd * per_x;
    const int64_t g_end = (g_begin + per_x < n_groups) ? g_begin + per_x : n_groups;
    const int64_t trips = (per_x + nslots - 1) / nslots;

    const bool cohort_lane = lane < Cc;
    const int64_t col = a.c0 + lane;

    for (int64_t it = 0; it < trips; ++it) {
        const int64_t g = g_begin + slot + it * nslots;
        const int64_t e = g * WAVES + wave;
        const bool active = (g < g_end) && (e < a.E);

        double mu = 0.0, var = 0.0;
        int robs = 0, flag = 0, rsize = 0, lsum = 0;
        unsigned class_has_zero = 0;      // bit q: some L[e, q, j] == 0 (wave-uniform)
        if (active) {
            const int64_t q0 = a.ov_ptr[e], q1 = a.ov_ptr[e + 1];
            int rc = 0;
            for (int64_t q = q0; q < q1; ++q) {
                const int64_t b = a.ov_idx[q];
                rc += a.bin_ctx[b * 64 + lane];
                if (cohort_lane) {
                    const int64_t o = b * a.C + col;
                    const double sd = a.bin_std[o];
                    mu += a.bin_mu[o];            // genic_driver_tools.py:265
                    var += mul_rn(sd, sd);      // :266
                    robs += a.bin_y[o];           // :267
                    flag |= (a.bin_flag[o] != 0); // :268 (numpy bool '+' is a logical OR)
                }
            }
            rsize = wave_sum_i32(rc);
            const int dst = a.strand_minus[e] ? revcomp_ctx(lane) : lane;   // sequence_tools.py:633-634
            st[NCLASS * 192 + dst] = (double)rc;
            const int32_t* Le = a.L + e * (int64_t)(NCLASS * 192);
#pragma unroll
            for (int r = 0; r < NCLASS * 3; ++r) {
                const int v = Le[r * 64 + lane];
                st[r * 64 + lane] = (double)v;
                lsum += v;
                if (__any(v == 0)) class_has_zero |= 1u << (r / 3);
            }
            lsum = wave_sum_i32(lsum);
        }
        __syncthreads();
        if (active) {
            if (cohort_lane) {
                const double* rcs = st + NCLASS * 192;
                double d0 = 0.0, d1 = 0.0, d2 = 0.0, d3 = 0.0;
#pragma unroll 4
                for (int j = 0; j < 64; j += 4) {   // sum(region_counts * d_pr), :361
                    d0 = fma(rcs[j + 0], d64T[(j + 0) * Cc + lane], d0);
                    d1 = fma(rcs[j + 1], d64T[(j + 1) * Cc + lane], d1);
                    d2 = fma(rcs[j + 2], d64T[(j + 2) * Cc + lane], d2);
                    d3 = fma(rcs[j + 3], d64T[(j + 3) * Cc + lane], d3);
                }
                const double denom = (d0 + d1) + (d2 + d3);
                const int64_t o = e * a.C + col;
                a.MU[o] = mu;
                a.SIGMA[o] = sqrt(var);             // :271
                a.R_OBS[o] = robs;
                a.FLAG[o] = flag;
#pragma unroll
                for (int q = 0; q < NCLASS; ++q) {
                    const double* Ls = st + q * 192;
                    double n0 = 0.0, n1 = 0.0, n2 = 0.0, n3 = 0.0;
#pragma unroll 4
                    for (int j = 0; j < 192; j += 4) {   // sum(t_pi * L), :364-366
                        n0 = fma(Ls[j + 0], dprT[(j + 0) * Cc + lane], n0);
                        n1 = fma(Ls[j + 1], dprT[(j + 1) * Cc + lane], n1);
                        n2 = fma(Ls[j + 2], dprT[(j + 2) * Cc + lane], n2);
                        n3 = fma(Ls[j + 3], dprT[(j + 3) * Cc + lane], n3);
                    }
                    // (denominator 0: the reference forms t_pi = d_pr / 0 = inf first, and inf * 0 is NaN -- see fix_zero_denominators)
                    const double numer = (denom == 0.0 && (((class_has_zero >> q) & 1u) || g_cohort_bad[lane]))
                                             ? __longlong_as_double(0x7ff8000000000000ll)
                                                                                        : (n0 + n1) + (n2 + n3);
                    a.P[(e * NCLASS + q) * a.C + col] = numer / denom;
                }
            }
            if (lane == 0 && a.c0 == 0) {
                const int esize = lsum / 3;                               // :380
                a.R_SIZE[e] = rsize;                                      // :375
                a.ELT_SIZE[e] = esize;
                const double num = a.gene_length ? (double)a.gene_length[e] : (double)esize;
                a.P_INDEL[e] = num / (double)rsize;                       // :381 / :159
            }
        }
        __syncthreads();
    }
}

template <int NCLASS>
static int launch_acc_v1(const AccArgs& a, hipStream_t stream)
{
    const size_t fixed = (size_t)(192 + 64) * a.Cc * sizeof(double);
    const size_t per_wave = (size_t)(NCLASS * 192 + 64) * sizeof(double);
    int waves = 16;
    while (waves > 4 && fixed + per_wave * waves > (size_t)kAccLdsBudget) waves >>= 1;
    const size_t lds = fixed + per_wave * waves;
    if (lds > (size_t)kAccLdsBudget) return set_error(DIG_EINVAL, "accumulate: LDS budget exceeded (Cc=%d)", a.Cc);
    const int64_t n_groups = (a.E + waves - 1) / waves;
    int grid = cu_count();
    if ((int64_t)grid > n_groups) grid = (int)(n_groups > 0 ? n_groups : 1);
    auto go = [&](auto kern) -> int {
        DIG_HIP_TRY(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(kern, dim3(grid), dim3(waves * 64), lds, stream, a);
        DIG_HIP_TRY(hipGetLastError());
        return DIG_OK;
    };
    if (waves == 16) return go(accumulate_kernel<NCLASS, 16>);
    if (waves == 8) return go(accumulate_kernel<NCLASS, 8>);
    return go(accumulate_kernel<NCLASS, 4>);
}

// =======================================================================================
// Workspace path: two kernels.
//
//   acc_region_kernel    bin-table sums (HBM-bound; flattened (element, cohort) pairs + 16-lane context slices),
//                        writes MU / SIGMA / R_OBS / FLAG / R_SIZE and the strand-permuted context counts (workspace);
//   acc_dot_mfma_kernel  the [E x 256] x [256 x C] FP64 product on the matrix cores, quotient and element sizes.
// =======================================================================================
typedef double double4_t __attribute__((ext_vector_type(4)));
#ifndef DIG_MFMA_WAVES
#define DIG_MFMA_WAVES 12
#endif
// waves per workgroup (one workgroup per CU: the table takes <= 96 KB LDS): three per SIMD hide the LDS / HBM waits better
// than two (127.7 -> 124.2 us per accumulate call); four would cap the kernel at 128 VGPRs and spill
constexpr int kMfmaWaves = DIG_MFMA_WAVES;
#ifndef DIG_MFMA_MINBLOCKS
#define DIG_MFMA_MINBLOCKS 1
#endif
constexpr int kMfmaSteps = 64;                // 256 K rows / 4
constexpr int kMfmaChunk = 48;                // cohorts per launch (3 B tiles)

// How the (at most 48) cohort columns of a chunk are cut: full 16-column tiles for v_mfma_f64_16x16x4, and, when what is
// left over is 1..8 columns, one or two QUADS of four columns for v_mfma_f64_4x4x4 (four 4x4 blocks = the same sixteen
// elements; a quad costs a quarter of a tile's matrix-pipe time, so 37 cohorts pay for 40 columns instead of 48).
struct ChunkCut {
    int nt, nq;      // full tiles, tail quads
};
__host__ __device__ inline ChunkCut chunk_cut(int C, int chunk)
{
    const int rem = C - chunk * 48 < 48 ? C - chunk * 48 : 48;
    const int full = rem >> 4, r = rem & 15;
    if (r == 0) return {full, 0};
    if (r <= 8) return {full, (r + 3) >> 2};
    return {full + 1, 0};
}

// tab[chunk][step = 4 t + u][nt][lane] = T[kappa = 16 t + 4 (lane / 16) + u][chunk * 48 + nt * 16 + lane % 16],
// T = per-context sums (kappa < 64, d_pr[c][3 ctx .. 3 ctx + 2] summed as (a + b) + c) then d_pr[c][kappa - 64].
// The slot after the full tiles holds the tail quads: entry 16 q + 4 k + j = T[16 t + 4 k + u][first tail column + 4 q + j].
// (written by the first phase of acc_region_kernel, which precedes the dot kernel on the stream)
__device__ __forceinline__ void acc_write_mfma_table(const double* __restrict__ d_pr, double* __restrict__ tab, int C,
                                                     int nchunk, int64_t first, int64_t step_)
{
    const int n = nchunk * kMfmaSteps * 3 * 64;
    for (int64_t idx64 = first; idx64 < n; idx64 += step_) {
        const int idx = (int)idx64;
        const int lane = idx & 63, nt = (idx >> 6) % 3, step = (idx / 192) % kMfmaSteps, chunk = idx / (192 * kMfmaSteps);
        const ChunkCut cut = chunk_cut(C, chunk);
        const bool tail = cut.nq > 0 && nt == cut.nt;
        const int kappa = 16 * (step >> 2) + 4 * (tail ? (lane >> 2) & 3 : lane >> 4) + (step & 3);
        const int c = chunk * kMfmaChunk + nt * 16 + (tail ? 4 * (lane >> 4) + (lane & 3) : lane & 15);
        double v = 0.0;
        if (c < C && (!tail || (lane >> 4) < cut.nq)) {
            const double* d = d_pr + (int64_t)c * 192;
            v = (kappa < 64) ? (d[3 * kappa] + d[3 * kappa + 1]) + d[3 * kappa + 2] : d[kappa - 64];
        }
        tab[idx] = v;
    }
}

constexpr int kRegionBlock = 256;

// acc_region_kernel: persistent grid, two independent phases per wave (no barriers, no LDS).
//   phase A (rates)    the (element, cohort) pairs are flattened to g = e * C + c and walked in 64-pair tiles:
//                      every lane is busy whatever C is, consecutive lanes read C*8 contiguous bytes of the [N, C]
//                      bin tables and the four outputs are written as fully coalesced rows of g.  The per-pair sums run
//                      over the element's bins in CSR order (the order of genic_driver_tools.py:262-268).
//   phase B (contexts) 16 lanes per element, one 16-byte slice of the 256-byte context row each: four elements per
//                      wave step, so the dependent ov_ptr -> ov_idx -> row chain is paid once per four elements.
//                      The strand permutation (sequence_tools.py:633-634) is applied on the store.
__global__ __launch_bounds__(kRegionBlock) void acc_region_kernel(
    const double* __restrict__ bin_mu, const double* __restrict__ bin_std, const int32_t* __restrict__ bin_y,
    const uint8_t* __restrict__ bin_flag, const int32_t* __restrict__ bin_ctx, const int64_t* __restrict__ ov_ptr,
    const int32_t* __restrict__ ov_idx, const uint8_t* __restrict__ strand_minus, double* __restrict__ MU,
    double* __restrict__ SIGMA, int32_t* __restrict__ R_OBS, int32_t* __restrict__ FLAG, int32_t* __restrict__ R_SIZE,
    int32_t* __restrict__ rcp, int64_t E, int64_t C, FastDiv divC, int use_fastdiv, const double* __restrict__ d_pr,
    double* __restrict__ tab, int n48, int do_rates, unsigned* __restrict__ zero_dwords, int n_zero)
{
    const int lane = threadIdx.x & 63;
    const int64_t wave0 = ((int64_t)blockIdx.x * kRegionBlock + threadIdx.x) >> 6;
    const int64_t nwaves = ((int64_t)gridDim.x * kRegionBlock) >> 6;

    // (pipeline only) clear the worklist header of the statistics stage that follows on the stream
    if (zero_dwords && blockIdx.x == 0 && (int)threadIdx.x < n_zero) zero_dwords[threadIdx.x] = 0u;

    // phase 0: the pre-swizzled parameter table of the dot stage (a few hundred KB, once per call)
    acc_write_mfma_table(d_pr, tab, (int)C, n48, (int64_t)blockIdx.x * kRegionBlock + threadIdx.x,
                         (int64_t)gridDim.x * kRegionBlock);

    const int64_t n = E * C;
    const int64_t n_tiles = do_rates ? (n + 63) >> 6 : 0;   // dig_element_pipeline forms the rate sums in its statistics kernel
    for (int64_t tile = wave0; tile < n_tiles; tile += nwaves) {
        const int64_t g = tile * 64 + lane;
        if (g < n) {
            const int64_t e = use_fastdiv ? fastdiv(g, divC) : g;
            const int64_t c = g - e * C;
            const int64_t q0 = ov_ptr[e], q1 = ov_ptr[e + 1];
            double mu = 0.0, var = 0.0;
            int robs = 0, flag = 0;
            for (int64_t q = q0; q < q1; ++q) {
                const int64_t o = (int64_t)ov_idx[q] * C + c;
                const double sd = bin_std[o];
                mu += bin_mu[o];                 // :265
                var += mul_rn(sd, sd);         // :266
                robs += bin_y[o];                // :267
                flag |= (bin_flag[o] != 0);      // :268 (numpy bool '+' is a logical OR)
            }
            MU[g] = mu;
            SIGMA[g] = sqrt(var);                // :271
            R_OBS[g] = robs;
            FLAG[g] = flag;
        }
    }

    // phase B, software-pipelined over the wave's quads: the chain CSR bounds -> bin indices -> context rows is three
    // dependent memory latencies, and a wave has only three or four quads; so while the rows of quad t are in flight the
    // bin indices of quad t + nwaves and the CSR bounds of quad t + 2 nwaves are requested (first two bins of an element
    // ahead of time; further bins, rare, in the plain loop).
    const int sub = lane >> 4, l16 = lane & 15;
    const int4* ctx4 = reinterpret_cast<const int4*>(bin_ctx);
    const int64_t n_quads = (E + 3) >> 2;
    struct Bounds { int64_t q0, q1; int minus; };
    struct First { int i0, i1; };
    auto load_bounds = [&](int64_t t) {
        Bounds r{0, 0, 0};
        const int64_t e = t * 4 + sub;
        if (t < n_quads && e < E) {   // uniform within each 16-lane group
            r.q0 = ov_ptr[e];
            r.q1 = ov_ptr[e + 1];
            r.minus = strand_minus[e];
        }
        return r;
    };
    auto load_first = [&](const Bounds& r) {
        First f{-1, -1};
        if (r.q1 > r.q0) f.i0 = ov_idx[r.q0];
        if (r.q1 > r.q0 + 1) f.i1 = ov_idx[r.q0 + 1];
        return f;
    };
    int64_t t = wave0;
    Bounds b0 = load_bounds(t), b1 = load_bounds(t + nwaves);
    First f0 = load_first(b0);
    for (; t < n_quads; t += nwaves) {
        const Bounds b2 = load_bounds(t + 2 * nwaves);
        const First f1 = load_first(b1);
        const int64_t e = t * 4 + sub;
        if (e < E) {
            int4 acc = make_int4(0, 0, 0, 0);
            if (f0.i0 >= 0) {
                const int4 v = ctx4[(int64_t)f0.i0 * 16 + l16];
                acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
            }
            if (f0.i1 >= 0) {
                const int4 v = ctx4[(int64_t)f0.i1 * 16 + l16];
                acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
            }
            for (int64_t q = b0.q0 + 2; q < b0.q1; ++q) {
                const int4 v = ctx4[(int64_t)ov_idx[q] * 16 + l16];
                acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
            }
            int s = (acc.x + acc.y) + (acc.z + acc.w);
#pragma unroll
            for (int off = 8; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
            if (l16 == 0) R_SIZE[e] = s;                              // genic_driver_tools.py:375
            if (!b0.minus) {
                reinterpret_cast<int4*>(rcp)[e * 16 + l16] = acc;
            } else {                                                  // sequence_tools.py:633-634
                int32_t* row = rcp + e * 64;
                const int c0 = 4 * l16;
                row[revcomp_ctx(c0 + 0)] = acc.x;
                row[revcomp_ctx(c0 + 1)] = acc.y;
                row[revcomp_ctx(c0 + 2)] = acc.z;
                row[revcomp_ctx(c0 + 3)] = acc.w;
            }
        }
        b0 = b1;
        b1 = b2;
        f0 = f1;
    }
}

// =======================================================================================
// v3 dot stage on the FP64 matrix cores.
//
// P[e, c] = sum_j L[e, j] d_pr[c, j] / sum_ctx rc[e, ctx] d64[c, ctx] is a [E x 256] x [256 x C] product in FP64 --
// the one GEMM-shaped step of the accumulation.  v_mfma_f64_16x16x4_f64 runs at the FP64 vector rate on gfx950, but
// each operand register feeds 16 FMAs, which removes the one-parameter-fetch-per-FMA limit of the scalar-operand
// form tried first (one lane per element, parameters as SGPR operands: 94 us, bound by scalar-cache misses on the
// 80 KB parameter table; this kernel: 65 us).
//
//   * operand layout (probed on the device, tools/probe/mfma_f64_layout.hip): A[i][k] in lane 16 k + i, B[k][j] in
//     lane 16 k + j, D[i][j] in lane 16 (i % 4) + j, register i / 4;
//   * a wave owns 16 consecutive elements (one A tile) and up to 48 cohorts (NT 16-column B tiles), 16 waves per
//     workgroup (4 per SIMD): lane (i, k) reads its element's count rows 16 bytes at a time (ints 16 t + 4 k .. + 3),
//     converts each int once and uses it for NT MFMAs; the K order inside a 16-int group is permuted accordingly
//     (a sum, so free); loads run one group of four slices (48 MFMAs) ahead;
//   * the parameter table is staged once per workgroup in LDS, pre-swizzled (acc_write_mfma_table) into the exact
//     per-lane order of the B operand (tab[step][nt][lane]): every ds_read_b64 is 512 contiguous bytes;
//   * denominators (64 context rows) and numerators (192 substitution rows) use separate accumulators; the quotient
//     and the integer element sizes are formed in registers and written once.
// =======================================================================================
template <int NT, int NQ>
__device__ __forceinline__ void mfma_group(const int4 (&a)[4], const double* __restrict__ tab, int step0, int lane,
                                           int toff, double4_t (&acc)[NT > 0 ? NT : 1], double (&accq)[NQ > 0 ? NQ : 1],
                                           int& isum, int& anyzero)
{
    constexpr int SL = NT + (NQ > 0 ? 1 : 0);     // LDS slots per step
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const int v[4] = {a[t].x, a[t].y, a[t].z, a[t].w};
        isum += (a[t].x + a[t].y) + (a[t].z + a[t].w);
        anyzero |= (a[t].x == 0) | (a[t].y == 0) | (a[t].z == 0) | (a[t].w == 0);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const double A = (double)v[u];
            const double* b = tab + ((step0 + 4 * t + u) * SL) * 64 + lane;
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
                acc[nt] = __builtin_amdgcn_mfma_f64_16x16x4f64(A, b[nt * 64], acc[nt], 0, 0, 0);
            if constexpr (NQ > 0) {
                // A[i][k] of block b sits in lane 16 k + 4 b + i: the same register as for the 16-row tile
                const double* bq = tab + ((step0 + 4 * t + u) * SL + NT) * 64 + toff;
#pragma unroll
                for (int q = 0; q < NQ; ++q)
                    accq[q] = __builtin_amdgcn_mfma_f64_4x4x4f64(A, bq[q * 16], accq[q], 0, 0, 0);
            }
        }
    }
}

// sum(region_counts * d_pr) == 0 (an element whose bins hold no countable context, or no bin at all): the reference forms
// t_pi = d_pr / 0 = inf FIRST and then sum(t_pi * L) (genic_driver_tools.py:361-366), which is NaN as soon as one L[j] is 0
// (inf * 0) and inf otherwise; numerator / 0 alone would say inf in both cases.  `lzero`: this lane's slices of the row's L
// hold a zero.  The numerators of such rows are replaced so that the quotient that follows gives the reference's value.
// Rare: one wave-uniform test per tile in front of it.
template <int NT, int NQ>
__device__ __forceinline__ void fix_zero_denominators(const double4_t (&den)[NT > 0 ? NT : 1], const double (&denq)[NQ > 0 ? NQ : 1],
                                                      double4_t (&num)[NT > 0 ? NT : 1], double (&numq)[NQ > 0 ? NQ : 1],
                                                      int lzero, int lane)
{
    bool z = false;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int r = 0; r < 4; ++r) z |= den[nt][r] == 0.0;
#pragma unroll
    for (int q = 0; q < NQ; ++q) z |= denq[q] == 0.0;
    if (!__any(z)) return;
    lzero |= __shfl_xor(lzero, 16, 64);                                  // the four k-lanes of a row together hold the row
    lzero |= __shfl_xor(lzero, 32, 64);
    const int kq = lane >> 4;
    const double nan = __longlong_as_double(0x7ff8000000000000ll);
#pragma unroll
    for (int r = 0; r < 4; ++r) {                                        // D[i][j]: row 4 r + kq
        const int rowzero = __shfl(lzero, 4 * r + kq, 64);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
            if (den[nt][r] == 0.0 && (rowzero || g_cohort_bad[nt * 16 + (lane & 15)])) num[nt][r] = nan;
    }
    if constexpr (NQ > 0) {
        const int rowzero = __shfl(lzero, 4 * ((lane >> 2) & 3) + kq, 64);   // D[i][j] of block b: row 4 b + i
#pragma unroll
        for (int q = 0; q < NQ; ++q)
            if (denq[q] == 0.0 && (rowzero || g_cohort_bad[NT * 16 + 4 * q + (lane & 3)])) numq[q] = nan;
    }
}

template <int NCLASS, int NT, int NQ>
__global__ __launch_bounds__(kMfmaWaves * 64, DIG_MFMA_MINBLOCKS) void acc_dot_mfma_kernel(
    const int32_t* __restrict__ rcp, const int32_t* __restrict__ L, const double* __restrict__ tab_g,
    const int32_t* __restrict__ R_SIZE, const int32_t* __restrict__ gene_length, double* __restrict__ P,
    int32_t* __restrict__ ELT_SIZE, double* __restrict__ P_INDEL, int64_t E, int C, int c0, int write_sizes,
    const double* __restrict__ d_pr)
{
    extern __shared__ double tab[];           // [kMfmaSteps][SL][64]
    constexpr int SL = NT + (NQ > 0 ? 1 : 0);
    constexpr int NTA = NT > 0 ? NT : 1, NQA = NQ > 0 ? NQ : 1;
    if (threadIdx.x < kChunkCohorts) g_cohort_bad[threadIdx.x] = 0u;
    __syncthreads();
    {   // the chunk's cohorts whose frequency table holds an entry that is not positive (see g_cohort_bad): the table this kernel
        // stages is the pre-swizzled one, so the raw frequencies are looked at here (9 216 values per workgroup, L2-resident)
        const int cc = min(C - c0, kMfmaChunk);
        for (int idx = threadIdx.x; idx < cc * 192; idx += kMfmaWaves * 64)
            if (!(d_pr[(int64_t)c0 * 192 + idx] > 0.0)) atomicOr(&g_cohort_bad[idx / 192], 1u);
    }
    {
        // stage this chunk's NT tiles of the pre-swizzled table (global layout: 3 tiles per step); all loads of a
        // thread are issued before its first LDS write
        constexpr int kTotal = kMfmaSteps * SL * 64;
        constexpr int kPer = (kTotal + kMfmaWaves * 64 - 1) / (kMfmaWaves * 64);   // doubles per thread
        double v[kPer];
#pragma unroll
        for (int k = 0; k < kPer; ++k) {
            const int idx = threadIdx.x + k * kMfmaWaves * 64;
            const int lane_ = idx & 63, nt = (idx >> 6) % SL, step = idx / (SL * 64);
            v[k] = idx < kTotal ? tab_g[(step * 3 + nt) * 64 + lane_] : 0.0;
        }
#pragma unroll
        for (int k = 0; k < kPer; ++k)
            if (threadIdx.x + k * kMfmaWaves * 64 < kTotal) tab[threadIdx.x + k * kMfmaWaves * 64] = v[k];
    }
    __syncthreads();

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i = lane & 15, kq = lane >> 4;
    const int toff = 4 * kq + (lane & 3);     // B[k][j] of every block of a quad: lane 16 k + 4 b + j
    const int64_t n_tiles = (E + 15) >> 4;
    const int64_t n_waves = (int64_t)gridDim.x * kMfmaWaves;
    constexpr int G = 1 + 3 * NCLASS;         // 16-row groups per tile: contexts, then 3 per mutation class

    // slice t of group gi of a tile (rows past the end replay row E-1 and are never stored)
    auto slice_ptr = [&](int64_t tile_, int gi) -> const int4* {
        const int64_t row = min(tile_ * 16 + i, E - 1);
        return gi == 0 ? reinterpret_cast<const int4*>(rcp + row * 64) + kq
                       : reinterpret_cast<const int4*>(L + row * NCLASS * 192) + (gi - 1) * 16 + kq;
    };
    // round r of wave w takes tile r * n_waves + (w's rank with workgroups interleaved), so a partial last round is
    // spread over all CUs instead of filling the first workgroups only
    const int64_t rank = (int64_t)wave * gridDim.x + blockIdx.x;
    int64_t tile = rank;
    if (tile >= n_tiles) return;
    int4 cur[4], nxt[4];
    {
        const int4* p0 = slice_ptr(tile, 0);
#pragma unroll
        for (int t = 0; t < 4; ++t) nxt[t] = p0[4 * t];
    }
    while (tile < n_tiles) {
        // The table reads are invariant across tiles; without this opaque zero the compiler hoists all 192 of them out
        // of the persistent loop and spills.
        int opaque_zero;
        asm volatile("s_mov_b32 %0, 0" : "=s"(opaque_zero));
        const double* tabw = tab + opaque_zero;
        const int64_t e0 = tile * 16;
        const int64_t tile_next = tile + n_waves;
        double4_t den[NTA];
        double denq[NQA];
        int rsum = 0, lsum = 0;
        // group 0: the 64 context rows -> denominators
#pragma unroll
        for (int t = 0; t < 4; ++t) cur[t] = nxt[t];
        {
            const int4* pn = slice_ptr(tile, 1);
#pragma unroll
            for (int t = 0; t < 4; ++t) nxt[t] = pn[4 * t];
        }
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) den[nt] = double4_t{0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int q = 0; q < NQ; ++q) denq[q] = 0.0;
        int unused_zero = 0;
        mfma_group<NT, NQ>(cur, tabw, 0, lane, toff, den, denq, rsum, unused_zero);     // sum(region_counts * d_pr), genic_driver_tools.py:361
#pragma unroll 1
        for (int q = 0; q < NCLASS; ++q) {
            int opaque_zero_q;                                           // (same hoisting guard, per class)
            asm volatile("s_mov_b32 %0, 0" : "=s"(opaque_zero_q));
            const double* tabq = tabw + opaque_zero_q;
            double4_t num[NTA];
            double numq[NQA];
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) num[nt] = double4_t{0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int qq = 0; qq < NQ; ++qq) numq[qq] = 0.0;
            int lzero = 0;                                               // some L[e, q, j] == 0 (this lane's slices)
#pragma unroll
            for (int g = 0; g < 3; ++g) {                                // sum(t_pi * L), :364-366
#pragma unroll
                for (int t = 0; t < 4; ++t) cur[t] = nxt[t];
                {   // request the next group of the stream (first group of the next tile after the last one)
                    const int gi_next = 2 + 3 * q + g;
                    const int4* pn = (gi_next < G) ? slice_ptr(tile, gi_next) : slice_ptr(min(tile_next, n_tiles - 1), 0);
#pragma unroll
                    for (int t = 0; t < 4; ++t) nxt[t] = pn[4 * t];
                }
                mfma_group<NT, NQ>(cur, tabq, 16 + 16 * g, lane, toff, num, numq, lsum, lzero);
            }
            fix_zero_denominators<NT, NQ>(den, denq, num, numq, lzero, lane);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int64_t e = e0 + 4 * r + kq;                      // D[i][j]: lane 16 (i % 4) + j, register i / 4
                if (e < E) {
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) {
                        const int c = c0 + nt * 16 + i;
                        if (c < C) P[(e * NCLASS + q) * C + c] = num[nt][r] / den[nt][r];
                    }
                }
            }
            if constexpr (NQ > 0) {
                const int64_t e = e0 + 4 * ((lane >> 2) & 3) + kq;      // D[i][j] of block b: lane 16 i + 4 b + j
                if (e < E) {
#pragma unroll
                    for (int qq = 0; qq < NQ; ++qq) {
                        const int c = c0 + NT * 16 + 4 * qq + (lane & 3);
                        if (c < C) P[(e * NCLASS + q) * C + c] = numq[qq] / denq[qq];
                    }
                }
            }
        }
        if (write_sizes) {
            lsum += __shfl_xor(lsum, 16, 64);
            lsum += __shfl_xor(lsum, 32, 64);
            const int64_t e = e0 + i;
            if (kq == 0 && e < E) {
                const int esize = lsum / 3;                                          // :380
                ELT_SIZE[e] = esize;
                const double numer = gene_length ? (double)gene_length[e] : (double)esize;
                P_INDEL[e] = numer / (double)R_SIZE[e];                              // :381 / :159
            }
        }
        tile = tile_next;
    }
}

// =======================================================================================
// Compact form: context-repeated L (every elementModel / tiledModel / quickDriver set).
//
// sequence_tools.py:560-564 builds the L_counts of an element as its 64 trinucleotide context counts, each written to the
// three substitutions of that context (adjacent in sorted "XYZ>XaZ" order).  Then
//     sum_j L[e, j] d_pr[c, j] = sum_ctx Lc[e, ctx] d64[c, ctx],   d64[c, ctx] = (d_pr[c, 3 ctx] + d_pr[c, 3 ctx + 1]) + d_pr[c, 3 ctx + 2]
// -- the numerator runs over the SAME 64 per-context sums the denominator sum(region_counts * d_pr) uses, so the
// [E x 256] x [256 x C] product collapses to two [E x 64] x [64 x C] products that share their B operand: half the matrix
// instructions, a quarter of the LDS table (24 KB: built by every workgroup straight from d_pr, no table kernel), and
// L is read as [E, 64] (compacted once per plan by compact_L_kernel, which also verifies the repetition).
// The context stage lives in the same kernel: lane (i, k) of a wave gathers the 16-byte slices 4 t + k of the context
// rows of its element's bins itself (CSR bounds two tiles ahead, bin indices one tile ahead, rows half a tile ahead: the
// loads of the main path are unconditional so that the memory counter stays exact), sums them as integers and feeds
// them to the matrix pipe; the '-' strand permutation (sequence_tools.py:633-634) is a register choice: the lane reads
// slice 4 t' + (3 - k) instead and takes component 3 - t of slice 3 - u where the '+' strand takes component u of slice t
// (revcomp(16 t + 4 k + u) = 16 (3 - u) + 4 (3 - k) + (3 - t)).  No context rows cross HBM between two kernels any more.
// =======================================================================================
#ifndef DIG_CTX_WAVES
#define DIG_CTX_WAVES 12
#endif
constexpr int kCtxWaves = DIG_CTX_WAVES;
constexpr int kCtxSteps = 16;                 // 64 context rows / 4

__global__ __launch_bounds__(256) void compact_L_kernel(const int32_t* __restrict__ L, int64_t n64, int32_t* __restrict__ Lc,
                                                        int* __restrict__ mismatch)
{
    const int64_t stride = (int64_t)gridDim.x * 256;
    int bad = 0;
    for (int64_t g = (int64_t)blockIdx.x * 256 + threadIdx.x; g < n64; g += stride) {
        const int32_t* p = L + 3 * g;                          // [E, 192] = [E * 64, 3]
        const int a = p[0], b = p[1], c = p[2];
        Lc[g] = a;
        bad |= (a != b) | (a != c);
    }
    if (__any(bad) && (threadIdx.x & 63) == 0) atomicOr(mismatch, 1);
}

template <int NT, int NQ>
__global__ __launch_bounds__(kCtxWaves * 64) void acc_dot_ctx_kernel(
    const int32_t* __restrict__ bin_ctx, const int64_t* __restrict__ ov_ptr, const int32_t* __restrict__ ov_idx,
    const uint8_t* __restrict__ strand_minus, const int32_t* __restrict__ Lc, const double* __restrict__ d_pr,
    const int32_t* __restrict__ gene_length, double* __restrict__ P, int32_t* __restrict__ R_SIZE,
    int32_t* __restrict__ ELT_SIZE, double* __restrict__ P_INDEL, int64_t E, int C, int c0, int write_sizes,
    unsigned* __restrict__ zero_dwords, int n_zero)
{
    constexpr int SL = NT + (NQ > 0 ? 1 : 0);
    constexpr int NTA = NT > 0 ? NT : 1, NQA = NQ > 0 ? NQ : 1;
    __shared__ double tab[kCtxSteps * SL * 64];       // tab[step = 4 t + u][slot][lane], as acc_write_mfma_table lays it out

    // (pipeline only) clear the worklist header of the statistics stage that follows on the stream
    if (zero_dwords && blockIdx.x == 0 && (int)threadIdx.x < n_zero) zero_dwords[threadIdx.x] = 0u;
    if (threadIdx.x < kChunkCohorts) g_cohort_bad[threadIdx.x] = 0u;
    __syncthreads();

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i = lane & 15, kq = lane >> 4;
    const int toff = 4 * kq + (lane & 3);     // B[k][j] of every block of a quad: lane 16 k + 4 b + j
    const int64_t n_tiles = (E + 15) >> 4;
    // XCD-aware walk: workgroups are dealt round-robin to the 8 XCDs; the workgroups of one XCD (equal blockIdx % 8) take one
    // contiguous eighth of the genome-ordered tiles, so that the bin rows neighbouring elements share stay in that L2
    const int nx = (gridDim.x & 7u) == 0u ? 8 : 1;
    const int xcd = blockIdx.x % nx, slot_b = blockIdx.x / nx, nslots = gridDim.x / nx;
    const int64_t per_x = (n_tiles + nx - 1) / nx;
    const int64_t t_begin = (int64_t)xcd * per_x;
    const int64_t t_end = t_begin + per_x < n_tiles ? t_begin + per_x : n_tiles;
    const int64_t stride = (int64_t)nslots * kCtxWaves;
    int64_t tile = t_begin + (int64_t)wave * nslots + slot_b;    // wave-major: a partial last round is spread over all CUs

    const int4* ctx4 = reinterpret_cast<const int4*>(bin_ctx);
    const int4* L4 = reinterpret_cast<const int4*>(Lc);
    // every element's CSR range lies inside [0, nnz); with an empty CSR the index loads replay ov_ptr[0] (= 0: row 0)
    const int64_t nnz = ov_ptr[E];
    const int32_t* oi_base = nnz > 0 ? ov_idx : reinterpret_cast<const int32_t*>(ov_ptr);
    const int64_t oi_last = nnz > 0 ? nnz - 1 : 0;

    struct Bounds { int64_t q0, row; int cnt, minus; };
    struct Idx { int i0, i1; };
    struct Rows { int4 a[4], b[4], l[4]; };
    auto load_bounds = [&](int64_t tile_) {          // tiles / rows past the end replay the last row (never stored)
        Bounds r;
        r.row = min(min(tile_, n_tiles - 1) * 16 + i, E - 1);
        r.q0 = ov_ptr[r.row];
        r.cnt = (int)(ov_ptr[r.row + 1] - r.q0);
        r.minus = strand_minus[r.row];
        return r;
    };
    auto load_idx = [&](const Bounds& b) {           // unconditional: bins an element does not have replay a valid entry
        Idx x;
        x.i0 = oi_base[min(b.q0, oi_last)];
        x.i1 = oi_base[min(b.q0 + 1, oi_last)];
        return x;
    };
    auto load_rows = [&](const Bounds& b, const Idx& x) {
        Rows r;
        const int kk = b.minus ? 3 - kq : kq;
        const int4* r0 = ctx4 + (int64_t)x.i0 * 16 + kk;
        const int4* r1 = ctx4 + (int64_t)x.i1 * 16 + kk;
        const int4* rl = L4 + b.row * 16 + kq;
#pragma unroll
        for (int t = 0; t < 4; ++t) r.a[t] = r0[4 * t];
#pragma unroll
        for (int t = 0; t < 4; ++t) r.b[t] = r1[4 * t];
#pragma unroll
        for (int t = 0; t < 4; ++t) r.l[t] = rl[4 * t];
        return r;
    };

    // pipeline fill (in flight while the table is staged)
    Bounds b_c = load_bounds(tile), b_n = load_bounds(tile + stride);
    Idx x_c = load_idx(b_c);

    // stage the B operand: per-context sums of d_pr in the per-lane order of the matrix instructions
    {   // (all loads of a thread are issued before its first LDS write: a plain loop waits for every entry's loads in turn,
        //  four L2 round trips at the start of every workgroup)
        constexpr int kTotal = kCtxSteps * SL * 64;
        constexpr int kPer = (kTotal + kCtxWaves * 64 - 1) / (kCtxWaves * 64);
        double d0[kPer], d1[kPer], d2[kPer];
#pragma unroll
        for (int j = 0; j < kPer; ++j) {
            const int idx = threadIdx.x + j * kCtxWaves * 64;
            const int lane_ = idx & 63, slot = (idx >> 6) % SL, step = idx / (SL * 64);
            const bool tail = NQ > 0 && slot == NT;
            const int kappa = 16 * (step >> 2) + 4 * (tail ? (lane_ >> 2) & 3 : lane_ >> 4) + (step & 3);
            const int c = c0 + slot * 16 + (tail ? 4 * (lane_ >> 4) + (lane_ & 3) : lane_ & 15);
            d0[j] = d1[j] = d2[j] = 0.0;
            if (idx < kTotal && c < C && (!tail || (lane_ >> 4) < NQ)) {
                const double* d = d_pr + (int64_t)c * 192 + 3 * kappa;
                d0[j] = d[0];
                d1[j] = d[1];
                d2[j] = d[2];
                if (!(d0[j] > 0.0 && d1[j] > 0.0 && d2[j] > 0.0)) atomicOr(&g_cohort_bad[c - c0], 1u);
            }
        }
#pragma unroll
        for (int j = 0; j < kPer; ++j) {
            const int idx = threadIdx.x + j * kCtxWaves * 64;
            if (idx < kTotal) tab[idx] = (d0[j] + d1[j]) + d2[j];
        }
    }
    Rows r_c = load_rows(b_c, x_c);
    __syncthreads();
    if (tile >= t_end) return;

    while (tile < t_end) {
        // (the table reads are invariant across tiles: without the opaque zero the compiler hoists them out of the loop and spills)
        int opaque_zero;
        asm volatile("s_mov_b32 %0, 0" : "=s"(opaque_zero));
        const double* tabw = tab + opaque_zero;
        const int64_t e0 = tile * 16;
        const Bounds b_nn = load_bounds(tile + 2 * stride);      // tile t+2: CSR bounds
        const Idx x_n = load_idx(b_n);                           // tile t+1: first two bin indices
        // region counts of the element: sum of its bins' context rows (sequence_tools.py:630-631), integers
        int rcv[4][4], lv[4][4];
        {
            const int m0 = b_c.cnt > 0 ? -1 : 0, m1 = b_c.cnt > 1 ? -1 : 0;
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                rcv[t][0] = (r_c.a[t].x & m0) + (r_c.b[t].x & m1);
                rcv[t][1] = (r_c.a[t].y & m0) + (r_c.b[t].y & m1);
                rcv[t][2] = (r_c.a[t].z & m0) + (r_c.b[t].z & m1);
                rcv[t][3] = (r_c.a[t].w & m0) + (r_c.b[t].w & m1);
                lv[t][0] = r_c.l[t].x; lv[t][1] = r_c.l[t].y; lv[t][2] = r_c.l[t].z; lv[t][3] = r_c.l[t].w;
            }
        }
        if (__any(b_c.cnt > 2)) {                                // elements over more than two bins (multi-block, gene-sized)
            const int kk = b_c.minus ? 3 - kq : kq;
            for (int j = 2; j < b_c.cnt; ++j) {
                const int4* r = ctx4 + (int64_t)ov_idx[b_c.q0 + j] * 16 + kk;
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const int4 v = r[4 * t];
                    rcv[t][0] += v.x; rcv[t][1] += v.y; rcv[t][2] += v.z; rcv[t][3] += v.w;
                }
            }
        }
        double4_t den[NTA], num[NTA];
        double denq[NQA], numq[NQA];
#pragma unroll
        for (int nt = 0; nt < NTA; ++nt) den[nt] = num[nt] = double4_t{0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int q = 0; q < NQA; ++q) denq[q] = numq[q] = 0.0;
        int rsum = 0, lsum = 0, lzero = 0;
        const bool minus = b_c.minus != 0;
        auto steps = [&](auto T0) {
            constexpr int t0 = decltype(T0)::value;
#pragma unroll
            for (int t = t0; t < t0 + 2; ++t) {
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int vr = minus ? rcv[3 - u][3 - t] : rcv[t][u];
                    rsum += rcv[t][u];
                    lsum += lv[t][u];
                    lzero |= lv[t][u] == 0;
                    const double Ar = (double)vr, Al = (double)lv[t][u];
                    const double* b = tabw + ((4 * t + u) * SL) * 64 + lane;
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) {
                        const double bv = b[nt * 64];
                        den[nt] = __builtin_amdgcn_mfma_f64_16x16x4f64(Ar, bv, den[nt], 0, 0, 0);   // sum(region_counts * d_pr), :361
                        num[nt] = __builtin_amdgcn_mfma_f64_16x16x4f64(Al, bv, num[nt], 0, 0, 0);   // sum(d_pr * L), :364-366
                    }
                    if constexpr (NQ > 0) {
                        const double* bq = tabw + ((4 * t + u) * SL + NT) * 64 + toff;
#pragma unroll
                        for (int q = 0; q < NQ; ++q) {
                            const double bv = bq[q * 16];
                            denq[q] = __builtin_amdgcn_mfma_f64_4x4x4f64(Ar, bv, denq[q], 0, 0, 0);
                            numq[q] = __builtin_amdgcn_mfma_f64_4x4x4f64(Al, bv, numq[q], 0, 0, 0);
                        }
                    }
                }
            }
        };
        steps(std::integral_constant<int, 0>{});
        r_c = load_rows(b_n, x_n);                               // tile t+1: context rows + L (indices have arrived meanwhile)
        steps(std::integral_constant<int, 2>{});
        fix_zero_denominators<NT, NQ>(den, denq, num, numq, lzero, lane);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int64_t e = e0 + 4 * r + kq;                   // D[i][j]: lane 16 (i % 4) + j, register i / 4
            if (e < E) {
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    const int c = c0 + nt * 16 + i;
                    if (c < C) P[e * C + c] = num[nt][r] / den[nt][r];
                }
            }
        }
        if constexpr (NQ > 0) {
            const int64_t e = e0 + 4 * ((lane >> 2) & 3) + kq;   // D[i][j] of block b: lane 16 i + 4 b + j
            if (e < E) {
#pragma unroll
                for (int q = 0; q < NQ; ++q) {
                    const int c = c0 + NT * 16 + 4 * q + (lane & 3);
                    if (c < C) P[e * C + c] = numq[q] / denq[q];
                }
            }
        }
        if (write_sizes) {
            rsum += __shfl_xor(rsum, 16, 64);
            rsum += __shfl_xor(rsum, 32, 64);
            lsum += __shfl_xor(lsum, 16, 64);
            lsum += __shfl_xor(lsum, 32, 64);
            const int64_t e = e0 + i;
            if (kq == 0 && e < E) {
                R_SIZE[e] = rsum;                                                    // genic_driver_tools.py:375
                ELT_SIZE[e] = lsum;                                                  // :380 (sum(L) / 3 with L = 3 x repeated)
                const double numer = gene_length ? (double)gene_length[e] : (double)lsum;
                P_INDEL[e] = numer / (double)rsum;                                   // :381 / :159
            }
        }
        b_c = b_n;
        b_n = b_nn;
        tile += stride;
    }
}

struct AccWorkspace {
    int32_t* rcp;    // strand-permuted context counts of the elements, [E][64]
    double* tab;     // pre-swizzled MFMA parameter table, [n48][64 steps][3][64]
    int n48;         // cohort chunks of 48 for the dot stage
    int64_t bytes;
};

static AccWorkspace acc_workspace_layout(void* base, int64_t E, int64_t C)
{
    AccWorkspace w{};
    auto up = [](int64_t v) { return (v + 255) / 256 * 256; };
    const int64_t o_rc = 0;
    const int64_t o_tab = up(o_rc + E * 64 * (int64_t)sizeof(int32_t));
    w.n48 = (int)((C + kMfmaChunk - 1) / kMfmaChunk);
    w.bytes = up(o_tab + (int64_t)w.n48 * kMfmaSteps * 3 * 64 * (int64_t)sizeof(double));
    char* b = (char*)base;
    w.rcp = (int32_t*)(b + o_rc);
    w.tab = (double*)(b + o_tab);
    return w;
}

template <int NCLASS>
static int launch_dot_mfma(const AccWorkspace& w, const int32_t* L, const int32_t* R_SIZE, const int32_t* gene_length,
                           double* P, int32_t* ELT_SIZE, double* P_INDEL, int64_t E, int64_t C, hipStream_t stream, const double* d_pr)
{
    const int64_t n_tiles = (E + 15) / 16;
    const int grid = (int)std::max<int64_t>(1, std::min<int64_t>(cu_count(), (n_tiles + kMfmaWaves - 1) / kMfmaWaves));
    for (int ch = 0; ch < w.n48; ++ch) {
        const int c0 = ch * kMfmaChunk;
        const ChunkCut cut = chunk_cut((int)C, ch);
        const size_t lds = (size_t)kMfmaSteps * (cut.nt + (cut.nq > 0)) * 64 * sizeof(double);
        const double* tab = w.tab + (int64_t)ch * kMfmaSteps * 3 * 64;
        auto go = [&](auto kern) -> int {
            DIG_HIP_TRY(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            DIG_LAUNCH_STAGE(DIG_PIPE_DOT, kern, dim3(grid), dim3(kMfmaWaves * 64), lds, stream, w.rcp, L, tab, R_SIZE, gene_length, P,
                               ELT_SIZE, P_INDEL, E, (int)C, c0, (int)(ch == 0), d_pr);
            DIG_HIP_TRY(hipGetLastError());
            return DIG_OK;
        };
        int rc;
        switch (cut.nt * 3 + cut.nq) {
        case 9: rc = go(acc_dot_mfma_kernel<NCLASS, 3, 0>); break;
        case 6: rc = go(acc_dot_mfma_kernel<NCLASS, 2, 0>); break;
        case 7: rc = go(acc_dot_mfma_kernel<NCLASS, 2, 1>); break;
        case 8: rc = go(acc_dot_mfma_kernel<NCLASS, 2, 2>); break;
        case 3: rc = go(acc_dot_mfma_kernel<NCLASS, 1, 0>); break;
        case 4: rc = go(acc_dot_mfma_kernel<NCLASS, 1, 1>); break;
        case 5: rc = go(acc_dot_mfma_kernel<NCLASS, 1, 2>); break;
        case 1: rc = go(acc_dot_mfma_kernel<NCLASS, 0, 1>); break;
        default: rc = go(acc_dot_mfma_kernel<NCLASS, 0, 2>); break;
        }
        if (rc) return rc;
    }
    return DIG_OK;
}

}  // namespace dig

using namespace dig;

extern "C" {

int64_t dig_accumulate_workspace(int64_t E, int64_t C)
{
    if (E <= 0 || C <= 0) return 0;
    return acc_workspace_layout(nullptr, E, C).bytes;
}

}  // extern "C"

namespace dig {

// Shared by dig_accumulate_elements and dig_element_pipeline (do_rates = 0: MU / SIGMA / R_OBS / FLAG are left to the
// caller's statistics kernel; needs the workspace path).
int accumulate_launch(const double* bin_mu, const double* bin_std, const int32_t* bin_y, const uint8_t* bin_flag,
                      const int32_t* bin_ctx, const int64_t* ov_ptr, const int32_t* ov_idx, const int32_t* L, int n_class,
                      const uint8_t* strand_minus, const int32_t* gene_length, const double* d_pr, double* MU,
                      double* SIGMA, int32_t* R_OBS, int32_t* FLAG, double* P, int32_t* R_SIZE, int32_t* ELT_SIZE,
                      double* P_INDEL, int64_t N, int64_t E, int64_t C, void* workspace, int64_t workspace_bytes,
                      void* stream, int do_rates, unsigned* zero_dwords, int n_zero, int parts)
{
    // parts: bit 0 = region kernel, bit 1 = dot kernel (the pipeline may enqueue them as two calls)
    DIG_REQUIRE(N >= 0 && E >= 0 && C >= 0, "N, E, C >= 0");
    DIG_REQUIRE(n_class == 1 || n_class == 4, "n_class must be 1 (elements) or 4 (genes)");
    if (E == 0 || C == 0) return DIG_OK;
    DIG_REQUIRE(bin_mu && bin_std && bin_y && bin_flag && bin_ctx && ov_ptr && ov_idx && L && strand_minus && d_pr,
                "non-null inputs");
    DIG_REQUIRE(MU && SIGMA && R_OBS && FLAG && P && R_SIZE && ELT_SIZE && P_INDEL, "non-null outputs");
    hipStream_t s = (hipStream_t)stream;
    if (!workspace) {
        DIG_REQUIRE(do_rates, "the fused pipeline needs a workspace");
        // no scratch: single-kernel LDS variant (slower; kept for callers that cannot provide a workspace)
        for (int64_t c0 = 0; c0 < C; c0 += 64) {
            AccArgs a{bin_mu, bin_std, bin_y, bin_flag, bin_ctx, ov_ptr, ov_idx, L, strand_minus, gene_length, d_pr,
                      MU, SIGMA, R_OBS, FLAG, P, R_SIZE, ELT_SIZE, P_INDEL, E, C, (int)c0,
                      (int)std::min<int64_t>(64, C - c0)};
            int rc = (n_class == 1) ? launch_acc_v1<1>(a, s) : launch_acc_v1<4>(a, s);
            if (rc) return rc;
        }
        return DIG_OK;
    }
    DIG_REQUIRE(((uintptr_t)workspace & 255u) == 0, "workspace 256-byte aligned");
    const AccWorkspace w = acc_workspace_layout(workspace, E, C);
    DIG_REQUIRE(workspace_bytes >= w.bytes, "workspace smaller than dig_accumulate_workspace(E, C)");
    if (parts & 1) {
        const int grid = grid_for(do_rates ? E * C : E * 16, kRegionBlock, 8);
        hipLaunchKernelGGL(acc_region_kernel, dim3(grid), dim3(kRegionBlock), 0, s, bin_mu, bin_std, bin_y, bin_flag,
                           bin_ctx, ov_ptr, ov_idx, strand_minus, MU, SIGMA, R_OBS, FLAG, R_SIZE, w.rcp, E, C,
                           make_fastdiv(C), (int)(C >= 2), d_pr, w.tab, w.n48, do_rates, zero_dwords, n_zero);
        DIG_HIP_TRY(hipGetLastError());
    }
    if (!(parts & 2)) return DIG_OK;
    return (n_class == 1) ? launch_dot_mfma<1>(w, L, R_SIZE, gene_length, P, ELT_SIZE, P_INDEL, E, C, s, d_pr)
                          : launch_dot_mfma<4>(w, L, R_SIZE, gene_length, P, ELT_SIZE, P_INDEL, E, C, s, d_pr);
}

// Compact form of the dot + context stages (dig_element_pipeline with DIG_PIPE_COMPACT_L): one kernel per 48-cohort chunk.
int accumulate_compact_launch(const int32_t* bin_ctx, const int64_t* ov_ptr, const int32_t* ov_idx, const uint8_t* strand_minus,
                              const int32_t* Lc, const int32_t* gene_length, const double* d_pr, double* P, int32_t* R_SIZE,
                              int32_t* ELT_SIZE, double* P_INDEL, int64_t E, int64_t C, void* stream, unsigned* zero_dwords,
                              int n_zero)
{
    if (E == 0 || C == 0) return DIG_OK;
    DIG_REQUIRE(bin_ctx && ov_ptr && ov_idx && strand_minus && Lc && d_pr, "non-null inputs");
    DIG_REQUIRE(P && R_SIZE && ELT_SIZE && P_INDEL, "non-null outputs");
    const int64_t n_tiles = (E + 15) / 16;
    int grid = (int)std::max<int64_t>(1, std::min<int64_t>(cu_count(), (n_tiles + kCtxWaves - 1) / kCtxWaves));
    if (grid >= 8) grid &= ~7;                 // whole rounds of the 8 XCDs (the walk is XCD-aware when it can be)
    const int n48 = (int)((C + kMfmaChunk - 1) / kMfmaChunk);
    for (int ch = 0; ch < n48; ++ch) {
        const ChunkCut cut = chunk_cut((int)C, ch);
        auto go = [&](auto kern) -> int {
            DIG_LAUNCH_STAGE(DIG_PIPE_DOT, kern, dim3(grid), dim3(kCtxWaves * 64), 0, (hipStream_t)stream, bin_ctx, ov_ptr, ov_idx, strand_minus,
                               Lc, d_pr, gene_length, P, R_SIZE, ELT_SIZE, P_INDEL, E, (int)C, ch * kMfmaChunk, (int)(ch == 0),
                               ch == 0 ? zero_dwords : nullptr, n_zero);
            DIG_HIP_TRY(hipGetLastError());
            return DIG_OK;
        };
        int rc;
        switch (cut.nt * 3 + cut.nq) {
        case 9: rc = go(acc_dot_ctx_kernel<3, 0>); break;
        case 6: rc = go(acc_dot_ctx_kernel<2, 0>); break;
        case 7: rc = go(acc_dot_ctx_kernel<2, 1>); break;
        case 8: rc = go(acc_dot_ctx_kernel<2, 2>); break;
        case 3: rc = go(acc_dot_ctx_kernel<1, 0>); break;
        case 4: rc = go(acc_dot_ctx_kernel<1, 1>); break;
        case 5: rc = go(acc_dot_ctx_kernel<1, 2>); break;
        case 1: rc = go(acc_dot_ctx_kernel<0, 1>); break;
        default: rc = go(acc_dot_ctx_kernel<0, 2>); break;
        }
        if (rc) return rc;
    }
    return DIG_OK;
}

// Plan-time: Lc[e, ctx] = L[e, 3 ctx]; *mismatch (device int, cleared here) becomes 1 when some context's three counts differ.
int compact_L_launch(const int32_t* L, int64_t E, int32_t* Lc, int* mismatch, void* stream)
{
    DIG_HIP_TRY(hipMemsetAsync(mismatch, 0, sizeof(int), (hipStream_t)stream));
    if (E == 0) return DIG_OK;
    hipLaunchKernelGGL(compact_L_kernel, dim3(grid_for(E * 64, 256)), dim3(256), 0, (hipStream_t)stream, L, E * 64, Lc, mismatch);
    DIG_HIP_TRY(hipGetLastError());
    return DIG_OK;
}

}  // namespace dig

extern "C" {

int dig_accumulate_elements(const double* bin_mu, const double* bin_std, const int32_t* bin_y, const uint8_t* bin_flag,
                            const int32_t* bin_ctx, const int64_t* ov_ptr, const int32_t* ov_idx, const int32_t* L,
                            int n_class, const uint8_t* strand_minus, const int32_t* gene_length, const double* d_pr,
                            double* MU, double* SIGMA, int32_t* R_OBS, int32_t* FLAG, double* P, int32_t* R_SIZE,
                            int32_t* ELT_SIZE, double* P_INDEL, int64_t N, int64_t E, int64_t C, void* workspace,
                            int64_t workspace_bytes, void* stream)
{
    return accumulate_launch(bin_mu, bin_std, bin_y, bin_flag, bin_ctx, ov_ptr, ov_idx, L, n_class, strand_minus,
                             gene_length, d_pr, MU, SIGMA, R_OBS, FLAG, P, R_SIZE, ELT_SIZE, P_INDEL, N, E, C, workspace,
                             workspace_bytes, stream, 1, nullptr, 0, 3);
}

int dig_accumulate_elements_host(const double* bin_mu, const double* bin_std, const int32_t* bin_y,
                                 const uint8_t* bin_flag, const int32_t* bin_ctx, const int64_t* ov_ptr,
                                 const int32_t* ov_idx, const int32_t* L, int n_class, const uint8_t* strand_minus,
                                 const int32_t* gene_length, const double* d_pr, double* MU, double* SIGMA,
                                 int32_t* R_OBS, int32_t* FLAG, double* P, int32_t* R_SIZE, int32_t* ELT_SIZE,
                                 double* P_INDEL, int64_t N, int64_t E, int64_t C, int device)
{
    DIG_REQUIRE(N >= 0 && E >= 0 && C >= 0, "N, E, C >= 0");
    DIG_REQUIRE(n_class == 1 || n_class == 4, "n_class must be 1 (elements) or 4 (genes)");
    if (E == 0 || C == 0) return DIG_OK;
    DIG_REQUIRE(bin_mu && bin_std && bin_y && bin_flag && bin_ctx && ov_ptr && ov_idx && L && strand_minus && d_pr,
                "non-null inputs");
    DIG_REQUIRE(MU && SIGMA && R_OBS && FLAG && P && R_SIZE && ELT_SIZE && P_INDEL, "non-null outputs");
    DIG_HIP_TRY(hipSetDevice(device));
    const int64_t nnz = ov_ptr[E];
    DIG_REQUIRE(nnz >= 0, "ov_ptr[E] >= 0");
    for (int64_t q = 0; q < nnz; ++q) DIG_REQUIRE(ov_idx[q] >= 0 && ov_idx[q] < N, "ov_idx within [0, N)");
    const size_t nNC = (size_t)N * C, nEC = (size_t)E * C;
    DevBuf d_mu, d_sd, d_y, d_fl, d_ctx, d_ptr, d_idx, d_L, d_st, d_gl, d_dpr;
    DevBuf o_mu, o_sg, o_ro, o_fl, o_p, o_rs, o_es, o_pi;
#define UP(buf, src, bytes)                       \
    DIG_HIP_TRY(buf.alloc(bytes));                \
    DIG_HIP_TRY(hipMemcpy(buf.p, src, bytes, hipMemcpyHostToDevice))
    UP(d_mu, bin_mu, nNC * 8);
    UP(d_sd, bin_std, nNC * 8);
    UP(d_y, bin_y, nNC * 4);
    UP(d_fl, bin_flag, nNC);
    UP(d_ctx, bin_ctx, (size_t)N * 64 * 4);
    UP(d_ptr, ov_ptr, (size_t)(E + 1) * 8);
    UP(d_idx, ov_idx, (size_t)nnz * 4);
    UP(d_L, L, (size_t)E * n_class * 192 * 4);
    UP(d_st, strand_minus, (size_t)E);
    if (gene_length) { UP(d_gl, gene_length, (size_t)E * 4); }
    UP(d_dpr, d_pr, (size_t)C * 192 * 8);
#undef UP
    DIG_HIP_TRY(o_mu.alloc(nEC * 8));
    DIG_HIP_TRY(o_sg.alloc(nEC * 8));
    DIG_HIP_TRY(o_ro.alloc(nEC * 4));
    DIG_HIP_TRY(o_fl.alloc(nEC * 4));
    DIG_HIP_TRY(o_p.alloc(nEC * n_class * 8));
    DIG_HIP_TRY(o_rs.alloc((size_t)E * 4));
    DIG_HIP_TRY(o_es.alloc((size_t)E * 4));
    DIG_HIP_TRY(o_pi.alloc((size_t)E * 8));
    DevBuf d_ws;
    const int64_t wsb = dig_accumulate_workspace(E, C);
    DIG_HIP_TRY(d_ws.alloc((size_t)wsb));
    int rc = dig_accumulate_elements(d_mu.as<double>(), d_sd.as<double>(), d_y.as<int32_t>(), d_fl.as<uint8_t>(),
                                     d_ctx.as<int32_t>(), d_ptr.as<int64_t>(), d_idx.as<int32_t>(), d_L.as<int32_t>(),
                                     n_class, d_st.as<uint8_t>(), gene_length ? d_gl.as<int32_t>() : nullptr,
                                     d_dpr.as<double>(), o_mu.as<double>(), o_sg.as<double>(), o_ro.as<int32_t>(),
                                     o_fl.as<int32_t>(), o_p.as<double>(), o_rs.as<int32_t>(), o_es.as<int32_t>(),
                                     o_pi.as<double>(), N, E, C, d_ws.p, wsb, nullptr);
    if (rc) return rc;
    DIG_HIP_TRY(hipDeviceSynchronize());
    DIG_HIP_TRY(hipMemcpy(MU, o_mu.p, nEC * 8, hipMemcpyDeviceToHost));
    DIG_HIP_TRY(hipMemcpy(SIGMA, o_sg.p, nEC * 8, hipMemcpyDeviceToHost));
    DIG_HIP_TRY(hipMemcpy(R_OBS, o_ro.p, nEC * 4, hipMemcpyDeviceToHost));
    DIG_HIP_TRY(hipMemcpy(FLAG, o_fl.p, nEC * 4, hipMemcpyDeviceToHost));
    DIG_HIP_TRY(hipMemcpy(P, o_p.p, nEC * n_class * 8, hipMemcpyDeviceToHost));
    DIG_HIP_TRY(hipMemcpy(R_SIZE, o_rs.p, (size_t)E * 4, hipMemcpyDeviceToHost));
    DIG_HIP_TRY(hipMemcpy(ELT_SIZE, o_es.p, (size_t)E * 4, hipMemcpyDeviceToHost));
    DIG_HIP_TRY(hipMemcpy(P_INDEL, o_pi.p, (size_t)E * 8, hipMemcpyDeviceToHost));
    return DIG_OK;
}

// Host-side integer index construction (genic_driver_tools.py:275-283).
int dig_ideal_overlaps_host(const int32_t* elt_chrom, const int64_t* blk_ptr, const int64_t* blk_start,
                            const int64_t* blk_end, int64_t E, int64_t window, const int32_t* bin_chrom,
                            const int64_t* bin_start, int64_t N, int64_t* ov_ptr, int32_t* ov_idx)
{
    DIG_REQUIRE(E >= 0 && N >= 0 && window > 0, "E, N >= 0 and window > 0");
    DIG_REQUIRE(ov_ptr, "ov_ptr non-null");
    if (E == 0) {
        ov_ptr[0] = 0;
        return DIG_OK;
    }
    DIG_REQUIRE(elt_chrom && blk_ptr && blk_start && blk_end && bin_chrom && bin_start, "non-null inputs");
    for (int64_t i = 1; i < N; ++i)
        DIG_REQUIRE(bin_chrom[i] > bin_chrom[i - 1] || (bin_chrom[i] == bin_chrom[i - 1] && bin_start[i] > bin_start[i - 1]),
                    "bin table sorted by (chrom, start), no duplicates");
    std::vector<int64_t> starts;
    int64_t nnz = 0;
    ov_ptr[0] = 0;
    for (int64_t e = 0; e < E; ++e) {
        starts.clear();
        for (int64_t b = blk_ptr[e]; b < blk_ptr[e + 1]; ++b) {
            const int64_t s = blk_start[b], en = blk_end[b];
            DIG_REQUIRE(s >= 0 && en >= 0, "non-negative coordinates");
            const int64_t low = (s / window) * window;                       // floor(start / w) * w
            const int64_t high = ((en + window - 1) / window) * window;      // ceil(end / w) * w
            for (int64_t x = low; x < high; x += window) starts.push_back(x);
        }
        std::sort(starts.begin(), starts.end());
        starts.erase(std::unique(starts.begin(), starts.end()), starts.end());   // list(set(...))
        if (ov_idx) {
            for (int64_t x : starts) {
                // binary search (chrom, start) in the sorted table
                int64_t lo = 0, hi = N;
                const int32_t ch = elt_chrom[e];
                while (lo < hi) {
                    const int64_t mid = (lo + hi) >> 1;
                    if (bin_chrom[mid] < ch || (bin_chrom[mid] == ch && bin_start[mid] < x)) lo = mid + 1;
                    else hi = mid;
                }
                if (lo >= N || bin_chrom[lo] != ch || bin_start[lo] != x)
                    return set_error(DIG_EINVAL, "element %lld overlaps bin chr%d:%lld which is not in the bin table",
                                     (long long)e, (int)ch, (long long)x);
                ov_idx[nnz++] = (int32_t)lo;
            }
        } else {
            nnz += (int64_t)starts.size();
        }
        ov_ptr[e + 1] = nnz;
    }
    return DIG_OK;
}

}  // extern "C"
