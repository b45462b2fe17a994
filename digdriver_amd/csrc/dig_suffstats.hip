// dig_suffstats.hip -- per-cohort sufficient statistics for the cohort scale factors.
//
// Reference: calc_scale_factor_efficient, genome mode (driver_model/transfer_tools.py:148-156):
//     regions_pass = regions[~regions.FLAG];  N_SNV_EXP = regions_pass.Y_PRED.sum()
//     cj_snv = N_SNV_OBS / N_SNV_EXP;  cj_ind = N_IND_OBS / N_SNV_EXP
// For C cohorts at once this is a column reduction of the [N, C] rate table with the FLAG mask.
//
// Two deterministic stages (fixed summation order -> bit-reproducible run to run and, with the
// rank-ordered sum of the all-gathered partials, across GPU counts):
//   stage 1: every workgroup owns a contiguous block of rows; thread t owns column t % C and
//            rows r0 + t / C, r0 + t / C + rpp, ... so a pass over rpp rows is one contiguous,
//            fully coalesced run of rpp * C doubles; per-thread sums are combined through LDS in
//            row-group order and written to partial[workgroup][C];
//   stage 2: one workgroup per column adds the partials with a fixed-shape tree.
#include "dig_common.hpp"

namespace dig {

constexpr int kSsBlock = 256;

__global__ __launch_bounds__(kSsBlock) void suffstats_stage1(const double* __restrict__ bin_mu,
                                                             const uint8_t* __restrict__ bin_flag, int64_t N,
                                                             int64_t C, int64_t rows_per_block,
                                                             double* __restrict__ partial)
{
    __shared__ double part[kSsBlock];
    const int tid = threadIdx.x;
    const int64_t r_begin = (int64_t)blockIdx.x * rows_per_block;
    const int64_t r_end = (r_begin + rows_per_block < N) ? r_begin + rows_per_block : N;
    if (C <= kSsBlock) {
        const int rpp = kSsBlock / (int)C;              // rows per pass
        const int col = tid % (int)C, rg = tid / (int)C;
        double acc = 0.0;
        if (rg < rpp) {
            int64_t r = r_begin + rg;
            // 8 independent row loads in flight per thread (the pass is pure streaming: what limits it is bytes in
            // flight per CU, not arithmetic)
            for (; r + 7 * rpp < r_end; r += 8 * rpp) {
                double v[8];
                uint8_t f[8];
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    const int64_t o = (r + (int64_t)q * rpp) * C + col;
                    v[q] = bin_mu[o];
                    f[q] = bin_flag[o];
                }
#pragma unroll
                for (int q = 0; q < 8; ++q) acc += f[q] ? 0.0 : v[q];
            }
            for (; r < r_end; r += rpp) {
                const int64_t o = r * C + col;
                acc += bin_flag[o] ? 0.0 : bin_mu[o];
            }
        }
        part[tid] = acc;
        __syncthreads();
        if (tid < C) {
            double s = 0.0;
            for (int g = 0; g < rpp; ++g) s += part[g * (int)C + tid];
            partial[(int64_t)blockIdx.x * C + tid] = s;
        }
    } else {
        for (int64_t c = tid; c < C; c += kSsBlock) {
            double acc = 0.0;
            for (int64_t r = r_begin; r < r_end; ++r) {
                const int64_t o = r * C + c;
                acc += bin_flag[o] ? 0.0 : bin_mu[o];
            }
            partial[(int64_t)blockIdx.x * C + c] = acc;
        }
    }
}

// stage 2: one workgroup per cohort column; thread t adds partial[t], partial[t + 256], ... (independent
// loads), then a fixed-shape LDS tree combines the 256 thread sums -> deterministic.
// n_snv / n_ind != NULL (single-shard form): the scale factors cj = n_snv / sum, cj_indel = n_ind / sum are written
// as well (transfer_tools.py:153-154), saving the separate division kernel when there is nothing to all-gather.
__global__ __launch_bounds__(kSsBlock) void suffstats_stage2(const double* __restrict__ partial, int nblocks, int64_t C,
                                                             double* __restrict__ out, const double* __restrict__ n_snv,
                                                             const double* __restrict__ n_ind, double* __restrict__ cj,
                                                             double* __restrict__ cj_indel)
{
    __shared__ double red[kSsBlock];
    const int64_t c = blockIdx.x;
    double s = 0.0;
    for (int b = threadIdx.x; b < nblocks; b += kSsBlock) s += partial[(int64_t)b * C + c];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int w = kSsBlock / 2; w > 0; w >>= 1) {
        if ((int)threadIdx.x < w) red[threadIdx.x] += red[threadIdx.x + w];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        out[c] = red[0];
        if (n_snv) {
            cj[c] = n_snv[c] / red[0];
            cj_indel[c] = n_ind[c] / red[0];
        }
    }
}

// scale factors from the all-gathered per-rank statistics: parts[r][0][c] = sum(Y_PRED[~FLAG]) of rank r's bins,
// parts[r][1][c] = its observed SNVs, parts[r][2][c] = its observed indels.  Summed in rank order on every rank
// (bit-reproducible, independent of the collective's reduction order), then cj = obs / expected.
__global__ void scale_factors_kernel(const double* __restrict__ parts, int world, int C, double* __restrict__ cj,
                                     double* __restrict__ cj_indel)
{
    for (int c = blockIdx.x * blockDim.x + threadIdx.x; c < C; c += gridDim.x * blockDim.x) {
        double e = 0.0, s = 0.0, d = 0.0;
        for (int r = 0; r < world; ++r) {
            const double* q = parts + (int64_t)r * 3 * C;
            e += q[c];
            s += q[C + c];
            d += q[2 * C + c];
        }
        cj[c] = s / e;            // transfer_tools.py:153
        cj_indel[c] = d / e;      // :154
    }
}

// ---- chunked form: sums that do not depend on how the bins are sharded ---------------------------------------------
// The plain form above fixes its summation order for ONE table; split the bins over ranks and the rank-ordered sum of the
// rank totals is a different floating-point expression, so the scale factors of a sharded run would differ from the
// single-GPU run in the last bits.  The chunked form defines the result as
//     S[c] = ((P_0[c] + P_1[c]) + ... ) + P_{K-1}[c],   P_j = sum over the bins of chunk j
// with K canonical chunks of the GLOBAL bin grid (boundaries floor(N j / K); K = 64 in digdriver_amd/parallel.py, so every
// shard boundary of 1, 2, 4, ... 64 ranks is a chunk boundary) and P_j summed in an order that depends on nothing but the
// chunk's own rows and C (fixed rows per workgroup, fixed lane assignment, partials added first to last).  A rank
// computes the P_j of the chunks it owns, the chunk sums are all-gathered in chunk order, and every rank adds the K
// of them first to last: identical bits for any number of ranks (tests/test_gpu_sharded.py).
constexpr int kSsMaxChunks = 256;
struct ChunkTable {
    int64_t row[kSsMaxChunks + 1];     // first row of every chunk in THIS table (+ end)
    int32_t blk[kSsMaxChunks + 1];     // first workgroup of every chunk (+ end)
    int n;
};

#ifndef DIG_SS_NT
#define DIG_SS_NT 1     // the background kernel's 85 MB stream is read past the caches (non-temporal loads): it evicted bin records the
#endif                  // statistics kernel re-reads -- step 0.1852 -> 0.1836 ms (records), 0.1969 -> 0.1927 (planes), same box, two runs each
#if DIG_SS_NT
#define DIG_SS_LOAD(p) __builtin_nontemporal_load(p)
#else
#define DIG_SS_LOAD(p) (*(p))
#endif
__global__ __launch_bounds__(kSsBlock) void suffstats_chunk_stage1(const double* __restrict__ bin_mu,
                                                                   const uint8_t* __restrict__ bin_flag, int64_t C,
                                                                   int64_t rows_per_block, ChunkTable tab,
                                                                   double* __restrict__ partial)
{
    __shared__ double part[kSsBlock];
    const int tid = threadIdx.x;
    int j = 0;
    while (j + 1 < tab.n && (int)blockIdx.x >= tab.blk[j + 1]) ++j;          // (uniform, at most kSsMaxChunks steps)
    const int64_t r_begin = tab.row[j] + (int64_t)((int)blockIdx.x - tab.blk[j]) * rows_per_block;
    const int64_t r_end = (r_begin + rows_per_block < tab.row[j + 1]) ? r_begin + rows_per_block : tab.row[j + 1];
    const int rpp = kSsBlock / (int)C;              // rows per pass (C <= kSsBlock)
    const int col = tid % (int)C, rg = tid / (int)C;
    // A BACKGROUND kernel: in the burden-test loop it runs on a side stream beside the statistics kernel, whose one
    // 1024-thread workgroup per CU leaves 32 of the 512 vector registers of a SIMD lane, one wave slot and a few KB of LDS
    // free.  With <= 32 VGPRs (four loads in flight per thread, 32-bit element offsets from the block's base) a workgroup of
    // this kernel fits into exactly that, so its 96 MB stream no longer waits for -- or holds up -- the big kernels
    // (round 3: 20 us of the step, tools/loop_probe.py).  Same additions in the same order per thread as before.
    double acc = 0.0;
    if (rg < rpp) {
        const double* mu0 = bin_mu + r_begin * C;
        const uint8_t* fl0 = bin_flag + r_begin * C;
        const int n_rows = (int)(r_end - r_begin);
        const int stride = rpp * (int)C;
        int r = rg;
        unsigned o = (unsigned)(rg * (int)C + col);
        if (bin_flag) {
            for (; r + 3 * rpp < n_rows; r += 4 * rpp, o += 4u * (unsigned)stride) {
                double v[4];
                uint8_t f[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    v[q] = mu0[o + (unsigned)(q * stride)];
                    f[q] = fl0[o + (unsigned)(q * stride)];
                }
#pragma unroll
                for (int q = 0; q < 4; ++q) acc += f[q] ? 0.0 : v[q];
            }
            for (; r < n_rows; r += rpp, o += (unsigned)stride) acc += fl0[o] ? 0.0 : mu0[o];
        } else {
            // bin_flag == NULL: the caller's table holds +0.0 where a bin is flagged (a plan-time copy: 8 instead of 9 bytes per
            // (bin, cohort) and step) -- the same additions, the same bits
            for (; r + 3 * rpp < n_rows; r += 4 * rpp, o += 4u * (unsigned)stride) {
                double v[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) v[q] = DIG_SS_LOAD(&mu0[o + (unsigned)(q * stride)]);
#pragma unroll
                for (int q = 0; q < 4; ++q) acc += v[q];
            }
            for (; r < n_rows; r += rpp, o += (unsigned)stride) acc += DIG_SS_LOAD(&mu0[o]);
        }
    }
    part[tid] = acc;
    __syncthreads();
    if (tid < C) {
        double s = 0.0;
        for (int g = 0; g < rpp; ++g) s += part[g * (int)C + tid];
        partial[(int64_t)blockIdx.x * C + tid] = s;
    }
}

// one workgroup per chunk: thread c adds the chunk's workgroup partials first to last
__global__ __launch_bounds__(kSsBlock) void suffstats_chunk_stage2(const double* __restrict__ partial, int64_t C, ChunkTable tab,
                                                                   double* __restrict__ out_chunks)
{
    const int j = blockIdx.x;
    for (int64_t c = threadIdx.x; c < C; c += kSsBlock) {
        double s = 0.0;
        for (int b = tab.blk[j]; b < tab.blk[j + 1]; ++b) s += partial[(int64_t)b * C + c];
        out_chunks[(int64_t)j * C + c] = s;
    }
}

// cj[c] = (sum over ranks of obs_snv) / (sum over ALL chunks, first to last); obs [world, 2, C] hold integer counts.
__global__ void scale_factors_chunked_kernel(const double* __restrict__ chunk_sums, int n_chunks, const double* __restrict__ obs,
                                             int world, int C, double* __restrict__ out_sum, double* __restrict__ cj,
                                             double* __restrict__ cj_indel)
{
    for (int c = blockIdx.x * blockDim.x + threadIdx.x; c < C; c += gridDim.x * blockDim.x) {
        double e = 0.0, s = 0.0, d = 0.0;
        for (int j = 0; j < n_chunks; ++j) e += chunk_sums[(int64_t)j * C + c];
        for (int r = 0; r < world; ++r) {
            s += obs[((int64_t)r * 2 + 0) * C + c];
            d += obs[((int64_t)r * 2 + 1) * C + c];
        }
        if (out_sum) out_sum[c] = e;
        cj[c] = s / e;            // transfer_tools.py:153
        cj_indel[c] = d / e;      // :154
    }
}

// Rows per workgroup of the chunked form: a function of C only (never of the device or of the table's size).
static int64_t ss_chunk_rows_per_block(int64_t C) { return 16 * (kSsBlock / C); }

static int ss_fill_chunk_table(const int64_t* chunk_rows, int n_chunks, int64_t C, ChunkTable* tab)
{
    const int64_t rpb = ss_chunk_rows_per_block(C);
    int64_t blocks = 0;
    tab->n = n_chunks;
    for (int j = 0; j <= n_chunks; ++j) {
        tab->row[j] = chunk_rows[j];
        tab->blk[j] = (int32_t)blocks;
        if (j < n_chunks) blocks += (chunk_rows[j + 1] - chunk_rows[j] + rpb - 1) / rpb;
    }
    return (int)blocks;
}

// Rows per workgroup: a multiple of 8 passes (the unrolled inner loop keeps 8 row loads in flight per thread; a
// ragged remainder would be walked one load at a time), sized for ~8 workgroups per CU.
static int64_t ss_rows_per_block(int64_t N, int64_t C)
{
    const int64_t rpp = (C <= kSsBlock) ? kSsBlock / C : 1;
    const int64_t unit = 8 * rpp;
    int64_t target = (N + (int64_t)cu_count() * 8 - 1) / ((int64_t)cu_count() * 8);
    if (target < unit) target = unit;
    return (target + unit - 1) / unit * unit;
}

static int ss_blocks(int64_t N, int64_t C)
{
    const int64_t rpb = ss_rows_per_block(N, C);
    const int64_t g = (N + rpb - 1) / rpb;
    return (int)(g < 1 ? 1 : g);
}

}  // namespace dig

using namespace dig;

extern "C" {

int64_t dig_scale_suffstats_workspace(int64_t N, int64_t C)
{
    if (N <= 0 || C <= 0) return 0;
    return (int64_t)ss_blocks(N, C) * C * (int64_t)sizeof(double);
}

int dig_scale_suffstats(const double* bin_mu, const uint8_t* bin_flag, int64_t N, int64_t C, double* out_sum,
                        void* workspace, int64_t workspace_bytes, void* stream)
{
    DIG_REQUIRE(N >= 0 && C >= 0, "N, C >= 0");
    if (C == 0) return DIG_OK;
    DIG_REQUIRE(out_sum, "non-null output");
    hipStream_t s = (hipStream_t)stream;
    if (N == 0) {
        DIG_HIP_TRY(hipMemsetAsync(out_sum, 0, (size_t)C * sizeof(double), s));
        return DIG_OK;
    }
    DIG_REQUIRE(bin_mu && bin_flag && workspace, "non-null inputs and workspace");
    const int g = ss_blocks(N, C);
    DIG_REQUIRE(workspace_bytes >= (int64_t)g * C * (int64_t)sizeof(double), "workspace smaller than dig_scale_suffstats_workspace(N, C)");
    DIG_REQUIRE(((uintptr_t)workspace & 7u) == 0, "workspace 8-byte aligned");
    const int64_t rpb = ss_rows_per_block(N, C);
    hipLaunchKernelGGL(suffstats_stage1, dim3(g), dim3(kSsBlock), 0, s, bin_mu, bin_flag, N, C, rpb, (double*)workspace);
    DIG_HIP_TRY(hipGetLastError());
    hipLaunchKernelGGL(suffstats_stage2, dim3((unsigned)C), dim3(kSsBlock), 0, s, (const double*)workspace, g, C, out_sum,
                       (const double*)nullptr, (const double*)nullptr, (double*)nullptr, (double*)nullptr);
    DIG_HIP_TRY(hipGetLastError());
    return DIG_OK;
}

int dig_scale_factors_local(const double* bin_mu, const uint8_t* bin_flag, int64_t N, int64_t C, const double* n_snv_obs,
                            const double* n_ind_obs, double* out_sum, double* cj, double* cj_indel, void* workspace,
                            int64_t workspace_bytes, void* stream)
{
    DIG_REQUIRE(N > 0 && C > 0, "N, C > 0");
    DIG_REQUIRE(bin_mu && bin_flag && n_snv_obs && n_ind_obs && out_sum && cj && cj_indel && workspace, "non-null pointers");
    hipStream_t s = (hipStream_t)stream;
    const int g = ss_blocks(N, C);
    DIG_REQUIRE(workspace_bytes >= (int64_t)g * C * (int64_t)sizeof(double), "workspace smaller than dig_scale_suffstats_workspace(N, C)");
    DIG_REQUIRE(((uintptr_t)workspace & 7u) == 0, "workspace 8-byte aligned");
    const int64_t rpb = ss_rows_per_block(N, C);
    hipLaunchKernelGGL(suffstats_stage1, dim3(g), dim3(kSsBlock), 0, s, bin_mu, bin_flag, N, C, rpb, (double*)workspace);
    DIG_HIP_TRY(hipGetLastError());
    hipLaunchKernelGGL(suffstats_stage2, dim3((unsigned)C), dim3(kSsBlock), 0, s, (const double*)workspace, g, C, out_sum,
                       n_snv_obs, n_ind_obs, cj, cj_indel);
    DIG_HIP_TRY(hipGetLastError());
    return DIG_OK;
}

int dig_scale_factors(const double* parts, int world, int64_t C, double* cj, double* cj_indel, void* stream)
{
    DIG_REQUIRE(world >= 1 && C >= 0, "world >= 1, C >= 0");
    if (C == 0) return DIG_OK;
    DIG_REQUIRE(parts && cj && cj_indel, "non-null pointers");
    DIG_REQUIRE(C <= 0x7fffffff, "C fits in 32 bits");
    hipLaunchKernelGGL(scale_factors_kernel, dim3((unsigned)((C + 255) / 256)), dim3(256), 0, (hipStream_t)stream, parts,
                       world, (int)C, cj, cj_indel);
    DIG_HIP_TRY(hipGetLastError());
    return DIG_OK;
}

int64_t dig_scale_suffstats_chunked_workspace(const int64_t* chunk_rows, int n_chunks, int64_t C)
{
    if (!chunk_rows || n_chunks <= 0 || n_chunks > kSsMaxChunks || C <= 0 || C > kSsBlock) return 0;
    ChunkTable tab;
    const int blocks = ss_fill_chunk_table(chunk_rows, n_chunks, C, &tab);
    return (int64_t)std::max(blocks, 1) * C * (int64_t)sizeof(double);
}

int dig_scale_suffstats_chunked(const double* bin_mu, const uint8_t* bin_flag, int64_t C, const int64_t* chunk_rows, int n_chunks,
                                double* out_chunks, void* workspace, int64_t workspace_bytes, void* stream)
{
    DIG_REQUIRE(chunk_rows && n_chunks >= 1 && n_chunks <= kSsMaxChunks, "1 .. 256 chunks, chunk_rows on the host");
    DIG_REQUIRE(C >= 1 && C <= kSsBlock, "1 <= C <= 256 for the chunked form");
    for (int j = 0; j < n_chunks; ++j) DIG_REQUIRE(chunk_rows[j] >= 0 && chunk_rows[j + 1] >= chunk_rows[j], "chunk_rows ascending");
    DIG_REQUIRE(out_chunks && workspace, "non-null output and workspace");
    ChunkTable tab;
    const int blocks = ss_fill_chunk_table(chunk_rows, n_chunks, C, &tab);
    DIG_REQUIRE(workspace_bytes >= (int64_t)std::max(blocks, 1) * C * (int64_t)sizeof(double), "workspace smaller than dig_scale_suffstats_chunked_workspace");
    hipStream_t s = (hipStream_t)stream;
    if (blocks > 0) {
        DIG_REQUIRE(bin_mu, "non-null bin_mu (bin_flag may be NULL: flagged entries of bin_mu are +0.0 then)");
        hipLaunchKernelGGL(suffstats_chunk_stage1, dim3(blocks), dim3(kSsBlock), 0, s, bin_mu, bin_flag, C, ss_chunk_rows_per_block(C), tab,
                           (double*)workspace);
        DIG_HIP_TRY(hipGetLastError());
    }
    hipLaunchKernelGGL(suffstats_chunk_stage2, dim3(n_chunks), dim3(kSsBlock), 0, s, (const double*)workspace, C, tab, out_chunks);
    DIG_HIP_TRY(hipGetLastError());
    return DIG_OK;
}

int dig_scale_factors_chunked(const double* chunk_sums, int n_chunks, const double* obs, int world, int64_t C, double* out_sum,
                              double* cj, double* cj_indel, void* stream)
{
    DIG_REQUIRE(n_chunks >= 1 && world >= 1 && C >= 0, "n_chunks >= 1, world >= 1, C >= 0");
    if (C == 0) return DIG_OK;
    DIG_REQUIRE(chunk_sums && obs && cj && cj_indel, "non-null pointers");
    DIG_REQUIRE(C <= 0x7fffffff, "C fits in 32 bits");
    hipLaunchKernelGGL(scale_factors_chunked_kernel, dim3((unsigned)((C + 255) / 256)), dim3(256), 0, (hipStream_t)stream, chunk_sums,
                       n_chunks, obs, world, (int)C, out_sum, cj, cj_indel);
    DIG_HIP_TRY(hipGetLastError());
    return DIG_OK;
}

int dig_scale_suffstats_host(const double* bin_mu, const uint8_t* bin_flag, int64_t N, int64_t C, double* out_sum,
                             int device)
{
    DIG_REQUIRE(N >= 0 && C >= 0, "N, C >= 0");
    if (C == 0) return DIG_OK;
    DIG_REQUIRE(out_sum, "non-null output");
    DIG_HIP_TRY(hipSetDevice(device));
    DevBuf dmu, dfl, dws, dout;
    const size_t n = (size_t)N * C;
    DIG_HIP_TRY(dmu.alloc(n * 8));
    DIG_HIP_TRY(dfl.alloc(n));
    DIG_HIP_TRY(dws.alloc((size_t)dig_scale_suffstats_workspace(N, C)));
    DIG_HIP_TRY(dout.alloc((size_t)C * 8));
    if (n) {
        DIG_REQUIRE(bin_mu && bin_flag, "non-null inputs");
        DIG_HIP_TRY(hipMemcpy(dmu.p, bin_mu, n * 8, hipMemcpyHostToDevice));
        DIG_HIP_TRY(hipMemcpy(dfl.p, bin_flag, n, hipMemcpyHostToDevice));
    }
    int rc = dig_scale_suffstats(dmu.as<double>(), dfl.as<uint8_t>(), N, C, dout.as<double>(), dws.p,
                                 dig_scale_suffstats_workspace(N, C), nullptr);
    if (rc) return rc;
    DIG_HIP_TRY(hipDeviceSynchronize());
    DIG_HIP_TRY(hipMemcpy(out_sum, dout.p, (size_t)C * 8, hipMemcpyDeviceToHost));
    return DIG_OK;
}

}  // extern "C"
