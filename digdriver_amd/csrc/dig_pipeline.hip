// dig_pipeline.hip -- accumulation + statistics block as ONE operation (the element / tile driver path).
//
// dig_element_pipeline == dig_accumulate_elements (n_class = 1) followed by dig_element_stats on its outputs
// (genic_driver_tools.py:300-431 then transfer_tools.py:1069-1087), with one fusion across the two: the per-pair rate
// sums MU / SIGMA / R_OBS / FLAG are formed inside the statistics streaming kernel, which needs them anyway, instead
// of being written by the region kernel and read back (24 B per (element, cohort) less HBM traffic each way and one
// HBM-bound pass fewer).  All outputs of both operations are still written; results are bit-identical to the two
// separate calls (tests/test_gpu_parity.py).  `stages` lets a caller enqueue the three stages (context kernel, dot
// kernel, statistics) as separate calls in that order on the same workspace, e.g. to form the scale factors -- which
// only the statistics stage needs -- on another stream beside the MFMA-bound dot kernel.
// Kernel sequence per call: acc_dot_ctx_kernel (contexts + dot, the compact form: DIG_PIPE_COMPACT_L after
// dig_element_pipeline_prepare) or acc_region_kernel -> acc_dot_mfma_kernel (general form), then
// element_stats_stream_fused_kernel (which finishes its own slow pairs).  Plan-time helpers live here too:
// dig_element_pipeline_prepare (compact L) and dig_bin_records_pack (the bin tables as 20-byte records).
#include <algorithm>
#include <vector>

#include "dig_common.hpp"

namespace dig {

struct FusedRates {
    const double *bin_mu, *bin_std;
    const int32_t* bin_y;
    const uint8_t* bin_flag;
    const int64_t* ov_ptr;
    const int32_t* ov_idx;
    double *mu_w, *sigma_w;
    int32_t *r_obs, *flag;
    int small_index;      // bin rows < 2^24 and rows * C < 2^32: bin-table offsets from one 24-bit multiply-add
    const double2* bin_pack;   // dig_bin_records_pack's records, or NULL: {Y_PRED, STD^2} ...
    const int32_t* bin_yf;     // ... and Y_TRUE | (FLAG != 0) << 31 per (bin, cohort)
    int records;               // DIG_PIPE_RECORDS: `out` holds one record of DIG_REC_DOUBLES doubles per pair
};

int accumulate_launch(const double* bin_mu, const double* bin_std, const int32_t* bin_y, const uint8_t* bin_flag,
                      const int32_t* bin_ctx, const int64_t* ov_ptr, const int32_t* ov_idx, const int32_t* L, int n_class,
                      const uint8_t* strand_minus, const int32_t* gene_length, const double* d_pr, double* MU,
                      double* SIGMA, int32_t* R_OBS, int32_t* FLAG, double* P, int32_t* R_SIZE, int32_t* ELT_SIZE,
                      double* P_INDEL, int64_t N, int64_t E, int64_t C, void* workspace, int64_t workspace_bytes,
                      void* stream, int do_rates, unsigned* zero_dwords, int n_zero, int parts);
int element_stats_launch(const double* mu, const double* sigma, const double* mu_indel, const double* sigma_indel,
                         const double* pi_sum, const double* pi_indel, int pi_indel_per_cohort, const int32_t* obs_snv,
                         const int32_t* obs_samples, const int32_t* obs_indel, const double* cj, const double* cj_indel,
                         double* out, int64_t E, int64_t C, void* workspace, int64_t workspace_bytes, void* stream,
                         const FusedRates* fused, int worklist_already_zero);
int accumulate_compact_launch(const int32_t* bin_ctx, const int64_t* ov_ptr, const int32_t* ov_idx, const uint8_t* strand_minus,
                              const int32_t* Lc, const int32_t* gene_length, const double* d_pr, double* P, int32_t* R_SIZE,
                              int32_t* ELT_SIZE, double* P_INDEL, int64_t E, int64_t C, void* stream, unsigned* zero_dwords,
                              int n_zero);
int compact_L_launch(const int32_t* L, int64_t E, int32_t* Lc, int* mismatch, void* stream);

// workspace of dig_element_pipeline: [accumulate: context rows + parameter table][statistics: worklist][compact L + flag]
struct PipeLayout {
    int64_t acc_bytes, stats_off, stats_bytes, lc_off, flag_off, bytes;
};
static PipeLayout pipe_layout(int64_t E, int64_t C, int64_t acc, int64_t stats)
{
    auto up = [](int64_t v) { return (v + 255) / 256 * 256; };
    PipeLayout p{};
    p.acc_bytes = up(acc);
    p.stats_off = p.acc_bytes;
    p.stats_bytes = stats;
    p.lc_off = up(p.stats_off + stats);
    p.flag_off = up(p.lc_off + E * 64 * (int64_t)sizeof(int32_t));
    p.bytes = p.flag_off + 256;
    return p;
}

// Plan-time packing of the four bin tables into the records the statistics kernel gathers from (dig_bin_records_pack).
__global__ __launch_bounds__(256) void pack_bins_kernel(const double* __restrict__ mu, const double* __restrict__ sd,
                                                        const int32_t* __restrict__ y, const uint8_t* __restrict__ fl, int64_t n,
                                                        double2* __restrict__ pack, int32_t* __restrict__ yf, int* __restrict__ bad)
{
#pragma clang fp contract(off)
    int neg = 0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const double s = sd[i];
        const int32_t yi = y[i];
        pack[i] = make_double2(mu[i], s * s);            // get_region_params_direct squares STD before summing (:266)
        yf[i] = (yi & 0x7fffffff) | (fl[i] ? (int32_t)0x80000000 : 0);
        neg |= yi < 0;
    }
    if (__any(neg) && (threadIdx.x & 63) == 0) atomicOr(bad, 1);
}

// DIG_PIPE_RECORDS -> the plane form.  One workgroup per 64 x 64 block of the [E, C] grid when cohort_major (planes [C, E]:
// a result frame's column is then one contiguous row), else a straight copy out of the records.
__global__ __launch_bounds__(256) void records_unpack_kernel(const double* __restrict__ rec, int64_t E, int64_t C, double* __restrict__ out7,
                                                             double* __restrict__ MU, double* __restrict__ SIGMA, int32_t* __restrict__ R_OBS,
                                                             int32_t* __restrict__ FLAG, int cohort_major)
{
    const int64_t n = E * C;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        int64_t o = i;
        if (cohort_major) {
            const int64_t e = i / C, c = i - e * C;
            o = c * E + e;
        }
        // field f of pair i: block i / 64 (5 rows x 64 lanes x 2 doubles), row f / 2, lane i % 64, half f % 2
#if defined(DIG_REC_LAYOUT) && DIG_REC_LAYOUT == 1
        const int64_t plane2 = ((n + 63) >> 6) << 7;
        auto field = [&](int f) { return rec[(int64_t)(f >> 1) * plane2 + (i << 1) + (f & 1)]; };
#else
        const double* blk = rec + (i >> 6) * (64 * DIG_REC_DOUBLES) + ((i & 63) << 1);
        auto field = [&](int f) { return blk[((f >> 1) << 7) + (f & 1)]; };
#endif
        if (out7)
#pragma unroll
            for (int pl = 0; pl < DIG_ES_NPLANES; ++pl) out7[(int64_t)pl * n + o] = field(pl);
        if (MU) MU[o] = field(DIG_REC_MU);
        if (SIGMA) SIGMA[o] = field(DIG_REC_SIGMA);
        const long long rf = __double_as_longlong(field(DIG_REC_ROBS_FLAG));
        if (R_OBS) R_OBS[o] = (int32_t)(rf & 0xffffffffll);
        if (FLAG) FLAG[o] = (int32_t)(rf >> 32);
    }
}

struct BinRecords {
    int64_t yf_off, flag_off, bytes;
};
static BinRecords bin_records_layout(int64_t N, int64_t C)
{
    auto up = [](int64_t v) { return (v + 255) / 256 * 256; };
    BinRecords r{};
    r.yf_off = up(N * C * (int64_t)sizeof(double2));
    r.flag_off = r.yf_off + up(N * C * (int64_t)sizeof(int32_t));
    r.bytes = r.flag_off + 256;
    return r;
}
}  // namespace dig

using namespace dig;

extern "C" {

int64_t dig_accumulate_workspace(int64_t E, int64_t C);
int64_t dig_element_stats_workspace(int64_t E, int64_t C);

int64_t dig_element_pipeline_workspace(int64_t E, int64_t C)
{
    if (E <= 0 || C <= 0) return 0;
    const int64_t a = dig_accumulate_workspace(E, C), b = dig_element_stats_workspace(E, C);
    if (a <= 0 || b <= 0) return 0;
    return pipe_layout(E, C, a, b).bytes;
}

int dig_element_pipeline_prepare(const int32_t* L, int64_t E, int64_t C, void* workspace, int64_t workspace_bytes,
                                 int* compact_ok, void* stream)
{
    DIG_REQUIRE(E >= 0 && C >= 0 && compact_ok, "E, C >= 0, compact_ok non-null");
    *compact_ok = 0;
    if (E == 0 || C == 0) return DIG_OK;
    DIG_REQUIRE(L, "L non-null");
    const int64_t need = dig_element_pipeline_workspace(E, C);
    DIG_REQUIRE(workspace && need > 0 && workspace_bytes >= need,
                "workspace of at least dig_element_pipeline_workspace(E, C) bytes (E * C must stay below 2^32 - 1)");
    DIG_REQUIRE(((uintptr_t)workspace & 255u) == 0, "workspace 256-byte aligned");
    const PipeLayout lay = pipe_layout(E, C, dig_accumulate_workspace(E, C), dig_element_stats_workspace(E, C));
    int* flag = (int*)((char*)workspace + lay.flag_off);
    int rc = compact_L_launch(L, E, (int32_t*)((char*)workspace + lay.lc_off), flag, stream);
    if (rc) return rc;
    int bad = 1;
    DIG_HIP_TRY(hipMemcpyAsync(&bad, flag, sizeof(int), hipMemcpyDeviceToHost, (hipStream_t)stream));
    DIG_HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
    *compact_ok = bad ? 0 : 1;
    return DIG_OK;
}

int64_t dig_element_records_bytes(int64_t E, int64_t C)
{
    if (E <= 0 || C <= 0) return 0;
    return (E * C + 63) / 64 * 64 * DIG_REC_DOUBLES * (int64_t)sizeof(double);
}

int dig_element_records_unpack(const double* records, int64_t E, int64_t C, double* out7, double* MU, double* SIGMA,
                               int32_t* R_OBS, int32_t* FLAG, int cohort_major, void* stream)
{
    DIG_REQUIRE(E >= 0 && C >= 0, "E, C >= 0");
    if (E == 0 || C == 0) return DIG_OK;
    DIG_REQUIRE(records, "records non-null");
    hipLaunchKernelGGL(records_unpack_kernel, dim3(grid_for(E * C, 256)), dim3(256), 0, (hipStream_t)stream, records, E, C, out7, MU,
                       SIGMA, R_OBS, FLAG, cohort_major);
    DIG_HIP_TRY(hipGetLastError());
    return DIG_OK;
}

int64_t dig_bin_records_bytes(int64_t N, int64_t C)
{
    if (N <= 0 || C <= 0) return 0;
    return bin_records_layout(N, C).bytes;
}

int dig_bin_records_pack(const double* bin_mu, const double* bin_std, const int32_t* bin_y, const uint8_t* bin_flag, int64_t N,
                         int64_t C, void* records, int64_t records_bytes, void* stream)
{
    DIG_REQUIRE(N >= 0 && C >= 0, "N, C >= 0");
    if (N == 0 || C == 0) return DIG_OK;
    DIG_REQUIRE(bin_mu && bin_std && bin_y && bin_flag && records, "non-null pointers");
    const BinRecords lay = bin_records_layout(N, C);
    DIG_REQUIRE(records_bytes >= lay.bytes, "records of at least dig_bin_records_bytes(N, C) bytes");
    DIG_REQUIRE(((uintptr_t)records & 255u) == 0, "records 256-byte aligned");
    int* bad = (int*)((char*)records + lay.flag_off);
    DIG_HIP_TRY(hipMemsetAsync(bad, 0, sizeof(int), (hipStream_t)stream));
    hipLaunchKernelGGL(pack_bins_kernel, dim3(grid_for(N * C, 256)), dim3(256), 0, (hipStream_t)stream, bin_mu, bin_std, bin_y,
                       bin_flag, N * C, (double2*)records, (int32_t*)((char*)records + lay.yf_off), bad);
    DIG_HIP_TRY(hipGetLastError());
    int neg = 0;
    DIG_HIP_TRY(hipMemcpyAsync(&neg, bad, sizeof(int), hipMemcpyDeviceToHost, (hipStream_t)stream));
    DIG_HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
    DIG_REQUIRE(!neg, "bin_y holds a negative count: the packed records keep Y_TRUE in 31 bits");
    return DIG_OK;
}

static int element_pipeline_run(const double* bin_mu, const double* bin_std, const int32_t* bin_y, const uint8_t* bin_flag,
                         const int32_t* bin_ctx, const int64_t* ov_ptr, const int32_t* ov_idx, const int32_t* L,
                         const uint8_t* strand_minus, const int32_t* gene_length, const double* d_pr,
                         const int32_t* obs_snv, const int32_t* obs_samples, const int32_t* obs_indel, const double* cj,
                         const double* cj_indel, double* MU, double* SIGMA, int32_t* R_OBS, int32_t* FLAG, double* P,
                         int32_t* R_SIZE, int32_t* ELT_SIZE, double* P_INDEL, double* out, int64_t N, int64_t E, int64_t C,
                         const void* bin_records, int stages, void* workspace, int64_t workspace_bytes, void* stream);

int dig_element_pipeline(const double* bin_mu, const double* bin_std, const int32_t* bin_y, const uint8_t* bin_flag,
                         const int32_t* bin_ctx, const int64_t* ov_ptr, const int32_t* ov_idx, const int32_t* L,
                         const uint8_t* strand_minus, const int32_t* gene_length, const double* d_pr,
                         const int32_t* obs_snv, const int32_t* obs_samples, const int32_t* obs_indel, const double* cj,
                         const double* cj_indel, double* MU, double* SIGMA, int32_t* R_OBS, int32_t* FLAG, double* P,
                         int32_t* R_SIZE, int32_t* ELT_SIZE, double* P_INDEL, double* out, int64_t N, int64_t E, int64_t C,
                         const void* bin_records, int stages, void* workspace, int64_t workspace_bytes, void* stream)
{
    const int rc = element_pipeline_run(bin_mu, bin_std, bin_y, bin_flag, bin_ctx, ov_ptr, ov_idx, L, strand_minus, gene_length, d_pr,
                                        obs_snv, obs_samples, obs_indel, cj, cj_indel, MU, SIGMA, R_OBS, FLAG, P, R_SIZE, ELT_SIZE,
                                        P_INDEL, out, N, E, C, bin_records, stages, workspace, workspace_bytes, stream);
    // a stage timer armed for a stage this call did not launch through the timed path (another form of the kernel, a stage
    // the call did not include, C > 48: only the FIRST chunk's dot launch takes the timer) does not stay armed
    dig::disarm_stage_timers();
    return rc;
}

}  // extern "C"

static int element_pipeline_run(const double* bin_mu, const double* bin_std, const int32_t* bin_y, const uint8_t* bin_flag,
                         const int32_t* bin_ctx, const int64_t* ov_ptr, const int32_t* ov_idx, const int32_t* L,
                         const uint8_t* strand_minus, const int32_t* gene_length, const double* d_pr,
                         const int32_t* obs_snv, const int32_t* obs_samples, const int32_t* obs_indel, const double* cj,
                         const double* cj_indel, double* MU, double* SIGMA, int32_t* R_OBS, int32_t* FLAG, double* P,
                         int32_t* R_SIZE, int32_t* ELT_SIZE, double* P_INDEL, double* out, int64_t N, int64_t E, int64_t C,
                         const void* bin_records, int stages, void* workspace, int64_t workspace_bytes, void* stream)
{
    const int worklist_clean = (stages & DIG_PIPE_WORKLIST_CLEAN) != 0;
    const int compact = (stages & DIG_PIPE_COMPACT_L) != 0 && N >= 1;
    const int records = (stages & DIG_PIPE_RECORDS) != 0;
    stages &= ~(DIG_PIPE_WORKLIST_CLEAN | DIG_PIPE_COMPACT_L | DIG_PIPE_RECORDS);
    DIG_REQUIRE(stages >= 1 && stages <= 7, "stages: bit mask of DIG_PIPE_CONTEXTS, DIG_PIPE_DOT, DIG_PIPE_STATISTICS");
    DIG_REQUIRE(N >= 0 && E >= 0 && C >= 0, "N, E, C >= 0");
    if (E == 0 || C == 0) return DIG_OK;
    DIG_REQUIRE(obs_snv && obs_samples && obs_indel && cj && cj_indel && out, "non-null statistics arguments");
    const int64_t need = dig_element_pipeline_workspace(E, C);
    DIG_REQUIRE(workspace && need > 0 && workspace_bytes >= need,
                "workspace of at least dig_element_pipeline_workspace(E, C) bytes (E * C must stay below 2^32 - 1)");
    DIG_REQUIRE(((uintptr_t)workspace & 255u) == 0, "workspace 256-byte aligned");
    const PipeLayout lay = pipe_layout(E, C, dig_accumulate_workspace(E, C), dig_element_stats_workspace(E, C));
    const int64_t acc_bytes = lay.acc_bytes;
    // the first kernel also clears the worklist header of the statistics stage (64 dwords): no separate memset node
    unsigned* wl = (unsigned*)((char*)workspace + acc_bytes);
    if (compact) {
        // context-repeated L, compacted by dig_element_pipeline_prepare: contexts + dot are ONE kernel (the DOT stage; a
        // CONTEXTS-only call has nothing to enqueue)
        if (stages & DIG_PIPE_DOT) {
            DIG_REQUIRE(bin_ctx && ov_ptr && ov_idx && strand_minus && d_pr && P && R_SIZE && ELT_SIZE && P_INDEL,
                        "non-null accumulation arguments");
            int rc = accumulate_compact_launch(bin_ctx, ov_ptr, ov_idx, strand_minus, (const int32_t*)((char*)workspace + lay.lc_off),
                                               gene_length, d_pr, P, R_SIZE, ELT_SIZE, P_INDEL, E, C, stream, wl, 64);
            if (rc) return rc;
        }
    } else if (stages & 3) {
        int rc = accumulate_launch(bin_mu, bin_std, bin_y, bin_flag, bin_ctx, ov_ptr, ov_idx, L, 1, strand_minus, gene_length,
                                   d_pr, MU, SIGMA, R_OBS, FLAG, P, R_SIZE, ELT_SIZE, P_INDEL, N, E, C, workspace, acc_bytes,
                                   stream, 0, wl, 64, stages & 3);
        if (rc) return rc;
    }
    if (!(stages & 4)) return DIG_OK;
    const int small_index = N < ((int64_t)1 << 24) && C < ((int64_t)1 << 24) && N * C < ((int64_t)1 << 32);
    FusedRates f{bin_mu, bin_std, bin_y, bin_flag, ov_ptr, ov_idx, MU, SIGMA, R_OBS, FLAG, small_index, nullptr, nullptr, records};
    if (records) {
        DIG_REQUIRE(bin_records, "DIG_PIPE_RECORDS needs bin_records (dig_bin_records_pack)");
        DIG_REQUIRE(((uintptr_t)out & 255u) == 0, "record-major `out` 256-byte aligned");
    }
    if (bin_records) {
        const BinRecords lay_r = bin_records_layout(N, C);
        f.bin_pack = (const double2*)bin_records;
        f.bin_yf = (const int32_t*)((const char*)bin_records + lay_r.yf_off);
    }
    return element_stats_launch(MU, SIGMA, nullptr, nullptr, P, P_INDEL, 0, obs_snv, obs_samples, obs_indel, cj, cj_indel,
                                out, E, C, (char*)workspace + acc_bytes, lay.stats_bytes, stream, &f,
                                /* worklist header cleared by the context kernel of THIS call; a statistics-only call
                                   (new scale factors on an existing accumulation) clears it itself */
                                (compact ? (stages & DIG_PIPE_DOT) != 0 : (stages & 1) != 0) || worklist_clean);
}
