// dig_nb.hip -- negative-binomial burden tests on gfx950 (FP64 VALU + HBM streaming).
//
// Kernels:
//   nb3_kernel<Op>          elementwise (k, alpha, p) -> p-value   (nb_model.py:243-337)
//   fisher_kernel           chi2.sf(-2(ln p1 + ln p2), 4)         (transfer_tools.py:1086-1087)
//   gamma_kernel            normal_params_to_gamma                (nb_model.py:237-241)
//   element_stats_stream_kernel / element_stats_slow_kernel (two-pass) and element_stats_single_pass_kernel:
//                           the seven-column statistics block per (element, cohort)
//                           (transfer_tools.py:17-19,300,343-344,473-482,594-615,731-747,1086-1087)
//   tiled_nb_kernel         per-base / per-tile exact test        (nb_model.py:141-178)
//
// The elementwise kernels are one work item per test with a grid-stride loop (8-byte coalesced loads, grid capped at
// 8 blocks/CU); the statistics block is a persistent streaming pass plus a compacted pass (see below).  The
// arithmetic per item is the FP64 recurrence / continued fraction in dig_math.hpp.
#include <algorithm>
#include <cstdlib>

#include "dig_common.hpp"
#include "dig_math.hpp"

namespace dig {

constexpr int kBlock = 256;

struct OpMidpUpper {
    __device__ static double apply(double k, double a, double p) { return nb_midp_upper(k, a, p); }
};
struct OpExact {
    __device__ static double apply(double k, double a, double p) { return nb_exact(k, a, p); }
};
struct OpGreater {
    __device__ static double apply(double k, double a, double p) { return nb_greater(k, a, p); }
};
struct OpMidpTwo {
    __device__ static double apply(double k, double a, double p) { return nb_midp_twosided(k, a, p); }
};

template <typename Op>
__global__ __launch_bounds__(kBlock) void nb3_kernel(const double* __restrict__ k, const double* __restrict__ alpha,
                                                     const double* __restrict__ p, double* __restrict__ out,
                                                     int64_t n)
{
    nb_tables_init();
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride)
        out[i] = Op::apply(k[i], alpha[i], p[i]);
}

__global__ __launch_bounds__(kBlock) void fisher_kernel(const double* __restrict__ p1, const double* __restrict__ p2,
                                                        double* __restrict__ out, int64_t n)
{
    nb_tables_init();
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride)
        out[i] = fisher_combine_fast(p1[i], p2[i]);   // same code path as the fused statistics kernels
}

__global__ __launch_bounds__(kBlock) void gamma_kernel(const double* __restrict__ mu, const double* __restrict__ sigma,
                                                       double* __restrict__ alpha, double* __restrict__ theta,
                                                       int64_t n)
{
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) {
        GammaParams g = normal_params_to_gamma(mu[i], sigma[i]);
        alpha[i] = g.alpha;
        theta[i] = g.theta;
    }
}

struct ElementStatsArgs {
    const double *mu, *sigma, *mu_indel, *sigma_indel, *pi_sum, *pi_indel;
    const int32_t *obs_snv, *obs_samples, *obs_indel;
    const double *cj, *cj_indel;
    double* out;
    int64_t E, C;
    int pi_indel_per_cohort;
    unsigned* worklist;   // [0] = count, entries from [kWorkHeader]; NULL -> slow lanes are resolved inline
    FastDiv divC;
    int use_fastdiv;
    // fused pipeline only (dig_element_pipeline): the rate sums of the accumulation are formed here from the bin
    // tables and written to mu_w / sigma_w / r_obs / flag (mu, sigma above alias mu_w, sigma_w for the compacted pass)
    const double *bin_mu, *bin_std;
    const int32_t* bin_y;
    const uint8_t* bin_flag;
    const int64_t* ov_ptr;
    const int32_t* ov_idx;
    double *mu_w, *sigma_w;
    int32_t *r_obs, *flag;
    int small_index;      // bin rows < 2^24 and rows * C < 2^32: bin-table offsets from one 24-bit multiply-add
    const double2* bin_pack;   // dig_bin_records_pack: {Y_PRED, STD^2} per (bin, cohort), or NULL
    const int32_t* bin_yf;     //                       Y_TRUE | (FLAG != 0) << 31
    int rec;              // DIG_PIPE_RECORDS: out is [ceil(n / 64) * 64][kRecOut] doubles (one record per pair) instead of seven planes
#ifdef DIG_DEV_ABLATE
    int ablate;           // developer build only (tools/variant_bench.py): 1 no stores, 2 no recurrence, 4 no bin loop, 8 no arithmetic
#endif
};

constexpr int kRecOut = 10;       // DIG_REC_DOUBLES: seven statistics, MU, SIGMA, {R_OBS, FLAG}
// DIG_PIPE_RECORDS: field f of pair i lives in block i / 64 of 5 rows x 64 lanes x 2 doubles, at row f / 2, lane i % 64
#ifndef DIG_REC_LAYOUT
#define DIG_REC_LAYOUT 0
#endif
__device__ __forceinline__ int64_t rec_index(int64_t i, int f, int64_t n)
{
#if DIG_REC_LAYOUT == 1      // developer A/B: five "pair planes" [5][n_pad][2]
    return (int64_t)(f >> 1) * (((n + 63) >> 6) << 7) + (i << 1) + (f & 1);
#else
    return (((i >> 6) * (kRecOut / 2) + (f >> 1)) << 7) + ((i & 63) << 1) + (f & 1);
#endif
}
// where plane `pl` of pair `i` lives (n pairs): the plane form, or field pl of the pair's record
__device__ __forceinline__ double* out_slot(const ElementStatsArgs& a, int pl, int64_t n, int64_t i)
{
    return a.rec ? a.out + rec_index(i, pl, n) : a.out + (int64_t)pl * n + i;
}

constexpr int kMaxDevices = 64;
constexpr int kWorkHeader = 64;   // dwords reserved in front of the worklist (count lives in [0])

struct PairRaw {
    double mu, sigma, pi_s, pi_i, mu_i, sigma_i, cj, cji;
    int k_snv, k_smp, k_ind;
    int64_t q0, q1;   // fused pipeline: CSR range of the pair's element (instead of mu, sigma)
    uint32_t c;       // cohort of the pair
};

struct PairInputs {
    double alpha, theta, p, exp_snv, alpha_i, theta_i, p_i, exp_ind, k_snv, k_smp, k_ind;
};

template <bool FUSED_RATES = false>
__device__ __forceinline__ PairRaw load_raw(const ElementStatsArgs& a, int64_t i, int64_t e, int64_t c)
{
    PairRaw r;
    r.c = (uint32_t)c;
    if (FUSED_RATES) {
        r.q0 = a.ov_ptr[e];
        r.q1 = a.ov_ptr[e + 1];
        r.mu = r.sigma = 0.0;
    } else {
        r.mu = a.rec ? a.out[rec_index(i, 7, a.E * a.C)] : a.mu[i];
        r.sigma = a.rec ? a.out[rec_index(i, 8, a.E * a.C)] : a.sigma[i];
        r.q0 = r.q1 = 0;
    }
    r.pi_s = a.pi_sum[i];
    r.pi_i = a.pi_indel_per_cohort ? a.pi_indel[i] : a.pi_indel[e];
    r.k_snv = a.obs_snv[i];
    r.k_smp = a.obs_samples[i];
    r.k_ind = a.obs_indel[i];
    r.cj = a.cj[c];
    r.cji = a.cj_indel[c];
    r.mu_i = a.mu_indel ? a.mu_indel[i] : r.mu;
    r.sigma_i = a.mu_indel ? a.sigma_indel[i] : r.sigma;
    return r;
}

template <bool FUSED_RATES = false>
__device__ __forceinline__ PairRaw load_raw(const ElementStatsArgs& a, int64_t i)
{
    const int64_t e = a.use_fastdiv ? fastdiv(i, a.divC) : i;   // use_fastdiv == 0 only when C == 1
    return load_raw<FUSED_RATES>(a, i, e, i - e * a.C);
}

// Input preparation for one (element, cohort) pair, bit-identical to the reference's numpy
// expressions (no FMA contraction): transfer_tools.py:17-19,46-48,300,343-344,476-481,737-745.
__device__ __forceinline__ PairInputs prepare_pair(const PairRaw& w, bool has_indel_params)
{
    PairInputs r;
    r.k_snv = (double)w.k_snv;
    r.k_smp = (double)w.k_smp;
    r.k_ind = (double)w.k_ind;
    const GammaParams g = normal_params_to_gamma(w.mu, w.sigma);
    r.alpha = g.alpha;
    r.theta = mul_rn(g.theta, w.cj);
    r.exp_snv = mul_rn(mul_rn(g.alpha, r.theta), w.pi_s);
    r.p = nb_success_prob(r.theta, w.pi_s);
    GammaParams gi = g;
    if (has_indel_params) gi = normal_params_to_gamma(w.mu_i, w.sigma_i);
    r.alpha_i = gi.alpha;
    r.theta_i = mul_rn(gi.theta, w.cji);
    r.exp_ind = mul_rn(mul_rn(gi.alpha, r.theta_i), w.pi_i);
    r.p_i = nb_success_prob(r.theta_i, w.pi_i);
    return r;
}

__device__ __forceinline__ PairInputs load_pair(const ElementStatsArgs& a, int64_t i)
{
    return prepare_pair(load_raw(a, i), a.mu_indel != nullptr);
}

// Single-pass form, used only when the caller provides no workspace (no worklist): one pair per thread, every pair
// through the fast recurrence (SNV and SAMPLE counts share one pass) and unresolved tests (k > kSmallK, or a p-value
// < kDirectMin where 1 - CDF cancels) finished inline.  The rare slow lanes stall their waves here; the two-pass path
// below exists because of that (measured: 210 us against 170 us).
__global__ __launch_bounds__(kBlock) void element_stats_single_pass_kernel(ElementStatsArgs a)
{
    nb_tables_init();
    const int64_t n = a.E * a.C;
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    const PairInputs q = load_pair(a, i);
    double pv_snv = 0.0, pv_smp = 0.0, pv_ind = 0.0, dummy = 0.0;
    const unsigned d1 = nb_midp_upper_fast2(q.k_snv, q.k_smp, 3u, q.alpha, q.p, pv_snv, pv_smp);
    const unsigned d2 = nb_midp_upper_fast2(q.k_ind, 0.0, 1u, q.alpha_i, q.p_i, pv_ind, dummy);
    if (!(d1 & 1u)) pv_snv = nb_midp_upper_unresolved(q.k_snv, q.alpha, q.p);
    if (!(d1 & 2u)) pv_smp = nb_midp_upper_unresolved(q.k_smp, q.alpha, q.p);
    if (!(d2 & 1u)) pv_ind = nb_midp_upper_unresolved(q.k_ind, q.alpha_i, q.p_i);
    a.out[0 * n + i] = q.exp_snv;
    a.out[1 * n + i] = pv_snv;
    a.out[2 * n + i] = pv_smp;
    a.out[3 * n + i] = q.theta_i;
    a.out[4 * n + i] = q.exp_ind;
    a.out[5 * n + i] = pv_ind;
    a.out[6 * n + i] = fisher_combine_fast(pv_snv, pv_ind);
}

// Pass 1, streaming form (used whenever a worklist is available): a persistent grid of 5 or 6 workgroups per CU, each wave
// walking 64-pair tiles with stride n_waves.  Profiling the one-shot form showed its waves spending two thirds of
// their life in s_waitcnt (input loads at the start, the worklist atomic's round trip at the end) with 5.3 of 8
// wave slots filled on average, i.e. a latency problem, not an arithmetic one.  Here
//   * the raw inputs of the NEXT tile are requested before the current tile's arithmetic starts,
//   * slow pairs are parked in a per-wave LDS buffer and handed to the global worklist with one atomic per
//     kParkCap pairs (normally one per wave lifetime) instead of one round trip per tile,
//   * every vector-memory operation of the loop body is unconditional (lanes past the end replay pair n-1, pass 2
//     overwrites the p-value planes of its pairs), so the in-order memory counter can be waited on exactly:
//     the loop waits for the prefetched loads only, never for its stores.
constexpr int kParkCap = 256;   // per-wave parking buffer (entries); flushed when fewer than 64 slots are free

__device__ __forceinline__ unsigned park_flush(unsigned* worklist, const unsigned* park, unsigned count, int lane)
{
    unsigned base = 0;
    if (lane == 0) base = atomicAdd(worklist, count);
    base = __shfl(base, 0, 64);
    for (unsigned j = lane; j < count; j += 64) worklist[kWorkHeader + base + j] = park[j];
    return 0;
}

// FUSED_RATES (dig_element_pipeline): MU = sum Y_PRED, SIGMA = sqrt(sum STD^2), R_OBS, FLAG of the pair are summed
// here over the element's bins (same CSR order and IEEE operations as acc_region_kernel, genic_driver_tools.py:262-271)
// and written out, instead of being read back from a previous kernel: 24 B per pair less HBM traffic each way.
// The streaming pass writes 80 bytes per pair that nothing reads back soon (the compacted pass revisits 1 % of the pairs):
// non-temporal stores keep them from displacing the bin tables and the next tiles' inputs in L2 (same-box A/B:
// 166 -> 157 us for dig_element_stats, 263 -> 256 us for dig_element_pipeline).
#ifdef DIG_ES_PLAIN_STORES                     // developer A/B: write-back stores instead of streaming ones
#define DIG_STREAM_STORE(ptr, val) (*(ptr) = (val))
#else
#define DIG_STREAM_STORE(ptr, val) __builtin_nontemporal_store((val), (ptr))
#endif
template <bool HAS_INDEL_PARAMS, bool FUSED_RATES = false>
__global__ __launch_bounds__(kBlock) void element_stats_stream_kernel(ElementStatsArgs a)
{
    __shared__ unsigned park_all[kBlock / 64][kParkCap];
    nb_tables_init();
    unsigned* park = park_all[threadIdx.x >> 6];
    const int64_t n = a.E * a.C;
    const int lane = threadIdx.x & 63;
    const unsigned long long lanes_below = (1ull << lane) - 1ull;
    const int64_t n_tiles = (n + 63) >> 6;
    const int64_t n_waves = ((int64_t)gridDim.x * kBlock) >> 6;
    int64_t tile = ((int64_t)blockIdx.x * kBlock + threadIdx.x) >> 6;
    if (tile >= n_tiles) return;
    // (element, cohort) of a lane's pair: split once by division, then advanced by the wave stride -- n_waves * 64 pairs
    // = step_e elements and step_c cohorts -- with one conditional carry (the 64-bit multiply-high of a per-tile
    // division is four quarter-rate integer multiplies).  Lanes past the end replay the last pair (E - 1, C - 1).
    const int64_t step_pairs = n_waves * 64;
    const int64_t step_e = a.use_fastdiv ? fastdiv(step_pairs, a.divC) : step_pairs;
    const uint32_t step_c = (uint32_t)(step_pairs - step_e * a.C);
    const uint32_t C32 = (uint32_t)a.C;
    int64_t iu = tile * 64 + lane;                    // unclamped flat index of the NEXT tile's pair
    int64_t eu = a.use_fastdiv ? fastdiv(iu, a.divC) : iu;
    uint32_t cu = (uint32_t)(iu - eu * a.C);
    auto clamped_load = [&]() {
        const bool past = iu >= n;
        return load_raw<FUSED_RATES>(a, past ? n - 1 : iu, past ? a.E - 1 : eu, past ? (int64_t)(C32 - 1) : (int64_t)cu);
    };
    PairRaw nxt = clamped_load();
    unsigned parked = 0;   // wave-uniform
    for (; tile < n_tiles; tile += n_waves) {
        if (parked > (unsigned)(kParkCap - 64)) parked = park_flush(a.worklist, park, parked, lane);
        const int64_t i_raw = tile * 64 + lane;
        const int64_t i = min(i_raw, n - 1);
        PairRaw cur = nxt;
        iu += step_pairs;
        eu += step_e;
        cu += step_c;
        if (cu >= C32) {
            cu -= C32;
            eu += 1;
        }
        nxt = clamped_load();
        if (FUSED_RATES) {
            double mu = 0.0, var = 0.0;
            int robs = 0, flag = 0;
            const int32_t* oi = a.ov_idx + cur.q0;
            uint32_t nb = (uint32_t)(cur.q1 - cur.q0);
#ifdef DIG_DEV_ABLATE
            if (a.ablate & 4) nb = 0;
#endif
            if (a.small_index) {
                for (uint32_t j = 0; j < nb; ++j) {
                    const uint32_t o = __umul24((uint32_t)oi[j], C32) + cur.c;   // bin row * C + cohort
                    const double sd = a.bin_std[o];
                    mu += a.bin_mu[o];
                    var += mul_rn(sd, sd);
                    robs += a.bin_y[o];
                    flag |= (a.bin_flag[o] != 0);
                }
            } else {
                for (uint32_t j = 0; j < nb; ++j) {
                    const int64_t o = (int64_t)oi[j] * a.C + cur.c;
                    const double sd = a.bin_std[o];
                    mu += a.bin_mu[o];
                    var += mul_rn(sd, sd);
                    robs += a.bin_y[o];
                    flag |= (a.bin_flag[o] != 0);
                }
            }
            cur.mu = mu;
            cur.sigma = sqrt(var);
            DIG_STREAM_STORE(&a.mu_w[i], cur.mu);
            DIG_STREAM_STORE(&a.sigma_w[i], cur.sigma);
            DIG_STREAM_STORE(&a.r_obs[i], robs);
            DIG_STREAM_STORE(&a.flag[i], flag);
        }
#ifdef DIG_DEV_ABLATE
        if (a.ablate & 2) cur.k_snv = cur.k_smp = cur.k_ind = 0;
        if (a.ablate & 8) {
            const double v = cur.mu + cur.sigma + cur.pi_s + cur.pi_i + (double)(cur.k_snv + cur.k_smp + cur.k_ind) + cur.cj + cur.cji;
            if (!(a.ablate & 1) || v == 12345.678) {
                for (int pl = 0; pl < 7; ++pl) DIG_STREAM_STORE(&a.out[pl * n + i], v);
            }
            continue;
        }
#endif
        const PairInputs q = prepare_pair(cur, HAS_INDEL_PARAMS);
        // (a test the recurrence cannot finish keeps a NEGATIVE value for pass 2: -pmf(k) when the direct form
        //  cancelled, -2 when it was not eligible at all; no p-value is negative)
        double pv_snv, pv_smp, pv_ind, dummy;
        const unsigned d1 = nb_fast2_counts<1>(cur.k_snv, cur.k_smp, true, q.alpha, q.p, pv_snv, pv_smp);
        const unsigned d2 = nb_fast2_counts<1>(cur.k_ind, 0, false, q.alpha_i, q.p_i, pv_ind, dummy);
#ifdef DIG_DEV_ABLATE
        if ((a.ablate & 1) && !(pv_snv + pv_smp + pv_ind + q.exp_snv + q.exp_ind + q.theta_i == 12345.678)) continue;
#endif
        const bool slow = ((d1 != 3u) || (d2 != 1u)) && i_raw < n;
        const unsigned long long m = __ballot(slow);
        if (slow) park[parked + __popcll(m & lanes_below)] = (unsigned)i;
        parked += (unsigned)__popcll(m);
        DIG_STREAM_STORE(&a.out[0 * n + i], q.exp_snv);
        DIG_STREAM_STORE(&a.out[1 * n + i], pv_snv);
        DIG_STREAM_STORE(&a.out[2 * n + i], pv_smp);
        DIG_STREAM_STORE(&a.out[3 * n + i], q.theta_i);
        DIG_STREAM_STORE(&a.out[4 * n + i], q.exp_ind);
        DIG_STREAM_STORE(&a.out[5 * n + i], pv_ind);
        // (a parked pair has placeholder zeros here: without the guard its lane sends the whole wave through the
        // library log / exp fallback of the combination -- 58 % of the tiles hold such a lane -- for a value the
        // compacted pass overwrites anyway)
        double pv_mut = 0.0;
        if (!slow) pv_mut = fisher_combine_fast(pv_snv, pv_ind);
        DIG_STREAM_STORE(&a.out[6 * n + i], pv_mut);
    }
    if (parked) park_flush(a.worklist, park, parked, lane);
}

// ---- Pass 1 of dig_element_pipeline: fused rates, three-deep software pipeline -----------------------------------------
// Ablation of the two-stage form above on the bench workload (tools/variant_bench.py): its loads and
// stores alone take 125 us (the practical HBM rate for this read/write mix), its arithmetic alone 89 us of VALU time
// per SIMD -- and together 165 us, because the rate sums start with two DEPENDENT loads per tile (bin index, then the
// bin's rates) that are issued after the previous tile's eleven stores: the in-order memory counter makes the wave sit
// out the stores' acknowledgement plus two round trips, once per tile.  Here every load of a tile is in flight a full
// tile ahead and none is issued behind a store it has to wait for:
//     top of iteration t :  CSR pointers of tile t+2;  bin indices + per-pair inputs of tile t+1 (pointers have arrived)
//     arithmetic of tile t  (its inputs and bin rates were requested during iteration t-1)
//     bin rates of tile t+1 (indices have arrived meanwhile)  ->  stores of tile t
// The first kPre bins of a pair travel through the pipeline in registers; a tile in which some pair overlaps more bins
// finishes those sums with the plain loop (wave-uniform branch; gene-sized elements).  Same operations in the same
// order per pair as acc_region_kernel (genic_driver_tools.py:262-271: mu += Y_PRED, var += STD**2 in CSR order).
// (kPre = 2 since round 3: elements of the element / tile routes overlap one or two 10-kb bins -- 1.3 % of the bench
//  workload's elements three -- and a third register slot cost more in replayed gathers than the loop costs the few:
//  142.4 -> 139.9 us, same bits; -DDIG_ES_KPRE=3 restores it)
#ifndef DIG_ES_KPRE
#define DIG_ES_KPRE 2
#endif
constexpr int kPre = DIG_ES_KPRE;

struct StagePtr {            // tile t+2
    int64_t q0, q1;
    uint32_t i, e, c;
    uint32_t ok;             // (one-kernel form) the tile exists: its chunk will be produced
};
struct StageIn {             // tile t+1
    int32_t idx[kPre];
    double mu, sg, mu_i, sg_i;   // (GIVEN form only: the pair's parameters as handed in)
    double pi_s, pi_i, cj, cji;
    int k_snv, k_smp, k_ind;
    int64_t q0;
    uint32_t nb, i, c;
};
struct StageBin {            // tile t+1
    double mu[kPre], sd[kPre];
    int32_t y[kPre];
    uint8_t fl[kPre];
};

// TICKETS: the tiles of a workgroup (tile = ticket * gridDim.x + blockIdx.x: the whole grid still sweeps the arrays as one
// moving window) are drawn by its waves from one LDS counter instead of being dealt out in advance.  With static
// shares the waves of a SIMD finish one after the other (the arbiter issues oldest-first): rocprofv3 shows an average wave
// lifetime of 68 % of the kernel's duration with the VALU 85 % busy while waves are resident -- the tail, where a SIMD is
// down to one or two waves, is where the pass loses its time.  With a shared queue all waves of a CU end together.
// ---- pass 2 inside the fused stream pass ---------------------------------------------------------------------------
// The pairs a wave cannot finish (0.8 % on the bench workload) go into a QUEUE OF RECORDS in the workgroup's LDS -- the
// three p-value slots as pass 1 left them, the counts and the two parameter pairs: everything pass 2 needs, nothing is
// read back from memory.  When the workgroup has run out of tiles its 16 waves empty the queue together, test by test
// (sixteen per draw, one quad each), with the quad series of the compacted pass, and overwrite their markers.  (Finishing them in the
// wave that found them was measured first: +30 us -- the 37 pairs of a large element park together, in ONE wave, and
// that wave then runs five rounds on its own while its SIMD waits for it.)  A queue that is full (more than
// kQueueCap parked pairs in one workgroup: 6 % of its pairs) overflows into the workgroup's OWN segment of the global
// worklist (pair indices only), which the same workgroup works off after its queue in the manner of the compacted kernel
// -- it reads back what its own waves wrote, through its own L1: no other workgroup is involved.  No kernel follows.
#ifndef DIG_ES_INWAVE
#define DIG_ES_INWAVE 1
#endif
constexpr int kQueueCap = 1024;        // records per workgroup
constexpr int kRecDoubles = 11;        // [0..2] p-value slots, [3..5] counts, [6] alpha, [7] p, [8] [9] the indel pair, [10] pair index
__shared__ double g_queue[kQueueCap * kRecDoubles];
typedef double v2d __attribute__((ext_vector_type(2)));
__shared__ unsigned g_queue_list[16][48];
__shared__ unsigned g_queue_len, g_ovf_len, g_ovf_next;
__shared__ unsigned g_tests[3 * kQueueCap];      // the open tests of the queue: record * 4 + role
__shared__ unsigned g_n_tests, g_next_test;
constexpr int kSlowBlock = 256;
constexpr int kSlowWaves = kSlowBlock / 64;
constexpr int kSlowPairsPerWave = 8;        // (<= 16: the open tests of a round are indexed by 16 role + slot)
constexpr int kOverflowSlack = 64 * 1024;      // entries behind the n of the worklist: each workgroup's segment is rounded up to whole tiles
__device__ __forceinline__ void slow_round(const ElementStatsArgs& a, const unsigned* __restrict__ items, unsigned base,
                                           unsigned count, bool have_first, unsigned first_item, double (*sp_all)[10],
                                           unsigned* list, int64_t n);

#ifdef DIG_ES_TIMING
// developer build: first / last clock (100 MHz) of every workgroup of the stream pass (tools/es_balance_probe.py)
__device__ unsigned long long g_es_t0[1024], g_es_t1[1024], g_es_b0[1024], g_es_b1[1024], g_es_q[1024];
#endif
#ifndef DIG_ES_XCD
#define DIG_ES_XCD 1
#endif
#ifndef DIG_ES_CONTIG
#define DIG_ES_CONTIG 0
#endif
#ifndef DIG_ES_ABL
#define DIG_ES_ABL 0     // developer ablation builds (tools/build_variant.sh): 1 no stores, 2 counts forced to 0, 8 no arithmetic, 16 no bin gathers,
                         // 32 no CSR / index loads, 64 no stores of the four rate outputs, 128 no stores of the seven planes
#endif
// GIVEN = 0: dig_element_pipeline (the rate sums of a pair are formed here from the bin tables and written out); 3: the same
// from the packed bin records of dig_bin_records_pack (two gathers per bin instead of four);
// GIVEN = 1 / 2: dig_element_stats (mu / sigma handed in per pair; 2: separate indel parameters) -- the same pipeline,
// tickets and in-kernel second pass without the CSR and bin stages.
template <int TB, bool TICKETS, int GIVEN = 0, bool REC = false>
__global__ __launch_bounds__(TB) void element_stats_stream_fused_kernel(ElementStatsArgs a)
{
    static_assert(!REC || (TB == 1024 && TICKETS && (GIVEN == 0 || GIVEN == 3)), "record outputs: the pipeline's kernel only");
    double* const queue = g_queue;
    unsigned* const tests = g_tests;
    constexpr int QCAP = kQueueCap;
#ifdef DIG_ES_TIMING
    if (threadIdx.x == 0) g_es_t0[blockIdx.x & 1023] = wall_clock64();
#endif
    constexpr bool FUSED = GIVEN == 0 || GIVEN == 3;      // rate sums formed here (3: from the packed bin records)
    constexpr bool PACKED = GIVEN == 3;
    __shared__ unsigned park_all[TB / 64][kParkCap];
    __shared__ unsigned s_ticket;
    if (TICKETS && threadIdx.x == 0) s_ticket = 0;
#if DIG_ES_INWAVE
    if (TB == 1024 && TICKETS && threadIdx.x == 0) g_queue_len = g_ovf_len = g_ovf_next = g_n_tests = g_next_test = 0;
#endif
    nb_tables_init();
    unsigned* park = park_all[threadIdx.x >> 6];
    const int64_t n = a.E * a.C;
    const int lane = threadIdx.x & 63;
    const unsigned long long lanes_below = (1ull << lane) - 1ull;
    const int64_t n_tiles = (n + 63) >> 6;
    const int64_t n_waves = ((int64_t)gridDim.x * TB) >> 6;
    int64_t tile = ((int64_t)blockIdx.x * TB + threadIdx.x) >> 6;
    if (!TICKETS && tile >= n_tiles) return;
    const int64_t step_pairs = n_waves * 64;
    const int64_t step_e = a.use_fastdiv ? fastdiv(step_pairs, a.divC) : step_pairs;
    const uint32_t step_c = (uint32_t)(step_pairs - step_e * a.C);
    const uint32_t C32 = (uint32_t)a.C;
    int64_t iu = tile * 64 + lane;                    // unclamped flat index of the pair the pointer stage fetches next
    int64_t eu = a.use_fastdiv ? fastdiv(iu, a.divC) : iu;
    uint32_t cu = (uint32_t)(iu - eu * a.C);
    auto draw = [&]() -> int64_t {                    // next tile of this workgroup (wave-uniform)
        unsigned t = 0;
        if (lane == 0) t = atomicAdd(&s_ticket, 1u);
        t = (unsigned)__builtin_amdgcn_readfirstlane((int)t);
#if DIG_ES_CONTIG      // developer A/B: every workgroup walks ONE contiguous range of tiles
        {
            const int64_t lo = n_tiles * blockIdx.x / gridDim.x, hi = n_tiles * (blockIdx.x + 1) / gridDim.x;
            return lo + t < hi ? lo + t : n_tiles;
        }
#endif
#if DIG_ES_XCD
        // Workgroups are dealt round-robin to the 8 XCDs (blockIdx % 8 labels the group that shares an L2).  A tile is 64
        // pairs = 1.7 elements at 37 cohorts, so neighbouring tiles read the same bin rows: the 8 groups take runs of
        // gridDim / 8 consecutive tiles instead of every 8th tile (the grid still sweeps the arrays as one window).
        if ((gridDim.x & 7u) == 0u) return ((int64_t)t * 8 + (blockIdx.x & 7u)) * (gridDim.x >> 3) + (blockIdx.x >> 3);
#endif
        return (int64_t)t * gridDim.x + blockIdx.x;
    };
    int64_t tile_ptr = 0;                             // TICKETS: tile the last pointer fetch was for
    auto fetch_ptr = [&]() {                          // lanes (and whole tiles) past the end replay the last pair
        StagePtr s;
        if (TICKETS) {
            tile_ptr = draw();
            iu = tile_ptr * 64 + lane;
            eu = a.use_fastdiv ? fastdiv(min(iu, n - 1), a.divC) : iu;
            cu = (uint32_t)(min(iu, n - 1) - eu * a.C);
        }
        const bool past = iu >= n;
        s.ok = TICKETS ? (uint32_t)(tile_ptr < n_tiles) : 1u;
        s.i = (uint32_t)(past ? n - 1 : iu);
        s.e = (uint32_t)(past ? a.E - 1 : eu);
        s.c = past ? C32 - 1 : cu;
#if DIG_ES_ABL & 32
        s.q0 = s.q1 = 0;
#else
        if (FUSED) {
            s.q0 = a.ov_ptr[s.e];
            s.q1 = a.ov_ptr[s.e + 1];
        } else
            s.q0 = s.q1 = 0;
#endif
        if (!TICKETS) {
            iu += step_pairs;
            eu += step_e;
            cu += step_c;
            if (cu >= C32) {
                cu -= C32;
                eu += 1;
            }
        }
        return s;
    };
    // every element's CSR range lies inside [0, nnz); with an empty CSR the index loads replay ov_ptr[0] (= 0: row 0)
    const int64_t nnz = FUSED ? a.ov_ptr[a.E] : 0;
    const int32_t* oi_base = nnz > 0 ? a.ov_idx : reinterpret_cast<const int32_t*>(a.ov_ptr);
    const int64_t oi_last = nnz > 0 ? nnz - 1 : 0;
    auto fetch_in = [&](const StagePtr& s) {
        StageIn r;
        r.q0 = s.q0;
        r.nb = (uint32_t)(s.q1 - s.q0);
        r.i = s.i;
        r.c = s.c;
        if (FUSED) {
#pragma unroll
            for (int j = 0; j < kPre; ++j)      // unconditional (the memory counter stays exact): bins past the pair's last replay a valid entry
#if DIG_ES_ABL & 32
                r.idx[j] = (int)(s.e & 1023u);
#else
                r.idx[j] = oi_base[min(s.q0 + j, oi_last)];
#endif
        } else {
            r.mu = a.mu[s.i];
            r.sg = a.sigma[s.i];
            if (GIVEN == 2) {
                r.mu_i = a.mu_indel[s.i];
                r.sg_i = a.sigma_indel[s.i];
            }
        }
        // the streamed inputs are read once: non-temporal loads keep them from displacing the bin records, which neighbouring
        // elements re-read, in L2 (same-box A/B, two pairs of runs: whole pass 172.8 -> 171.1 us; profiles/r05_stats_kernel_probes.txt)
        r.pi_s = __builtin_nontemporal_load(&a.pi_sum[s.i]);
        r.pi_i = a.pi_indel_per_cohort ? a.pi_indel[s.i] : a.pi_indel[s.e];
        r.k_snv = __builtin_nontemporal_load(&a.obs_snv[s.i]);
        r.k_smp = __builtin_nontemporal_load(&a.obs_samples[s.i]);
        r.k_ind = __builtin_nontemporal_load(&a.obs_indel[s.i]);
        r.cj = a.cj[s.c];
        r.cji = a.cj_indel[s.c];
        return r;
    };
    auto fetch_bin = [&](const StageIn& r) {
        StageBin b;
        if (!FUSED) return b;
#if DIG_ES_ABL & 16
        for (int j = 0; j < kPre; ++j) { b.mu[j] = 1.0 + r.c; b.sd[j] = 0.5; b.y[j] = 1; b.fl[j] = 0; }
        return b;
#endif
#pragma unroll
        for (int j = 0; j < kPre; ++j) {
            const int64_t o = a.small_index ? (int64_t)(__umul24((uint32_t)r.idx[j], C32) + r.c)
                                            : (int64_t)r.idx[j] * a.C + r.c;      // bin row * C + cohort
            if (PACKED) {
            const double2 mv = a.bin_pack[o];
            const int32_t yf = a.bin_yf[o];
            b.mu[j] = mv.x;
            b.sd[j] = mv.y;                       // (the square already)
            b.y[j] = yf & 0x7fffffff;
            b.fl[j] = (uint8_t)((uint32_t)yf >> 31);
            } else {
            b.mu[j] = a.bin_mu[o];
            b.sd[j] = a.bin_std[o];
            b.y[j] = a.bin_y[o];
            b.fl[j] = a.bin_flag[o];
            }
        }
        return b;
    };
    // pipeline fill: pointers of the first two tiles, inputs and bin rates of the first
    if (TICKETS) __syncthreads();        // the counter is zero
    StagePtr ptr_n = fetch_ptr();
    if (TICKETS) tile = tile_ptr;        // tickets come out in ascending order: the wave's tiles are tile, tile_n1, tile_ptr
    StageIn in_c = fetch_in(ptr_n);
    ptr_n = fetch_ptr();
    int64_t tile_n1 = tile_ptr;
    StageBin bin_c = fetch_bin(in_c);
    unsigned parked = 0;   // wave-uniform
    // the wave's parked pair indices leave LDS for the global worklist; the in-kernel form keeps one segment per workgroup
    // (ceil(tiles per workgroup) x 64 entries: it cannot overflow)
    const bool own_segment = DIG_ES_INWAVE && TB == 1024 && TICKETS;
    unsigned* segment = a.worklist + kWorkHeader + (own_segment ? (int64_t)blockIdx.x * (((n_tiles + gridDim.x - 1) / gridDim.x) << 6) : 0);
    auto flush = [&](unsigned count) -> unsigned {
        if (!own_segment) return park_flush(a.worklist, park, count, lane);
        unsigned base = 0;
        if (lane == 0) base = atomicAdd(&g_ovf_len, count);
        base = (unsigned)__builtin_amdgcn_readfirstlane((int)base);
        for (unsigned j = lane; j < count; j += 64) segment[base + j] = park[j];
        return 0;
    };
    for (; tile < n_tiles; tile = TICKETS ? tile_n1 : tile + n_waves) {
        if (parked > (unsigned)(kParkCap - 64)) parked = flush(parked);
        const bool live = tile * 64 + lane < n;
        const StageIn cur = in_c;
        const StageBin bin = bin_c;
        in_c = fetch_in(ptr_n);          // tile t+1: indices + inputs
        if (TICKETS) tile_n1 = tile_ptr;
        ptr_n = fetch_ptr();             // tile t+2: pointers
        const int64_t i = cur.i;
        // rate sums (genic_driver_tools.py:262-271)
        double mu = 0.0, var = 0.0;
        int robs = 0, flag = 0;
        if (FUSED) {
#pragma unroll
        for (int j = 0; j < kPre; ++j) {      // (bins past the pair's last were fetched from a replayed row: skipped here)
            if ((uint32_t)j < cur.nb) {
                mu += bin.mu[j];
                var += PACKED ? bin.sd[j] : mul_rn(bin.sd[j], bin.sd[j]);
                robs += bin.y[j];
                flag |= (bin.fl[j] != 0);
            }
        }
        if (__any(cur.nb > (uint32_t)kPre)) {
            const int32_t* oi = a.ov_idx + cur.q0;
            for (uint32_t j = kPre; j < cur.nb; ++j) {
                const int64_t o = (int64_t)oi[j] * a.C + cur.c;
                if (PACKED) {
                    const double2 mv = a.bin_pack[o];
                    const int32_t yf = a.bin_yf[o];
                    mu += mv.x;
                    var += mv.y;
                    robs += yf & 0x7fffffff;
                    flag |= (int)((uint32_t)yf >> 31);
                } else {
                    const double sd = a.bin_std[o];
                    mu += a.bin_mu[o];
                    var += mul_rn(sd, sd);
                    robs += a.bin_y[o];
                    flag |= (a.bin_flag[o] != 0);
                }
            }
        }
        }
        PairRaw w;
        if (FUSED) {
            w.mu = w.mu_i = mu;
            w.sigma = w.sigma_i = sqrt(var);
        } else {
            w.mu = w.mu_i = cur.mu;
            w.sigma = w.sigma_i = cur.sg;
            if (GIVEN == 2) {
                w.mu_i = cur.mu_i;
                w.sigma_i = cur.sg_i;
            }
        }
        w.pi_s = cur.pi_s; w.pi_i = cur.pi_i; w.cj = cur.cj; w.cji = cur.cji;
        w.k_snv = cur.k_snv; w.k_smp = cur.k_smp; w.k_ind = cur.k_ind;
#if DIG_ES_ABL & 2
        w.k_snv = w.k_smp = w.k_ind = 0;
#endif
#if DIG_ES_ABL & 256                                   // counts capped at 28: what a sorted wave's trip count would cost (timing only)
#ifndef DIG_ES_CAPK
#define DIG_ES_CAPK 28
#endif
        w.k_snv = min(w.k_snv, DIG_ES_CAPK); w.k_smp = min(w.k_smp, DIG_ES_CAPK);
#endif
        w.q0 = w.q1 = 0; w.c = cur.c;
#if DIG_ES_ABL & 8
        PairInputs q;
        q.alpha = q.alpha_i = w.mu; q.p = q.p_i = 0.5; q.exp_snv = w.sigma; q.theta_i = w.pi_s; q.exp_ind = w.pi_i + w.cj + w.cji;
        q.k_snv = q.k_smp = q.k_ind = 0.0;
        double pv_snv = (double)w.k_snv, pv_smp = (double)w.k_smp, pv_ind = (double)w.k_ind, pv_mut = 0.25;
        const bool slow = false;
#else
        const PairInputs q = prepare_pair(w, GIVEN == 2);
        // (a test the recurrence cannot finish keeps a NEGATIVE value for pass 2: -pmf(k) when the direct form
        //  cancelled, -2 when it was not eligible at all; no p-value is negative)
        double pv_snv, pv_smp, pv_ind, dummy;
        const unsigned d1 = nb_fast2_counts<1>(w.k_snv, w.k_smp, true, q.alpha, q.p, pv_snv, pv_smp);
        const unsigned d2 = nb_fast2_counts<1>(w.k_ind, 0, false, q.alpha_i, q.p_i, pv_ind, dummy);
        const bool slow = ((d1 != 3u) || (d2 != 1u)) && live;
        double pv_mut = 0.0;
        if (!slow) pv_mut = fisher_combine_fast(pv_snv, pv_ind);
#endif
        bin_c = fetch_bin(in_c);         // tile t+1: bin rates (requested BEFORE this tile's stores)
        unsigned long long m = __ballot(slow);
#if DIG_ES_INWAVE
        if (TB == 1024 && TICKETS && m) {                   // into the workgroup's queue; what does not fit goes the old way
            const unsigned cnt = (unsigned)__popcll(m);
            unsigned qb = 0;
            if (lane == 0) qb = atomicAdd(&g_queue_len, cnt);
            qb = (unsigned)__builtin_amdgcn_readfirstlane((int)qb);
            const unsigned slot = qb + (unsigned)__popcll(m & lanes_below);
            const bool fits = slot < (unsigned)QCAP;
            if (slow && fits) {
                double* r = queue + slot * kRecDoubles;
                r[0] = pv_snv; r[1] = pv_smp; r[2] = pv_ind;
                r[3] = q.k_snv; r[4] = q.k_smp; r[5] = q.k_ind;
                r[6] = q.alpha; r[7] = q.p; r[8] = q.alpha_i; r[9] = q.p_i;
                r[10] = __longlong_as_double((long long)i);
            }
            m = __ballot(slow && !fits);
        }
#endif
        if (slow && ((m >> lane) & 1ull)) park[parked + __popcll(m & lanes_below)] = (unsigned)i;
        parked += (unsigned)__popcll(m);
#if DIG_ES_ABL & 1
        if (pv_snv + pv_smp + pv_ind + pv_mut + q.exp_snv + q.exp_ind + q.theta_i + w.mu + w.sigma + robs + flag != 12345.678) continue;
#endif
        if (REC) {
            // tile-blocked records: the tile's block is 5 rows of 64 x 16 bytes; lane l writes its fields (2 j, 2 j + 1) to row j --
            // five store instructions of 1 KB each, one aligned 5 120-byte run per tile, one base address
            const unsigned long long rf = (unsigned long long)(unsigned)robs | ((unsigned long long)(unsigned)flag << 32);
#if DIG_REC_LAYOUT == 1
            v2d* g = reinterpret_cast<v2d*>(a.out) + tile * 64 + lane;
            const int64_t row = ((n + 63) >> 6);              // v2d elements per pair plane / 64
#define DIG_REC_ROW(j) ((j) * row * 64)
#else
            v2d* g = reinterpret_cast<v2d*>(a.out + tile * (64 * kRecOut)) + lane;
#define DIG_REC_ROW(j) ((j) * 64)
#endif
#ifdef DIG_REC_PLAIN_STORES                     // developer A/B: write-back stores
#define DIG_REC_STORE(j, x, y) g[DIG_REC_ROW(j)] = v2d{(x), (y)}
#else
#define DIG_REC_STORE(j, x, y) __builtin_nontemporal_store(v2d{(x), (y)}, &g[DIG_REC_ROW(j)])
#endif
            DIG_REC_STORE(0, q.exp_snv, pv_snv);
            DIG_REC_STORE(1, pv_smp, q.theta_i);
            DIG_REC_STORE(2, q.exp_ind, pv_ind);
            DIG_REC_STORE(3, pv_mut, w.mu);
            DIG_REC_STORE(4, w.sigma, __longlong_as_double((long long)rf));
            continue;
        }
#if !(DIG_ES_ABL & 64)
        if (FUSED) {
            DIG_STREAM_STORE(&a.mu_w[i], w.mu);
            DIG_STREAM_STORE(&a.sigma_w[i], w.sigma);
            DIG_STREAM_STORE(&a.r_obs[i], robs);
            DIG_STREAM_STORE(&a.flag[i], flag);
        }
#else
        if (w.mu + w.sigma + robs + flag == 12345.678) DIG_STREAM_STORE(&a.mu_w[i], w.mu);
#endif
#if DIG_ES_ABL & 128
        if (pv_snv + pv_smp + pv_ind + pv_mut + q.exp_snv + q.exp_ind + q.theta_i != 12345.678) continue;
#endif
        DIG_STREAM_STORE(&a.out[0 * n + i], q.exp_snv);
        DIG_STREAM_STORE(&a.out[1 * n + i], pv_snv);
        DIG_STREAM_STORE(&a.out[2 * n + i], pv_smp);
        DIG_STREAM_STORE(&a.out[3 * n + i], q.theta_i);
        DIG_STREAM_STORE(&a.out[4 * n + i], q.exp_ind);
        DIG_STREAM_STORE(&a.out[5 * n + i], pv_ind);
        DIG_STREAM_STORE(&a.out[6 * n + i], pv_mut);
    }
    if (parked) flush(parked);
#if DIG_ES_INWAVE
    if (TB == 1024 && TICKETS) {
#ifdef DIG_ES_TIMING
        if (lane == 0) {
            const unsigned long long now = wall_clock64();
            atomicMin(&g_es_b0[blockIdx.x & 1023], now);
            atomicMax(&g_es_b1[blockIdx.x & 1023], now);
        }
#endif
        __syncthreads();                                    // every wave of the workgroup is out of tiles: the queue is complete
        const int total = (int)min(g_queue_len, (unsigned)QCAP);
        if (threadIdx.x == 0 && total) atomicAdd(&a.worklist[3], (unsigned)total);     // diagnostic: pairs finished here
        // The queue is taken apart TEST by test, not pair by pair: the workgroups that end last are the ones with the most
        // records (probe: 14-18 us from the barrier to the end against a median of 8), and with whole pairs dealt to
        // the waves a wave with sixteen pairs runs three rounds of sixteen quads while its neighbours run two.
        // (a) one thread per record lists the record's open tests; (b) the waves draw sixteen tests at a time, one quad
        // each, and put the p-value back into the record; (c) one thread per record combines and writes the four planes.
        if (total) {
            const int rec = (int)threadIdx.x;               // TB == kQueueCap
            double* r = queue + (rec < QCAP ? rec : 0) * kRecDoubles;
            unsigned open = 0;
            if (rec < total)
                open = (__double_as_longlong(r[0]) < 0 ? 1u : 0u) | (__double_as_longlong(r[1]) < 0 ? 2u : 0u) |
                       (__double_as_longlong(r[2]) < 0 ? 4u : 0u);
            const unsigned long long b0 = __ballot(open & 1u), b1 = __ballot(open & 2u), b2 = __ballot(open & 4u);
            const unsigned c0 = (unsigned)__popcll(b0), c1 = (unsigned)__popcll(b1), c2 = (unsigned)__popcll(b2);
            unsigned tb = 0;
            if (lane == 0 && c0 + c1 + c2) tb = atomicAdd(&g_n_tests, c0 + c1 + c2);
            tb = (unsigned)__builtin_amdgcn_readfirstlane((int)tb);
            if (open & 1u) tests[tb + (unsigned)__popcll(b0 & lanes_below)] = (unsigned)rec * 4u;
            if (open & 2u) tests[tb + c0 + (unsigned)__popcll(b1 & lanes_below)] = (unsigned)rec * 4u + 1u;
            if (open & 4u) tests[tb + c0 + c1 + (unsigned)__popcll(b2 & lanes_below)] = (unsigned)rec * 4u + 2u;
            __syncthreads();
            const unsigned n_tests = g_n_tests;
#ifdef DIG_ES_TIMING
            if (threadIdx.x == 0) g_es_q[blockIdx.x & 1023] = (unsigned long long)total | ((unsigned long long)n_tests << 32);
#endif
            const int quad = lane >> 2, sub = lane & 3;
            for (;;) {
                unsigned t0 = 0;
                if (lane == 0) t0 = atomicAdd(&g_next_test, 16u);
                t0 = (unsigned)__builtin_amdgcn_readfirstlane((int)t0);
                if (t0 >= n_tests) break;
                const bool active = t0 + (unsigned)quad < n_tests;
                const unsigned id = tests[active ? t0 + (unsigned)quad : t0];
                const int role = (int)(id & 3u);
                double* sp = queue + (id >> 2) * kRecDoubles;
                const double marker = sp[role];             // -pmf(k), or -2: pmf(k) not known
                const double pv = nb_midp_upper_quad(sp[3 + role], sp[role == 2 ? 8 : 6], sp[role == 2 ? 9 : 7],
                                                     marker == -2.0 ? -1.0 : -marker, sub);
                if (active && sub == 0) sp[role] = pv;      // (nobody else reads this slot before the barrier)
            }
            __syncthreads();
            if (rec < total) {
                const int64_t item = __double_as_longlong(r[10]);
                const double pv_snv = r[0], pv_smp = r[1], pv_ind = r[2];
                if (REC) {
                    a.out[rec_index(item, 1, n)] = pv_snv;
                    a.out[rec_index(item, 2, n)] = pv_smp;
                    a.out[rec_index(item, 5, n)] = pv_ind;
                    a.out[rec_index(item, 6, n)] = fisher_combine_fast(pv_snv, pv_ind);
                } else {
                    a.out[1 * n + item] = pv_snv;
                    a.out[2 * n + item] = pv_smp;
                    a.out[5 * n + item] = pv_ind;
                    a.out[6 * n + item] = fisher_combine_fast(pv_snv, pv_ind);
                }
            }
        }
        const unsigned ovf = g_ovf_len;                     // (final since the barrier above)
        if (ovf) {
            // the pairs the queue had no room for: their indices are in this workgroup's segment, their markers in the
            // planes -- written by this workgroup's waves, all of which have passed the barrier (vmcnt(0) in front of it)
            __syncthreads();                                // the queue's LDS is free: 80 doubles per wave of it become the pair buffers
            if (threadIdx.x == 0) atomicAdd(&a.worklist[2], ovf);
            double (*sp)[10] = reinterpret_cast<double (*)[10]>(queue + (threadIdx.x >> 6) * (kSlowPairsPerWave * 10));
            for (;;) {
                unsigned b = 0;
                if (lane == 0) b = atomicAdd(&g_ovf_next, (unsigned)kSlowPairsPerWave);
                b = (unsigned)__builtin_amdgcn_readfirstlane((int)b);
                if (b >= ovf) break;
                slow_round(a, segment, b, ovf, false, 0u, sp, g_queue_list[threadIdx.x >> 6], n);
            }
        }
    }
#endif
#ifdef DIG_ES_TIMING
    if (lane == 0) atomicMax(&g_es_t1[blockIdx.x & 1023], (unsigned long long)wall_clock64());
#endif
}

// Pass 2: the compacted slow pairs.  Pass 1 left every test it could not finish NEGATIVE in its p-value plane (counts
// above kSmallK, p^alpha out of range, or a p-value below kDirectMin where 1 - CDF cancels: -pmf(k) then) and the values
// of the others in place.  A wave takes eight pairs per round; their open tests (at most 24) are compacted through LDS
// and evaluated sixteen at a time, ONE QUAD PER TEST (nb_midp_upper_quad: the series is split over the lanes, so a count
// of 500 costs a few hundred dependent instructions instead of 4 500); then one lane per pair combines SNV and indel and
// writes the four p-value planes.  Rounds 1 and 2 of this kernel gave a test one lane and sorted the pairs of a
// workgroup by count: 30 us whatever the number of pairs, the length of its longest lane.

// One round of pass 2 on listed pairs: the pairs items[base .. base + 8) (fewer at the end of the list), by one wave.
// sp: 8 x 10 doubles and list: 48 dwords of LDS owned by the wave.
__device__ __forceinline__ void slow_round(const ElementStatsArgs& a, const unsigned* __restrict__ items, unsigned base,
                                           unsigned count, bool have_first, unsigned first_item, double (*sp_all)[10],
                                           unsigned* list, int64_t n)
{
    const int lane = threadIdx.x & 63, quad = lane >> 2, sub = lane & 3;
    // ---- the first lanes: one pair each: its inputs, and which of its three tests are open (sign bit set) ----
    const bool owner = lane < kSlowPairsPerWave && base + lane < count;
    int64_t item = 0;
    unsigned open = 0;
    if (owner) {
        item = have_first ? first_item : items[base + lane];
        const double v1 = *out_slot(a, 1, n, item), v2 = *out_slot(a, 2, n, item), v5 = *out_slot(a, 5, n, item);
        const PairInputs q = load_pair(a, item);
        double* sp = sp_all[lane];
        sp[0] = v1; sp[1] = v2; sp[2] = v5;
        sp[3] = q.k_snv; sp[4] = q.k_smp; sp[5] = q.k_ind;
        sp[6] = q.alpha; sp[7] = q.p; sp[8] = q.alpha_i; sp[9] = q.p_i;
        open = (__double_as_longlong(v1) < 0 ? 1u : 0u) | (__double_as_longlong(v2) < 0 ? 2u : 0u) |
               (__double_as_longlong(v5) < 0 ? 4u : 0u);
    }
    // bit 16 role + slot of W <-> test (role, pair slot)
    const unsigned long long W = ((unsigned long long)__ballot(open & 1u) & 0xffffull) |
                                 (((unsigned long long)__ballot(open & 2u) & 0xffffull) << 16) |
                                 (((unsigned long long)__ballot(open & 4u) & 0xffffull) << 32);
    const int n_tests = __popcll(W);
    if (lane < 48 && ((W >> lane) & 1ull)) list[__popcll(W & ((1ull << lane) - 1ull))] = (unsigned)lane;
    __builtin_amdgcn_s_waitcnt(0xc07f);                    // lgkmcnt(0): the wave's own LDS writes are in
    __builtin_amdgcn_wave_barrier();
    // ---- the open tests, sixteen at a time, one quad each ----
    for (int t0 = 0; t0 < n_tests; t0 += 16) {
        const int t = t0 + quad;
        const bool active = t < n_tests;
        const unsigned bit = list[active ? t : 0];
        const int role = (int)(bit >> 4), slot = (int)(bit & 15u);
        const double* sp = sp_all[slot];
        const double marker = sp[role];                     // -pmf(k), or -2: pmf(k) not known
        const double k = sp[3 + role];
        const double al = sp[role == 2 ? 8 : 6], pp = sp[role == 2 ? 9 : 7];
        const double pv = nb_midp_upper_quad(k, al, pp, marker == -2.0 ? -1.0 : -marker, sub);
        if (active && sub == 0) sp_all[slot][role] = pv;     // (nobody else reads this test's slot)
    }
    __builtin_amdgcn_s_waitcnt(0xc07f);
    __builtin_amdgcn_wave_barrier();
    // ---- the first lanes: combine and write ----
    if (owner) {
        const double* sp = sp_all[lane];
        const double pv_snv = sp[0], pv_smp = sp[1], pv_ind = sp[2];
        *out_slot(a, 1, n, item) = pv_snv;
        *out_slot(a, 2, n, item) = pv_smp;
        *out_slot(a, 5, n, item) = pv_ind;
        *out_slot(a, 6, n, item) = fisher_combine_fast(pv_snv, pv_ind);
    }
    __builtin_amdgcn_s_waitcnt(0xc07f);                    // (sp / list are rewritten in the next round)
    __builtin_amdgcn_wave_barrier();
}

__global__ __launch_bounds__(kSlowBlock) void element_stats_slow_kernel(ElementStatsArgs a)
{
    // per pair: [0..2] the three p-value slots (pass 1's values, then the finished ones), [3..5] the counts,
    // [6] alpha, [7] p, [8] alpha of the indel test, [9] its p
    __shared__ double s_pair[kSlowWaves][kSlowPairsPerWave][10];
    __shared__ unsigned s_list[kSlowWaves][48];
    const int64_t n = a.E * a.C;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const unsigned n_waves = gridDim.x * kSlowWaves;
    const unsigned base0 = (blockIdx.x * kSlowWaves + wave) * kSlowPairsPerWave;
    // the length of the worklist and this wave's first entries are requested together, in front of the table set-up: one
    // round trip instead of three (an entry past the end is stale and never used; n pairs bound the list's capacity)
    const unsigned count_raw = a.worklist[0];
    unsigned first_item = 0;
    if (lane < kSlowPairsPerWave && (int64_t)(base0 + lane) < n) first_item = a.worklist[kWorkHeader + base0 + lane];
    nb_tables_init();
    const unsigned count = (unsigned)min((int64_t)count_raw, n);   // (never more entries than pairs, whatever the header holds)
    if (blockIdx.x == 0 && tid == 0) a.worklist[2] = count;      // diagnostic: length of the last worklist
    for (unsigned base = base0; base < count; base += n_waves * kSlowPairsPerWave)
        slow_round(a, a.worklist + kWorkHeader, base, count, base == base0, first_item, s_pair[wave], s_list[wave], n);
}

// Generic form: any shape, one item per thread and grid-stride step, everything recomputed per item.
__global__ __launch_bounds__(kBlock) void tiled_nb_generic_kernel(const double* __restrict__ pt, int pt_per_cohort,
                                                                  const int32_t* __restrict__ k,
                                                                  const double* __restrict__ mu,
                                                                  const double* __restrict__ sigma,
                                                                  double* __restrict__ pval, double* __restrict__ exp_out,
                                                                  int64_t C, int64_t n_bins, int64_t n_tiles)
{
    nb_tables_init();
    const int64_t per_cohort = n_bins * n_tiles;
    const int64_t n = C * per_cohort;
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) {
        const int64_t cb = i / n_tiles;   // c * n_bins + b
        const double m = mu[cb], s = sigma[cb];
        const double ptv = pt_per_cohort ? pt[i] : pt[i % per_cohort];
        const GammaParams g = normal_params_to_gamma(m, s);
        const double p = nb_success_prob(ptv, g.theta);   // 1 / (pt * theta + 1), nb_model.py:151
        pval[i] = nb_exact((double)k[i], g.alpha, p);      // :152
        exp_out[i] = mul_rn(ptv, m);                       // :157
    }
}

// The shape that matters (200 tiles per 10-kb bin, 10^7..10^9 tiles): a workgroup takes 1024 consecutive tiles per
// step.  They span a handful of (cohort, bin) rows, whose Gamma parameters (two IEEE divisions each) are formed once,
// by one thread per row, into LDS; the flat index is split with the host-computed magic multipliers (the two 64-bit
// divisions of the generic form cost more than the test itself); the four items of a thread are loaded before any of
// them is evaluated.  Same arithmetic per item, same bits.
constexpr int kTiledItems = 4;
constexpr int kTiledChunk = kBlock * kTiledItems;
constexpr int kTiledRowsMax = 64;     // rows a chunk can touch: n_tiles >= 17  =>  <= 1024 / 17 + 2

__global__ __launch_bounds__(kBlock) void tiled_nb_kernel(const double* __restrict__ pt, int pt_per_cohort,
                                                          const int32_t* __restrict__ k, const double* __restrict__ mu,
                                                          const double* __restrict__ sigma, double* __restrict__ pval,
                                                          double* __restrict__ exp_out, int64_t C, int64_t n_bins,
                                                          int64_t n_tiles, FastDiv div_tiles, FastDiv div_bins)
{
    __shared__ double s_alpha[kTiledRowsMax], s_theta[kTiledRowsMax], s_mu[kTiledRowsMax];
    nb_tables_init();
    const int64_t per_cohort = n_bins * n_tiles;
    const int64_t n = C * per_cohort;
    const int tid = threadIdx.x;
    for (int64_t chunk0 = (int64_t)blockIdx.x * kTiledChunk; chunk0 < n; chunk0 += (int64_t)gridDim.x * kTiledChunk) {
        const int64_t last = (chunk0 + kTiledChunk - 1 < n - 1) ? chunk0 + kTiledChunk - 1 : n - 1;
        const int64_t cb0 = fastdiv(chunk0, div_tiles);
        const int nrows = (int)(fastdiv(last, div_tiles) - cb0) + 1;
        if (tid < nrows) {
            const double m = mu[cb0 + tid];
            const GammaParams g = normal_params_to_gamma(m, sigma[cb0 + tid]);
            s_alpha[tid] = g.alpha;
            s_theta[tid] = g.theta;
            s_mu[tid] = m;
        }
        double ptv[kTiledItems];
        int kk[kTiledItems], row[kTiledItems];
#pragma unroll
        for (int u = 0; u < kTiledItems; ++u) {
            const int64_t i = chunk0 + u * kBlock + tid;
            const int64_t ic = i < n ? i : n - 1;
            const int64_t cb = fastdiv(ic, div_tiles);
            row[u] = (int)(cb - cb0);
            const int64_t c = n_bins > 1 ? fastdiv(cb, div_bins) : cb;
            ptv[u] = pt_per_cohort ? pt[ic] : pt[ic - c * per_cohort];
            kk[u] = k[ic];
        }
        __syncthreads();
#pragma unroll
        for (int u = 0; u < kTiledItems; ++u) {
            const int64_t i = chunk0 + u * kBlock + tid;
            if (i < n) {
                const double p = nb_success_prob(ptv[u], s_theta[row[u]]);   // 1 / (pt * theta + 1), nb_model.py:151
                // (written once, read by nobody on the device: non-temporal)
                __builtin_nontemporal_store(nb_exact((double)kk[u], s_alpha[row[u]], p), &pval[i]);       // :152
                __builtin_nontemporal_store(mul_rn(ptv[u], s_mu[row[u]]), &exp_out[i]);                   // :157
            }
        }
        __syncthreads();
    }
}

template <typename Op>
static int launch_nb3(const double* k, const double* alpha, const double* p, double* out, int64_t n, void* stream)
{
    if (n == 0) return DIG_OK;
    if (!k || !alpha || !p || !out || n < 0) return set_error(DIG_EINVAL, "nb3: null pointer or negative n");
    hipLaunchKernelGGL(nb3_kernel<Op>, dim3(grid_for(n, kBlock)), dim3(kBlock), 0, (hipStream_t)stream, k, alpha, p,
                       out, n);
    DIG_HIP_TRY(hipGetLastError());
    return DIG_OK;
}

template <typename Op>
static int host_nb3(const double* k, const double* alpha, const double* p, double* out, int64_t n, int device)
{
    if (n == 0) return DIG_OK;
    if (!k || !alpha || !p || !out || n < 0) return set_error(DIG_EINVAL, "nb3_host: null pointer or negative n");
    DIG_HIP_TRY(hipSetDevice(device));
    DevBuf dk, da, dp, dout;
    const size_t bytes = (size_t)n * sizeof(double);
    DIG_HIP_TRY(dk.alloc(bytes));
    DIG_HIP_TRY(da.alloc(bytes));
    DIG_HIP_TRY(dp.alloc(bytes));
    DIG_HIP_TRY(dout.alloc(bytes));
    DIG_HIP_TRY(hipMemcpy(dk.p, k, bytes, hipMemcpyHostToDevice));
    DIG_HIP_TRY(hipMemcpy(da.p, alpha, bytes, hipMemcpyHostToDevice));
    DIG_HIP_TRY(hipMemcpy(dp.p, p, bytes, hipMemcpyHostToDevice));
    int rc = launch_nb3<Op>(dk.as<double>(), da.as<double>(), dp.as<double>(), dout.as<double>(), n, nullptr);
    if (rc) return rc;
    DIG_HIP_TRY(hipDeviceSynchronize());
    DIG_HIP_TRY(hipMemcpy(out, dout.p, bytes, hipMemcpyDeviceToHost));
    return DIG_OK;
}

}  // namespace dig

using namespace dig;

extern "C" {

#ifdef DIG_ES_TIMING
int dig_debug_es_timing(unsigned long long* out2048)
{
    DIG_HIP_TRY(hipDeviceSynchronize());
    DIG_HIP_TRY(hipMemcpyFromSymbol(out2048, HIP_SYMBOL(dig::g_es_t0), 1024 * sizeof(unsigned long long)));
    DIG_HIP_TRY(hipMemcpyFromSymbol(out2048 + 1024, HIP_SYMBOL(dig::g_es_t1), 1024 * sizeof(unsigned long long)));
    static unsigned long long z[1024] = {};
    DIG_HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(dig::g_es_t1), z, sizeof(z)));
    // [2048, 3072): first, [3072, 4096): last arrival of a wave at the workgroup's end-of-tiles barrier
    DIG_HIP_TRY(hipMemcpyFromSymbol(out2048 + 2048, HIP_SYMBOL(dig::g_es_b0), 1024 * sizeof(unsigned long long)));
    DIG_HIP_TRY(hipMemcpyFromSymbol(out2048 + 3072, HIP_SYMBOL(dig::g_es_b1), 1024 * sizeof(unsigned long long)));
    DIG_HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(dig::g_es_b1), z, sizeof(z)));
    static unsigned long long big[1024];
    for (auto& v : big) v = ~0ull;
    DIG_HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(dig::g_es_b0), big, sizeof(big)));
    return DIG_OK;
}
int dig_debug_es_queue(unsigned long long* out1024)      // records | open tests << 32 of every workgroup's queue, last launch
{
    DIG_HIP_TRY(hipMemcpyFromSymbol(out1024, HIP_SYMBOL(dig::g_es_q), 1024 * sizeof(unsigned long long)));
    return DIG_OK;
}
#endif


int dig_nb_midp_upper(const double* k, const double* alpha, const double* p, double* out, int64_t n, void* stream)
{
    return launch_nb3<OpMidpUpper>(k, alpha, p, out, n, stream);
}
int dig_nb_midp_upper_host(const double* k, const double* alpha, const double* p, double* out, int64_t n, int device)
{
    return host_nb3<OpMidpUpper>(k, alpha, p, out, n, device);
}
int dig_nb_exact(const double* k, const double* alpha, const double* p, double* out, int64_t n, void* stream)
{
    return launch_nb3<OpExact>(k, alpha, p, out, n, stream);
}
int dig_nb_exact_host(const double* k, const double* alpha, const double* p, double* out, int64_t n, int device)
{
    return host_nb3<OpExact>(k, alpha, p, out, n, device);
}
int dig_nb_greater(const double* k, const double* alpha, const double* p, double* out, int64_t n, void* stream)
{
    return launch_nb3<OpGreater>(k, alpha, p, out, n, stream);
}
int dig_nb_greater_host(const double* k, const double* alpha, const double* p, double* out, int64_t n, int device)
{
    return host_nb3<OpGreater>(k, alpha, p, out, n, device);
}
int dig_nb_midp_twosided(const double* k, const double* alpha, const double* p, double* out, int64_t n, void* stream)
{
    return launch_nb3<OpMidpTwo>(k, alpha, p, out, n, stream);
}
int dig_nb_midp_twosided_host(const double* k, const double* alpha, const double* p, double* out, int64_t n,
                              int device)
{
    return host_nb3<OpMidpTwo>(k, alpha, p, out, n, device);
}

int dig_fisher(const double* p1, const double* p2, double* out, int64_t n, void* stream)
{
    if (n == 0) return DIG_OK;
    DIG_REQUIRE(p1 && p2 && out && n > 0, "non-null pointers, n >= 0");
    hipLaunchKernelGGL(fisher_kernel, dim3(grid_for(n, kBlock)), dim3(kBlock), 0, (hipStream_t)stream, p1, p2, out, n);
    DIG_HIP_TRY(hipGetLastError());
    return DIG_OK;
}

int dig_fisher_host(const double* p1, const double* p2, double* out, int64_t n, int device)
{
    if (n == 0) return DIG_OK;
    DIG_REQUIRE(p1 && p2 && out && n > 0, "non-null pointers, n >= 0");
    DIG_HIP_TRY(hipSetDevice(device));
    DevBuf d1, d2, dout;
    const size_t bytes = (size_t)n * sizeof(double);
    DIG_HIP_TRY(d1.alloc(bytes));
    DIG_HIP_TRY(d2.alloc(bytes));
    DIG_HIP_TRY(dout.alloc(bytes));
    DIG_HIP_TRY(hipMemcpy(d1.p, p1, bytes, hipMemcpyHostToDevice));
    DIG_HIP_TRY(hipMemcpy(d2.p, p2, bytes, hipMemcpyHostToDevice));
    int rc = dig_fisher(d1.as<double>(), d2.as<double>(), dout.as<double>(), n, nullptr);
    if (rc) return rc;
    DIG_HIP_TRY(hipDeviceSynchronize());
    DIG_HIP_TRY(hipMemcpy(out, dout.p, bytes, hipMemcpyDeviceToHost));
    return DIG_OK;
}

int dig_normal_params_to_gamma(const double* mu, const double* sigma, double* alpha, double* theta, int64_t n,
                               void* stream)
{
    if (n == 0) return DIG_OK;
    DIG_REQUIRE(mu && sigma && alpha && theta && n > 0, "non-null pointers, n >= 0");
    hipLaunchKernelGGL(gamma_kernel, dim3(grid_for(n, kBlock)), dim3(kBlock), 0, (hipStream_t)stream, mu, sigma, alpha,
                       theta, n);
    DIG_HIP_TRY(hipGetLastError());
    return DIG_OK;
}

int dig_normal_params_to_gamma_host(const double* mu, const double* sigma, double* alpha, double* theta, int64_t n,
                                    int device)
{
    if (n == 0) return DIG_OK;
    DIG_REQUIRE(mu && sigma && alpha && theta && n > 0, "non-null pointers, n >= 0");
    DIG_HIP_TRY(hipSetDevice(device));
    DevBuf dm, ds, da, dt;
    const size_t bytes = (size_t)n * sizeof(double);
    DIG_HIP_TRY(dm.alloc(bytes));
    DIG_HIP_TRY(ds.alloc(bytes));
    DIG_HIP_TRY(da.alloc(bytes));
    DIG_HIP_TRY(dt.alloc(bytes));
    DIG_HIP_TRY(hipMemcpy(dm.p, mu, bytes, hipMemcpyHostToDevice));
    DIG_HIP_TRY(hipMemcpy(ds.p, sigma, bytes, hipMemcpyHostToDevice));
    int rc = dig_normal_params_to_gamma(dm.as<double>(), ds.as<double>(), da.as<double>(), dt.as<double>(), n, nullptr);
    if (rc) return rc;
    DIG_HIP_TRY(hipDeviceSynchronize());
    DIG_HIP_TRY(hipMemcpy(alpha, da.p, bytes, hipMemcpyDeviceToHost));
    DIG_HIP_TRY(hipMemcpy(theta, dt.p, bytes, hipMemcpyDeviceToHost));
    return DIG_OK;
}

int64_t dig_element_stats_workspace(int64_t E, int64_t C)
{
    if (E < 0 || C < 0) return 0;
    const int64_t n = E * C;
    if (n >= (int64_t)0xffffffffu) return 0;   // 32-bit worklist indices; larger problems run single-pass
    return (int64_t)sizeof(unsigned) * (kWorkHeader + n + kOverflowSlack);
}

}  // extern "C"

namespace dig {

// Shared launcher of dig_element_stats and of the statistics stage of dig_element_pipeline (fused != NULL: the rate
// sums are formed inside the streaming kernel from the bin tables and written to fused->mu_w etc.).
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
    int records;               // DIG_PIPE_RECORDS: `out` holds one record of kRecOut doubles per pair
};

// DIG_ES_FORM / DIG_ES_BLOCKS_PER_CU are developer knobs for A/B runs (tools/variant_bench.py).
static int stream_form()      // 1 = three-deep pipelined fused kernel (default), 0 = two-stage form
{
    const char* e = getenv("DIG_ES_FORM");
    return e ? atoi(e) : 1;
}
static int stream_blocks_per_cu(int dflt)
{
    const char* e = getenv("DIG_ES_BLOCKS_PER_CU");
    return e ? std::max(1, atoi(e)) : dflt;
}

int element_stats_launch(const double* mu, const double* sigma, const double* mu_indel, const double* sigma_indel,
                         const double* pi_sum, const double* pi_indel, int pi_indel_per_cohort, const int32_t* obs_snv,
                         const int32_t* obs_samples, const int32_t* obs_indel, const double* cj, const double* cj_indel,
                         double* out, int64_t E, int64_t C, void* workspace, int64_t workspace_bytes, void* stream,
                         const FusedRates* fused, int worklist_already_zero)
{
    const int64_t need = dig_element_stats_workspace(E, C);
    unsigned* wl = nullptr;
    if (workspace && need > 0) {
        DIG_REQUIRE(workspace_bytes >= need, "workspace smaller than dig_element_stats_workspace(E, C)");
        DIG_REQUIRE(((uintptr_t)workspace & 3u) == 0, "workspace 4-byte aligned");
        wl = (unsigned*)workspace;
    }
    DIG_REQUIRE(!fused || (wl && !mu_indel), "the fused pipeline needs the worklist workspace and shares the SNV parameters");
    hipStream_t s = (hipStream_t)stream;
    const int use_fd = (C >= 2);   // exact: E * C * C < 2^64 for any problem that fits in memory
    ElementStatsArgs a{mu, sigma, mu_indel, sigma_indel, pi_sum, pi_indel, obs_snv, obs_samples, obs_indel,
                       cj, cj_indel, out, E, C, pi_indel_per_cohort, wl, make_fastdiv(C), use_fd,
                       nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0, nullptr, nullptr, 0,
#ifdef DIG_DEV_ABLATE
                       , 0
#endif
    };
    if (fused) {
        a.bin_mu = fused->bin_mu; a.bin_std = fused->bin_std; a.bin_y = fused->bin_y; a.bin_flag = fused->bin_flag;
        a.ov_ptr = fused->ov_ptr; a.ov_idx = fused->ov_idx;
        a.mu_w = fused->mu_w; a.sigma_w = fused->sigma_w; a.r_obs = fused->r_obs; a.flag = fused->flag;
        a.small_index = fused->small_index;
        a.bin_pack = fused->bin_pack; a.bin_yf = fused->bin_yf;
        a.rec = fused->records;
    }
#ifdef DIG_DEV_ABLATE
    a.ablate = getenv("DIG_ABLATE") ? atoi(getenv("DIG_ABLATE")) : 0;
#endif
    DIG_REQUIRE(!a.rec || (wl && stream_form() == 1 && (!getenv("DIG_ES_TICKETS") || atoi(getenv("DIG_ES_TICKETS")) == 1024)),
                "DIG_PIPE_RECORDS: only the default form of the statistics kernel writes records");
    if (wl && !worklist_already_zero) DIG_HIP_TRY(hipMemsetAsync(wl, 0, sizeof(unsigned) * kWorkHeader, s));
    bool finished_in_wave = false;
    const int64_t want_blocks = (E * C + kBlock - 1) / kBlock;
    DIG_REQUIRE(want_blocks <= 0x7fffffff, "E * C too large for one launch");
    const int grid = (int)want_blocks;
    if (wl) {
        // Persistent grids: as many workgroups as are resident at once.  The occupancy figures are cached per (device,
        // kernel form): a process may drive several GPUs.
        const int which = fused ? 2 : (mu_indel ? 1 : 0);
        static int resident[kMaxDevices][4] = {};
        int dev_id = 0;
        DIG_HIP_TRY(hipGetDevice(&dev_id));
        DIG_REQUIRE(dev_id >= 0 && dev_id < kMaxDevices, "device index below 64");
        auto occupancy = [&](int slot, auto kernel, int block) -> int {
            int& r = resident[dev_id][slot];
            if (!r) {
                int per_cu = 0;
                if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, block, 0) != hipSuccess || per_cu < 1) per_cu = 4;
                r = per_cu;
            }
            return r;
        };
        static const int form = stream_form(), tickets = getenv("DIG_ES_TICKETS") ? atoi(getenv("DIG_ES_TICKETS")) : 1024;
        static const int given_form = getenv("DIG_ES_GIVEN_FORM") ? atoi(getenv("DIG_ES_GIVEN_FORM")) : 1;
        // the pipelined kernel (one 1024-thread workgroup per CU, tile tickets; it finishes its own parked pairs)
        auto launch_fused = [&](auto kernel) -> int {
            DIG_REQUIRE(cu_count() <= 1024, "at most 1024 workgroups (overflow segments)");
            DIG_LAUNCH_STAGE(DIG_PIPE_STATISTICS, kernel, dim3(grid_for(E * C, 1024, 1)), dim3(1024), 0, s, a);
            return DIG_OK;
        };
        if (which == 2 && form == 1 && tickets == 1024) {
            // one 1024-thread workgroup per CU drawing tiles from an LDS counter (default)
            int rc;
            if (a.rec) {
                DIG_REQUIRE(a.bin_pack && DIG_ES_INWAVE, "DIG_PIPE_RECORDS needs the packed bin records (dig_bin_records_pack)");
                rc = launch_fused(element_stats_stream_fused_kernel<1024, true, 3, true>);
            } else if (a.bin_pack)
                rc = launch_fused(element_stats_stream_fused_kernel<1024, true, 3>);
            else
                rc = launch_fused(element_stats_stream_fused_kernel<1024, true>);
            if (rc) return rc;
            finished_in_wave = DIG_ES_INWAVE != 0;
        } else if (which == 2 && form == 1 && tickets == 256) {
            const int g = grid_for(E * C, 256, std::min(occupancy(3, element_stats_stream_fused_kernel<256, true>, 256), stream_blocks_per_cu(8)));
            hipLaunchKernelGGL((element_stats_stream_fused_kernel<256, true>), dim3(g), dim3(256), 0, s, a);
        } else if (which == 2 && form == 1) {
            const int g = grid_for(E * C, 256, std::min(occupancy(3, element_stats_stream_fused_kernel<256, false>, 256), stream_blocks_per_cu(8)));
            hipLaunchKernelGGL((element_stats_stream_fused_kernel<256, false>), dim3(g), dim3(256), 0, s, a);
        } else if (which == 2) {
            const int g = grid_for(E * C, kBlock, std::min(occupancy(2, element_stats_stream_kernel<false, true>, kBlock), 5));
            hipLaunchKernelGGL((element_stats_stream_kernel<false, true>), dim3(g), dim3(kBlock), 0, s, a);
        } else if (which != 2 && DIG_ES_INWAVE && given_form) {
            // dig_element_stats on the pipelined kernel of dig_element_pipeline (one 1024-thread workgroup per CU, tickets,
            // second pass inside): DIG_ES_GIVEN_FORM=0 keeps round 1's two-stage kernels + the compacted kernel
            const int rc = which == 1 ? launch_fused(element_stats_stream_fused_kernel<1024, true, 2>)
                                      : launch_fused(element_stats_stream_fused_kernel<1024, true, 1>);
            if (rc) return rc;
            finished_in_wave = true;
        } else if (which == 1) {
            // (the two-stage forms are bound by VALU issue, not by latency hiding: slightly fewer than the maximum of
            //  resident workgroups measured best -- fewer waves contend for the scalar unit and the instruction cache)
            const int g = grid_for(E * C, kBlock, std::min(occupancy(1, element_stats_stream_kernel<true, false>, kBlock), 6));
            hipLaunchKernelGGL((element_stats_stream_kernel<true, false>), dim3(g), dim3(kBlock), 0, s, a);
        } else {
            const int g = grid_for(E * C, kBlock, std::min(occupancy(0, element_stats_stream_kernel<false, false>, kBlock), 6));
            hipLaunchKernelGGL((element_stats_stream_kernel<false, false>), dim3(g), dim3(kBlock), 0, s, a);
        }
    } else
        hipLaunchKernelGGL(element_stats_single_pass_kernel, dim3(grid), dim3(kBlock), 0, s, a);
    DIG_HIP_TRY(hipGetLastError());
    if (wl) {
        // (the worklist length is only known on the device: a grid that covers 1 % of the pairs in one round, but never more
        //  workgroups than are resident at once -- the waves of a second batch would start when the first ones end and
        //  double the length of a kernel whose waves all live equally long: 18 against 12 us, measured)
        if (finished_in_wave) return DIG_OK;      // the fused stream pass finishes its own slow pairs
        const int64_t slow_pairs = std::max<int64_t>(E * C / 100, 1);
        static int slow_resident[kMaxDevices] = {};
        int slow_dev = 0;
        DIG_HIP_TRY(hipGetDevice(&slow_dev));
        DIG_REQUIRE(slow_dev >= 0 && slow_dev < kMaxDevices, "device index below 64");
        if (!slow_resident[slow_dev]) {
            int per_cu = 0;
            if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, element_stats_slow_kernel, kSlowBlock, 0) != hipSuccess || per_cu < 1) per_cu = 4;
            slow_resident[slow_dev] = per_cu;
        }
        const int slow_grid = (int)std::min<int64_t>(std::max<int64_t>((slow_pairs + kSlowWaves * kSlowPairsPerWave - 1) / (kSlowWaves * kSlowPairsPerWave), 1),
                                                     (int64_t)cu_count() * slow_resident[slow_dev]);
        hipLaunchKernelGGL(element_stats_slow_kernel, dim3(slow_grid), dim3(kSlowBlock), 0, s, a);
        // (the header keeps the count: the next launch sequence clears it in front -- context kernel or memset node.  "The
        //  last workgroup clears behind itself" costs one device-scope atomic per workgroup on ONE address, 13 ns each:
        //  18 us for this grid, measured.)
        DIG_HIP_TRY(hipGetLastError());
    }
    return DIG_OK;
}

}  // namespace dig

extern "C" {

int dig_element_stats(const double* mu, const double* sigma, const double* mu_indel, const double* sigma_indel,
                      const double* pi_sum, const double* pi_indel, int pi_indel_per_cohort, const int32_t* obs_snv,
                      const int32_t* obs_samples, const int32_t* obs_indel, const double* cj, const double* cj_indel,
                      double* out, int64_t E, int64_t C, void* workspace, int64_t workspace_bytes, void* stream)
{
    DIG_REQUIRE(E >= 0 && C >= 0, "E, C >= 0");
    if (E == 0 || C == 0) return DIG_OK;
    DIG_REQUIRE(mu && sigma && pi_sum && pi_indel && obs_snv && obs_samples && obs_indel && cj && cj_indel && out,
                "non-null pointers");
    DIG_REQUIRE((mu_indel == nullptr) == (sigma_indel == nullptr), "mu_indel and sigma_indel both set or both NULL");
    return element_stats_launch(mu, sigma, mu_indel, sigma_indel, pi_sum, pi_indel, pi_indel_per_cohort, obs_snv,
                                obs_samples, obs_indel, cj, cj_indel, out, E, C, workspace, workspace_bytes, stream, nullptr, 0);
}

int dig_element_stats_host(const double* mu, const double* sigma, const double* mu_indel, const double* sigma_indel,
                           const double* pi_sum, const double* pi_indel, int pi_indel_per_cohort,
                           const int32_t* obs_snv, const int32_t* obs_samples, const int32_t* obs_indel,
                           const double* cj, const double* cj_indel, double* out, int64_t E, int64_t C, int device)
{
    DIG_REQUIRE(E >= 0 && C >= 0, "E, C >= 0");
    if (E == 0 || C == 0) return DIG_OK;
    DIG_REQUIRE(mu && sigma && pi_sum && pi_indel && obs_snv && obs_samples && obs_indel && cj && cj_indel && out,
                "non-null pointers");
    DIG_HIP_TRY(hipSetDevice(device));
    const size_t n = (size_t)E * (size_t)C;
    const size_t nd = n * sizeof(double), ni = n * sizeof(int32_t);
    const size_t npi = (pi_indel_per_cohort ? n : (size_t)E) * sizeof(double);
    DevBuf dmu, dsg, dmui, dsgi, dps, dpi, dk1, dk2, dk3, dcj, dcji, dout;
    DIG_HIP_TRY(dmu.alloc(nd));
    DIG_HIP_TRY(dsg.alloc(nd));
    DIG_HIP_TRY(dps.alloc(nd));
    DIG_HIP_TRY(dpi.alloc(npi));
    DIG_HIP_TRY(dk1.alloc(ni));
    DIG_HIP_TRY(dk2.alloc(ni));
    DIG_HIP_TRY(dk3.alloc(ni));
    DIG_HIP_TRY(dcj.alloc((size_t)C * sizeof(double)));
    DIG_HIP_TRY(dcji.alloc((size_t)C * sizeof(double)));
    DIG_HIP_TRY(dout.alloc(nd * DIG_ES_NPLANES));
    DIG_HIP_TRY(hipMemcpy(dmu.p, mu, nd, hipMemcpyHostToDevice));
    DIG_HIP_TRY(hipMemcpy(dsg.p, sigma, nd, hipMemcpyHostToDevice));
    if (mu_indel) {
        DIG_REQUIRE(sigma_indel, "sigma_indel with mu_indel");
        DIG_HIP_TRY(dmui.alloc(nd));
        DIG_HIP_TRY(dsgi.alloc(nd));
        DIG_HIP_TRY(hipMemcpy(dmui.p, mu_indel, nd, hipMemcpyHostToDevice));
        DIG_HIP_TRY(hipMemcpy(dsgi.p, sigma_indel, nd, hipMemcpyHostToDevice));
    }
    DIG_HIP_TRY(hipMemcpy(dps.p, pi_sum, nd, hipMemcpyHostToDevice));
    DIG_HIP_TRY(hipMemcpy(dpi.p, pi_indel, npi, hipMemcpyHostToDevice));
    DIG_HIP_TRY(hipMemcpy(dk1.p, obs_snv, ni, hipMemcpyHostToDevice));
    DIG_HIP_TRY(hipMemcpy(dk2.p, obs_samples, ni, hipMemcpyHostToDevice));
    DIG_HIP_TRY(hipMemcpy(dk3.p, obs_indel, ni, hipMemcpyHostToDevice));
    DIG_HIP_TRY(hipMemcpy(dcj.p, cj, (size_t)C * sizeof(double), hipMemcpyHostToDevice));
    DIG_HIP_TRY(hipMemcpy(dcji.p, cj_indel, (size_t)C * sizeof(double), hipMemcpyHostToDevice));
    DevBuf dws;
    const int64_t wsb = dig_element_stats_workspace(E, C);
    if (wsb > 0) DIG_HIP_TRY(dws.alloc((size_t)wsb));
    int rc = dig_element_stats(dmu.as<double>(), dsg.as<double>(), mu_indel ? dmui.as<double>() : nullptr,
                               mu_indel ? dsgi.as<double>() : nullptr, dps.as<double>(), dpi.as<double>(),
                               pi_indel_per_cohort, dk1.as<int32_t>(), dk2.as<int32_t>(), dk3.as<int32_t>(),
                               dcj.as<double>(), dcji.as<double>(), dout.as<double>(), E, C, wsb > 0 ? dws.p : nullptr,
                               wsb, nullptr);
    if (rc) return rc;
    DIG_HIP_TRY(hipDeviceSynchronize());
    DIG_HIP_TRY(hipMemcpy(out, dout.p, nd * DIG_ES_NPLANES, hipMemcpyDeviceToHost));
    return DIG_OK;
}

int dig_tiled_nb_test(const double* pt, int pt_per_cohort, const int32_t* k, const double* mu, const double* sigma,
                      double* pval, double* exp_out, int64_t C, int64_t n_bins, int64_t n_tiles, void* stream)
{
    DIG_REQUIRE(C >= 0 && n_bins >= 0 && n_tiles >= 0, "non-negative sizes");
    const int64_t n = C * n_bins * n_tiles;
    if (n == 0) return DIG_OK;
    DIG_REQUIRE(pt && k && mu && sigma && pval && exp_out, "non-null pointers");
    // fastdiv is exact while i * d < 2^64 (and needs d >= 2): n * n_tiles and (C n_bins) * n_bins stay far below that
    if (n_tiles >= 17 && n < ((int64_t)1 << 40) && n_bins < ((int64_t)1 << 24))
        hipLaunchKernelGGL(tiled_nb_kernel, dim3(grid_for(n, kTiledChunk)), dim3(kBlock), 0, (hipStream_t)stream, pt,
                           pt_per_cohort, k, mu, sigma, pval, exp_out, C, n_bins, n_tiles, make_fastdiv(n_tiles),
                           make_fastdiv(n_bins));
    else
        hipLaunchKernelGGL(tiled_nb_generic_kernel, dim3(grid_for(n, kBlock)), dim3(kBlock), 0, (hipStream_t)stream, pt,
                           pt_per_cohort, k, mu, sigma, pval, exp_out, C, n_bins, n_tiles);
    DIG_HIP_TRY(hipGetLastError());
    return DIG_OK;
}

int dig_tiled_nb_test_host(const double* pt, int pt_per_cohort, const int32_t* k, const double* mu,
                           const double* sigma, double* pval, double* exp_out, int64_t C, int64_t n_bins,
                           int64_t n_tiles, int device)
{
    DIG_REQUIRE(C >= 0 && n_bins >= 0 && n_tiles >= 0, "non-negative sizes");
    const size_t n = (size_t)C * n_bins * n_tiles;
    if (n == 0) return DIG_OK;
    DIG_REQUIRE(pt && k && mu && sigma && pval && exp_out, "non-null pointers");
    DIG_HIP_TRY(hipSetDevice(device));
    const size_t npt = (pt_per_cohort ? n : (size_t)n_bins * n_tiles) * sizeof(double);
    const size_t ncb = (size_t)C * n_bins * sizeof(double);
    DevBuf dpt, dk, dmu, dsg, dpv, dex;
    DIG_HIP_TRY(dpt.alloc(npt));
    DIG_HIP_TRY(dk.alloc(n * sizeof(int32_t)));
    DIG_HIP_TRY(dmu.alloc(ncb));
    DIG_HIP_TRY(dsg.alloc(ncb));
    DIG_HIP_TRY(dpv.alloc(n * sizeof(double)));
    DIG_HIP_TRY(dex.alloc(n * sizeof(double)));
    DIG_HIP_TRY(hipMemcpy(dpt.p, pt, npt, hipMemcpyHostToDevice));
    DIG_HIP_TRY(hipMemcpy(dk.p, k, n * sizeof(int32_t), hipMemcpyHostToDevice));
    DIG_HIP_TRY(hipMemcpy(dmu.p, mu, ncb, hipMemcpyHostToDevice));
    DIG_HIP_TRY(hipMemcpy(dsg.p, sigma, ncb, hipMemcpyHostToDevice));
    int rc = dig_tiled_nb_test(dpt.as<double>(), pt_per_cohort, dk.as<int32_t>(), dmu.as<double>(), dsg.as<double>(),
                               dpv.as<double>(), dex.as<double>(), C, n_bins, n_tiles, nullptr);
    if (rc) return rc;
    DIG_HIP_TRY(hipDeviceSynchronize());
    DIG_HIP_TRY(hipMemcpy(pval, dpv.p, n * sizeof(double), hipMemcpyDeviceToHost));
    DIG_HIP_TRY(hipMemcpy(exp_out, dex.p, n * sizeof(double), hipMemcpyDeviceToHost));
    return DIG_OK;
}

}  // extern "C"
