// dig_gene.hip -- the gene route's statistics block as ONE launch, and accumulation + statistics as one operation.
//
// Reference (DIGDriver/driver_model/transfer_tools.py), per gene and cohort, six mutation classes
// SYN, MIS, NONS, SPL, TRUNC = NONS + SPL, NONSYN = MIS + TRUNC:
//   gene_expected_muts_nb            :331-340   EXP_c = ALPHA * THETA * Pi_c
//   gene_pvalue_burden_nb            :394-456   PVAL_c_BURDEN        = nb_pvalue_greater_midp(OBS_c,    ALPHA, 1 / (THETA Pi_c + 1))
//   gene_pvalue_burden_nb_by_sample  :554-583   PVAL_c_BURDEN_SAMPLE = nb_pvalue_greater_midp(N_SAMP_c, ALPHA, 1 / (THETA Pi_c + 1))
//   gene_pvalue_indel                :709-729   THETA_INDEL *= t_indel; EXP_INDEL; PVAL_INDEL_BURDEN
//   Fisher                           :860-861   PVAL_MUT_BURDEN = chi2.sf(-2 (ln PVAL_TRUNC_BURDEN + ln PVAL_INDEL_BURDEN), 4)
// with ALPHA, THETA = normal_params_to_gamma(MU, SIGMA), THETA *= cj (:17-19, :266).  The reference makes thirteen scipy
// calls per cohort from pandas; the package's column-by-column mirror makes four launches.  Here one thread takes one
// (gene, cohort) pair through all thirteen tests: the count and the sample count of a class share one pass of the scaled
// recurrence (same p), anything it cannot finish takes the scalar path with scipy's semantics.  Same device functions,
// same bits as the elementwise entry points (dig_nb_midp_upper, dig_fisher).
#include "dig_common.hpp"
#include "dig_math.hpp"

namespace dig {

int accumulate_launch(const double* bin_mu, const double* bin_std, const int32_t* bin_y, const uint8_t* bin_flag,
                      const int32_t* bin_ctx, const int64_t* ov_ptr, const int32_t* ov_idx, const int32_t* L, int n_class,
                      const uint8_t* strand_minus, const int32_t* gene_length, const double* d_pr, double* MU,
                      double* SIGMA, int32_t* R_OBS, int32_t* FLAG, double* P, int32_t* R_SIZE, int32_t* ELT_SIZE,
                      double* P_INDEL, int64_t N, int64_t E, int64_t C, void* workspace, int64_t workspace_bytes,
                      void* stream, int do_rates, unsigned* zero_dwords, int n_zero, int parts);

struct GeneStatsArgs {
    const double *mu, *sigma, *mu_indel, *sigma_indel;   // [G, C]; the indel pair may be NULL (= mu, sigma)
    const double* pi;                                    // [G, n_pi, C]
    const double* pi_indel;                              // [G] or [G, C]
    const int32_t* obs;                                  // [G, 5, C]: SYN, MIS, NONS, SPL, INDEL
    const int32_t* n_samp;                               // [G, 6, C]
    const double *cj, *t_indel;                          // [C]
    double* out;                                         // [22, G, C]
    int64_t G, C;
    int n_pi, pi_indel_per_cohort, with_indel;
};

constexpr int kGeneBlock = 256;

__global__ __launch_bounds__(kGeneBlock) void gene_stats_kernel(GeneStatsArgs a)
{
    nb_tables_init();
    const int64_t n = a.G * a.C;
    const int64_t stride = (int64_t)gridDim.x * kGeneBlock;
    for (int64_t i = (int64_t)blockIdx.x * kGeneBlock + threadIdx.x; i < n; i += stride) {
        const int64_t g = i / a.C, c = i - g * a.C;
        const GammaParams gp = normal_params_to_gamma(a.mu[i], a.sigma[i]);
        const double alpha = gp.alpha, theta = mul_rn(gp.theta, a.cj[c]);          // :266 THETA * cj
        double pi[6];
        for (int q = 0; q < a.n_pi; ++q) pi[q] = a.pi[(g * a.n_pi + q) * a.C + c];
        if (a.n_pi == 4) {                         // straight from the accumulation: P_TRUNC = P_NONS + P_SPLICE
            pi[4] = pi[2] + pi[3];                 // (genic_driver_tools.py:199), Pi_NONSYN = Pi_MIS + Pi_TRUNC (transfer_tools.py:48)
            pi[5] = pi[1] + pi[4];
        }
        int k[6], ns[6];
        for (int q = 0; q < 4; ++q) k[q] = a.obs[(g * 5 + q) * a.C + c];
        k[4] = k[2] + k[3];                        // OBS_TRUNC, OBS_NONSYN (:253-254)
        k[5] = k[1] + k[4];
        for (int q = 0; q < 6; ++q) ns[q] = a.n_samp[(g * 6 + q) * a.C + c];
        const double rate = mul_rn(alpha, theta);
        double pv_trunc = 0.0;
#pragma unroll 1
        for (int q = 0; q < 6; ++q) {
            const double p = nb_success_prob(theta, pi[q]);
            double r1 = 0.0, r2 = 0.0;
            const double k1 = (double)k[q], k2 = (double)ns[q];
            const unsigned done = nb_midp_upper_fast2<1>(k1, k2, 3u, alpha, p, r1, r2);
            if (!(done & 1u)) r1 = nb_midp_upper_unresolved(k1, alpha, p);
            if (!(done & 2u)) r2 = nb_midp_upper_unresolved(k2, alpha, p);
            a.out[(0 + q) * n + i] = mul_rn(rate, pi[q]);                            // EXP_c
            a.out[(6 + q) * n + i] = r1;                                             // PVAL_c_BURDEN
            a.out[(12 + q) * n + i] = r2;                                            // PVAL_c_BURDEN_SAMPLE
            if (q == 4) pv_trunc = r1;
        }
        double theta_i = dnan(), exp_i = dnan(), pv_i = dnan(), pv_mut = dnan();
        if (a.with_indel) {
            const GammaParams gi = a.mu_indel ? normal_params_to_gamma(a.mu_indel[i], a.sigma_indel[i]) : gp;
            const double pii = a.pi_indel_per_cohort ? a.pi_indel[i] : a.pi_indel[g];
            theta_i = mul_rn(gi.theta, a.t_indel[c]);                                // :723
            exp_i = mul_rn(mul_rn(gi.alpha, theta_i), pii);
            const double p_i = nb_success_prob(theta_i, pii);
            const double ki = (double)a.obs[(g * 5 + 4) * a.C + c];
            double r1 = 0.0, dummy = 0.0;
            const unsigned done = nb_midp_upper_fast2<1>(ki, 0.0, 1u, gi.alpha, p_i, r1, dummy);
            pv_i = (done & 1u) ? r1 : nb_midp_upper_unresolved(ki, gi.alpha, p_i);
            pv_mut = fisher_combine_fast(pv_trunc, pv_i);                            // :860-861
        }
        a.out[18 * n + i] = theta_i;
        a.out[19 * n + i] = exp_i;
        a.out[20 * n + i] = pv_i;
        a.out[21 * n + i] = pv_mut;
    }
}

static int gene_stats_launch(const GeneStatsArgs& a, hipStream_t stream)
{
    hipLaunchKernelGGL(gene_stats_kernel, dim3(grid_for(a.G * a.C, kGeneBlock, 8)), dim3(kGeneBlock), 0, stream, a);
    DIG_HIP_TRY(hipGetLastError());
    return DIG_OK;
}

}  // namespace dig

using namespace dig;

extern "C" {

int dig_gene_stats(const double* mu, const double* sigma, const double* mu_indel, const double* sigma_indel, const double* pi,
                   int n_pi, const double* pi_indel, int pi_indel_per_cohort, const int32_t* obs, const int32_t* n_samp,
                   const double* cj, const double* t_indel, int with_indel, double* out, int64_t G, int64_t C, void* stream)
{
    DIG_REQUIRE(G >= 0 && C >= 0, "G, C >= 0");
    DIG_REQUIRE(n_pi == 4 || n_pi == 6, "n_pi: 4 (SYN, MIS, NONS, SPL: TRUNC and NONSYN are formed here) or 6");
    if (G == 0 || C == 0) return DIG_OK;
    DIG_REQUIRE(mu && sigma && pi && obs && n_samp && cj && out, "non-null pointers");
    DIG_REQUIRE((mu_indel == nullptr) == (sigma_indel == nullptr), "mu_indel and sigma_indel together");
    DIG_REQUIRE(!with_indel || (pi_indel && t_indel), "pi_indel and t_indel for the indel block");
    const GeneStatsArgs a{mu, sigma, mu_indel, sigma_indel, pi, pi_indel, obs, n_samp, cj, t_indel, out, G, C,
                          n_pi, pi_indel_per_cohort, with_indel};
    return gene_stats_launch(a, (hipStream_t)stream);
}

int dig_gene_stats_host(const double* mu, const double* sigma, const double* mu_indel, const double* sigma_indel, const double* pi,
                        int n_pi, const double* pi_indel, int pi_indel_per_cohort, const int32_t* obs, const int32_t* n_samp,
                        const double* cj, const double* t_indel, int with_indel, double* out, int64_t G, int64_t C, int device)
{
    DIG_REQUIRE(G >= 0 && C >= 0, "G, C >= 0");
    DIG_REQUIRE(n_pi == 4 || n_pi == 6, "n_pi: 4 or 6");
    if (G == 0 || C == 0) return DIG_OK;
    DIG_REQUIRE(mu && sigma && pi && obs && n_samp && cj && out, "non-null pointers");
    DIG_REQUIRE(!with_indel || (pi_indel && t_indel), "pi_indel and t_indel for the indel block");
    DIG_HIP_TRY(hipSetDevice(device));
    const size_t nGC = (size_t)G * C;
    DevBuf dmu, dsg, dmi, dsi, dpi, dpii, dob, dns, dcj, dti, dout;
#define UP(buf, src, bytes)                                                    \
    DIG_HIP_TRY(buf.alloc(bytes));                                             \
    if (src) DIG_HIP_TRY(hipMemcpy(buf.p, src, bytes, hipMemcpyHostToDevice))
    UP(dmu, mu, nGC * 8);
    UP(dsg, sigma, nGC * 8);
    UP(dmi, mu_indel, nGC * 8);
    UP(dsi, sigma_indel, nGC * 8);
    UP(dpi, pi, nGC * n_pi * 8);
    UP(dpii, pi_indel, (pi_indel_per_cohort ? nGC : (size_t)G) * 8);
    UP(dob, obs, nGC * 5 * 4);
    UP(dns, n_samp, nGC * 6 * 4);
    UP(dcj, cj, (size_t)C * 8);
    UP(dti, t_indel, (size_t)C * 8);
#undef UP
    DIG_HIP_TRY(dout.alloc(nGC * 22 * 8));
    int rc = dig_gene_stats(dmu.as<double>(), dsg.as<double>(), mu_indel ? dmi.as<double>() : nullptr,
                            sigma_indel ? dsi.as<double>() : nullptr, dpi.as<double>(), n_pi, pi_indel ? dpii.as<double>() : nullptr,
                            pi_indel_per_cohort, dob.as<int32_t>(), dns.as<int32_t>(), dcj.as<double>(),
                            t_indel ? dti.as<double>() : nullptr, with_indel, dout.as<double>(), G, C, nullptr);
    if (rc) return rc;
    DIG_HIP_TRY(hipDeviceSynchronize());
    DIG_HIP_TRY(hipMemcpy(out, dout.p, nGC * 22 * 8, hipMemcpyDeviceToHost));
    return DIG_OK;
}

int64_t dig_accumulate_workspace(int64_t E, int64_t C);

int dig_gene_pipeline(const double* bin_mu, const double* bin_std, const int32_t* bin_y, const uint8_t* bin_flag,
                      const int32_t* bin_ctx, const int64_t* ov_ptr, const int32_t* ov_idx, const int32_t* L,
                      const uint8_t* strand_minus, const int32_t* gene_length, const double* d_pr, const int32_t* obs,
                      const int32_t* n_samp, const double* cj, const double* t_indel, int with_indel, double* MU, double* SIGMA,
                      int32_t* R_OBS, int32_t* FLAG, double* P, int32_t* R_SIZE, int32_t* ELT_SIZE, double* P_INDEL, double* out,
                      int64_t N, int64_t G, int64_t C, void* workspace, int64_t workspace_bytes, void* stream)
{
    DIG_REQUIRE(N >= 0 && G >= 0 && C >= 0, "N, G, C >= 0");
    if (G == 0 || C == 0) return DIG_OK;
    DIG_REQUIRE(obs && n_samp && cj && out, "non-null statistics arguments");
    DIG_REQUIRE(workspace && workspace_bytes >= dig_accumulate_workspace(G, C), "workspace of dig_accumulate_workspace(G, C) bytes");
    // genic_model (genic_driver_tools.py:31-203): four class columns of L, P_INDEL = GENE_LENGTH / R_SIZE
    int rc = accumulate_launch(bin_mu, bin_std, bin_y, bin_flag, bin_ctx, ov_ptr, ov_idx, L, 4, strand_minus, gene_length, d_pr, MU,
                               SIGMA, R_OBS, FLAG, P, R_SIZE, ELT_SIZE, P_INDEL, N, G, C, workspace, workspace_bytes, stream, 1,
                               nullptr, 0, 3);
    if (rc) return rc;
    return dig_gene_stats(MU, SIGMA, nullptr, nullptr, P, 4, P_INDEL, 0, obs, n_samp, cj, t_indel, with_indel, out, G, C, stream);
}

}  // extern "C"
