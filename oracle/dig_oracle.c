/*
 * dig_oracle.c -- plain-C CPU restatement of the DIGDriver burden-test hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  This file is the checker, never the product: only
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load the
 * library built from it.  The product (digdriver_amd/csrc) never links it.
 *
 * Pinning: every entry point is checked against golden vectors produced by the
 * real reference in the build container (tests/golden/make_golden.py, scipy
 * 1.15.3) in tests/test_oracle_golden.py.
 *
 * The reference calls third-party arithmetic for the NB tests:
 *   scipy.special.betainc(a, b, x)   -> regularised incomplete beta I_x(a, b)
 *   scipy.stats.nbinom.pmf(k, n, p)  -> Gamma(k+n)/(k! Gamma(n)) p^n (1-p)^k
 *   scipy.stats.chi2.sf(x, df=4)     -> exp(-x/2) (1 + x/2)
 * (scipy pinned 1.5.3 in conda-recipe/meta.yaml:61; goldens made with 1.15.3).
 * Their published definitions are restated here: I_x(a,b) by the classical
 * continued fraction (modified Lentz) with the x <-> 1-x switch at
 * x = (a+1)/(a+b+2); pmf through lgamma in log space.
 *
 * file:line citations are relative to the reference tree.
 *
 * Build:  make -C oracle    (gcc -O2 -fopenmp -shared)
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define DIG_CF_EPS 1e-15
#define DIG_CF_MAXIT 20000
#define DIG_FPMIN 1e-300

/* continued fraction for I_x(a,b), valid (fast) for x < (a+1)/(a+b+2) */
static double betacf(double a, double b, double x)
{
    double qab = a + b, qap = a + 1.0, qam = a - 1.0;
    double c = 1.0, d = 1.0 - qab * x / qap;
    if (fabs(d) < DIG_FPMIN) d = DIG_FPMIN;
    d = 1.0 / d;
    double h = d;
    for (int m = 1; m <= DIG_CF_MAXIT; ++m) {
        double m2 = 2.0 * m;
        double aa = m * (b - m) * x / ((qam + m2) * (a + m2));
        d = 1.0 + aa * d; if (fabs(d) < DIG_FPMIN) d = DIG_FPMIN;
        c = 1.0 + aa / c; if (fabs(c) < DIG_FPMIN) c = DIG_FPMIN;
        d = 1.0 / d;
        h *= d * c;
        aa = -(a + m) * (qab + m) * x / ((a + m2) * (qap + m2));
        d = 1.0 + aa * d; if (fabs(d) < DIG_FPMIN) d = DIG_FPMIN;
        c = 1.0 + aa / c; if (fabs(c) < DIG_FPMIN) c = DIG_FPMIN;
        d = 1.0 / d;
        double del = d * c;
        h *= del;
        if (fabs(del - 1.0) <= DIG_CF_EPS) break;
    }
    return h;
}

/* scipy.special.betainc(a, b, x) */
double dig_oracle_betainc(double a, double b, double x)
{
    if (isnan(a) || isnan(b) || isnan(x)) return NAN;
    if (a <= 0.0 || b <= 0.0 || x < 0.0 || x > 1.0) return NAN;   /* scipy 1.15.3: a==0 or b==0 -> nan */
    if (isinf(a) || isinf(b)) return NAN;
    if (x == 0.0) return 0.0;
    if (x == 1.0) return 1.0;
    double y = 1.0 - x;
    double lfront = a * log(x) + b * log(y) + lgamma(a + b) - lgamma(a) - lgamma(b);
    if (x < (a + 1.0) / (a + b + 2.0))
        return exp(lfront) * betacf(a, b, x) / a;
    return 1.0 - exp(lfront) * betacf(b, a, y) / b;
}

/* scipy.stats.nbinom.pmf(k, n, p): argcheck (n>0)&(p>0)&(p<=1) else nan; non-integer or
 * negative k -> 0 (rv_discrete support check) */
double dig_oracle_nbinom_pmf(double k, double n, double p)
{
    if (isnan(k) || isnan(n) || isnan(p)) return NAN;
    if (!(n > 0.0) || !(p > 0.0) || !(p <= 1.0) || isinf(n)) return NAN;
    if (k < 0.0 || floor(k) != k) return 0.0;
    if (p == 1.0) return k == 0.0 ? 1.0 : 0.0;
    double l = lgamma(k + n) - lgamma(k + 1.0) - lgamma(n) + n * log(p) + k * log1p(-p);
    return exp(l);
}

/* nb_model.py:271-278 */
double dig_oracle_nb_midp_upper(double k, double alpha, double p)
{
    return 0.5 * dig_oracle_nbinom_pmf(k, alpha, p) + dig_oracle_betainc(k + 1.0, alpha, 1.0 - p);
}

/* nb_model.py:243-256 */
double dig_oracle_nb_greater(double k, double alpha, double p)
{
    if (k == 0.0) return 1.0;
    double pv = dig_oracle_betainc(k, alpha, 1.0 - p);
    if (pv == 0.0) pv = dig_oracle_nbinom_pmf(k, alpha, p);
    return pv;
}

/* nb_model.py:298-314 */
double dig_oracle_nb_exact(double k, double alpha, double p)
{
    double mu = alpha * (1.0 - p) / p;
    if (k < mu) return dig_oracle_betainc(alpha, k + 1.0, p);
    double pv = dig_oracle_betainc(k, alpha, 1.0 - p);
    if (pv == 0.0) pv = dig_oracle_nbinom_pmf(k, alpha, p);
    return pv;
}

/* nb_model.py:316-337 */
double dig_oracle_nb_midp_twosided(double k, double alpha, double p)
{
    double mu = alpha * (1.0 - p) / p;
    double pmf = dig_oracle_nbinom_pmf(k, alpha, p);
    if (k < mu) {
        if (k > 0.0) return 0.5 * pmf + dig_oracle_betainc(alpha, k, p);
        return 0.5 * pmf;
    }
    return 0.5 * pmf + dig_oracle_betainc(k + 1.0, alpha, 1.0 - p);
}

/* transfer_tools.py:1086-1087: chi2.sf(-2 (ln p1 + ln p2), df=4) */
double dig_oracle_fisher(double p1, double p2)
{
    if (isnan(p1) || isnan(p2)) return NAN;
    double h = -(log(p1) + log(p2));   /* x2 / 2 */
    if (isnan(h)) return NAN;
    if (h < 0.0) return 1.0;           /* chi2.sf of a negative argument */
    if (isinf(h)) return 0.0;
    return exp(-h) * (1.0 + h);
}

void dig_oracle_set_threads(int n)
{
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
#else
    (void)n;
#endif
}

#define VEC3(NAME, FN)                                                                   \
    void NAME(const double *k, const double *alpha, const double *p, double *out, int64_t n) \
    {                                                                                    \
        _Pragma("omp parallel for schedule(dynamic, 1024)")                              \
        for (int64_t i = 0; i < n; ++i) out[i] = FN(k[i], alpha[i], p[i]);               \
    }
VEC3(dig_oracle_nb_midp_upper_v, dig_oracle_nb_midp_upper)
VEC3(dig_oracle_nb_greater_v, dig_oracle_nb_greater)
VEC3(dig_oracle_nb_exact_v, dig_oracle_nb_exact)
VEC3(dig_oracle_nb_midp_twosided_v, dig_oracle_nb_midp_twosided)
VEC3(dig_oracle_betainc_v, dig_oracle_betainc)
VEC3(dig_oracle_nbinom_pmf_v, dig_oracle_nbinom_pmf)

void dig_oracle_fisher_v(const double *p1, const double *p2, double *out, int64_t n)
{
#pragma omp parallel for
    for (int64_t i = 0; i < n; ++i) out[i] = dig_oracle_fisher(p1[i], p2[i]);
}

/*
 * Element statistics block over a dense [E, C] problem (cohort fastest):
 * transfer_tools.py:17-19 (ALPHA, THETA), :300 (THETA *= cj), :343-344 (EXP_SNV),
 * :473-482 / :594-615 (mid-p SNV and sample tests), :737-745 (indel), :1086-1087 (Fisher).
 * pi_indel_stride: 0 -> pi_indel is [E]; 1 -> [E, C].
 * out: seven [E, C] planes in the order EXP_SNV, PVAL_SNV_BURDEN, PVAL_SAMPLE_BURDEN,
 * THETA_INDEL, EXP_INDEL, PVAL_INDEL_BURDEN, PVAL_MUT_BURDEN.
 */
void dig_oracle_element_stats(const double *mu, const double *sigma, const double *pi_sum, const double *pi_indel,
                              int pi_indel_per_cohort, const int32_t *obs_snv, const int32_t *obs_samples,
                              const int32_t *obs_indel, const double *cj, const double *cj_indel, double *out,
                              int64_t E, int64_t C)
{
    int64_t n = E * C;
#pragma omp parallel for schedule(dynamic, 1024)
    for (int64_t i = 0; i < n; ++i) {
        int64_t e = i / C, c = i % C;
        double m = mu[i], s = sigma[i];
        double alpha = (m * m) / (s * s);
        double theta0 = (s * s) / m;
        double theta = theta0 * cj[c];
        double ps = pi_sum[i];
        double pi_i = pi_indel_per_cohort ? pi_indel[i] : pi_indel[e];
        double p = 1.0 / (theta * ps + 1.0);
        double pv_snv = dig_oracle_nb_midp_upper((double)obs_snv[i], alpha, p);
        double pv_smp = dig_oracle_nb_midp_upper((double)obs_samples[i], alpha, p);
        double theta_i = theta0 * cj_indel[c];
        double p_i = 1.0 / (theta_i * pi_i + 1.0);
        double pv_ind = dig_oracle_nb_midp_upper((double)obs_indel[i], alpha, p_i);
        out[0 * n + i] = alpha * theta * ps;
        out[1 * n + i] = pv_snv;
        out[2 * n + i] = pv_smp;
        out[3 * n + i] = theta_i;
        out[4 * n + i] = alpha * theta_i * pi_i;
        out[5 * n + i] = pv_ind;
        out[6 * n + i] = dig_oracle_fisher(pv_snv, pv_ind);
    }
}

/*
 * Per-element accumulation (genic_driver_tools.py:258-272 region part; :361-381 sequence
 * part; '-' strand: sequence_tools.py:633-634 == reverse-complement permutation rho of the
 * 64 context counts).  Layouts: bin_* [N, C] cohort fastest; bin_ctx [N, 64]; CSR overlaps;
 * L [E, n_class, 192] int32; d_pr [C, 192]; rho[64].
 * Outputs: MU, SIGMA [E,C] f64; R_OBS, FLAG [E,C] i32; P [E, n_class, C] f64;
 * R_SIZE, ELT_SIZE [E] i32; P_INDEL [E] f64.
 */
void dig_oracle_accumulate_elements(const double *bin_mu, const double *bin_std, const int32_t *bin_y,
                                    const uint8_t *bin_flag, const int32_t *bin_ctx, const int64_t *ov_ptr,
                                    const int32_t *ov_idx, const int32_t *L, int n_class, const uint8_t *strand_minus,
                                    const double *d_pr, const int32_t *rho, double *MU, double *SIGMA, int32_t *R_OBS,
                                    int32_t *FLAG, double *P, int32_t *R_SIZE, int32_t *ELT_SIZE, double *P_INDEL,
                                    int64_t E, int64_t C)
{
#pragma omp parallel for schedule(dynamic, 64)
    for (int64_t e = 0; e < E; ++e) {
        int64_t rc[64];
        memset(rc, 0, sizeof rc);
        for (int64_t c = 0; c < C; ++c) {
            double mu = 0.0, var = 0.0;
            int32_t ro = 0, fl = 0;
            for (int64_t q = ov_ptr[e]; q < ov_ptr[e + 1]; ++q) {
                int64_t b = ov_idx[q];
                mu += bin_mu[b * C + c];
                var += bin_std[b * C + c] * bin_std[b * C + c];
                ro += bin_y[b * C + c];
                fl |= (bin_flag[b * C + c] != 0);   /* numpy bool '+' is a logical OR (:268) */
            }
            MU[e * C + c] = mu; SIGMA[e * C + c] = sqrt(var); R_OBS[e * C + c] = ro; FLAG[e * C + c] = fl;
        }
        for (int64_t q = ov_ptr[e]; q < ov_ptr[e + 1]; ++q)
            for (int j = 0; j < 64; ++j) rc[j] += bin_ctx[(int64_t)ov_idx[q] * 64 + j];
        int64_t rsize = 0;
        for (int j = 0; j < 64; ++j) rsize += rc[j];
        int64_t lsum = 0;
        for (int q = 0; q < n_class; ++q)
            for (int j = 0; j < 192; ++j) lsum += L[(e * n_class + q) * 192 + j];
        R_SIZE[e] = (int32_t)rsize;
        ELT_SIZE[e] = (int32_t)(lsum / 3);
        P_INDEL[e] = (double)(lsum / 3) / (double)rsize;
        for (int64_t c = 0; c < C; ++c) {
            const double *d = d_pr + c * 192;
            double denom = 0.0;
            for (int j = 0; j < 192; ++j) {
                int ctx = j / 3;
                int64_t cnt = strand_minus[e] ? rc[rho[ctx]] : rc[ctx];
                denom += (double)cnt * d[j];
            }
            for (int q = 0; q < n_class; ++q) {
                double num = 0.0;
                const int32_t *Lq = L + (e * n_class + q) * 192;
                for (int j = 0; j < 192; ++j) num += (d[j] / denom) * (double)Lq[j];
                P[(e * n_class + q) * C + c] = num;
            }
        }
    }
}
