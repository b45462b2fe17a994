// dig_math.hpp -- FP64 device math for the DIGDriver burden tests on gfx950.
//
// Everything here is __device__ code for CDNA4; there is no host/CPU fallback.
//
// What is computed (reference: DIGDriver/sequence_model/nb_model.py:237-337):
//   I_x(a, b)            regularised incomplete beta  (scipy.special.betainc)
//   pmf(k; n, p)         negative-binomial pmf        (scipy.stats.nbinom.pmf)
//   mid-p upper tail     0.5 pmf(k) + I_{1-p}(k+1, alpha)          (nb_model.py:271-278)
//   exact / greater / two-sided mid-p siblings         (nb_model.py:243-256,298-337)
//   Fisher(p1, p2)       chi2.sf(-2 (ln p1 + ln p2), 4) = q (1 - ln q)   (transfer_tools.py:1086-1087)
//
// Numerical design (tolerance contract: <=1e-6 relative for p >= 1e-250):
//   * k is an integer count in every live call, so the NB tail is evaluated from the pmf
//     recurrence  t_{j+1} = t_j (alpha + j) x / (j + 1),  t_0 = p^alpha :
//       - small k  : S = sum_{j<k} t_j directly; result 1 - S - t_k/2 when that is not small
//                    (no lgamma, no continued fraction; ~4 FP64 ops per step);
//       - otherwise: modified-Lentz continued fraction for I_x(a,b) with the usual
//                    x <-> 1-x switch at x = (a+1)/(a+b+2); the prefactor
//                    x^a y^b / (a B(a,b)) is pmf(k) (k+alpha) x / (k+1), so one pmf serves
//                    both terms of the mid-p sum.
//   * the inputs alpha = mu^2/sigma^2, theta = sigma^2/mu * cj, p = 1/(theta Pi + 1) and
//     x = 1 - p are formed with FP contraction OFF so they are bit-identical to the
//     reference's numpy expressions (the tail is ~x^(k+1): a 1-ulp change in p matters).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace dig {

constexpr double kCfEps = 1e-15;
constexpr int kCfMaxIt = 20000;
constexpr double kFpMin = 1e-300;
constexpr int kSmallK = 128;         // direct-summation limit of the fast recurrence (see kFastLp0Min)
constexpr double kDirectMin = 1e-6;  // accept 1 - S - t/2 when >= this: abs err <= ~65 ulp(1) = 7e-15 -> rel 7e-9

__device__ __forceinline__ double dnan() { return __longlong_as_double(0x7ff8000000000000LL); }

// ---- exact-rounding input preparation (no FMA contraction) -------------------------
struct GammaParams {
    double alpha, theta;
};

// nb_model.py:237-241
__device__ __forceinline__ GammaParams normal_params_to_gamma(double mu, double sigma)
{
#pragma clang fp contract(off)
    GammaParams g;
    double m2 = mu * mu;
    double s2 = sigma * sigma;
    g.alpha = m2 / s2;
    g.theta = s2 / mu;
    return g;
}

// p = 1 / (theta * Pi + 1)   (transfer_tools.py:476-481)
__device__ __forceinline__ double nb_success_prob(double theta, double pi)
{
#pragma clang fp contract(off)
    double t = theta * pi;
    double d = t + 1.0;
#ifdef DIG_FAST_P          // developer A/B: hardware reciprocal + two Newton steps (p within an ulp) instead of the IEEE division
    double r = __builtin_amdgcn_rcp(d);
    r = __builtin_fma(__builtin_fma(-d, r, 1.0), r, r);
    return __builtin_fma(__builtin_fma(-d, r, 1.0), r, r);
#else
    return 1.0 / d;
#endif
}

__device__ __forceinline__ double mul_rn(double a, double b)
{
#pragma clang fp contract(off)
    return a * b;
}

// 1 / v to full precision from the hardware reciprocal estimate + two Newton steps (5 FP64 ops instead of the
// ~12 of an IEEE division; used only where the last-bit rounding of the quotient does not matter).
__device__ __forceinline__ double recip_nr(double v)
{
    double r = __builtin_amdgcn_rcp(v);
    r = fma(fma(-v, r, 1.0), r, r);
    r = fma(fma(-v, r, 1.0), r, r);
    return r;
}

// a * b + c with the wave-uniform constant c held in scalar registers: the compiler's own choice for a Horner step
// with a literal is v_mov_b32 x2 + v_fmac_f64 (three vector instructions); this is one, the constant costs two
// scalar moves that issue beside the vector pipe.
__device__ __forceinline__ double fma_sconst(double a, double b, double c)
{
    double d;
    asm("v_fma_f64 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "s"(c));
    return d;
}

// ---- exponential for arguments in [-745, 0] (p^alpha = exp(alpha log p)), < 1 ulp ----------------
// x = n ln2 + r, |r| <= ln2 / 2;  e^r from the degree-13 Taylor polynomial (remainder r^14 / 14! < 5e-18), scaled by
// 2^n with ldexp.  Anything outside (-745, 0] or NaN goes to the library exp.
// x in (-745, 0] (no checks).
__device__ __forceinline__ double fast_exp_neg_core(double x)
{
    const double n = rint(x * 1.4426950408889634);                       // log2(e)
    double r = fma(n, -6.93147180369123816490e-01, x);                   // ln2 split: high part has 32 zero low bits
    r = fma(n, -1.90821492927058770002e-10, r);
    double q = 1.6059043836821613e-10;                                   // 1/13!
    q = fma_sconst(q, r, 2.08767569878681e-09);                          // 1/12!
    q = fma_sconst(q, r, 2.505210838544172e-08);                         // 1/11!
    q = fma_sconst(q, r, 2.755731922398589e-07);                         // 1/10!
    q = fma_sconst(q, r, 2.7557319223985893e-06);                        // 1/9!
    q = fma_sconst(q, r, 2.48015873015873e-05);                          // 1/8!
    q = fma_sconst(q, r, 1.984126984126984e-04);                         // 1/7!
    q = fma_sconst(q, r, 1.3888888888888889e-03);                        // 1/6!
    q = fma_sconst(q, r, 8.333333333333333e-03);                         // 1/5!
    q = fma_sconst(q, r, 4.1666666666666664e-02);                        // 1/4!
    q = fma_sconst(q, r, 1.6666666666666666e-01);                        // 1/3!
    q = fma_sconst(q, r, 0.5);
    q = fma(q, r, 1.0);
    q = fma(q, r, 1.0);
    return ldexp(q, (int)n);
}

__device__ __forceinline__ double fast_exp_neg(double x)
{
    if (!(x <= 0.0 && x > -745.0)) return exp(x);
    return fast_exp_neg_core(x);
}

// ---- natural logarithm from a 128-entry LDS table + a short polynomial ---------------------------------------------
// x = 2^k z, z in [0.6875, 1.375) (bits(x) - bits(0.6875): the exponent field of the difference is k, its top seven
// mantissa bits select the bin); with c the bin centre, invc = double(1 / c) and logc = double(-log(invc)):
//     log x = k ln2 + logc + log1p(r),   r = z * invc - 1 (one fma, |r| <= 0.0046),
//     log1p(r) = r + r^2 P(r), P of degree 4 (truncation < 1.3e-19 absolute).
// 18 vector instructions and one 16-byte LDS read against ~40 for the classical atanh form this replaces (three logs
// per (element, cohort) pair in the statistics kernels).  Error: <= 1.2e-16 * max(|log x|, 0.005) absolute -- the
// quantity that matters here, since every use multiplies the result by alpha and exponentiates (or adds 1 to it).
// Table and coefficients: tools/gen_log_table.py (mpmath, 80 digits).  Every kernel that reaches this function calls
// nb_tables_init() first.  x must be a positive normal number (no checks).
static __device__ __constant__ const double kLogTabRom[128][2] = {
    {0x1.734f0c541fe8dp+0, -0x1.7cc7f7db46a0ep-2},
    {0x1.713786d9c7c09p+0, -0x1.76feecb947176p-2},
    {0x1.6f26016f26017p+0, -0x1.713e33a46a17cp-2},
    {0x1.6d1a62681c861p+0, -0x1.6b85b4cffa3fdp-2},
    {0x1.6b1490aa31a3dp+0, -0x1.65d558d4ce00bp-2},
    {0x1.691473a88d0c0p+0, -0x1.602d08af091ecp-2},
    {0x1.6719f3601671ap+0, -0x1.5a8cadbbedfa1p-2},
    {0x1.6524f853b4aa3p+0, -0x1.54f431b7be1a8p-2},
    {0x1.63356b88ac0dep+0, -0x1.4f637ebba9810p-2},
    {0x1.614b36831ae94p+0, -0x1.49da7f3bcc420p-2},
    {0x1.5f66434292dfcp+0, -0x1.44591e0539f49p-2},
    {0x1.5d867c3ece2a5p+0, -0x1.3edf463c1683ep-2},
    {0x1.5babcc647fa91p+0, -0x1.396ce359bbf53p-2},
    {0x1.59d61f123ccaap+0, -0x1.3401e12aecba0p-2},
    {0x1.5805601580560p+0, -0x1.2e9e2bce12286p-2},
    {0x1.56397ba7c52e2p+0, -0x1.2941afb186b7cp-2},
    {0x1.54725e6bb82fep+0, -0x1.23ec5991eba49p-2},
    {0x1.52aff56a8054bp+0, -0x1.1e9e1678899f5p-2},
    {0x1.50f22e111c4c5p+0, -0x1.1956d3b9bc2f9p-2},
    {0x1.4f38f62dd4c9bp+0, -0x1.14167ef367784p-2},
    {0x1.4d843bedc2c4cp+0, -0x1.0edd060b78082p-2},
    {0x1.4bd3edda68fe1p+0, -0x1.09aa572e6c6d4p-2},
    {0x1.4a27fad76014ap+0, -0x1.047e60cde83b7p-2},
    {0x1.4880522014880p+0, -0x1.feb2233ea07cbp-3},
    {0x1.46dce34596066p+0, -0x1.f474b134df228p-3},
    {0x1.453d9e2c776cap+0, -0x1.ea4449f04aaf5p-3},
    {0x1.43a2730abee4dp+0, -0x1.e020cc6235ab5p-3},
    {0x1.420b5265e5951p+0, -0x1.d60a17f903514p-3},
    {0x1.40782d10e6566p+0, -0x1.cc000c9db3c52p-3},
    {0x1.3ee8f42a5af07p+0, -0x1.c2028ab17f9b5p-3},
    {0x1.3d5d991aa75c6p+0, -0x1.b811730b823d4p-3},
    {0x1.3bd60d9232955p+0, -0x1.ae2ca6f672bd8p-3},
    {0x1.3a524387ac822p+0, -0x1.a454082e6ab03p-3},
    {0x1.38d22d366088ep+0, -0x1.9a8778debaa3ap-3},
    {0x1.3755bd1c945eep+0, -0x1.90c6db9fcbcdbp-3},
    {0x1.35dce5f9f2af8p+0, -0x1.871213750e994p-3},
    {0x1.34679ace01346p+0, -0x1.7d6903caf5acdp-3},
    {0x1.32f5ced6a1dfap+0, -0x1.73cb9074fd14dp-3},
    {0x1.3187758e9ebb6p+0, -0x1.6a399dabbd383p-3},
    {0x1.301c82ac40260p+0, -0x1.60b3100b09474p-3},
    {0x1.2eb4ea1fed14bp+0, -0x1.5737cc9018cddp-3},
    {0x1.2d50a012d50a0p+0, -0x1.4dc7b897bc1c7p-3},
    {0x1.2bef98e5a3711p+0, -0x1.4462b9dc9b3dcp-3},
    {0x1.2a91c92f3c105p+0, -0x1.3b08b6757f2a7p-3},
    {0x1.293725bb804a5p+0, -0x1.31b994d3a4f86p-3},
    {0x1.27dfa38a1ce4dp+0, -0x1.28753bc11aba2p-3},
    {0x1.268b37cd60127p+0, -0x1.1f3b925f25d44p-3},
    {0x1.2539d7e9177b2p+0, -0x1.160c8024b27b0p-3},
    {0x1.23eb79717605bp+0, -0x1.0ce7ecdccc28bp-3},
    {0x1.22a0122a0122ap+0, -0x1.03cdc0a51ec0dp-3},
    {0x1.21579804855e6p+0, -0x1.f57bc7d9005dbp-4},
    {0x1.2012012012012p+0, -0x1.e3707ee30487bp-4},
    {0x1.1ecf43c7fb84cp+0, -0x1.d179788219362p-4},
    {0x1.1d8f5672e4abdp+0, -0x1.bf968769fca18p-4},
    {0x1.1c522fc1ce059p+0, -0x1.adc77ee5aea8ep-4},
    {0x1.1b17c67f2bae3p+0, -0x1.9c0c32d4d254dp-4},
    {0x1.19e0119e0119ep+0, -0x1.8a6477a91dc29p-4},
    {0x1.18ab083902bdbp+0, -0x1.78d02263d82d7p-4},
    {0x1.1778a191bd684p+0, -0x1.674f089365a78p-4},
    {0x1.1648d50fc3201p+0, -0x1.55e10050e0382p-4},
    {0x1.151b9a3fdd5c9p+0, -0x1.4485e03dbdfb0p-4},
    {0x1.13f0e8d344724p+0, -0x1.333d7f8183f4ap-4},
    {0x1.12c8b89edc0acp+0, -0x1.2207b5c7854a1p-4},
    {0x1.11a3019a74826p+0, -0x1.10e45b3cae829p-4},
    {0x1.107fbbe011080p+0, -0x1.ffa6911ab9309p-5},
    {0x1.0f5edfab325a2p+0, -0x1.dda8adc67ee59p-5},
    {0x1.0e40655826011p+0, -0x1.bbcebfc68f424p-5},
    {0x1.0d24456359e3ap+0, -0x1.9a187b573de81p-5},
    {0x1.0c0a7868b4171p+0, -0x1.788595a3577c8p-5},
    {0x1.0af2f722eecb5p+0, -0x1.5715c4c03cee1p-5},
    {0x1.09ddba6af8360p+0, -0x1.35c8bfaa13069p-5},
    {0x1.08cabb37565e2p+0, -0x1.149e3e4005a8dp-5},
    {0x1.07b9f29b8eae2p+0, -0x1.e72bf2813ce6ap-6},
    {0x1.06ab59c7912fbp+0, -0x1.a55f548c5c427p-6},
    {0x1.059eea0727586p+0, -0x1.63d6178690bbep-6},
    {0x1.04949cc1664c5p+0, -0x1.228fb1fea2e0ap-6},
    {0x1.038c6b78247fcp+0, -0x1.c317384c75f0dp-7},
    {0x1.02864fc7729e9p+0, -0x1.41929f968330cp-7},
    {0x1.0182436517a37p+0, -0x1.8121214586b02p-8},
    {0x1.0000000000000p+0, 0x0.0p+0}   /* the bin below 1, [1 - 2^-8, 1): c = 1, so that log x = log1p(x - 1) keeps its RELATIVE accuracy for x -> 1 (log p of a success probability near 1 is multiplied by alpha ~ 1e6) */,
    {0x1.fe01fe01fe020p-1, 0x1.ff00aa2b10ba0p-9},
    {0x1.fa11caa01fa12p-1, 0x1.7dc475f810a69p-7},
    {0x1.f6310aca0dbb5p-1, 0x1.3cea44346a584p-6},
    {0x1.f25f644230ab5p-1, 0x1.b9fc027af919ap-6},
    {0x1.ee9c7f8458e02p-1, 0x1.1b0d98923d97fp-5},
    {0x1.eae807aba01ebp-1, 0x1.58a5bafc8e4d3p-5},
    {0x1.e741aa59750e4p-1, 0x1.95c830ec8e3f2p-5},
    {0x1.e3a9179dc1a73p-1, 0x1.d276b8adb0b56p-5},
    {0x1.e01e01e01e01ep-1, 0x1.075983598e471p-4},
    {0x1.dca01dca01dcap-1, 0x1.253f62f0a1417p-4},
    {0x1.d92f2231e7f8ap-1, 0x1.42edcbea646eep-4},
    {0x1.d5cac807572b2p-1, 0x1.60658a93750c4p-4},
    {0x1.d272ca3fc5b1ap-1, 0x1.7da766d7b12d0p-4},
    {0x1.cf26e5c44bfc6p-1, 0x1.9ab42462033aep-4},
    {0x1.cbe6d9601cbe7p-1, 0x1.b78c82bb0eda0p-4},
    {0x1.c8b265afb8a42p-1, 0x1.d4313d66cb35dp-4},
    {0x1.c5894d10d4986p-1, 0x1.f0a30c01162a4p-4},
    {0x1.c26b5392ea01cp-1, 0x1.0671512ca596fp-3},
    {0x1.bf583ee868d8bp-1, 0x1.14785846742acp-3},
    {0x1.bc4fd65883e7bp-1, 0x1.2266f190a5acdp-3},
    {0x1.b951e2b18ff23p-1, 0x1.303d718e47fd5p-3},
    {0x1.b65e2e3beee05p-1, 0x1.3dfc2b0ecc62ap-3},
    {0x1.b37484ad806cep-1, 0x1.4ba36f39a55e5p-3},
    {0x1.b094b31d922a4p-1, 0x1.59338d9982085p-3},
    {0x1.adbe87f94905ep-1, 0x1.66acd4272ad51p-3},
    {0x1.aaf1d2f87ebfdp-1, 0x1.740f8f54037a3p-3},
    {0x1.a82e65130e159p-1, 0x1.815c0a14357e9p-3},
    {0x1.a574107688a4ap-1, 0x1.8e928de886d41p-3},
    {0x1.a2c2a87c51ca0p-1, 0x1.9bb362e7dfb85p-3},
    {0x1.a01a01a01a01ap-1, 0x1.a8becfc882f19p-3},
    {0x1.9d79f176b682dp-1, 0x1.b5b519e8fb5a6p-3},
    {0x1.9ae24ea5510dap-1, 0x1.c2968558c18c2p-3},
    {0x1.9852f0d8ec0ffp-1, 0x1.cf6354e09c5ddp-3},
    {0x1.95cbb0be377aep-1, 0x1.dc1bca0abec7bp-3},
    {0x1.934c67f9b2ce6p-1, 0x1.e8c0252aa5a60p-3},
    {0x1.90d4f120190d5p-1, 0x1.f550a564b7b37p-3},
    {0x1.8e6527af1373fp-1, 0x1.00e6c45ad501dp-2},
    {0x1.8bfce8062ff3ap-1, 0x1.071b85fcd590dp-2},
    {0x1.899c0f601899cp-1, 0x1.0d46b579ab74bp-2},
    {0x1.87427bcc092b9p-1, 0x1.136870293a8b0p-2},
    {0x1.84f00c2780614p-1, 0x1.1980d2dd4236fp-2},
    {0x1.82a4a0182a4a0p-1, 0x1.1f8ff9e48a2f3p-2},
    {0x1.8060180601806p-1, 0x1.2596010df763ap-2},
    {0x1.7e225515a4f1dp-1, 0x1.2b9303ab89d25p-2},
    {0x1.7beb3922e017cp-1, 0x1.31871c9544185p-2},
    {0x1.79baa6bb6398bp-1, 0x1.3772662bfd85cp-2},
    {0x1.77908119ac60dp-1, 0x1.3d54fa5c1f710p-2},
    {0x1.756cac201756dp-1, 0x1.432ef2a04e813p-2},
};
__shared__ __attribute__((aligned(16))) double g_log_tab[128][2];

__device__ __forceinline__ double fast_log_normal_classic(double x);
__device__ __forceinline__ double fast_log_normal(double x)
{
#ifdef DIG_CLASSIC_LOG
    return fast_log_normal_classic(x);
#endif
    const unsigned long long bits = (unsigned long long)__double_as_longlong(x);
    const uint32_t hx = (uint32_t)(bits >> 32);
    const uint32_t tmp = hx - 0x3fe60000u;                     // high word of bits(x) - bits(0.6875) (low word of OFF is 0)
    const int k = (int)tmp >> 20;
    const uint32_t zh = hx - (tmp & 0xfff00000u);
    const double z = __longlong_as_double((long long)(((unsigned long long)zh << 32) | (bits & 0xffffffffull)));
    const double2 t = *reinterpret_cast<const double2*>(&g_log_tab[(tmp >> 13) & 127u][0]);   // {invc, logc}
    const double r = fma(z, t.x, -1.0);
    const double r2 = r * r;
    double P = -0.16666950550827794;
    P = fma_sconst(P, r, 0.20000270365390174);
    P = fma_sconst(P, r, -0.24999999998388212);
    P = fma_sconst(P, r, 0.33333333332309978);
    P = fma(P, r, -0.5);
    const double y = fma(r2, P, r);
    const double dk = (double)k;
    const double w = fma(dk, 6.93147180369123816490e-01, t.y);  // exact product (ln2_hi has 32 trailing zero bits)
    return w + fma(dk, 1.90821492927058770002e-10, y);
}

// Classical form (x = 2^k m, log m = 2 atanh(s) with the degree-14 polynomial in s^2; < 1 ulp, ~40 instructions).
__device__ __forceinline__ double fast_log_normal_classic(double x)
{
    const long long bits = __double_as_longlong(x);
    int hx = (int)(bits >> 32);
    int k = (hx >> 20) - 1023;
    hx &= 0x000fffff;
    const int i = (hx + 0x95f64) & 0x100000;                   // m >= sqrt(2) -> halve it, k += 1
    k += i >> 20;
    const long long mbits = ((long long)(hx | (i ^ 0x3ff00000)) << 32) | (bits & 0xffffffffll);
    const double f = __longlong_as_double(mbits) - 1.0;
    const double s = f * recip_nr(2.0 + f);
    const double z = s * s, w = z * z;
    double t1 = 1.531383769920937332e-01;
    t1 = fma_sconst(t1, w, 2.222219843214978396e-01);
    t1 = fma_sconst(t1, w, 3.999999999940941908e-01);
    t1 *= w;
    double t2 = 1.479819860511658591e-01;
    t2 = fma_sconst(t2, w, 1.818357216161805012e-01);
    t2 = fma_sconst(t2, w, 2.857142874366239149e-01);
    t2 = fma_sconst(t2, w, 6.666666666666735130e-01);
    t2 *= z;
    const double R = t1 + t2;
    const double hfsq = 0.5 * f * f;
    const double dk = (double)k;
    return dk * 6.93147180369123816490e-01 - ((hfsq - (s * (hfsq + R) + dk * 1.90821492927058770002e-10)) - f);
}


__device__ __forceinline__ double fast_log(double x)
{
    const int hx = (int)(__double_as_longlong(x) >> 32);
    if (hx < 0x00100000 || hx >= 0x7ff00000) return log(x);   // zero, negative, subnormal, inf, nan
    return fast_log_normal(x);
}

// log Gamma(z) for z > 0: arguments below 16 are shifted up by 16 (Gamma(z) = Gamma(z + 16) / (z (z + 1) ... (z + 15))),
// then Stirling's series with five correction terms (the first one left out, 691 / (360360 z^11), is 1e-16 at z = 16).
// Absolute error ~ 2e-16 max(1, z log z): the size of the rounding of its leading term.
__device__ __forceinline__ double lgamma_stirling(double z)
{
    const bool small = z < 16.0;
    double shift = 1.0, zz = z;
    if (small) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            shift *= zz;
            zz += 1.0;
        }
    }
    const double r = 1.0 / zz, r2 = r * r;
    const double corr = r * fma(r2, fma(r2, fma(r2, fma(r2, 1.0 / 1188.0, -1.0 / 1680.0), 1.0 / 1260.0), -1.0 / 360.0), 1.0 / 12.0);
    double v = (zz - 0.5) * fast_log_normal(zz) - zz + 0.91893853320467274178 + corr;
    if (small) v -= fast_log_normal(shift);
    return v;
}

// ---- pmf of the negative binomial without cancellation (Loader's saddle-point form, as R's dnbinom) ------------------
// scipy's nbinom._logpmf -- lgamma(k + n) - lgamma(k + 1) - lgamma(n) + n log p + k log1p(-p) -- subtracts numbers of the
// size of k log k: at k ~ 10^6 its double-precision value is off by 2e-6 (measured against 50-digit arithmetic), whatever
// the quality of lgamma.  Here  pmf(k) = n / (n + k) * dbinom_raw(n, n + k, p, 1 - p)  with
//   dbinom_raw(x, N, p, q) = exp(stirlerr(N) - stirlerr(x) - stirlerr(N - x) - bd0(x, N p) - bd0(N - x, N q)) sqrt(N / (2 pi x (N - x)))
//   stirlerr(z) = lgamma(z + 1) - ((z + 1/2) log z - z + log sqrt(2 pi))        (the Stirling series from z = 16 on)
//   bd0(x, M)   = x log(x / M) + M - x                                           (a series in (x - M) / (x + M) near x = M)
// every term is small when the pmf is not: 3e-10 worst case over n in [1e-3, 1e7], means in [1e-3, 3e6], counts from five
// standard deviations below the mean to twenty above (the lgamma form: 1.8e-6).  k >= 0 integer, n > 0 finite, 0 < p < 1.
__device__ __forceinline__ double stirlerr_dev(double z)
{
    if (z >= 16.0) {
        const double r = 1.0 / z, r2 = r * r;
        return r * fma(-r2, fma(-r2, fma(-r2, fma(-r2, 1.0 / 1188.0, 1.0 / 1680.0), 1.0 / 1260.0), 1.0 / 360.0), 1.0 / 12.0);
    }
    return lgamma_stirling(z + 1.0) - (z + 0.5) * fast_log_normal(z) + z - 0.91893853320467274178;
}

__device__ inline double bd0_dev(double x, double M)
{
    const double d = x - M;
    if (fabs(d) < 0.1 * (x + M)) {
        const double v = d / (x + M), v2 = v * v;
        double s = d * v, ej = 2.0 * x * v;
        for (int j = 1; j < 1000; ++j) {
            ej *= v2;
            const double s1 = s + ej / (double)(2 * j + 1);
            if (s1 == s) return s1;
            s = s1;
        }
        return s;
    }
    return x * fast_log_normal(x / M) + M - x;
}

__device__ inline double nb_pmf_saddle(double k, double n, double p)
{
    if (k == 0.0) return exp(n * fast_log_normal(p));
    const double q = 1.0 - p, N = k + n;
    const double lc = stirlerr_dev(N) - stirlerr_dev(n) - stirlerr_dev(k) - bd0_dev(n, N * p) - bd0_dev(k, N * q);
    const double lf = 1.83787706640934548356 + fast_log_normal(n) + log1p(-n / N);          // log(2 pi n (N - n) / N)
    return (n / N) * exp(lc - 0.5 * lf);
}

// ---- continued fraction for I_x(a,b) (fast for x < (a+1)/(a+b+2)) -------------------
__device__ inline double betacf(double a, double b, double x)
{
    const double qab = a + b, qap = a + 1.0, qam = a - 1.0;
    double c = 1.0, d = 1.0 - qab * x / qap;
    if (fabs(d) < kFpMin) d = kFpMin;
    d = 1.0 / d;
    double h = d;
    for (int m = 1; m <= kCfMaxIt; ++m) {
        const double dm = (double)m, m2 = 2.0 * dm;
        double aa = dm * (b - dm) * x / ((qam + m2) * (a + m2));
        d = 1.0 + aa * d;
        if (fabs(d) < kFpMin) d = kFpMin;
        c = 1.0 + aa / c;
        if (fabs(c) < kFpMin) c = kFpMin;
        d = 1.0 / d;
        h *= d * c;
        aa = -(a + dm) * (qab + dm) * x / ((a + m2) * (qap + m2));
        d = 1.0 + aa * d;
        if (fabs(d) < kFpMin) d = kFpMin;
        c = 1.0 + aa / c;
        if (fabs(c) < kFpMin) c = kFpMin;
        d = 1.0 / d;
        const double del = d * c;
        h *= del;
        if (fabs(del - 1.0) <= kCfEps) break;
    }
    return h;
}

// The same continued fraction evaluated with the forward (A_n, B_n) recurrence, renormalised every double step:
// two reciprocals per double step instead of the six divisions of the Lentz form.  Falls back to betacf() if the
// recurrence degenerates (A_n ~ 0).
__device__ inline double betacf_fast(double a, double b, double x)
{
    const double qab = a + b;
    const double d1 = -qab * x * recip_nr(a + 1.0);
    double Ap = 1.0, Bp = 1.0;            // convergent n-1 after normalisation: (A_{n-1}, B_{n-1})
    double A = 1.0 + d1, B = 1.0;         // convergent n
    {
        const double s0 = recip_nr(A);
        Ap *= s0; Bp *= s0; B *= s0; A = 1.0;
    }
    double hold = B;
    for (int m = 1; m <= kCfMaxIt; ++m) {
        const double dm = (double)m, a2m = a + 2.0 * dm;
        const double rT = recip_nr((a2m - 1.0) * a2m * (a2m + 1.0));
        const double de = dm * (b - dm) * x * (rT * (a2m + 1.0));
        const double dd = -(a + dm) * (qab + dm) * x * (rT * (a2m - 1.0));
        const double A1 = fma(de, Ap, A), B1 = fma(de, Bp, B);
        const double A2 = fma(dd, A, A1), B2 = fma(dd, B, B1);
        const double sc = recip_nr(A2);
        if (!(fabs(sc) < 1e300)) return betacf(a, b, x);
        Ap = A1 * sc;
        Bp = B1 * sc;
        A = 1.0;
        B = B2 * sc;
        if (fabs(B - hold) <= kCfEps * fabs(B)) break;
        hold = B;
    }
    return B;
}

// scipy.special.betainc(a, b, x), general real a, b
__device__ inline double betainc(double a, double b, double x)
{
    if (isnan(a) || isnan(b) || isnan(x)) return dnan();
    if (a <= 0.0 || b <= 0.0 || x < 0.0 || x > 1.0) return dnan();  // scipy 1.15.3: a==0 or b==0 -> nan
    if (isinf(a) || isinf(b)) return dnan();
    if (x == 0.0) return 0.0;
    if (x == 1.0) return 1.0;
    const double y = 1.0 - x;
    // x^a y^b / B(a, b) = a b / (a + b) * dbinom_raw(a, a + b, x, y) in Loader's form (see nb_pmf_saddle): the textbook
    // a log x + b log y + lgamma(a + b) - lgamma(a) - lgamma(b) cancels numbers of the size of a log a (2e-6 off at a ~ 1e6)
    double front;
    if (x >= 2.2250738585072014e-308 && y >= 2.2250738585072014e-308 && a >= 1e-300 && b >= 1e-300 && a < 1e300 && b < 1e300) {
        const double nn = a + b;
        const double lc = stirlerr_dev(nn) - stirlerr_dev(a) - stirlerr_dev(b) - bd0_dev(a, nn * x) - bd0_dev(b, nn * y);
        const double lf = 1.83787706640934548356 + fast_log_normal(a) + log1p(-a / nn);
        front = (a * (b / nn)) * exp(lc - 0.5 * lf);
    } else {
        front = exp(a * log(x) + b * log(y) + lgamma(a + b) - lgamma(a) - lgamma(b));
    }
    if (x < (a + 1.0) / (a + b + 2.0)) return front * betacf(a, b, x) / a;
    return 1.0 - front * betacf(b, a, y) / b;
}

// scipy.stats.nbinom.pmf(k, n, p)
__device__ inline double nbinom_pmf(double k, double n, double p)
{
    if (isnan(k) || isnan(n) || isnan(p)) return dnan();
    if (!(n > 0.0) || !(p > 0.0) || !(p <= 1.0) || isinf(n)) return dnan();
    if (k < 0.0 || floor(k) != k) return 0.0;
    if (p == 1.0) return k == 0.0 ? 1.0 : 0.0;
    // (scipy's own expression here is lgamma(k + n) - lgamma(k + 1) - lgamma(n) + n log p + k log1p(-p): it cancels k log k,
    //  2e-6 off at k ~ 1e6; scipy 1.15 evaluates the pmf through boost instead, and so does this -- Loader's form)
    if (p >= 2.2250738585072014e-308 && k < 4.0e15) return nb_pmf_saddle(k, n, p);
    const double l = lgamma(k + n) - lgamma(k + 1.0) - lgamma(n) + n * log(p) + k * log1p(-p);
    return exp(l);
}

// log pmf(k) for integer k >= 0, 0 < p < 1, finite n > 0 (no argument checks)
__device__ __forceinline__ double nbinom_logpmf_unchecked(double k, double n, double p)
{
    return lgamma(k + n) - lgamma(k + 1.0) - lgamma(n) + n * log(p) + k * log1p(-p);
}

// P(X > k) for integer k >= 0 given pmf(k):  I_x(k+1, alpha) with prefactor folded into pmf(k).
__device__ inline double nb_upper_tail_from_pmf(double k, double alpha, double p, double x, double pmfk)
{
    const double a = k + 1.0, b = alpha;
    if (x < (a + 1.0) / (a + b + 2.0)) {
        // I = pmf(k+1) * cf(a,b,x),  pmf(k+1) = pmf(k) (k+alpha) x / (k+1)
        return pmfk * ((k + alpha) * x / a) * betacf_fast(a, b, x);
    }
    // 1 - I_y(b, a),  prefactor y^b x^a /(b B(a,b)) = pmf(k) (k+alpha) x / alpha
    return 1.0 - pmfk * ((k + alpha) * x / alpha) * betacf_fast(b, a, p);
}

// The recurrences below evaluate  1 - S_k - (W2 / 2) t_k  (S_k = sum_{j<k} t_j):  W2 = 1 is the mid-p statistic
// 0.5 pmf(k) + P(X > k);  W2 = 0 is P(X >= k) = betainc(k, alpha, 1 - p), the upper tail of nb_pvalue_exact and
// nb_pvalue_greater.
template <int W2 = 1>
__device__ __forceinline__ unsigned nb_midp_upper_fast2(double k1, double k2, unsigned want, double alpha, double p,
                                                        double& r1, double& r2);

// The expensive tail of nb_midp_upper: integer k >= 0, 0 < p < 1, finite alpha > 0, and either
// k > kSmallK, p^alpha underflows, or the p-value is < kDirectMin (1 - CDF would cancel).
constexpr int kRecurK = 2048;   // direct summation limit of the slow pass

__device__ __forceinline__ void pmf_scaled_step(double& A, double& N, double& D, double& u, double& jj, double x, double ax);
template <int W2 = 1>
__device__ __forceinline__ double midp_from_state(double A, double N, double D, double k, double t0);

// Slow-pass evaluation for integer k >= 0, 0 < p < 1, finite alpha > 0.
//  (1) Most items get here only because k > kSmallK while sitting near their mean: the same scaled recurrence as
//      the fast pass (5 FP64 ops per step, no division), with (A, N, D) rescaled by the exact power of two
//      2^-exponent(D) every 16 steps so that D = j! never overflows, gives S_k and t_k in k steps; 1 - S_k - t_k/2
//      is accepted under the same >= kDirectMin rule (absolute error grows like k ulp: 2048 * 1.1e-16 / 1e-4 = 2e-9
//      worst case).  Range: after rescaling N/D = t_j/t_0 <= 1/t_0 and one block of 16 steps multiplies by at most
//      (2 * 2048)^16 = 1e58, so p^alpha >= e^-500 keeps everything finite.
//  (2) A small tail (1 - S_k cancels) is summed directly instead: the recurrence simply continues past k with a
//      fresh accumulator, B_{J+1} = B_J (J+1) + N_{J+1}, so that sum_{k<j<=J} t_j = t_0 B_J / D_J -- all terms
//      positive, no cancellation, relative error ~ (number of terms) ulp -- until the last term is below 2^-56 of
//      the sum (the terms fall geometrically this far above the mean; the test runs every 8 steps).
//  (3) Only a smaller p^alpha, k > kRecurK or a series that has not converged after kTailMax terms goes on to
//      lgamma + the continued fraction.
constexpr int kTailMax = 4096;

// Two counts sharing (alpha, p) in ONE pass of the recurrence (the SNV and SAMPLE tests of a pair): the state is
// evaluated at the smaller count on the way to the larger one.  `want` selects the requested counts (bit 0: k1,
// bit 1: k2); both must be integers >= 0.  All rescalings are exact powers of two, so the result for a count does
// not depend on whether it was computed alone or together with another one.
template <int W2 = 1>
__device__ inline void nb_midp_upper_slow2(double k1, double k2, unsigned want, double alpha, double p, double& r1,
                                           double& r2)
{
    const double x = 1.0 - p;
    const double lp0 = alpha * fast_log(p);
    unsigned todo = want;
    if (lp0 > -500.0) {
        const double t0 = fast_exp_neg(lp0);
        const bool el1 = (want & 1u) && k1 <= (double)kRecurK, el2 = (want & 2u) && k2 <= (double)kRecurK;
        // targets in ascending order; a single eligible count is visited once
        const double klo = (el1 && el2) ? fmin(k1, k2) : (el1 ? k1 : k2);
        const double khi = (el1 && el2) ? fmax(k1, k2) : klo;
        const int n_phase = (el1 || el2) ? ((el1 && el2 && k1 != k2) ? 2 : 1) : 0;
        const double ax = alpha * x;
        double N = 1.0, A = 0.0, D = 1.0, u = ax, jj = 0.0;
        for (int phase = 0; phase < n_phase; ++phase) {
            const double k = phase == 0 ? klo : khi;
            while (jj < k) {
                const double stop = fmin(k, jj + 16.0);
                while (jj < stop) pmf_scaled_step(A, N, D, u, jj, x, ax);
                const int e = -__builtin_amdgcn_frexp_exp(D);
                D = ldexp(D, e);
                N = ldexp(N, e);
                A = ldexp(A, e);
            }
            double res = midp_from_state<W2>(A, N, D, jj, t0);
            bool ok = res >= (k <= 256.0 ? kDirectMin : 1e-4);     // long sums: keep a wider safety margin
            if (!ok) {
                // (2) tail series from a copy of the state at k:  (1 - W2/2) t_k + sum_{j>k} t_j
                double Nt = N, Dt = D, ut = u, jt = jj, B = 0.0, H = (1.0 - 0.5 * W2) * N;
                const double jend = k + (double)kTailMax;
                bool converged = false;
                while (jt < jend) {
#pragma unroll
                    for (int s = 0; s < 8; ++s) {
                        Nt *= ut;                // N_{j+1}
                        jt += 1.0;
                        ut = fma(jt, x, ax);     // (not ut += x: see pmf_scaled_step)
                        Dt *= jt;                // D_{j+1}
                        B = fma(B, jt, Nt);      // B_{j+1} = B_j (j+1) + N_{j+1}
                        H *= jt;                 // keeps 0.5 t_k on the same scale
                    }
                    if (Nt <= B * 0x1p-56) { converged = true; break; }
                    const int e = -__builtin_amdgcn_frexp_exp(Dt);
                    Dt = ldexp(Dt, e);
                    Nt = ldexp(Nt, e);
                    B = ldexp(B, e);
                    H = ldexp(H, e);
                }
                if (converged) {
                    const double v = (B + H) * (t0 * recip_nr(Dt));
                    if (v > 1e-290) { res = v; ok = true; }        // below that the scaled terms may have underflowed
                }
            }
            if (ok) {
                if (el1 && k1 == k && (todo & 1u)) { r1 = res; todo &= ~1u; }
                if (el2 && k2 == k && (todo & 2u)) { r2 = res; todo &= ~2u; }
            }
        }
    }
    // (3) lgamma + continued fraction for whatever is left
    if (todo & 1u) {
        const double pmfk = nb_pmf_saddle(k1, alpha, p);
        r1 = (1.0 - 0.5 * W2) * pmfk + nb_upper_tail_from_pmf(k1, alpha, p, x, pmfk);
    }
    if (todo & 2u) {
        const double pmfk = nb_pmf_saddle(k2, alpha, p);
        r2 = (1.0 - 0.5 * W2) * pmfk + nb_upper_tail_from_pmf(k2, alpha, p, x, pmfk);
    }
}

__device__ inline double nb_midp_upper_slow(double k, double alpha, double p)
{
    double r = 0.0, dummy = 0.0;
    nb_midp_upper_slow2(k, 0.0, 1u, alpha, p, r, dummy);
    return r;
}

// For a test nb_midp_upper_fast2 left unresolved (its argument checks already passed: finite alpha > 0,
// 0 < p < 1, k not NaN): counts outside the support keep scipy's semantics, the rest take the slow tail.
__device__ inline double nb_midp_upper_unresolved(double k, double alpha, double p)
{
    if (k < 0.0 || floor(k) != k || isinf(k)) return betainc(k + 1.0, alpha, 1.0 - p);   // pmf = 0 off the support
    return nb_midp_upper_slow(k, alpha, p);
}

// nb_model.py:271-278:  0.5 * nbinom.pmf(k, alpha, p) + betainc(k + 1, alpha, 1 - p)
__device__ inline double nb_midp_upper(double k, double alpha, double p)
{
    if (isnan(k) || isnan(alpha) || isnan(p)) return dnan();
    if (!(alpha > 0.0) || !(p > 0.0) || !(p <= 1.0) || isinf(alpha)) return dnan();
    if (k < 0.0 || floor(k) != k || isinf(k)) {
        // never produced by the live callers (counts); keep scipy's semantics
        const double pmf = 0.0;   // negative or non-integer k is outside the support
        return 0.5 * pmf + betainc(k + 1.0, alpha, 1.0 - p);
    }
    double r = 0.0, dummy = 0.0;
    if (nb_midp_upper_fast2(k, 0.0, 1u, alpha, p, r, dummy) & 1u) return r;   // same bits as the fused kernels
    return nb_midp_upper_slow(k, alpha, p);
}

// ---- mid-p statistic by the FOUR lanes of a quad (the compacted pass of the statistics block) ----------------------
// The serial slow path above walks the recurrence from 0 to k and then on into the tail: 500 counts = 4 500 dependent
// FP64 instructions in ONE lane, and the compacted pass lasted as long as its longest lane.  Here a test is anchored at
// k and only the side of the distribution that is short is summed, as a series RELATIVE to pmf(k), in blocks of 64
// terms split over the four lanes of a quad (sixteen terms per lane):
//   r_k = (alpha + k) x / (k + 1) < 1  (k at or above the mode):  0.5 pmf(k) + P(X > k) = pmf(k) (0.5 + sum_{m>=1} prod_{i<m} r_{k+i})
//   otherwise                                                  :  1 - pmf(k) (0.5 + sum_{m>=1} prod_{i<=m} 1 / r_{k-i})   (ends at j = 0)
// Both series fall monotonically (alpha > 1: the ratios move away from 1; alpha <= 1: r < x < 1 always and only the
// first form occurs).  A lane runs the scaled recurrence over its sixteen ratios (products of numerators N and
// denominators D apart, B = running sum times D: no division inside), the lanes' products are combined by an
// inclusive scan and the sums by a butterfly, both as quad-permute DPP moves (no LDS traffic); a block is the last one
// when the geometric bound of the remainder, last term x rho / (1 - rho), is below 2^-54 of the sum.
// pmf(k): the streaming pass hands it over when it has it (a count <= kSmallK whose direct form cancelled: the usual
// case); otherwise p^alpha times the product of the ratios below k (counts up to 64: one block), or, for counts and
// dispersions up to 4096, the lgamma form with the three Stirling terms on three lanes of the quad (a count of 500 would
// take eight product blocks, and the 37 pairs of a large element reach the pass together: its waves set the length of
// the kernel), or Loader's saddle-point form (nb_pmf_saddle) beyond, where the lgamma differences cancel.  Anything unusual (arguments outside the support, a series still open after
// kQuadBlocks blocks: heavy tails with alpha << 1) goes to the scalar nb_midp_upper, which keeps scipy's semantics.
// Every lane of the quad must call with the same arguments; the result is the same in all four.
constexpr int kQuadTerms = 16;         // terms per lane and block
constexpr int kQuadBlocks = 64;        // 4 096 terms

__device__ inline double nb_midp_upper(double k, double alpha, double p);

template <int CTRL>
__device__ __forceinline__ double quad_perm(double v)
{
    const long long b = __double_as_longlong(v);
    const int lo = __builtin_amdgcn_mov_dpp((int)b, CTRL, 0xf, 0xf, true);
    const int hi = __builtin_amdgcn_mov_dpp((int)(b >> 32), CTRL, 0xf, 0xf, true);
    return __longlong_as_double(((long long)hi << 32) | (unsigned)lo);
}
// quad_perm selectors: lane i reads lane sel[i]
constexpr int kQuadUp1 = 0 | (0 << 2) | (1 << 4) | (2 << 6);     // i - 1 (lane 0 itself)
constexpr int kQuadUp2 = 0 | (0 << 2) | (0 << 4) | (1 << 6);     // i - 2
constexpr int kQuadLast = 3 | (3 << 2) | (3 << 4) | (3 << 6);
constexpr int kQuadXor1 = 1 | (0 << 2) | (3 << 4) | (2 << 6);
constexpr int kQuadXor2 = 2 | (3 << 2) | (0 << 4) | (1 << 6);
constexpr int kQuadLane0 = 0, kQuadLane1 = 1 | (1 << 2) | (1 << 4) | (1 << 6), kQuadLane2 = 2 | (2 << 2) | (2 << 4) | (2 << 6);

// inclusive scan of a product over the four lanes of a quad; every lane returns the total, `excl` the product of the lanes below
__device__ __forceinline__ double quad_scan_product(double P, int sub, double& excl)
{
    double v = quad_perm<kQuadUp1>(P);
    if (sub >= 1) P *= v;
    v = quad_perm<kQuadUp2>(P);
    if (sub >= 2) P *= v;
    excl = quad_perm<kQuadUp1>(P);
    if (sub == 0) excl = 1.0;
    return quad_perm<kQuadLast>(P);
}

// pmf_k < 0: not known.
__device__ __forceinline__ double nb_midp_upper_quad(double k, double alpha, double p, double pmf_k, int sub)
{
    const double inf = __longlong_as_double(0x7ff0000000000000LL);
    if (!(alpha > 0.0 && alpha < inf && p >= 2.2250738585072014e-308 && p < 1.0 && k >= 0.0 && k < 4.0e15 && floor(k) == k))
        return nb_midp_upper(k, alpha, p);
    const double x = 1.0 - p;
    const double ax = alpha * x;
    double tk = pmf_k;
    if (!(tk >= 0.0)) {
        const double lp = fast_log_normal(p);
        const double lp0 = alpha * lp;
        if (k <= 64.0 && lp0 > -690.0) {
            // pmf(k) = p^alpha prod_{j<k} (alpha + j) x / (j + 1): one block; at most 1 / p^alpha < e^690 on the way
            double j = (double)(sub * kQuadTerms);
            double N = 1.0, D = 1.0;
#pragma unroll
            for (int s = 0; s < kQuadTerms; ++s) {
                const bool in = j < k;
                const double num = in ? fma(j, x, ax) : 1.0;
                j += 1.0;
                N *= num;
                D *= in ? j : 1.0;
            }
            double excl;
            tk = fast_exp_neg(lp0) * quad_scan_product(N / D, sub, excl);
        } else if (k <= 4096.0 && alpha <= 4096.0) {
            // log pmf(k) = lgamma(k + alpha) - lgamma(k + 1) - lgamma(alpha) + alpha log p + k log(1 - p)  (scipy nbinom._logpmf), the
            // three lgamma terms from Stirling's series on three lanes of the quad; at these sizes its terms are below 4e4
            // and the cancellation costs at most 1e-11 (the usual case of a large element: a count of a few hundred)
            const double lg = lgamma_stirling(sub == 0 ? k + alpha : sub == 1 ? k + 1.0 : alpha);
            const double coeff = quad_perm<kQuadLane0>(lg) - quad_perm<kQuadLane1>(lg) - quad_perm<kQuadLane2>(lg);
            const double lx = x >= 0.5 ? fast_log_normal(x) : log1p(-p);
            tk = exp(coeff + lp0 + k * lx);
        } else {
            tk = nb_pmf_saddle(k, alpha, p);        // no cancellation at any size (the lgamma form loses 2e-6 at k ~ 1e6)
        }
    }
    const bool upper = (alpha + k) * x < k + 1.0;
    double base = 1.0, V = 0.0;
    bool converged = false;
    for (int blk = 0; blk < kQuadBlocks && !converged; ++blk) {
        const double m0 = (double)(blk * 4 * kQuadTerms + sub * kQuadTerms);      // ratios m0 .. m0 + 15 of the series
        double N = 1.0, D = 1.0, B = 0.0;
        if (upper) {
            double j = k + m0;
#pragma unroll
            for (int s = 0; s < kQuadTerms; ++s) {
                const double num = fma(j, x, ax);          // (alpha + j) x
                j += 1.0;                                   // den = j + 1
                N *= num;
                D *= j;
                B = fma(B, j, N);
            }
        } else {
            double j = k - m0;                              // ratio j / ((alpha + j - 1) x), none below j = 1
#pragma unroll
            for (int s = 0; s < kQuadTerms; ++s) {
                const bool in = j >= 1.0;
                const double num = in ? j : 0.0;
                j -= 1.0;
                const double den = in ? fma(j, x, ax) : 1.0;
                N *= num;
                D *= den;
                B = fma(B, den, N);
            }
        }
        const double rD = 1.0 / D;
        double E;
        const double tot = quad_scan_product(N * rD, sub, E);       // product of the block's ratios; E: of the lanes below
        double c = E * (B * rD);                                    // the lane's terms relative to the block's first factor
        c += quad_perm<kQuadXor1>(c);
        c += quad_perm<kQuadXor2>(c);
        V = fma(base, c, V);
        base *= tot;                                                // = the last term of the block
        // the ratio num / den that follows the block bounds all later ones:  base rho / (1 - rho) <= 2^-54 (0.5 + V)
        const double mn = (double)((blk + 1) * 4 * kQuadTerms);
        double num, den;
        if (upper) {
            const double j = k + mn;
            num = alpha > 1.0 ? fma(j, x, ax) : x;
            den = alpha > 1.0 ? j + 1.0 : 1.0;
        } else {
            const double j = k - mn;                         // next ratio j / ((alpha + j - 1) x)
            num = j >= 1.0 ? j : 0.0;
            den = j >= 1.0 ? fma(j - 1.0, x, ax) : 1.0;
        }
        converged = num < den && base * num <= (0.5 + V) * 0x1p-54 * (den - num);
    }
    if (!converged) return nb_midp_upper(k, alpha, p);
    const double r = tk * (0.5 + V);
    return upper ? r : 1.0 - r;
}

// ---- fast mid-p evaluation for small integer counts sharing (alpha, p) ----------------
// 1 - S_k - t_k / 2 from the scaled state (A_k, N_k, D_k = k!, k):  t_k = t_0 N_k / D_k,  S_k = t_0 A_k k / D_k.
// Single definition: both counts of a pair and every entry point go through these exact operations.
template <int W2>
__device__ __forceinline__ double midp_from_state(double A, double N, double D, double k, double t0)
{
#pragma clang fp contract(off)
    const double rD = t0 * recip_nr(D);
    const double S = (A * k) * rD;
    if (W2 == 0) return 1.0 - S;
    const double t = N * rD;
    const double res = (1.0 - S) - 0.5 * t;
    return res >= kDirectMin ? res : -t;          // not accepted: hand pmf(k) on (sign bit set) for the series of the compacted pass
}

// 1 / k! for k = 0 .. kSmallK, correctly rounded (generated from exact rationals).  The fast recurrence reads it from
// LDS (one ds_read per evaluated count; LDS waits are independent of the in-order vector-memory counter), which
// removes the running factorial from the loop and the reciprocal from the evaluation.  Every kernel that reaches
// nb_midp_upper_fast2 calls nb_tables_init() first.
static __device__ __constant__ const double kInvFactorialRom[kSmallK + 1] = {
    0x1.0000000000000p+0, 0x1.0000000000000p+0, 0x1.0000000000000p-1, 0x1.5555555555555p-3,
    0x1.5555555555555p-5, 0x1.1111111111111p-7, 0x1.6c16c16c16c17p-10, 0x1.a01a01a01a01ap-13,
    0x1.a01a01a01a01ap-16, 0x1.71de3a556c734p-19, 0x1.27e4fb7789f5cp-22, 0x1.ae64567f544e4p-26,
    0x1.1eed8eff8d898p-29, 0x1.6124613a86d09p-33, 0x1.93974a8c07c9dp-37, 0x1.ae7f3e733b81fp-41,
    0x1.ae7f3e733b81fp-45, 0x1.952c77030ad4ap-49, 0x1.6827863b97d97p-53, 0x1.2f49b46814157p-57,
    0x1.e542ba4020225p-62, 0x1.71b8ef6dcf572p-66, 0x1.0ce396db7f853p-70, 0x1.761b41316381ap-75,
    0x1.f2cf01972f578p-80, 0x1.3f3ccdd165fa9p-84, 0x1.88e85fc6a4e5ap-89, 0x1.d1ab1c2dccea3p-94,
    0x1.0a18a2635085dp-98, 0x1.259f98b4358adp-103, 0x1.3932c5047d60ep-108, 0x1.434d2e783f5bcp-113,
    0x1.434d2e783f5bcp-118, 0x1.3981254dd0d52p-123, 0x1.2710231c0fd7ap-128, 0x1.0dc59c716d91fp-133,
    0x1.df983290c2ca9p-139, 0x1.9ec8d1c94e85bp-144, 0x1.5d4acb9c0c3abp-149, 0x1.1e99449a4bacep-154,
    0x1.ca8ed42a12ae3p-160, 0x1.65e61c39d0241p-165, 0x1.10af527530de8p-170, 0x1.95db45257e512p-176,
    0x1.272b1b03fec6ap-181, 0x1.a3cb872220648p-187, 0x1.240804f659510p-192, 0x1.8da8e0a127ebap-198,
    0x1.091b406b6ff26p-203, 0x1.5a42f0dfeb086p-209, 0x1.bb36f6e12cd78p-215, 0x1.161872bf7b823p-220,
    0x1.56457989358c9p-226, 0x1.9d4f1058674dfp-232, 0x1.e9d8f6ed83eaap-238, 0x1.1d008faac5c50p-243,
    0x1.45b77f9e98e12p-249, 0x1.6db793c887b97p-255, 0x1.938cc661b03f6p-261, 0x1.b5bfc17fa97d3p-267,
    0x1.d2eeac43e7fcfp-273, 0x1.e9e56d649f768p-279, 0x1.f9b3059128bc7p-285, 0x1.00dcf6a320e1cp-290,
    0x1.00dcf6a320e1cp-296, 0x1.f9d2a2bb5471bp-303, 0x1.ea7ead50ce01ap-309, 0x1.d48849da8f4a3p-315,
    0x1.b8f8bdfae136cp-321, 0x1.99046602abcaep-327, 0x1.75f56494ba532p-333, 0x1.5116e3adb9fb9p-339,
    0x1.2ba2917dfaa6cp-345, 0x1.06b1981a48762p-351, 0x1.c6639f500ea2dp-358, 0x1.83bed30a49edfp-364,
    0x1.4685bf3115d5dp-370, 0x1.0f653132c5ae6p-376, 0x1.bd5dda94f5a18p-383, 0x1.68cda75b82f10p-389,
    0x1.20a485e2cf273p-395, 0x1.c8206e6fe560bp-402, 0x1.64005631debbep-408, 0x1.1281cd42368abp-414,
    0x1.a24be3711628bp-421, 0x1.3af3de7343e26p-427, 0x1.d4c44522a0927p-434, 0x1.58d700d5cb749p-440,
    0x1.f595d2ab567b0p-447, 0x1.68b0c583d6a34p-453, 0x1.007db446ff080p-459, 0x1.68c751f8f632ap-466,
    0x1.f5f3ec7bc5d72p-473, 0x1.596e0e189e2b7p-479, 0x1.d65f64e59b771p-486, 0x1.3ce1f3216b6dep-492,
    0x1.a6829981e4928p-499, 0x1.16c503a23d142p-505, 0x1.6c1b7275dcd65p-512, 0x1.d6c3cf76c59bap-519,
    0x1.2d4a1e607e781p-525, 0x1.7dd50faf84657p-532, 0x1.df297d187dfcdp-539, 0x1.29bb552f8772dp-545,
    0x1.6e7068d8092aep-552, 0x1.beb4eb15fc8c1p-559, 0x1.0db5afbffdea5p-565, 0x1.42a4b5885d350p-572,
    0x1.7e64655f3f0f6p-579, 0x1.c10c3547ec1b7p-586, 0x1.05439cac2c47dp-592, 0x1.2d470c4e9d270p-599,
    0x1.585132a2fcbeep-606, 0x1.8605e345153bep-613, 0x1.b5eba9d8cb7dap-620, 0x1.e76cb424808bdp-627,
    0x1.0cec86b309210p-633, 0x1.263516a53c4fep-640, 0x1.3f23e47f2bba7p-647, 0x1.5746e043f2ccep-654,
    0x1.6e2977bff1eb9p-661, 0x1.83584be68daafp-668, 0x1.9665084a05f24p-675, 0x1.a6ea2e16eb219p-682,
    0x1.b48ea3306e964p-689, 0x1.bf08d841fa750p-696, 0x1.c6215db8ddeccp-703, 0x1.c9b4c7476cc64p-710,
    0x1.c9b4c7476cc64p-717,
};
__shared__ double g_inv_factorial[kSmallK + 1];

__device__ __forceinline__ void nb_tables_init()
{
    const unsigned t = threadIdx.x + blockDim.x * (threadIdx.y + blockDim.y * threadIdx.z);
    if (t <= (unsigned)kSmallK) g_inv_factorial[t] = kInvFactorialRom[t];
    for (unsigned i = t; i < 256u; i += blockDim.x * blockDim.y * blockDim.z) (&g_log_tab[0][0])[i] = (&kLogTabRom[0][0])[i];
    __syncthreads();
}

// 1 - S_k - (W2/2) t_k from the scaled state (A_k, N_k) with 1/k! from the table:  t_k = t_0 N_k / k!,
// S_k = t_0 A_k k / k!.
template <int W2>
__device__ __forceinline__ double tail_from_state_tab(double A, double N, double k, double t0, double thr = kDirectMin)
{
#pragma clang fp contract(off)
    const double rD = t0 * g_inv_factorial[(int)k];
    const double S = (A * k) * rD;
    if (W2 == 0) return 1.0 - S;
    const double t = N * rD;
    const double res = (1.0 - S) - 0.5 * t;
    return res >= thr ? res : -t;          // not accepted: hand pmf(k) on (sign bit set) for the series of the compacted pass
}

// One step of the scaled recurrence with the running factorial (compacted pass, where D is rescaled on the way).  The
// accumulator update A <- A * j + N is issued as the three-address v_fma_f64 (the compiler's v_fmac form needs three
// register copies per step to keep N alive).
__device__ __forceinline__ void pmf_scaled_step(double& A, double& N, double& D, double& u, double& jj, double x, double ax)
{
    double An;
    asm("v_fma_f64 %0, %1, %2, %3" : "=v"(An) : "v"(A), "v"(jj), "v"(N));
    A = An;                 // A_{j+1} = A_j * j + N_j
    N *= u;                 // N_{j+1}
    jj += 1.0;
    u = fma(jj, x, ax);     // (alpha + j) x afresh, one rounding: an incremental u += x lets its errors pile up, k^2 / 2 ulp in N_k
    D *= jj;                // D_{j+1} = (j+1)!
}

// One step of the scaled recurrence without the factorial (fast pass): 4 FP64 operations.
__device__ __forceinline__ void pmf_scaled_step_nofact(double& A, double& N, double& u, double& jj, double x)
{
    double An;
    asm("v_fma_f64 %0, %1, %2, %3" : "=v"(An) : "v"(A), "v"(jj), "v"(N));
    A = An;                 // A_{j+1} = A_j * j + N_j
    N *= u;                 // N_{j+1}
    u += x;                 // (incremental: its rounding errors add up to ~k^1.5 / 3 ulp in N_k, 5e-14 at k = 128 -- 5e-8 of a
    jj += 1.0;              //  p-value at the 1e-6 acceptance edge; forming (alpha + j) x afresh costs 2 % of the pass)
}

// Range of the scaled recurrence: N_k <= k! / t_0 must stay finite, i.e. log(k!) - lp0 < 709.  64! = e^205 allows
// lp0 > -400; 128! = e^496 allows lp0 > -200 (bench workload: lp0 > -100 for 99 % of the pairs with counts above 64).
__device__ __forceinline__ double fast_lp0_min(double kmax) { return kmax <= 64.0 ? -400.0 : -200.0; }

// The recurrence itself for eligible counts (integers 0 .. kSmallK; -1 = not requested), valid (alpha, p) and
// lp0 = alpha log p > fast_lp0_min(kmax).  TWO = false skips the intermediate evaluation.
//   N_j = prod_{i<j} (alpha + i) x,  A_j = S_j (j-1)! / t_0  (A_{j+1} = A_j * j + N_j):  one step is 4 full-rate FP64
//   operations with no memory access and no division; the trip count is tested on the FP64 counter itself.
template <int W2, bool TWO>
__device__ __forceinline__ void nb_fast_recurrence(double kmin, double kmax, double alpha, double x, double lp0,
                                                   double& r_min, double& r_max, double thr = kDirectMin)
{
    const double t0 = fast_exp_neg_core(lp0);                    // -400 < lp0 <= 0 here
    double N = 1.0, A = 0.0, u = alpha * x, jj = 0.0;
    // steps are taken two at a time (one trip test per pair), then at most one single step
    if (TWO) {
        const double kmin_m1 = kmin - 1.0;
        while (jj < kmin_m1) {
            pmf_scaled_step_nofact(A, N, u, jj, x);
            pmf_scaled_step_nofact(A, N, u, jj, x);
        }
        if (jj < kmin) pmf_scaled_step_nofact(A, N, u, jj, x);
        r_min = tail_from_state_tab<W2>(A, N, jj, t0, thr);
    }
    const double kmax_m1 = kmax - 1.0;
    while (jj < kmax_m1) {
        pmf_scaled_step_nofact(A, N, u, jj, x);
        pmf_scaled_step_nofact(A, N, u, jj, x);
    }
    if (jj < kmax) pmf_scaled_step_nofact(A, N, u, jj, x);
    r_max = tail_from_state_tab<W2>(A, N, jj, t0, thr);
}

// Shared tail of the two front ends below: e1 / e2 say which counts are eligible for the fast recurrence.
template <int W2>
__device__ __forceinline__ unsigned nb_fast2_run(double k1, double k2, bool e1, bool e2, bool two, double alpha, double p,
                                                 double& r1, double& r2)
{
    // A count that is not resolved leaves a NEGATIVE value in its result for the compacted pass of the statistics block:
    // -pmf(k) when the direct form cancelled (tail_from_state_tab, W2 == 1), -2 when the recurrence never ran for it.
    if (!(p >= 2.2250738585072014e-308)) {   // subnormal p: leave it to the general path
        r1 = r2 = -2.0;
        return 0u;
    }
    const double lp0 = alpha * fast_log_normal(p);
    const double k1d = e1 ? k1 : -1.0, k2d = e2 ? k2 : -1.0;
    // the lane's loop ends at the larger count; only the smaller one needs recording on the way
    const double kmax = fmax(k1d, k2d), kmin = fmin(k1d, k2d);
    if (!(lp0 > fast_lp0_min(kmax))) {
        r1 = r2 = -2.0;
        return 0u;
    }
    const double x = 1.0 - p;
    double r_min = 0.0, r_max = 0.0;
    if (two) nb_fast_recurrence<W2, true>(kmin, kmax, alpha, x, lp0, r_min, r_max);
    else nb_fast_recurrence<W2, false>(kmin, kmax, alpha, x, lp0, r_min, r_max);
    const bool k1_is_max = !two || k1d >= k2d;
    const double ra = k1_is_max ? r_max : r_min;   // result for k1
    const double rb = k1_is_max ? r_min : r_max;   // result for k2
    unsigned done = 0;
    r1 = e1 ? ra : -2.0;
    r2 = e2 ? rb : -2.0;
    if (e1 && ra >= kDirectMin) done |= 1u;
    if (e2 && rb >= kDirectMin) done |= 2u;
    return done;
}

template <int W2>
__device__ __forceinline__ unsigned nb_midp_upper_fast2(double k1, double k2, unsigned want, double alpha, double p,
                                                        double& r1, double& r2)
{
    if (isnan(alpha) || isnan(p) || !(alpha > 0.0) || !(p > 0.0) || !(p <= 1.0) || isinf(alpha)) {
        // pmf is NaN whatever k is (scipy argcheck) -> NaN, resolved
        r1 = r2 = dnan();
        return want;
    }
    unsigned done = 0;
    if (isnan(k1)) { r1 = dnan(); done |= 1u; }
    if (isnan(k2)) { r2 = dnan(); done |= 2u; }
    if (p == 1.0) {
        if ((want & 1u) && k1 >= 0.0 && floor(k1) == k1) { r1 = (k1 == 0.0) ? (W2 ? 0.5 : 1.0) : 0.0; done |= 1u; }
        if ((want & 2u) && k2 >= 0.0 && floor(k2) == k2) { r2 = (k2 == 0.0) ? (W2 ? 0.5 : 1.0) : 0.0; done |= 2u; }
        return done & want;
    }
    const bool e1 = (want & 1u) && !(done & 1u) && k1 >= 0.0 && k1 <= (double)kSmallK && floor(k1) == k1;
    const bool e2 = (want & 2u) && !(done & 2u) && k2 >= 0.0 && k2 <= (double)kSmallK && floor(k2) == k2;
    if (!(e1 || e2)) return done & want;
    double t1 = 0.0, t2 = 0.0;                  // (the run writes both results; a count settled above keeps its value)
    done |= nb_fast2_run<W2>(k1, k2, e1, e2, (want & 2u) != 0u, alpha, p, t1, t2);
    if (e1) r1 = t1;
    if (e2) r2 = t2;
    return done & want;
}

// Front end for integer counts (the fused statistics kernels): anything unusual -- alpha or p not a finite number in
// the open range, p == 1, a negative count -- is simply left unresolved; the compacted pass takes those pairs through
// nb_midp_upper with the full scipy semantics.  Same recurrence, same bits as nb_midp_upper_fast2 on valid inputs.
template <int W2>
__device__ __forceinline__ unsigned nb_fast2_counts(int k1, int k2, bool two, double alpha, double p, double& r1,
                                                    double& r2)
{
    const bool e1 = k1 >= 0 && k1 <= kSmallK, e2 = two && k2 >= 0 && k2 <= kSmallK;
    if (!(alpha > 0.0 && alpha < __longlong_as_double(0x7ff0000000000000LL) && p > 0.0 && p < 1.0) || !(e1 || e2)) {
        r1 = r2 = -2.0;                      // (see nb_fast2_run)
        return 0u;
    }
    return nb_fast2_run<W2>((double)k1, (double)k2, e1, e2, two, alpha, p, r1, r2);
}

// Fisher via q = p1 p2:  chi2.sf(-2 ln q, 4) = q (1 - ln q); falls back to the log form when q
// is subnormal/zero or an argument is out of (0, 1].
__device__ __forceinline__ double fisher_combine_fast(double p1, double p2);

// betainc(k, alpha, 1 - p) = P(X >= k) for an integer count k >= 1 and valid (alpha, p) from the same recurrences as
// the mid-p statistic (weight 0 on pmf(k) in 1 - S_k - w t_k); anything else takes the general incomplete beta.
__device__ inline double nb_upper_incl(double k, double alpha, double p)
{
    if (!(k >= 1.0) || floor(k) != k || isinf(k) || isnan(alpha) || isnan(p) || !(alpha > 0.0) || !(p > 0.0) ||
        !(p < 1.0) || isinf(alpha))
        return betainc(k, alpha, 1.0 - p);
    double r = 0.0, dummy = 0.0;
    if (nb_midp_upper_fast2<0>(k, 0.0, 1u, alpha, p, r, dummy) & 1u) return r;
    nb_midp_upper_slow2<0>(k, 0.0, 1u, alpha, p, r, dummy);
    return r;
}

// nb_model.py:243-256
__device__ inline double nb_greater(double k, double alpha, double p)
{
    if (k == 0.0) return 1.0;
    double pv = nb_upper_incl(k, alpha, p);
    if (pv == 0.0) pv = nbinom_pmf(k, alpha, p);
    return pv;
}

// P(X <= k) = I_p(alpha, k+1) by direct summation for small integer k (all terms positive);
// returns false when the fast path does not apply.
__device__ __forceinline__ bool nb_lower_cdf_small(double k, double alpha, double p, double* out)
{
    if (!(alpha > 0.0) || !(p > 0.0) || !(p < 1.0) || isinf(alpha)) return false;
    if (!(k >= 0.0) || k > (double)kSmallK || floor(k) != k) return false;
    const double lp0 = alpha * fast_log(p);
    if (!(lp0 > -690.0)) return false;
    const double x = 1.0 - p;
    const double ax = alpha * x;
    double t = fast_exp_neg(lp0), S = t, u = ax, jj = 1.0;
    while (jj <= k) {
        t *= u * recip_nr(jj);
        S += t;
        u = fma(jj, x, ax);
        jj += 1.0;
    }
    *out = S;
    return true;
}

// Both tails of nb_pvalue_exact from ONE pass of the scaled recurrence, for the common case (valid alpha, 0 < p < 1,
// integer 0 <= k <= kSmallK, p^alpha not tiny):  S_k = sum_{j<k} pmf(j) and t_k = pmf(k) come out of the same state, the
// lower tail I_p(alpha, k + 1) is S_k + t_k and the upper tail I_{1-p}(k, alpha) is 1 - S_k.  One log and one exp per
// test; without this a wave whose lanes fall on both sides of the mean ran the two tail routines one after the other,
// each with its own log / exp.  Returns false when the general routines must take over.
__device__ __forceinline__ bool nb_exact_fast(double k, double alpha, double p, double mu, double* out)
{
    if (!(alpha > 0.0 && alpha < __longlong_as_double(0x7ff0000000000000LL) && p >= 2.2250738585072014e-308 && p < 1.0))
        return false;
    if (!(k >= 0.0) || k > (double)kSmallK || floor(k) != k) return false;
    const double lp0 = alpha * fast_log_normal(p);
    if (!(lp0 > fast_lp0_min(k))) return false;
    const double x = 1.0 - p;
    const double t0 = fast_exp_neg_core(lp0);
    double N = 1.0, A = 0.0, u = alpha * x, jj = 0.0;
    while (jj < k) pmf_scaled_step_nofact(A, N, u, jj, x);
    const double rD = t0 * g_inv_factorial[(int)k];
    const double S = (A * k) * rD;
    if (k < mu) {
        *out = S + N * rD;
        return true;
    }
    if (k == 0.0) return false;                // (mean underflowed to 0: betainc(0, ., .) is the general path's business)
    const double r = 1.0 - S;
    if (!(r >= kDirectMin)) return false;      // cancellation: the general path sums the tail directly
    *out = r;
    return true;
}

// nb_model.py:298-314 (mu defaults to alpha (1-p)/p)
__device__ inline double nb_exact(double k, double alpha, double p)
{
    const double mu = alpha * (1.0 - p) / p;
    {
        double r;
        if (nb_exact_fast(k, alpha, p, mu, &r)) return r;
    }
    if (k < mu) {
        double s;
        if (nb_lower_cdf_small(k, alpha, p, &s)) return s;
        return betainc(alpha, k + 1.0, p);
    }
    double pv = nb_upper_incl(k, alpha, p);
    if (pv == 0.0) pv = nbinom_pmf(k, alpha, p);
    return pv;
}

// nb_model.py:316-337
__device__ inline double nb_midp_twosided(double k, double alpha, double p)
{
    const double mu = alpha * (1.0 - p) / p;
    if (!(k < mu) && !isnan(k) && !isnan(mu))
        return nb_midp_upper(k, alpha, p);      // 0.5 pmf(k) + betainc(k + 1, alpha, 1 - p): the statistic of the burden test, by its own routes
    const double pmf = nbinom_pmf(k, alpha, p);
    if (k < mu) {
        if (k > 0.0) return 0.5 * pmf + betainc(alpha, k, p);
        return 0.5 * pmf;
    }
    return 0.5 * pmf + betainc(k + 1.0, alpha, 1.0 - p);
}

// transfer_tools.py:860-861,1086-1087: chi2.sf(-2 (ln p1 + ln p2), df=4)
__device__ __forceinline__ double fisher_combine(double p1, double p2)
{
    if (isnan(p1) || isnan(p2)) return dnan();
    const double h = -(log(p1) + log(p2));
    if (isnan(h)) return dnan();
    if (h < 0.0) return 1.0;
    if (isinf(h)) return 0.0;
    return exp(-h) * (1.0 + h);
}

__device__ __forceinline__ double fisher_combine_fast(double p1, double p2)
{
    const double q = p1 * p2;
    if (q > 1e-290 && p1 <= 1.0 && p2 <= 1.0) return q * (1.0 - fast_log_normal(q));
    return fisher_combine(p1, p2);
}

}  // namespace dig
