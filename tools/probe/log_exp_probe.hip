// Probe: accuracy of the device log / exp helpers of dig_math.hpp at chosen arguments (developer tool).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include "../../digdriver_amd/csrc/dig_common.hpp"
#include "../../digdriver_amd/csrc/dig_math.hpp"
using namespace dig;
__global__ void k(const double* in, double* out, int n)
{
    nb_tables_init();
    const int i = threadIdx.x;
    if (i < n) {
        out[4 * i + 0] = fast_log(in[i]);
        out[4 * i + 1] = fast_log_normal(in[i]);
        out[4 * i + 2] = fast_exp_neg(94716.93653374125 * fast_log(in[i]));
        out[4 * i + 3] = nb_upper_incl(220.0, 94716.93653374125, in[i]);
        if (i == 0) {
            const double alpha = 94716.93653374125, p = in[i], x = 1.0 - p, k = 220.0;
            const double lp0 = alpha * fast_log(p), t0 = fast_exp_neg(lp0);
            double N = 1.0, A = 0.0, D = 1.0, u = alpha * x, jj = 0.0;
            while (jj < k) {
                const double stop = fmin(k, jj + 16.0);
                while (jj < stop) pmf_scaled_step(A, N, D, u, jj, x, alpha * x);
                const int e = -__builtin_amdgcn_frexp_exp(D);
                D = ldexp(D, e); N = ldexp(N, e); A = ldexp(A, e);
            }
            const double rD = t0 * recip_nr(D), S = (A * k) * rD;
            printf("dev: lp0 %.17g t0 %.17g A %.17g N %.17g D %.17g rD %.17g S %.17g 1-S %.17g  1/D exact-ish %.17g\n", lp0, t0, A, N, D, rD, S, 1.0 - S, t0 / D);
        }
    }
}
int main()
{
    const int n = 4;
    double h[n] = {0.998341839400936, 0.5, 0.9999, 0.123456789};
    double *din, *dout, out[4 * n];
    hipMalloc(&din, sizeof(h)); hipMalloc(&dout, sizeof(out));
    hipMemcpy(din, h, sizeof(h), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(256), 0, 0, din, dout, n);
    hipMemcpy(out, dout, sizeof(out), hipMemcpyDeviceToHost);
    for (int i = 0; i < n; ++i) {
        const double l = std::log(h[i]);
        printf("x=%.17g  log host %.17g  fast_log %.17g (diff %.3g)  fast_log_normal %.17g (diff %.3g)  exp(alpha log) dev %.17g host %.17g rel %.3g  upper_incl %.17g\n",
               h[i], l, out[4 * i], out[4 * i] - l, out[4 * i + 1], out[4 * i + 1] - l, out[4 * i + 2], std::exp(94716.93653374125 * l),
               (out[4 * i + 2] - std::exp(94716.93653374125 * l)) / std::exp(94716.93653374125 * l), out[4 * i + 3]);
    }
    return 0;
}
