// Probe: operand / result lane layout of v_mfma_f64_16x16x4_f64 on gfx950 (developer tool).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef double double4_t __attribute__((ext_vector_type(4)));
__global__ void probe(const double* A, const double* B, double* D)   // A[16][4], B[4][16] row-major; D raw [64][4]
{
    const int l = threadIdx.x, i = l & 15, k = l >> 4;
    double4_t acc = {0, 0, 0, 0};
    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(A[i * 4 + k], B[k * 16 + i], acc, 0, 0, 0);
    for (int r = 0; r < 4; ++r) D[l * 4 + r] = acc[r];
}
int main()
{
    std::vector<double> A(64), B(64), D(256), ref(256, 0.0);
    for (int i = 0; i < 16; ++i) for (int k = 0; k < 4; ++k) A[i * 4 + k] = 1 + i + 17 * k;
    for (int k = 0; k < 4; ++k) for (int j = 0; j < 16; ++j) B[k * 16 + j] = 3 + 5 * j + 101 * k;
    for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) for (int k = 0; k < 4; ++k) ref[i * 16 + j] += A[i * 4 + k] * B[k * 16 + j];
    double *dA, *dB, *dD;
    hipMalloc(&dA, 512); hipMalloc(&dB, 512); hipMalloc(&dD, 2048);
    hipMemcpy(dA, A.data(), 512, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), 512, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, dA, dB, dD);
    hipMemcpy(D.data(), dD, 2048, hipMemcpyDeviceToHost);
    int ok1 = 1, ok2 = 1, ok3 = 1;
    for (int l = 0; l < 64; ++l) for (int r = 0; r < 4; ++r) {
        const double v = D[l * 4 + r];
        if (v != ref[(4 * (l >> 4) + r) * 16 + (l & 15)]) ok1 = 0;     // i = 4*(l/16)+r, j = l%16
        if (v != ref[((l >> 4) + 4 * r) * 16 + (l & 15)]) ok2 = 0;     // i = l/16 + 4r,  j = l%16
        if (v != ref[(l & 15) * 16 + 4 * (l >> 4) + r]) ok3 = 0;       // i = l%16, j = 4*(l/16)+r
    }
    printf("layout i=4*(l/16)+r,j=l%%16: %d ; i=l/16+4r,j=l%%16: %d ; i=l%%16,j=4*(l/16)+r: %d\n", ok1, ok2, ok3);
    for (int l = 0; l < 64; l += 16) printf("lane %d: %g %g %g %g   ref[0][0]=%g ref[1][0]=%g ref[4][0]=%g ref[0][1]=%g\n", l, D[l*4], D[l*4+1], D[l*4+2], D[l*4+3], ref[0], ref[16], ref[64], ref[1]);
    return 0;
}
