// dig_gather.hip -- per-bin epigenomic-track gather feeding the CNN.
//
// Reference: LazyLoadDatasetFromH5.__getitem__ (region_model/data_aux/mut_dataset.py:76-81)
// re-opens the h5 file per sample and slices x_data[bin, :, selected_tracks]; here the
// bin x position x track matrix is resident in HBM and a batch of bins is gathered per launch.
//
// Pure HBM streaming (L*T*4 B read + L*T_sel*{2,4} B written per bin):
//   * row-major output [B, L, T_sel]: one wave per (bin, position) row, lanes sweep the tracks,
//     so both the read (a contiguous T-row when the track list is a range) and the write are
//     256-B coalesced; no integer division in the inner loop;
//   * channels-first output [B, T_sel, L] (what conv1d consumes after the reference's
//     transpose(x, 1, 2), cnn_predictors.py:131): a 64-track x L tile is staged through LDS
//     ([L][65] floats, padded against bank conflicts) so reads stay track-contiguous and
//     writes position-contiguous.
// Values are round(x, 2) * 100 (DataExtractor.py:220): exact in f32 and i16; bf16 output is
// exact only up to 256 and is offered for the bf16 CNN path.
#include <hip/hip_bf16.h>

#include "dig_common.hpp"

namespace dig {

template <typename T>
__device__ __forceinline__ float load_as_float(const T* p, int64_t i)
{
    return (float)p[i];
}

template <typename D>
__device__ __forceinline__ void store_from_float(D* p, int64_t i, float v);
template <>
__device__ __forceinline__ void store_from_float<float>(float* p, int64_t i, float v)
{
    p[i] = v;
}
template <>
__device__ __forceinline__ void store_from_float<__hip_bfloat16>(__hip_bfloat16* p, int64_t i, float v)
{
    p[i] = __float2bfloat16(v);
}

constexpr int kGatherBlock = 256;

// out[b, l, t] row-major.  grid.x = bins, each workgroup sweeps the L rows of one bin.
template <typename S, typename D>
__global__ __launch_bounds__(kGatherBlock) void gather_rows_kernel(const S* __restrict__ x, int64_t L, int64_t T,
                                                                   const int64_t* __restrict__ rows, int64_t B,
                                                                   const int32_t* __restrict__ tracks, int64_t T_sel,
                                                                   D* __restrict__ out)
{
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, nw = kGatherBlock >> 6;
    for (int64_t b = blockIdx.x; b < B; b += gridDim.x) {
        const int64_t src0 = rows[b] * L * T;
        const int64_t dst0 = b * L * T_sel;
        for (int64_t l = wave; l < L; l += nw) {
            const S* xs = x + src0 + l * T;
            D* od = out + dst0 + l * T_sel;
            for (int64_t t = lane; t < T_sel; t += 64) store_from_float<D>(od, t, load_as_float<S>(xs, tracks[t]));
        }
    }
}

// out[b, t, l] channels-first.  grid = (track tiles of 64, bins)
template <typename S, typename D>
__global__ __launch_bounds__(kGatherBlock) void gather_transpose_kernel(const S* __restrict__ x, int64_t L, int64_t T,
                                                                        const int64_t* __restrict__ rows, int64_t B,
                                                                        const int32_t* __restrict__ tracks,
                                                                        int64_t T_sel, D* __restrict__ out)
{
    extern __shared__ float tile[];   // [L][65]
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, nw = kGatherBlock >> 6;
    const int64_t t0 = (int64_t)blockIdx.x * 64;
    const int nt = (int)((T_sel - t0 < 64) ? (T_sel - t0) : 64);
    for (int64_t b = blockIdx.y; b < B; b += gridDim.y) {
        const int64_t src0 = rows[b] * L * T;
        const int tr = (lane < nt) ? tracks[t0 + lane] : 0;
        for (int64_t l = wave; l < L; l += nw)
            if (lane < nt) tile[l * 65 + lane] = load_as_float<S>(x, src0 + l * T + tr);
        __syncthreads();
        for (int tt = wave; tt < nt; tt += nw) {
            D* od = out + (b * T_sel + t0 + tt) * L;
            for (int64_t l = lane; l < L; l += 64) store_from_float<D>(od, l, tile[l * 65 + tt]);
        }
        __syncthreads();
    }
}

template <typename S, typename D>
static int launch_gather(const void* x, int64_t L, int64_t T, const int64_t* rows, int64_t B, const int32_t* tracks,
                         int64_t T_sel, void* out, int transpose_out, hipStream_t stream)
{
    if (!transpose_out) {
        int grid = (int)std::min<int64_t>(B, (int64_t)cu_count() * 8);
        hipLaunchKernelGGL((gather_rows_kernel<S, D>), dim3(grid), dim3(kGatherBlock), 0, stream, (const S*)x, L, T, rows,
                           B, tracks, T_sel, (D*)out);
    } else {
        const size_t lds = (size_t)L * 65 * sizeof(float);
        if (lds > 150 * 1024) return set_error(DIG_EINVAL, "gather: L=%lld too long for the LDS transpose tile", (long long)L);
        const int tiles = (int)((T_sel + 63) / 64);
        int gy = (int)std::min<int64_t>(B, std::max<int64_t>(1, (int64_t)cu_count() * 8 / tiles));
        DIG_HIP_TRY(hipFuncSetAttribute((const void*)gather_transpose_kernel<S, D>,
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL((gather_transpose_kernel<S, D>), dim3(tiles, gy), dim3(kGatherBlock), lds, stream,
                           (const S*)x, L, T, rows, B, tracks, T_sel, (D*)out);
    }
    DIG_HIP_TRY(hipGetLastError());
    return DIG_OK;
}

static size_t dtype_size(int dt)
{
    switch (dt) {
        case DIG_F32: return 4;
        case DIG_F64: return 8;
        case DIG_I16: return 2;
        case DIG_BF16: return 2;
        default: return 0;
    }
}

}  // namespace dig

using namespace dig;

extern "C" {

int dig_gather_bins(const void* x_data, int src_dtype, int64_t N, int64_t L, int64_t T, const int64_t* bin_rows,
                    int64_t B, const int32_t* tracks, int64_t T_sel, void* out, int out_dtype, int transpose_out,
                    void* stream)
{
    DIG_REQUIRE(N >= 0 && L > 0 && T > 0 && B >= 0 && T_sel >= 0, "sizes");
    if (B == 0 || T_sel == 0) return DIG_OK;
    DIG_REQUIRE(x_data && bin_rows && tracks && out, "non-null pointers");
    DIG_REQUIRE(src_dtype == DIG_F32 || src_dtype == DIG_F64 || src_dtype == DIG_I16, "src_dtype f32|f64|i16");
    DIG_REQUIRE(out_dtype == DIG_F32 || out_dtype == DIG_BF16, "out_dtype f32|bf16");
    hipStream_t s = (hipStream_t)stream;
#define GO(S, D) return launch_gather<S, D>(x_data, L, T, bin_rows, B, tracks, T_sel, out, transpose_out, s)
    if (out_dtype == DIG_F32) {
        if (src_dtype == DIG_F32) GO(float, float);
        if (src_dtype == DIG_F64) GO(double, float);
        GO(int16_t, float);
    }
    if (src_dtype == DIG_F32) GO(float, __hip_bfloat16);
    if (src_dtype == DIG_F64) GO(double, __hip_bfloat16);
    GO(int16_t, __hip_bfloat16);
#undef GO
}

int dig_gather_bins_host(const void* x_data, int src_dtype, int64_t N, int64_t L, int64_t T, const int64_t* bin_rows,
                         int64_t B, const int32_t* tracks, int64_t T_sel, void* out, int out_dtype, int transpose_out,
                         int device)
{
    DIG_REQUIRE(N >= 0 && L > 0 && T > 0 && B >= 0 && T_sel >= 0, "sizes");
    if (B == 0 || T_sel == 0) return DIG_OK;
    DIG_REQUIRE(x_data && bin_rows && tracks && out, "non-null pointers");
    const size_t ss = dtype_size(src_dtype), ds = dtype_size(out_dtype);
    DIG_REQUIRE(ss && ds, "known dtypes");
    for (int64_t b = 0; b < B; ++b) DIG_REQUIRE(bin_rows[b] >= 0 && bin_rows[b] < N, "bin_rows within [0, N)");
    for (int64_t t = 0; t < T_sel; ++t) DIG_REQUIRE(tracks[t] >= 0 && tracks[t] < T, "tracks within [0, T)");
    DIG_HIP_TRY(hipSetDevice(device));
    DevBuf dx, dr, dt, dout;
    const size_t xb = (size_t)N * L * T * ss, ob = (size_t)B * L * T_sel * ds;
    DIG_HIP_TRY(dx.alloc(xb));
    DIG_HIP_TRY(dr.alloc((size_t)B * 8));
    DIG_HIP_TRY(dt.alloc((size_t)T_sel * 4));
    DIG_HIP_TRY(dout.alloc(ob));
    DIG_HIP_TRY(hipMemcpy(dx.p, x_data, xb, hipMemcpyHostToDevice));
    DIG_HIP_TRY(hipMemcpy(dr.p, bin_rows, (size_t)B * 8, hipMemcpyHostToDevice));
    DIG_HIP_TRY(hipMemcpy(dt.p, tracks, (size_t)T_sel * 4, hipMemcpyHostToDevice));
    int rc = dig_gather_bins(dx.p, src_dtype, N, L, T, dr.as<int64_t>(), B, dt.as<int32_t>(), T_sel, dout.p, out_dtype,
                             transpose_out, nullptr);
    if (rc) return rc;
    DIG_HIP_TRY(hipDeviceSynchronize());
    DIG_HIP_TRY(hipMemcpy(out, dout.p, ob, hipMemcpyDeviceToHost));
    return DIG_OK;
}

}  // extern "C"
