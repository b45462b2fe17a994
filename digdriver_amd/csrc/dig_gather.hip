// dig_gather.hip -- per-bin epigenomic-track gather feeding the CNN.
//
// Reference: LazyLoadDatasetFromH5.__getitem__ (region_model/data_aux/mut_dataset.py:76-81)
// re-opens the h5 file per sample and slices x_data[bin, :, selected_tracks]; here the
// bin x position x track matrix is resident in HBM and a batch of bins is gathered per launch.
//
// Pure HBM streaming (L*T*4 B read + L*T_sel*{2,4} B written per bin):
//   * row-major output [B, L, T], all tracks (tracks == NULL): batched contiguous block copy with
//     conversion, 8 / 16-byte accesses (gather_block_kernel);
//   * row-major output [B, L, T_sel]: one wave per (bin, position) row, lanes sweep the tracks,
//     so both the read (a contiguous T-row when the track list is a range) and the write are
//     256-B coalesced; no integer division in the inner loop;
//   * channels-first output [B, T, L], all tracks: 64-track x L tiles transposed through LDS with
//     4-value vector accesses on both sides (gather_transpose_all_kernel);
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
            for (int64_t t = lane; t < T_sel; t += 64) store_from_float<D>(od, t, load_as_float<S>(xs, tracks ? tracks[t] : (int)t));
        }
    }
}

// Track SUBSET (a track-selection file, dataset_generator.py:57-80), row-major output.  The plain kernel above issues one
// index load, one scattered source load and one scalar store per value.  Here a wave takes one (bin, position) row:
// the whole source row is copied to LDS with coalesced loads, every lane then assembles FOUR consecutive outputs from
// LDS (the track list is read once per lane and kept in registers for the kernel's lifetime: the selection does not change
// from row to row) and writes them with one vector store.  Needs T <= kSubsetMaxT and T_sel % 4 == 0 with a 16-byte
// aligned output; anything else takes the plain kernel.
__device__ __forceinline__ void store_vec4(float* p, float a, float b, float c, float d);
__device__ __forceinline__ void store_vec4(__hip_bfloat16* p, float a, float b, float c, float d);
constexpr int kSubsetMaxT = 2048;
constexpr int kSubsetMaxVec = 8;          // vectors of four outputs per lane: T_sel <= 2048

template <typename S, typename D>
__global__ __launch_bounds__(kGatherBlock) void gather_rows_subset_kernel(const S* __restrict__ x, int64_t L, int64_t T,
                                                                          const int64_t* __restrict__ rows, int64_t B,
                                                                          const int32_t* __restrict__ tracks, int64_t T_sel,
                                                                          D* __restrict__ out)
{
    __shared__ float s_row[kGatherBlock / 64][kSubsetMaxT];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, nw = kGatherBlock >> 6;
    float* row = s_row[wave];
    const int n_vec = (int)(T_sel >> 2);
    int4 sel[kSubsetMaxVec];
#pragma unroll
    for (int v = 0; v < kSubsetMaxVec; ++v) {
        const int q = lane + 64 * v;
        sel[v] = q < n_vec ? *reinterpret_cast<const int4*>(tracks + 4 * q) : make_int4(0, 0, 0, 0);
    }
    for (int64_t b = blockIdx.x; b < B; b += gridDim.x) {
        const int64_t src0 = rows[b] * L * T;
        const int64_t dst0 = b * L * T_sel;
        for (int64_t l = wave; l < L; l += nw) {
            const S* xs = x + src0 + l * T;
            for (int64_t t = lane; t < T; t += 64) row[t] = load_as_float<S>(xs, t);
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_s_waitcnt(0xc07f);                  // lgkmcnt(0): the row is in LDS (one wave: no s_barrier needed)
            D* od = out + dst0 + l * T_sel;
#pragma unroll
            for (int v = 0; v < kSubsetMaxVec; ++v) {
                const int q = lane + 64 * v;
                if (q < n_vec) store_vec4(od + 4 * q, row[sel[v].x], row[sel[v].y], row[sel[v].z], row[sel[v].w]);
            }
            __builtin_amdgcn_wave_barrier();
        }
    }
}

// Wide form of the subset gather (round 3).  The kernel above reads a source row value by value (2-byte accesses for the
// int16 matrix: 128 bytes per load instruction).  Rows of a bin are contiguous, and G = 8 / sizeof(S) consecutive rows
// (4 of int16, 2 of float, 1 of double) are one block of 8 T bytes that starts 8-byte aligned whenever a bin does
// (L T sizeof(S) % 8 == 0) -- whatever T is.  A wave therefore takes GROUPS of G positions: T 8-byte loads (512 bytes per
// instruction) put the block into LDS in the source type; then, row by row, every lane assembles four consecutive outputs
// from LDS with the track list it keeps in registers and writes them with one vector store.  Needs L % G == 0 and
// T <= kWideMaxT; anything else takes the kernel above.
constexpr int kWideMaxT = 1024;

template <typename S, typename D>
__global__ __launch_bounds__(kGatherBlock) void gather_rows_subset_wide_kernel(const S* __restrict__ x, int64_t L, int64_t T,
                                                                               const int64_t* __restrict__ rows, int64_t B,
                                                                               const int32_t* __restrict__ tracks, int64_t T_sel,
                                                                               D* __restrict__ out)
{
    constexpr int G = 8 / (int)sizeof(S);
    __shared__ uint2 s_blk[kGatherBlock / 64][kWideMaxT];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, nw = kGatherBlock >> 6;
    uint2* blk = s_blk[wave];
    const S* vals = reinterpret_cast<const S*>(blk);             // [G][T]
    const int n_vec = (int)(T_sel >> 2);
    int4 sel[kSubsetMaxVec];
#pragma unroll
    for (int v = 0; v < kSubsetMaxVec; ++v) {
        const int q = lane + 64 * v;
        sel[v] = q < n_vec ? *reinterpret_cast<const int4*>(tracks + 4 * q) : make_int4(0, 0, 0, 0);
    }
    const int64_t groups_per_bin = L / G, n_groups = B * groups_per_bin;
    for (int64_t grp = (int64_t)blockIdx.x * nw + wave; grp < n_groups; grp += (int64_t)gridDim.x * nw) {
        const int64_t b = grp / groups_per_bin, g = grp - b * groups_per_bin;
        const uint2* src = reinterpret_cast<const uint2*>(x + (rows[b] * L + g * G) * T);
        {   // all loads of the block are issued before the first LDS write (a plain loop waits for every load in turn)
            uint2 tmp[kWideMaxT / 64];
#pragma unroll
            for (int j = 0; j < kWideMaxT / 64; ++j) {
                const int t = lane + 64 * j;
                if (t < (int)T) tmp[j] = src[t];
            }
#pragma unroll
            for (int j = 0; j < kWideMaxT / 64; ++j) {
                const int t = lane + 64 * j;
                if (t < (int)T) blk[t] = tmp[j];
            }
        }
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_s_waitcnt(0xc07f);                      // lgkmcnt(0): the block is in LDS (one wave: no s_barrier needed)
#pragma unroll
        for (int r = 0; r < G; ++r) {
            const S* row = vals + (int64_t)r * T;
            D* od = out + ((b * L + g * G + r) * T_sel);
#pragma unroll
            for (int v = 0; v < kSubsetMaxVec; ++v) {
                const int q = lane + 64 * v;
                if (q < n_vec)
                    store_vec4(od + 4 * q, (float)row[sel[v].x], (float)row[sel[v].y], (float)row[sel[v].z], (float)row[sel[v].w]);
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
}

// All tracks selected (tracks == NULL), row-major output: a bin is ONE contiguous block of L * T values on both sides, so
// the gather is a batched block copy with conversion.  Four values per lane and access (8-byte loads of i16, 16-byte
// of f32; 8-byte stores of bf16, 16-byte of f32), kBlockUnroll independent accesses per lane in flight.
// grid = (chunks of a bin, bins).  Needs (L * T) % 4 == 0 so that every bin starts on a vector boundary.
constexpr int kBlockUnroll = 4;

template <typename S>
struct Vec4In;
template <>
struct Vec4In<int16_t> {
    short4 v;
    __device__ __forceinline__ float get(int i) const { return (float)(i == 0 ? v.x : i == 1 ? v.y : i == 2 ? v.z : v.w); }
};
template <>
struct Vec4In<float> {
    float4 v;
    __device__ __forceinline__ float get(int i) const { return i == 0 ? v.x : i == 1 ? v.y : i == 2 ? v.z : v.w; }
};
template <>
struct Vec4In<double> {
    double4 v;
    __device__ __forceinline__ float get(int i) const { return (float)(i == 0 ? v.x : i == 1 ? v.y : i == 2 ? v.z : v.w); }
};

// The outputs are written once and read by another kernel much later: non-temporal stores (no write-allocate in L2).
// Measured on the channels-first f32 form: 0.62 -> 0.37 ms per 4096 bins; the row-major forms gain 3-5 %.
typedef float float4_nt __attribute__((ext_vector_type(4)));
typedef unsigned uint2_nt __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void store_vec4(float* p, float a, float b, float c, float d)
{
    float4_nt v = {a, b, c, d};
    __builtin_nontemporal_store(v, reinterpret_cast<float4_nt*>(p));
}
__device__ __forceinline__ void store_vec4(__hip_bfloat16* p, float a, float b, float c, float d)
{
    union { __hip_bfloat16 h[4]; uint2_nt u; } pk;
    pk.h[0] = __float2bfloat16(a); pk.h[1] = __float2bfloat16(b); pk.h[2] = __float2bfloat16(c); pk.h[3] = __float2bfloat16(d);
    __builtin_nontemporal_store(pk.u, reinterpret_cast<uint2_nt*>(p));
}

template <typename S, typename D>
__global__ __launch_bounds__(kGatherBlock) void gather_block_kernel(const S* __restrict__ x, int64_t n_per_bin,
                                                                    const int64_t* __restrict__ rows, int64_t B,
                                                                    D* __restrict__ out)
{
    const int64_t nv = n_per_bin >> 2;
    for (int64_t b = blockIdx.y; b < B; b += gridDim.y) {
        const Vec4In<S>* src = reinterpret_cast<const Vec4In<S>*>(x + rows[b] * n_per_bin);
        D* dst = out + b * n_per_bin;
        for (int64_t v0 = (int64_t)blockIdx.x * (kGatherBlock * kBlockUnroll) + threadIdx.x; v0 < nv;
             v0 += (int64_t)gridDim.x * (kGatherBlock * kBlockUnroll)) {
            Vec4In<S> in[kBlockUnroll];
#pragma unroll
            for (int u = 0; u < kBlockUnroll; ++u)
                if (v0 + u * kGatherBlock < nv) in[u] = src[v0 + u * kGatherBlock];
#pragma unroll
            for (int u = 0; u < kBlockUnroll; ++u)
                if (v0 + u * kGatherBlock < nv)
                    store_vec4(dst + 4 * (v0 + u * kGatherBlock), in[u].get(0), in[u].get(1), in[u].get(2), in[u].get(3));
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
        const int tr = (lane < nt) ? (tracks ? tracks[t0 + lane] : (int)(t0 + lane)) : 0;
        for (int64_t l0 = wave; l0 < L; l0 += 8 * nw) {           // eight rows of the tile in flight per wave (a plain loop waits
            float v[8];                                             //  for every load in turn)
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int64_t l = l0 + u * nw;
                v[u] = (lane < nt && l < L) ? load_as_float<S>(x, src0 + l * T + tr) : 0.0f;
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int64_t l = l0 + u * nw;
                if (lane < nt && l < L) tile[l * 65 + lane] = v[u];
            }
        }
        __syncthreads();
        for (int tt = wave; tt < nt; tt += nw) {
            D* od = out + (b * T_sel + t0 + tt) * L;
            for (int64_t l = lane; l < L; l += 64) store_from_float<D>(od, l, tile[l * 65 + tt]);
        }
        __syncthreads();
    }
}

// out[b, t, l] channels-first, all tracks (tracks == NULL), L % 4 == 0.  grid = (track tiles of 64, bins).
// Both sides of the tile move as 4-value vectors.  A source row of the tile (64 consecutive tracks of one position) is
// only element-aligned when T is odd.  Misaligned 8-byte loads of i16 run at half speed, so a 2-byte source is read as
// the 17 ALIGNED 4-value vectors that cover the row (it starts `delta` = 0..3 values into the first one; values outside
// the tile are dropped); 16-byte loads of f32 at 4-byte alignment measured faster than the 17-vector form and stay.  A lane scatters its four values into a transposed LDS tile [64][L + 1] (odd row stride: the
// track quads x positions of a wave spread over all 32 banks) and, after the barrier, reads four consecutive positions
// of one track and stores them as one 8 / 16-byte vector -- the 64 output rows of a tile are one contiguous block.
constexpr int kTransUnroll = 4;

template <typename S>
struct Quad {
    S v[4];
};

template <typename S, typename D>
__global__ __launch_bounds__(kGatherBlock) void gather_transpose_all_kernel(const S* __restrict__ x, int64_t n_total, int L,
                                                                            int64_t T, const int64_t* __restrict__ rows,
                                                                            int64_t B, D* __restrict__ out)
{
    extern __shared__ float tile[];   // [64][L + 1]
    const int Sr = L + 1;
    const int64_t t0 = (int64_t)blockIdx.x * 64;
    const int nt = (int)((T - t0 < 64) ? (T - t0) : 64);
    constexpr bool kAligned = sizeof(S) == 2;
    constexpr int kTransQuads = kAligned ? 17 : 16;
    const int n_items = L * kTransQuads;           // (position, quad)
    const float inv_L = 1.0f / (float)L;
    for (int64_t b = blockIdx.y; b < B; b += gridDim.y) {
        const int64_t src0 = rows[b] * L * T + t0;          // element index of (position 0, track t0)
        for (int base = threadIdx.x; base < n_items; base += kGatherBlock * kTransUnroll) {
            Quad<S> q[kTransUnroll];
#pragma unroll
            for (int u = 0; u < kTransUnroll; ++u) {
                const int idx = base + u * kGatherBlock;
                if (idx < n_items) {
                    const int l = idx / kTransQuads, k = idx - l * kTransQuads;
                    const int64_t row = src0 + (int64_t)l * T;
                    const int64_t g = (kAligned ? (row & ~(int64_t)3) : row) + 4 * k;   // (x itself is 4-value aligned)
                    if (g + 3 < n_total) {
                        __builtin_memcpy(&q[u], x + g, sizeof(Quad<S>));
                    } else {
#pragma unroll
                        for (int j = 0; j < 4; ++j) q[u].v[j] = (g + j < n_total) ? x[g + j] : (S)0;
                    }
                }
            }
#pragma unroll
            for (int u = 0; u < kTransUnroll; ++u) {
                const int idx = base + u * kGatherBlock;
                if (idx < n_items) {
                    const int l = idx / kTransQuads, k = idx - l * kTransQuads;
                    const int delta = kAligned ? (int)((src0 + (int64_t)l * T) & 3) : 0;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int t = 4 * k - delta + j;
                        if (t >= 0 && t < nt) tile[t * Sr + l] = (float)q[u].v[j];
                    }
                }
            }
        }
        __syncthreads();
        D* dst = out + (b * T + t0) * L;
        const int nvec = nt * L / 4;
        for (int vi = threadIdx.x; vi < nvec; vi += kGatherBlock) {
            const int e = 4 * vi;
            const int tr = (int)(((float)e + 0.5f) * inv_L);      // e / L, exact for e < 64 * 256
            const float* r = tile + tr * Sr + (e - tr * L);
            store_vec4(dst + e, r[0], r[1], r[2], r[3]);
        }
        __syncthreads();
    }
}

template <typename S, typename D>
static int launch_gather(const void* x, int64_t n_total, int64_t L, int64_t T, const int64_t* rows, int64_t B, const int32_t* tracks,
                         int64_t T_sel, void* out, int transpose_out, hipStream_t stream)
{
    if (!transpose_out && !tracks && ((L * T) & 3) == 0 && ((uintptr_t)x & 31) == 0 && ((uintptr_t)out & 15) == 0) {
        const int64_t nv = (L * T) >> 2;
        const int chunks = (int)std::min<int64_t>((nv + kGatherBlock * kBlockUnroll - 1) / (kGatherBlock * kBlockUnroll), 64);
        const int gy = (int)std::min<int64_t>(B, 65535);
        hipLaunchKernelGGL((gather_block_kernel<S, D>), dim3(chunks, gy), dim3(kGatherBlock), 0, stream, (const S*)x, L * T,
                           rows, B, (D*)out);
    } else if (!transpose_out && tracks && T <= kWideMaxT && (T_sel & 3) == 0 && T_sel > 0 && T_sel <= 256 * kSubsetMaxVec &&
               ((uintptr_t)out & 15) == 0 && ((uintptr_t)tracks & 15) == 0 && ((uintptr_t)x & 7) == 0 &&
               (L * T * (int64_t)sizeof(S)) % 8 == 0 && L % (8 / (int64_t)sizeof(S)) == 0 && (T_sel * (int64_t)sizeof(D)) % 16 == 0) {
        const int64_t n_groups = B * (L / (8 / (int64_t)sizeof(S)));
        int grid = (int)std::max<int64_t>(1, std::min<int64_t>((n_groups + 3) / 4, (int64_t)cu_count() * 4));
        hipLaunchKernelGGL((gather_rows_subset_wide_kernel<S, D>), dim3(grid), dim3(kGatherBlock), 0, stream, (const S*)x, L, T, rows,
                           B, tracks, T_sel, (D*)out);
    } else if (!transpose_out && tracks && T <= kSubsetMaxT && (T_sel & 3) == 0 && T_sel > 0 && T_sel <= 256 * kSubsetMaxVec &&
               ((uintptr_t)out & 15) == 0 && ((uintptr_t)tracks & 15) == 0) {
        int grid = (int)std::min<int64_t>(B, (int64_t)cu_count() * 8);
        hipLaunchKernelGGL((gather_rows_subset_kernel<S, D>), dim3(grid), dim3(kGatherBlock), 0, stream, (const S*)x, L, T, rows,
                           B, tracks, T_sel, (D*)out);
    } else if (!transpose_out) {
        int grid = (int)std::min<int64_t>(B, (int64_t)cu_count() * 8);
        hipLaunchKernelGGL((gather_rows_kernel<S, D>), dim3(grid), dim3(kGatherBlock), 0, stream, (const S*)x, L, T, rows,
                           B, tracks, T_sel, (D*)out);
    } else if (!tracks && (L & 3) == 0 && L <= 252 && ((uintptr_t)out & 15) == 0 && ((uintptr_t)x & (4 * sizeof(S) - 1)) == 0) {
        const size_t lds = (size_t)64 * (L + 1) * sizeof(float);
        const int tiles = (int)((T + 63) / 64);
        const int gy = (int)std::min<int64_t>(B, std::max<int64_t>(1, (int64_t)cu_count() * 16 / tiles));
        DIG_HIP_TRY(hipFuncSetAttribute((const void*)gather_transpose_all_kernel<S, D>,
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL((gather_transpose_all_kernel<S, D>), dim3(tiles, gy), dim3(kGatherBlock), lds, stream,
                           (const S*)x, n_total, (int)L, T, rows, B, (D*)out);
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
    DIG_REQUIRE(x_data && bin_rows && out, "non-null pointers");
    DIG_REQUIRE(tracks || T_sel == T, "tracks == NULL selects all tracks: T_sel must equal T");
    DIG_REQUIRE(src_dtype == DIG_F32 || src_dtype == DIG_F64 || src_dtype == DIG_I16, "src_dtype f32|f64|i16");
    DIG_REQUIRE(out_dtype == DIG_F32 || out_dtype == DIG_BF16, "out_dtype f32|bf16");
    hipStream_t s = (hipStream_t)stream;
#define GO(S, D) return launch_gather<S, D>(x_data, N * L * T, L, T, bin_rows, B, tracks, T_sel, out, transpose_out, s)
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
    DIG_REQUIRE(x_data && bin_rows && out, "non-null pointers");
    DIG_REQUIRE(tracks || T_sel == T, "tracks == NULL selects all tracks: T_sel must equal T");
    const size_t ss = dtype_size(src_dtype), ds = dtype_size(out_dtype);
    DIG_REQUIRE(ss && ds, "known dtypes");
    for (int64_t b = 0; b < B; ++b) DIG_REQUIRE(bin_rows[b] >= 0 && bin_rows[b] < N, "bin_rows within [0, N)");
    for (int64_t t = 0; tracks && t < T_sel; ++t) DIG_REQUIRE(tracks[t] >= 0 && tracks[t] < T, "tracks within [0, T)");
    DIG_HIP_TRY(hipSetDevice(device));
    DevBuf dx, dr, dt, dout;
    const size_t xb = (size_t)N * L * T * ss, ob = (size_t)B * L * T_sel * ds;
    DIG_HIP_TRY(dx.alloc(xb));
    DIG_HIP_TRY(dr.alloc((size_t)B * 8));
    if (tracks) DIG_HIP_TRY(dt.alloc((size_t)T_sel * 4));
    DIG_HIP_TRY(dout.alloc(ob));
    DIG_HIP_TRY(hipMemcpy(dx.p, x_data, xb, hipMemcpyHostToDevice));
    DIG_HIP_TRY(hipMemcpy(dr.p, bin_rows, (size_t)B * 8, hipMemcpyHostToDevice));
    if (tracks) DIG_HIP_TRY(hipMemcpy(dt.p, tracks, (size_t)T_sel * 4, hipMemcpyHostToDevice));
    int rc = dig_gather_bins(dx.p, src_dtype, N, L, T, dr.as<int64_t>(), B, tracks ? dt.as<int32_t>() : nullptr, T_sel, dout.p, out_dtype,
                             transpose_out, nullptr);
    if (rc) return rc;
    DIG_HIP_TRY(hipDeviceSynchronize());
    DIG_HIP_TRY(hipMemcpy(out, dout.p, ob, hipMemcpyDeviceToHost));
    return DIG_OK;
}

}  // extern "C"
