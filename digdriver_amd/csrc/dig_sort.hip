// dig_sort.hip -- ranking the p-values of many lists at once on gfx950: a batched LSD radix sort written for this step, and the
// Benjamini-Hochberg pass behind it (nb_model.get_q_vals, nb_model.py:340-342 = statsmodels' fdrcorrection, method 'indep').
//
// Until round 5 the per-base route of BASELINE configs[4] ended in torch.sort (rocPRIM: 64-bit keys + 64-bit indices, eight
// 8-bit passes) + dig_bh_qvalues_sorted + scatter_: 40 of the route's 43 ms per eighth of the genome x 37 cohorts.  Here (15.7 -
// 16.0 ms; what was measured on the way: profiles/r06_sort_probes.txt):
//   * rows = lists (cohorts), ragged (row_ptr): all rows in one launch sequence; a row's keys never leave its range;
//   * key = the double's bits made monotone (sign handled; every NaN = the largest key, as torch.sort places it; -0 = +0);
//     payload = the 32-bit position in the row (rows < 2^30 elements): 12 bytes per element and pass instead of 16;
//   * a pass = the digit histograms of the ranges (65 536 elements), one scan per row, and the scatter: a workgroup walks its
//     range tile by tile (8 192 elements), ranks a tile's elements stably (wave-wide match of the 9-bit digit by ballots, per-wave
//     digit counters in LDS), regroups the tile by digit in LDS and writes every digit's run contiguously behind the runs its
//     earlier tiles wrote; no workgroup waits for another one;
//   * FOUR passes over bits 27 .. 62 (a fifth over the sign only when a device flag says a negative value exists), then a
//     fix-up of the short runs that share those 36 bits; a long run with two different keys sets a device flag and eight gated
//     "careful" passes over all 63 bits redo the sort (launches that return at once otherwise);
//   * the Benjamini-Hochberg pass reads the sorted keys, and the q-values leave through the payload to their places: no
//     separate scatter.  The same IEEE operations in the same order as the host form -- p / (rank / n), reverse running
//     minimum (NaN-propagating), cap at 1 -- so the same bits as statsmodels' operations (q depends on the VALUE of p only:
//     equal p-values get equal q whatever order a sort leaves them in).
//   * rank0 / n_global / carry per row: a rank of parallel.ShardedTiles holds one contiguous range of the global order (sample
//     sort) and finishes it with the minimum of the ranks behind it.
#include <stdlib.h>

#include "dig_common.hpp"

namespace dig {

constexpr int kSortBits = 9, kSortBins = 1 << kSortBits;        // digit
constexpr int kSortBlock = 512, kSortItems = 8;                 // threads per workgroup, elements per thread
constexpr int kSortTile = kSortBlock * kSortItems;              // elements per tile
static_assert(kSortBlock == kSortBins, "a thread of the pass kernel = a digit");
constexpr int kSortPasses = 8;                                  // 7 x 9 bits + the sign bit

__device__ __forceinline__ uint64_t sort_key(double p)
{
    const uint64_t b = (uint64_t)__double_as_longlong(p), mag = b & 0x7fffffffffffffffull;      // (integer selects only: no branches)
    uint64_t k = (b >> 63) ? ~b : b | 0x8000000000000000ull;
    k = mag == 0ull ? 0x8000000000000000ull : k;                 // -0.0 = +0.0, as a comparison sort has it
    k = mag > 0x7ff0000000000000ull ? ~0ull : k;                 // every NaN: the largest key
    return k;
}
__device__ __forceinline__ double sort_value(uint64_t k)
{
    if (k == ~0ull) return __longlong_as_double(0x7ff8000000000000LL);
    return __longlong_as_double((int64_t)((k >> 63) ? k ^ 0x8000000000000000ull : ~k));
}
__device__ __forceinline__ unsigned sort_digit(uint64_t k, int pass)
{
    return pass < 7 ? (unsigned)(k >> (kSortBits * pass)) & (kSortBins - 1) : (unsigned)(k >> 63);
}

// row of a global tile number: tile_start[r] <= tile < tile_start[r + 1] (rows without elements have no tiles)
__device__ __forceinline__ int sort_row_of(const int64_t* __restrict__ tile_start, int rows, int64_t tile)
{
    int lo = 0, hi = rows;                                       // tile_start[lo] <= tile < tile_start[hi]
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (tile_start[mid] <= tile) lo = mid;
        else hi = mid;
    }
    return lo;
}

// ---- a pass = three launches: (1) the digit histogram of every RANGE (sixteen consecutive tiles of a row), (2) per row: where
// every (digit, range) starts -- digit-major, range-minor: the order of the pass's output --, (3) the ranges regrouped: a
// workgroup walks the tiles of its range and keeps the running places of the 512 digits itself.  No workgroup waits for another
// one.  (First build of the round: one kernel per pass with decoupled look-back over per-tile status words -- 2.6 ms per pass of
// 266 M elements, of which 0.9 ms were the look-back: with two tiles per CU in flight the tiles whose counts are published
// but whose prefix is not yet are dozens deep, and every step back is a trip to the L2; and 0.65 ms the scatter pattern -- 512
// runs of eight elements per tile, neighbouring runs written by other CUs.  A range's runs are sixteen tiles long and written by
// one CU.  profiles/r06_sort_probes.txt.) ----
constexpr int kRangeTiles = 16;
constexpr int kRangeElems = kRangeTiles * kSortTile;
// the scatter kernel's own workgroup: 16 waves x 8 keys = tiles of 8 192, so that a digit's run out of a tile is sixteen elements
// on average with random digits (128 B of keys, 64 B of payloads: the passes over the mantissa's digits are bound by how the
// memory system takes these runs -- with runs of eight they took 2.4 ms, the passes over the concentrated upper digits 1.3 ms,
// whatever the occupancy); a range stays 65 536 elements (the histogram and scan kernels do not change)
#ifndef DIG_SORT_SCAT_WAVES
#define DIG_SORT_SCAT_WAVES 16
#endif
constexpr int kScatWaves = DIG_SORT_SCAT_WAVES, kScatBlock = kScatWaves * 64, kScatTile = kScatBlock * kSortItems;
static_assert(kRangeElems % kScatTile == 0 && kScatBlock >= kSortBins, "whole scatter tiles per range; a thread per digit");

// a barrier that orders LDS only: __syncthreads() also waits for every global load and store of the wave (its fence covers global
// memory), which would end the scatter kernel's look-ahead loads at the first barrier behind them
__device__ __forceinline__ void sort_lds_barrier()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

// wave-aggregated LDS histogram add: p-values crowd into one or two binades, the high digits of a wave's 64 keys are mostly ONE
// value, and 64 atomic adds to one counter are 64 turns of the LDS -- a wave whose digit is uniform adds its count once
__device__ __forceinline__ void sort_hist_add(unsigned* h, unsigned d, bool live, bool try_uniform)
{
    if (try_uniform) {
        const uint64_t on = __ballot(live);
        if (on) {
            const unsigned d0 = __builtin_amdgcn_readlane(d, __builtin_ctzll(on));
            const uint64_t same = __ballot(live && d == d0);
            if (same == on) {
                if ((threadIdx.x & 63) == (unsigned)__builtin_ctzll(on)) atomicAdd(&h[d0], (unsigned)__popcll(on));
                return;
            }
        }
    }
    if (live) atomicAdd(&h[d], 1u);
}

template <bool FIRST>
__global__ __launch_bounds__(kSortBlock) void sort_range_hist_kernel(const double* __restrict__ p_in, const uint64_t* __restrict__ k_in,
                                                                     const int64_t* __restrict__ row_ptr, const int64_t* __restrict__ range_start,
                                                                     int rows, int pass, unsigned* __restrict__ rhist, unsigned* __restrict__ flags, int gate)
{
    if (gate && !(*flags & 2u)) return;                          // the careful passes: only behind a fix-up that gave up
    if (pass == 7 && !(*flags & 1u)) return;                     // no negative value: bit 63 is the same everywhere
    __shared__ unsigned s_h[kSortBins];
    s_h[threadIdx.x] = 0u;
    __syncthreads();
    const int row = sort_row_of(range_start, rows, blockIdx.x);
    const int64_t lo = row_ptr[row] + ((int64_t)blockIdx.x - range_start[row]) * kRangeElems;
    const int64_t hi = lo + kRangeElems < row_ptr[row + 1] ? lo + kRangeElems : row_ptr[row + 1];
    bool neg = false;
    for (int64_t i0 = lo; i0 < hi; i0 += kSortBlock) {
        const int64_t i = i0 + threadIdx.x;
        const bool live = i < hi;
        uint64_t k = ~0ull;
        if (live) k = FIRST ? sort_key(__builtin_nontemporal_load(p_in + i)) : __builtin_nontemporal_load(k_in + i);
        neg |= live && !(k >> 63);
        sort_hist_add(s_h, sort_digit(k, pass), live, pass >= 3);
    }
    if (FIRST && __any(neg) && (threadIdx.x & 63) == 0) atomicOr(flags, 1u);
    __syncthreads();
    rhist[(int64_t)blockIdx.x * kSortBins + threadIdx.x] = s_h[threadIdx.x];
}

// one workgroup per row, thread = digit: the digit totals, their exclusive scan, then the running place of the digit over the ranges
__global__ __launch_bounds__(kSortBins) void sort_range_scan_kernel(unsigned* __restrict__ rhist, const int64_t* __restrict__ range_start, int pass,
                                                                    const unsigned* __restrict__ flags, int gate)
{
    if (gate && !(*flags & 2u)) return;
    if (pass == 7 && !(*flags & 1u)) return;
    __shared__ unsigned s[kSortBins];
    const int64_t g0 = range_start[blockIdx.x], g1 = range_start[blockIdx.x + 1];
    unsigned* h = rhist + g0 * kSortBins + threadIdx.x;
    const int64_t ng = g1 - g0;
    unsigned total = 0u;
    int64_t g = 0;
    for (; g + 8 <= ng; g += 8) {                                // (eight loads in flight: one after the other they are a memory trip each)
        unsigned v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = h[(g + j) * kSortBins];
#pragma unroll
        for (int j = 0; j < 8; ++j) total += v[j];
    }
    for (; g < ng; ++g) total += h[g * kSortBins];
    s[threadIdx.x] = total;
    __syncthreads();
    for (int d = 1; d < kSortBins; d <<= 1) {
        const unsigned v = (int)threadIdx.x >= d ? s[threadIdx.x - d] : 0u;
        __syncthreads();
        s[threadIdx.x] += v;
        __syncthreads();
    }
    unsigned run = s[threadIdx.x] - total;
    g = 0;
    for (; g + 8 <= ng; g += 8) {
        unsigned v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = h[(g + j) * kSortBins];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            h[(g + j) * kSortBins] = run;
            run += v[j];
        }
    }
    for (; g < ng; ++g) {
        const unsigned v = h[g * kSortBins];
        h[g * kSortBins] = run;
        run += v;
    }
}

// PAY: the 32-bit payloads travel with the keys (the order is wanted: dig_sort_rows, or the q-values go to their places through
// it); without, only the keys are sorted (the q-values are looked up by value afterwards: bh_lookup_kernel)
template <bool FIRST, bool PAY>
__global__ __launch_bounds__(kScatBlock, 4) void sort_range_scatter_kernel(
    const double* __restrict__ p_in, const uint64_t* __restrict__ k_in, const unsigned* __restrict__ v_in, uint64_t* __restrict__ k_out,
    unsigned* __restrict__ v_out, const int64_t* __restrict__ row_ptr, const int64_t* __restrict__ range_start, int rows, int pass,
    const unsigned* __restrict__ roff, const unsigned* __restrict__ flags, int gate)
{
    if (gate && !(*flags & 2u)) return;
    if (pass == 7 && !(*flags & 1u)) return;
    __shared__ unsigned s_cnt[kScatWaves][kSortBins];            // per-wave digit counts, then their exclusive prefix over the waves
    __shared__ unsigned s_tpre[kSortBins];                       // where a digit starts in the regrouped tile
    __shared__ int64_t s_gbase[kSortBins];                       // where the tile's elements of a digit go, minus s_tpre
    __shared__ uint64_t s_key[kScatTile];
    __shared__ unsigned s_val[PAY ? kScatTile : 1];
    __shared__ unsigned s_wsum[kSortBins / 64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int row = sort_row_of(range_start, rows, blockIdx.x);
    const int64_t r0 = row_ptr[row], n_row = row_ptr[row + 1] - r0;
    const int64_t range_e0 = ((int64_t)blockIdx.x - range_start[row]) * kRangeElems;     // first element of the range, in the row
    const int64_t range_e1 = range_e0 + kRangeElems < n_row ? range_e0 + kRangeElems : n_row;
    const int64_t whole_e1 = range_e0 + ((range_e1 - range_e0) / kScatTile) * kScatTile;  // behind the range's last whole tile
    unsigned running = tid < kSortBins ? roff[(int64_t)blockIdx.x * kSortBins + tid] : 0u;                      // where this range's elements of digit tid go, in the row
    asm volatile("" : "+v"(running) : : "memory");               // (arrived before the loop: its wait inside would end every look-ahead)
    // Element j of a tile = (wave, item, lane): j = wave * 512 + item * 64 + lane (the order ranks are taken in).  A WHOLE tile's
    // loads are issued a tile ahead and cross the barriers of the tile in front of them (the barriers order LDS only:
    // sort_lds_barrier) -- with two workgroups per CU and load -> rank -> regroup -> store in turn, a CU had loads in flight a
    // third of the time.  One scalar base per tile + one lane offset + constant distances: eight 64-bit addresses per stream
    // were spilled.  A row's last, partial tile is loaded when its turn comes.
    uint64_t nkey[kSortItems], key[kSortItems];
    unsigned nval[kSortItems], val[kSortItems], loc[kSortItems];
    const unsigned j0 = (unsigned)(wave * (64 * kSortItems) + lane);
    auto request = [&](int64_t e0) {
        const double* pb = p_in + r0 + e0;
        const uint64_t* kb = k_in + r0 + e0;
        const unsigned* vb = v_in + r0 + e0;
#pragma unroll
        for (int e = 0; e < kSortItems; ++e) {
            if (FIRST) {
                nkey[e] = (uint64_t)__double_as_longlong(__builtin_nontemporal_load(pb + j0 + e * 64));
            } else {
                nkey[e] = __builtin_nontemporal_load(kb + j0 + e * 64);
                if (PAY) nval[e] = __builtin_nontemporal_load(vb + j0 + e * 64);
            }
        }
    };
    auto take = [&](int64_t e0) {                                // the requested tile becomes the current one (waits for its loads)
#pragma unroll
        for (int e = 0; e < kSortItems; ++e) {
            key[e] = FIRST ? sort_key(__longlong_as_double((int64_t)nkey[e])) : nkey[e];
            val[e] = !PAY ? 0u : (FIRST ? (unsigned)(e0 + j0 + e * 64) : nval[e]);
            if (PAY) asm volatile("" : "+v"(key[e]), "+v"(val[e]) : : "memory");         // (here, not sunk to the next use)
            else asm volatile("" : "+v"(key[e]) : : "memory");
        }
    };
    // one tile: key / val hold it; `ahead`: the tile behind it has been requested and is taken in front of this tile's stores (the
    // wait for its loads would otherwise wait for the stores as well: one counter for both)
    auto tile = [&](int64_t e0, int n_here, bool ahead) {
        // the wave's own row of counters: nobody else touches it between the prefix of the tile before and this tile's prefix
#pragma unroll
        for (int i = 0; i < kSortBins / 64; ++i) s_cnt[wave][i * 64 + lane] = 0u;
        // ---- stable ranks inside the wave: the lanes with the same digit (nine ballots), in lane order, behind what the wave's
        // earlier items counted ----
#pragma unroll
        for (int e = 0; e < kSortItems; ++e) {
            const bool live = (int)(j0 + e * 64) < n_here;
            const unsigned d = sort_digit(key[e], pass);
            uint64_t peers = __ballot(live);
#pragma unroll
            for (int b = 0; b < kSortBits; ++b) {
                const bool bit = (d >> b) & 1u;
                const uint64_t vote = __ballot(bit);
                peers &= bit ? vote : ~vote;
            }
            const unsigned below = __builtin_amdgcn_mbcnt_hi((unsigned)(peers >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)peers, 0u));
            const unsigned old = s_cnt[wave][d];
            loc[e] = old + below;
            if (live && below == 0u) s_cnt[wave][d] = old + (unsigned)__popcll(peers);
        }
        sort_lds_barrier();
        // ---- the tile's count of digit tid: the waves' counts become their exclusive prefix (thread = digit: the first 512) ----
        unsigned h = 0u, incl = 0u;
        if (tid < kSortBins) {
#pragma unroll
            for (int w = 0; w < kScatWaves; ++w) {
                const unsigned c = s_cnt[w][tid];
                s_cnt[w][tid] = h;
                h += c;
            }
            // (exclusive scan of h over the 512 digits: inside the wave by shuffles, then over the eight waves)
            incl = h;
#pragma unroll
            for (int dd = 1; dd < 64; dd <<= 1) {
                const unsigned v = __shfl_up(incl, dd, 64);
                if (lane >= dd) incl += v;
            }
            if (lane == 63) s_wsum[wave] = incl;
        }
        sort_lds_barrier();
        if (tid < kSortBins) {
            unsigned wbase = 0u;
#pragma unroll
            for (int w = 0; w < kSortBins / 64; ++w) wbase += w < wave ? s_wsum[w] : 0u;
            const unsigned tpre = wbase + incl - h;
            s_tpre[tid] = tpre;
            s_gbase[tid] = r0 + (int64_t)running - tpre;
            running += h;
        }
        sort_lds_barrier();
        // ---- regroup the tile by digit in LDS, then every digit's run leaves in one piece ----
#pragma unroll
        for (int e = 0; e < kSortItems; ++e) {
            if ((int)(j0 + e * 64) < n_here) {
                const unsigned d = sort_digit(key[e], pass);
                const unsigned pos = s_tpre[d] + s_cnt[wave][d] + loc[e];
                s_key[pos] = key[e];
                if (PAY) s_val[pos] = val[e];
            }
        }
        if (ahead) take(e0 + kScatTile);
        sort_lds_barrier();
#pragma unroll
        for (int e = 0; e < kSortItems; ++e) {                   // (unconditional stores: a lane past the end of a row's last tile
            int pos = e * kScatBlock + tid;                      //  writes the tile's last element once more -- stores that may
            pos = pos < n_here ? pos : n_here - 1;               //  or may not happen cannot be counted past by the waits in front
            const uint64_t k = s_key[pos];                       //  of the next tile's loads)
            const int64_t dst = s_gbase[sort_digit(k, pass)] + pos;
            k_out[dst] = k;
            if (PAY) v_out[dst] = s_val[pos];
        }
    };
    if (range_e0 < whole_e1) {
        request(range_e0);
        take(range_e0);
        for (int64_t e0 = range_e0; e0 < whole_e1; e0 += kScatTile) {
            const bool ahead = e0 + kScatTile < whole_e1;
            if (ahead) request(e0 + kScatTile);
            tile(e0, kScatTile, ahead);
        }
    }
    if (whole_e1 < range_e1) {                                   // the row's last tile: a lane past the end holds the largest key, masked
        const int n_here = (int)(range_e1 - whole_e1);
#pragma unroll
        for (int e = 0; e < kSortItems; ++e) {
            const int64_t j = j0 + e * 64;
            key[e] = ~0ull;
            val[e] = 0u;
            if (j < n_here) {
                key[e] = FIRST ? sort_key(p_in[r0 + whole_e1 + j]) : k_in[r0 + whole_e1 + j];
                val[e] = !PAY ? 0u : (FIRST ? (unsigned)(whole_e1 + j) : v_in[r0 + whole_e1 + j]);
            }
        }
        tile(whole_e1, n_here, false);
    }
}

// ---- the fix-up behind FOUR passes.  Passes 3 .. 6 (and 7) sort by bits 27 .. 63 -- sign, exponent and the upper 25 bits of the
// mantissa: with p-values spread over a few binades, elements that share all 36 bits are pairs and triples (a tenth of 7.2 M
// uniform values sits in such runs) or exact ties.  A run of equal upper bits of at most kFixRun elements is put in order here
// (rank among the run by full key, then by position: a stable sort of the run), out of place; longer runs stay as they are, and
// if one of them holds two different keys the fix-up GIVES UP: flags bit 1, and the seven careful passes that follow on the
// stream (gated by that bit: 21 launches that return at once otherwise) sort the lists from the start.  Either way the sorted
// lists end in the same buffer: (k0, v0), or (k1, v1) when a negative value made the sign pass run. ----
__global__ void sort_giveup_kernel(unsigned* flags) { atomicOr(flags, 2u); }       // developer switch DIG_SORT_FORM=careful

constexpr int kFixRun = 256;                                     // longest run the fix-up sorts = the halo on either side of a tile
constexpr int kFixLowBits = 27;

template <bool PAY>
__global__ __launch_bounds__(kSortBlock) void sort_fixup_kernel(uint64_t* __restrict__ k0, uint64_t* __restrict__ k1, unsigned* __restrict__ v0,
                                                                unsigned* __restrict__ v1, const int64_t* __restrict__ row_ptr,
                                                                const int64_t* __restrict__ tile_start, int rows, unsigned* __restrict__ flags)
{
    const bool neg = *flags & 1u;                                // pass 7 ran: the four-pass order is in (k0, v0), else in (k1, v1)
    const uint64_t* ks = neg ? k0 : k1;
    const unsigned* vs = neg ? v0 : v1;
    uint64_t* kd = neg ? k1 : k0;
    unsigned* vd = neg ? v1 : v0;
    constexpr int kWin = kSortTile + 2 * kFixRun;
    __shared__ uint64_t s_k[kWin];
    const int tid = threadIdx.x;
    const int row = sort_row_of(tile_start, rows, blockIdx.x);
    const int64_t r0 = row_ptr[row], n_row = row_ptr[row + 1] - r0;
    const int64_t e0 = ((int64_t)blockIdx.x - tile_start[row]) * kSortTile;          // first element the tile owns, in the row
    const int64_t w0 = e0 - kFixRun;                             // first element of the window (may lie in front of the row)
    unsigned val[kSortItems];                                    // (the payloads travel while the keys are looked at: a load per element
    {                                                            //  inside the loop was a memory trip per element, 16 of the tile's 20 us)
        // the window's nine keys and the tile's eight payloads per thread in flight together: clamped addresses, masked afterwards
        constexpr int kPer = kWin / kSortBlock;
        static_assert(kWin % kSortBlock == 0, "whole rounds of the workgroup over the window");
        uint64_t lk[kPer];
#pragma unroll
        for (int c = 0; c < kPer; ++c) {
            int64_t i = w0 + c * kSortBlock + tid;
            i = i < 0 ? 0 : (i < n_row ? i : n_row - 1);
            lk[c] = ks[r0 + i];
        }
#pragma unroll
        for (int e = 0; e < kSortItems; ++e) {
            int64_t i = e0 + e * kSortBlock + tid;
            i = i < n_row ? i : n_row - 1;
            val[e] = PAY ? __builtin_nontemporal_load(vs + r0 + i) : 0u;
        }
#pragma unroll
        for (int c = 0; c < kPer; ++c) {
            const int64_t i = w0 + c * kSortBlock + tid;
            s_k[c * kSortBlock + tid] = (i >= 0 && i < n_row) ? lk[c] : 0ull;
        }
    }
    __shared__ unsigned short s_list[kSortTile];                 // window indices of the tile's elements that sit in a run
    __shared__ unsigned s_n;
    if (tid == 0) s_n = 0u;
    __syncthreads();
    // ---- phase 1, every element: a neighbour with the same upper bits?  No (nine in ten, as a rule): the element stays where it is.
    // Yes: its window index goes to a list (one LDS atomic per wave and step) -- the walk along the run and the rank inside it are
    // done by ALL lanes on list entries afterwards, not by the one lane in ten of a wave that meets a run. ----
#pragma unroll
    for (int e = 0; e < kSortItems; ++e) {
        const int w = kFixRun + e * kSortBlock + tid;            // window index of the element
        const int64_t i = w0 + w;
        const bool on = i < n_row;
        const uint64_t key = s_k[w], pre = key >> kFixLowBits;
        const bool in_run = on && ((i > 0 && (s_k[w - 1] >> kFixLowBits) == pre) || (i + 1 < n_row && (s_k[w + 1] >> kFixLowBits) == pre));
        if (on && !in_run) {
            kd[r0 + i] = key;
            if (PAY) vd[r0 + i] = val[e];
        }
        const uint64_t m = __ballot(in_run);
        if (m) {
            unsigned base = 0u;
            if ((tid & 63) == 0) base = atomicAdd(&s_n, (unsigned)__popcll(m));
            base = __builtin_amdgcn_readfirstlane(base);
            if (in_run) s_list[base + __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u))] = (unsigned short)w;
        }
    }
    __syncthreads();
    // ---- phase 2, the members of runs ----
    bool give_up = false;
    const int n_list = (int)s_n;
    for (int t = tid; t < n_list; t += kSortBlock) {
        const int w = s_list[t];
        const int64_t i = w0 + w;
        const uint64_t key = s_k[w], pre = key >> kFixLowBits;
        // the run of equal upper bits around the element.  The lists are in order of those bits, so the element kFixRun places
        // away tells at once whether the run is a long one (exact ties by the hundred thousand -- every tile without a mutation
        // has p = 1 -- would otherwise walk 2 x 256 places per element); a short run is walked, one or two steps as a rule
        int back = 0, ahead = 0;
        if (i > 0 && (s_k[w - 1] >> kFixLowBits) == pre) {
            if (i >= kFixRun && (s_k[w - kFixRun] >> kFixLowBits) == pre) back = kFixRun;
            else { back = 1; while (i - back - 1 >= 0 && (s_k[w - back - 1] >> kFixLowBits) == pre) ++back; }
        }
        if (i + 1 < n_row && (s_k[w + 1] >> kFixLowBits) == pre) {
            if (i + kFixRun < n_row && (s_k[w + kFixRun] >> kFixLowBits) == pre) ahead = kFixRun;
            else { ahead = 1; while (i + ahead + 1 < n_row && (s_k[w + ahead + 1] >> kFixLowBits) == pre) ++ahead; }
        }
        int64_t dst = i;
        if (back + ahead + 1 <= kFixRun) {                       // the whole run is in sight (every element of it sees the same)
            int rank = 0;
            for (int j = -back; j <= ahead; ++j) {
                const uint64_t o = s_k[w + j];
                rank += (o < key) || (o == key && j < 0);
            }
            dst = i - back + rank;
        } else if (back > 0 && s_k[w - 1] != key) {              // a long run with two different keys: not for this kernel
            give_up = true;
        }
        kd[r0 + dst] = key;
        if (PAY) vd[r0 + dst] = vs[r0 + i];                      // (the tile's payloads were read a moment ago: an L2 hit)
    }
    if (__any(give_up) && (tid & 63) == 0) atomicOr(flags, 2u);
}

// ---- Benjamini-Hochberg over the sorted keys of ragged rows ----
constexpr int kBhrBlock = 256, kBhrItems = 16, kBhrChunk = kBhrBlock * kBhrItems;

__device__ __forceinline__ double bhr_nan_min(double a, double b)      // np.minimum: NaN if either is
{
    return (a != a) ? a : ((b != b) ? b : (b < a ? b : a));
}

// reverse inclusive running minimum over a chunk of a row; MODE 0: the chunk's minimum -> chunk_min; MODE 1: q-values through
// the payload (or in sorted order when `scatter` is 0), with the chunks behind (suffix) and the ranks behind (carry).
// Keys and payloads are read lane after lane (whole 512-byte runs per wave-instruction) and change to the thread-owns-sixteen
// arrangement of the running minimum in LDS (one pad per sixteen: the lanes' reads fall into different banks); the threads'
// minima are combined by a shuffle scan per wave and one exchange of the four wave minima -- two barriers where the round-5
// kernel (sixteen strided loads per thread, a 256-wide Hillis-Steele scan) had eighteen.
template <int MODE>
__global__ __launch_bounds__(kBhrBlock) void bhr_chunk_kernel(const uint64_t* __restrict__ k0, const uint64_t* __restrict__ k1,
                                                              const unsigned* __restrict__ v0, const unsigned* __restrict__ v1,
                                                              const unsigned* __restrict__ flags, const int64_t* __restrict__ row_ptr,
                                                              const int64_t* __restrict__ chunk_start, int rows, const double* __restrict__ n_global,
                                                              const int64_t* __restrict__ rank0, const double* __restrict__ carry,
                                                              double* __restrict__ chunk_min, const double* __restrict__ suffix,
                                                              double* __restrict__ q, int scatter, unsigned* __restrict__ rec_cnt = nullptr,
                                                              unsigned short* __restrict__ rec_mask = nullptr)
{
    const bool eight = *flags & 1u;                              // the eighth pass ran: the sorted lists are in the other buffer
    const uint64_t* ks = eight ? k1 : k0;
    const unsigned* vs = eight ? v1 : v0;
    constexpr int kPad = kBhrChunk + kBhrChunk / 16;
    __shared__ uint64_t s_k[kPad];
    __shared__ unsigned s_v[MODE == 1 ? kPad : 1];
    __shared__ double s_wave[kBhrBlock / 64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int row = sort_row_of(chunk_start, rows, blockIdx.x);
    const int64_t r0 = row_ptr[row], n_row = row_ptr[row + 1] - r0;
    const int64_t c0 = ((int64_t)blockIdx.x - chunk_start[row]) * kBhrChunk;       // first element of the chunk, in the row
    const int n_here = (int)(n_row - c0 < kBhrChunk ? n_row - c0 : kBhrChunk);
    const double inf = __longlong_as_double(0x7ff0000000000000LL);
    {   // (every load issued before the first LDS write, and unconditional -- a lane past the end reads the chunk's last element:
        //  a guarded load-then-store loop is a memory round trip per element, sixteen of them in turn)
        uint64_t lk[kBhrItems];
        unsigned lv[kBhrItems];
        const uint64_t* kb = ks + r0 + c0;
        const unsigned* vb = vs + r0 + c0;
#pragma unroll
        for (int k = 0; k < kBhrItems; ++k) {
            int j = k * kBhrBlock + tid;
            j = j < n_here ? j : n_here - 1;
            lk[k] = __builtin_nontemporal_load(kb + j);
            if (MODE == 1) lv[k] = scatter ? __builtin_nontemporal_load(vb + j) : 0u;
        }
#pragma unroll
        for (int k = 0; k < kBhrItems; ++k) {
            const int j = k * kBhrBlock + tid;
            if (j < n_here) {
                s_k[j + (j >> 4)] = lk[k];
                if (MODE == 1) s_v[j + (j >> 4)] = lv[k];
            }
        }
    }
    __syncthreads();
    const double n_f = n_global[row];
    const int64_t rk = rank0[row] + c0;
    const int base = tid * kBhrItems;                            // this thread's sixteen elements of the chunk
    double v[kBhrItems];
    double vo[MODE == 2 ? kBhrItems : 1];                        // (MODE 2: the quotients themselves, next to their running minima)
    if (MODE == 0) {
        // Only the chunk's minimum is wanted here, and two IEEE divisions per element (266 M elements: 0.75 of this kernel's 0.8 ms)
        // are the cost: the quotients are first formed approximately (reciprocal of the rank + two Newton steps: relative
        // error < 1e-15), a thread's candidates for its minimum are the elements within 1e-12 of the smallest approximate
        // quotient -- one, as a rule -- and only those are divided exactly.  A thread that meets a NaN, or magnitudes where the
        // approximation is not to be trusted (subnormal quotients, overflow), divides all sixteen.
        double a[kBhrItems], pv[kBhrItems];
        double amin = inf;
        bool all_exact = false;
#pragma unroll
        for (int k = 0; k < kBhrItems; ++k) {
            const bool on = base + k < n_here;
            pv[k] = on ? sort_value(s_k[base + k + tid]) : inf;
            const double r = (double)(rk + base + k + 1);
            double y = __builtin_amdgcn_rcp(r);
            y = __builtin_fma(__builtin_fma(-r, y, 1.0), y, y);
            y = __builtin_fma(__builtin_fma(-r, y, 1.0), y, y);
            a[k] = on ? pv[k] * (n_f * y) : inf;
            all_exact |= on && !(a[k] == a[k]);
            amin = a[k] < amin ? a[k] : amin;
        }
        const double mag = amin < 0.0 ? -amin : amin;
        all_exact |= !(mag == 0.0 || (mag > 1e-280 && mag < 1e280));
        const double thr = amin + mag * 1e-12;
        double pc = inf, rc = 1.0;
        int cnt = 0;
#pragma unroll
        for (int k = 0; k < kBhrItems; ++k) {
            const bool c = base + k < n_here && a[k] <= thr;
            if (c && cnt == 0) {
                pc = pv[k];
                rc = (double)(rk + base + k + 1);
            }
            cnt += c;
        }
        double vmin;
        {
#pragma clang fp contract(off)
            vmin = pc / (rc / n_f);
        }
        if (__any(all_exact || cnt > 1)) {                       // (rare: near-ties of the quotient, or the cases above)
#pragma unroll
            for (int k = 0; k < kBhrItems; ++k) {
                if (base + k < n_here && (all_exact || (cnt > 1 && a[k] <= thr))) {
#pragma clang fp contract(off)
                    const double e = pv[k] / ((double)(rk + base + k + 1) / n_f);
                    vmin = bhr_nan_min(e, vmin);
                }
            }
        }
        v[0] = vmin;
    } else {
#pragma unroll
        for (int k = 0; k < kBhrItems; ++k) {
#pragma clang fp contract(off)
            v[k] = base + k < n_here ? sort_value(s_k[base + k + tid]) / ((double)(rk + base + k + 1) / n_f) : inf;
        }
        if (MODE == 2) {
#pragma unroll
            for (int k = 0; k < kBhrItems; ++k) vo[k] = v[k];
        }
#pragma unroll
        for (int k = kBhrItems - 2; k >= 0; --k) v[k] = bhr_nan_min(v[k], v[k + 1]);
    }
    // reverse inclusive scan of the threads' minima: down the wave by shuffles, then over the four waves
    double sc = v[0];
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const double o = __shfl_down(sc, d, 64);
        if (lane + d < 64) sc = bhr_nan_min(sc, o);
    }
    if (lane == 0) s_wave[wave] = sc;
    __syncthreads();
    double after_waves = inf;                                    // the waves behind this one
#pragma unroll
    for (int w = kBhrBlock / 64 - 1; w >= 0; --w)
        if (w > wave) after_waves = bhr_nan_min(s_wave[w], after_waves);
    if (MODE == 0) {
        if (tid == 0) chunk_min[blockIdx.x] = bhr_nan_min(sc, after_waves);
        return;
    }
    double behind = __shfl_down(sc, 1, 64);                      // the threads behind this one in the wave ...
    if (lane == 63) behind = inf;
    behind = bhr_nan_min(behind, after_waves);                   // ... the waves behind, the chunks behind, the ranks behind
    behind = bhr_nan_min(behind, suffix[blockIdx.x]);
    behind = bhr_nan_min(behind, carry[row]);
    if (MODE == 2) {
        // RECORDS of the reverse running minimum: element r with v_r < min(v_s, s > r) -- its q-value is its own quotient, and every
        // element between the record in front of it and itself shares it (equal p-values: only the last of them can be a record).
        // q is therefore a step function of p with one step per record: sixteen record bits per thread and the chunk's count leave here,
        // bh_table_kernel writes the (key, quotient) pairs, bh_lookup_kernel finds every element's q by its VALUE.
        unsigned bits = 0u;
#pragma unroll
        for (int k = 0; k < kBhrItems; ++k) {
            const double later = k + 1 < kBhrItems ? bhr_nan_min(v[k + 1 < kBhrItems ? k + 1 : k], behind) : behind;
            bits |= (base + k < n_here && vo[k] < later) ? 1u << k : 0u;
        }
        rec_mask[(int64_t)blockIdx.x * kBhrBlock + tid] = (unsigned short)bits;
        unsigned c = (unsigned)__popc(bits);
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) c += __shfl_xor(c, d, 64);
        __shared__ unsigned s_c[kBhrBlock / 64];
        if (lane == 0) s_c[wave] = c;
        __syncthreads();
        if (tid == 0) {
            unsigned t = 0u;
#pragma unroll
            for (int w = 0; w < kBhrBlock / 64; ++w) t += s_c[w];
            rec_cnt[blockIdx.x] = t;
        }
        return;
    }
#pragma unroll
    for (int k = 0; k < kBhrItems; ++k)
        if (base + k < n_here) {
            const double m = bhr_nan_min(v[k], behind);
            const double out = (m != m) ? m : (m < 1.0 ? m : 1.0);
            q[r0 + (scatter ? (int64_t)s_v[base + k + tid] : c0 + base + k)] = out;
        }
}

// suffix[c] = min of the chunk minima behind chunk c in its row (inf for the last); row_min[row] = min of all (may be NULL)
__global__ __launch_bounds__(kBhrBlock) void bhr_suffix_kernel(const double* __restrict__ chunk_min, const int64_t* __restrict__ chunk_start,
                                                               double* __restrict__ suffix, double* __restrict__ row_min)
{
    const int row = blockIdx.x;
    const int64_t c0 = chunk_start[row], n_chunks = chunk_start[row + 1] - c0;
    chunk_min += c0;
    suffix += c0;
    __shared__ double s_tot[kBhrBlock];
    const double inf = __longlong_as_double(0x7ff0000000000000LL);
    double carry = inf;
    for (int64_t hi = n_chunks; hi > 0; hi -= kBhrBlock) {
        const int64_t c = hi - kBhrBlock + threadIdx.x;
        s_tot[threadIdx.x] = c >= 0 ? chunk_min[c] : inf;
        __syncthreads();
        for (int d = 1; d < kBhrBlock; d <<= 1) {
            const double mine = s_tot[threadIdx.x];
            const double other = (int)threadIdx.x + d < kBhrBlock ? s_tot[threadIdx.x + d] : inf;
            __syncthreads();
            s_tot[threadIdx.x] = bhr_nan_min(mine, other);
            __syncthreads();
        }
        const double after = (int)threadIdx.x + 1 < kBhrBlock ? s_tot[threadIdx.x + 1] : inf;
        if (c >= 0) suffix[c] = bhr_nan_min(after, carry);
        const double all = s_tot[0];
        __syncthreads();
        carry = bhr_nan_min(carry, all);
    }
    if (row_min && threadIdx.x == 0) row_min[row] = carry;
}

// ---- q-values by VALUE: the table of records (bhr_chunk_kernel<2>) and the lookup ----
// where row r's table lies in the payload buffers a keys-only sort leaves unused (8-byte entries over the row's 4-byte slots)
__device__ __host__ __forceinline__ int64_t bh_table_start(int64_t r0) { return (r0 + 1) >> 1; }
__device__ __host__ __forceinline__ int64_t bh_table_cap(int64_t r0, int64_t len) { return ((r0 + len) >> 1) - ((r0 + 1) >> 1); }

// per row: where every chunk's records start in the row's table, the number of records, and whether they fit (flags bit 2 if not)
__global__ __launch_bounds__(kBhrBlock) void bh_count_scan_kernel(const unsigned* __restrict__ rec_cnt, const int64_t* __restrict__ chunk_start,
                                                                  const int64_t* __restrict__ row_ptr, unsigned* __restrict__ rec_off,
                                                                  unsigned* __restrict__ row_k, unsigned* __restrict__ flags)
{
    const int row = blockIdx.x;
    const int64_t c0 = chunk_start[row], n_chunks = chunk_start[row + 1] - c0;
    __shared__ unsigned s[kBhrBlock];
    unsigned carry = 0u;
    for (int64_t lo = 0; lo < n_chunks; lo += kBhrBlock) {
        const int64_t c = lo + threadIdx.x;
        const unsigned mine = c < n_chunks ? rec_cnt[c0 + c] : 0u;
        s[threadIdx.x] = mine;
        __syncthreads();
        for (int d = 1; d < kBhrBlock; d <<= 1) {
            const unsigned o = (int)threadIdx.x >= d ? s[threadIdx.x - d] : 0u;
            __syncthreads();
            s[threadIdx.x] += o;
            __syncthreads();
        }
        if (c < n_chunks) rec_off[c0 + c] = carry + s[threadIdx.x] - mine;
        const unsigned all = s[kBhrBlock - 1];
        __syncthreads();
        carry += all;
    }
    if (threadIdx.x == 0) {
        row_k[row] = carry;
        const int64_t r0 = row_ptr[row];
        if ((int64_t)carry > bh_table_cap(r0, row_ptr[row + 1] - r0)) atomicOr(flags, 4u);
    }
}

// the records of a chunk -> the row's table: ascending keys, the record's own quotient beside its key
__global__ __launch_bounds__(kBhrBlock) void bh_table_kernel(const uint64_t* __restrict__ k0, const uint64_t* __restrict__ k1,
                                                             const unsigned* __restrict__ flags, const int64_t* __restrict__ row_ptr,
                                                             const int64_t* __restrict__ chunk_start, int rows, const double* __restrict__ n_global,
                                                             const int64_t* __restrict__ rank0, const unsigned short* __restrict__ rec_mask,
                                                             const unsigned* __restrict__ rec_off, uint64_t* __restrict__ tab_k,
                                                             double* __restrict__ tab_q)
{
    const uint64_t* ks = (*flags & 1u) ? k1 : k0;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const unsigned bits = rec_mask[(int64_t)blockIdx.x * kBhrBlock + tid];
    unsigned cnt = (unsigned)__popc(bits), incl = cnt;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const unsigned o = __shfl_up(incl, d, 64);
        if (lane >= d) incl += o;
    }
    __shared__ unsigned s_w[kBhrBlock / 64];
    if (lane == 63) s_w[wave] = incl;
    __syncthreads();
    unsigned before = incl - cnt;
#pragma unroll
    for (int w = 0; w < kBhrBlock / 64; ++w) before += w < wave ? s_w[w] : 0u;
    if (!bits) return;
    const int row = sort_row_of(chunk_start, rows, blockIdx.x);
    const int64_t r0 = row_ptr[row];
    const int64_t c0 = ((int64_t)blockIdx.x - chunk_start[row]) * kBhrChunk;
    const double n_f = n_global[row];
    const int64_t rk = rank0[row] + c0;
    int64_t at = bh_table_start(r0) + rec_off[blockIdx.x] + before;
    for (int k = 0; k < kBhrItems; ++k)
        if (bits >> k & 1u) {
            const int j = tid * kBhrItems + k;
            const uint64_t key = ks[r0 + c0 + j];
            double v;
            {
#pragma clang fp contract(off)
                v = sort_value(key) / ((double)(rk + j + 1) / n_f);
            }
            tab_k[at] = key;
            tab_q[at] = v;
            ++at;
        }
}

// q of every element, in the elements' own order: the first record whose key is not below the element's key (behind the last
// record: the carry of the ranks behind the row).  256 evenly spaced keys of the row's table in LDS, the rest of the search in the
// table itself (a few thousand records as a rule: it lives in the L2); sixteen independent searches per thread.
__global__ __launch_bounds__(kBhrBlock) void bh_lookup_kernel(const double* __restrict__ p, const int64_t* __restrict__ row_ptr,
                                                              const int64_t* __restrict__ chunk_start, int rows,
                                                              const unsigned* __restrict__ row_k, const uint64_t* __restrict__ tab_k,
                                                              const double* __restrict__ tab_q, const double* __restrict__ row_min,
                                                              const double* __restrict__ carry, double* __restrict__ q)
{
    constexpr int kSamp = 256;
    __shared__ uint64_t s_samp[kSamp];
    const int tid = threadIdx.x;
    const int row = sort_row_of(chunk_start, rows, blockIdx.x);
    const int64_t r0 = row_ptr[row], n_row = row_ptr[row + 1] - r0;
    const int64_t c0 = ((int64_t)blockIdx.x - chunk_start[row]) * kBhrChunk;
    const int n_here = (int)(n_row - c0 < kBhrChunk ? n_row - c0 : kBhrChunk);
    const int64_t K = row_k[row], t0 = bh_table_start(r0);
    const int64_t stride = (K + kSamp - 1) / kSamp > 0 ? (K + kSamp - 1) / kSamp : 1;       // records per sample block
    {   // sample j = the LAST key of block j (blocks of `stride` records); blocks past the table: the largest key
        const int64_t last = (int64_t)(tid + 1) * stride - 1;
        s_samp[tid] = (int64_t)tid * stride < K ? tab_k[t0 + (last < K ? last : K - 1)] : ~0ull;
    }
    uint64_t key[kBhrItems];
    const double* pb = p + r0 + c0;
#pragma unroll
    for (int k = 0; k < kBhrItems; ++k) {
        int j = k * kBhrBlock + tid;
        j = j < n_here ? j : n_here - 1;
        key[k] = sort_key(__builtin_nontemporal_load(pb + j));
    }
    __syncthreads();
    const double cr = carry[row], rm = row_min[row];
    const bool nan_row = rm != rm || cr != cr;                   // a NaN anywhere in the list (or behind it) makes every q NaN
    int64_t lo[kBhrItems], len[kBhrItems];
#pragma unroll
    for (int k = 0; k < kBhrItems; ++k) {                        // the block: the number of samples below the key
        int pos = 0;
#pragma unroll
        for (int st = kSamp / 2; st >= 1; st >>= 1) pos += s_samp[pos + st - 1] < key[k] ? st : 0;
        pos += s_samp[pos] < key[k] ? 1 : 0;
        lo[k] = (int64_t)pos * stride;
        lo[k] = lo[k] < K ? lo[k] : K;
        const int64_t hi = lo[k] + stride < K ? lo[k] + stride : K;
        len[k] = hi - lo[k];
    }
    for (int64_t span = stride; span > 0; span >>= 1) {          // (the same number of steps for every search: ceil(log2(stride + 1)))
#pragma unroll
        for (int k = 0; k < kBhrItems; ++k) {
            const int64_t half = len[k] >> 1;
            const bool go = len[k] > 0 && tab_k[t0 + lo[k] + half] < key[k];
            lo[k] = go ? lo[k] + half + 1 : lo[k];
            len[k] = len[k] > 0 ? (go ? len[k] - half - 1 : half) : 0;
        }
    }
#pragma unroll
    for (int k = 0; k < kBhrItems; ++k) {
        const int j = k * kBhrBlock + tid;
        if (j < n_here) {
            const double m = lo[k] < K ? tab_q[t0 + lo[k]] : cr;
            const double out = nan_row ? __longlong_as_double(0x7ff8000000000000LL) : (m != m) ? m : (m < 1.0 ? m : 1.0);
            q[r0 + c0 + j] = out;
        }
    }
}

__global__ void sort_unpack_kernel(const uint64_t* __restrict__ k0, const uint64_t* __restrict__ k1, const unsigned* __restrict__ v0,
                                   const unsigned* __restrict__ v1, const unsigned* __restrict__ flags, int64_t n, double* __restrict__ p_sorted,
                                   unsigned* __restrict__ order)
{
    const bool eight = *flags & 1u;
    const uint64_t* ks = eight ? k1 : k0;
    const unsigned* vs = eight ? v1 : v0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        if (p_sorted) p_sorted[i] = sort_value(ks[i]);
        if (order) order[i] = vs[i];
    }
}

// layout of the workspace (all offsets from its 256-byte aligned start)
struct SortLayout {
    int64_t total_ranges, total_tiles, bh_chunks;
    int64_t off_k0, off_k1, off_v0, off_v1, off_rhist, off_small, off_rowptr, off_rangestart, off_tilestart, off_bhstart, off_nglob,
        off_rank0, off_carry, off_cmin, off_suffix, off_reccnt, off_recoff, off_recmask, off_rowk, off_rowmin, bytes;
};
static int64_t up256(int64_t x) { return (x + 255) & ~(int64_t)255; }
static SortLayout sort_layout(const int64_t* row_ptr, int64_t rows)
{
    SortLayout L{};
    const int64_t n = rows > 0 ? row_ptr[rows] - row_ptr[0] : 0;
    for (int64_t r = 0; r < rows; ++r) {
        const int64_t len = row_ptr[r + 1] - row_ptr[r];
        L.total_ranges += (len + kRangeElems - 1) / kRangeElems;
        L.total_tiles += (len + kSortTile - 1) / kSortTile;
        L.bh_chunks += (len + kBhrChunk - 1) / kBhrChunk;
    }
    int64_t o = 0;
    auto take = [&](int64_t bytes) {
        const int64_t at = o;
        o += up256(bytes);
        return at;
    };
    L.off_k0 = take(n * 8);
    L.off_k1 = take(n * 8);
    L.off_v0 = take(n * 4);
    L.off_v1 = take(n * 4);
    L.off_rhist = take(L.total_ranges * kSortBins * 4);
    L.off_small = take(256);                                          // flags
    L.off_rowptr = take((rows + 1) * 8);
    L.off_rangestart = take((rows + 1) * 8);
    L.off_tilestart = take((rows + 1) * 8);
    L.off_bhstart = take((rows + 1) * 8);
    L.off_nglob = take(rows * 8);
    L.off_rank0 = take(rows * 8);
    L.off_carry = take(rows * 8);
    L.off_cmin = take(L.bh_chunks * 8);
    L.off_suffix = take(L.bh_chunks * 8);
    L.off_reccnt = take(L.bh_chunks * 4);                              // (the lookup form of the q-values: records per chunk, ...)
    L.off_recoff = take(L.bh_chunks * 4);
    L.off_recmask = take(L.bh_chunks * kBhrBlock * 2);
    L.off_rowk = take(rows * 4);
    L.off_rowmin = take(rows * 8);
    L.bytes = o + 256;
    return L;
}

}  // namespace dig

using namespace dig;

namespace {

struct SortPlan {
    SortLayout L;
    char* base;
    int64_t n, rows;
};

// uploads the row tables and sorts; afterwards the sorted keys / payloads are in (k0, v0) -- or (k1, v1) when flags & 1
int sort_rows(const double* p, const int64_t* row_ptr, int64_t rows, void* workspace, int64_t workspace_bytes, hipStream_t s, SortPlan& plan,
              bool pay)
{
    plan.L = sort_layout(row_ptr, rows);
    const SortLayout& L = plan.L;
    DIG_REQUIRE(workspace && workspace_bytes >= L.bytes, "workspace of dig_bh_ragged_workspace(row_ptr, rows) bytes");
    plan.base = (char*)(((uintptr_t)workspace + 255) & ~(uintptr_t)255);
    plan.n = row_ptr[rows] - row_ptr[0];
    plan.rows = rows;
    char* b = plan.base;
    // row tables (relative to row_ptr[0] = 0 on the device side)
    std::string host((size_t)(4 * (rows + 1) * 8), '\0');
    int64_t* h = (int64_t*)host.data();
    int64_t *h_rp = h, *h_rs = h + (rows + 1), *h_bs = h + 2 * (rows + 1), *h_ts = h + 3 * (rows + 1);
    h_rs[0] = h_bs[0] = h_ts[0] = 0;
    for (int64_t r = 0; r <= rows; ++r) h_rp[r] = row_ptr[r] - row_ptr[0];
    for (int64_t r = 0; r < rows; ++r) {
        const int64_t len = row_ptr[r + 1] - row_ptr[r];
        h_rs[r + 1] = h_rs[r] + (len + kRangeElems - 1) / kRangeElems;
        h_bs[r + 1] = h_bs[r] + (len + kBhrChunk - 1) / kBhrChunk;
        h_ts[r + 1] = h_ts[r] + (len + kSortTile - 1) / kSortTile;
    }
    DIG_HIP_TRY(hipMemcpyAsync(b + L.off_tilestart, h_ts, (rows + 1) * 8, hipMemcpyHostToDevice, s));
    DIG_HIP_TRY(hipMemcpyAsync(b + L.off_rowptr, h_rp, (rows + 1) * 8, hipMemcpyHostToDevice, s));
    DIG_HIP_TRY(hipMemcpyAsync(b + L.off_rangestart, h_rs, (rows + 1) * 8, hipMemcpyHostToDevice, s));
    DIG_HIP_TRY(hipMemcpyAsync(b + L.off_bhstart, h_bs, (rows + 1) * 8, hipMemcpyHostToDevice, s));
    DIG_HIP_TRY(hipStreamSynchronize(s));            // (the host tables above go out of scope; 3 small copies)
    if (plan.n == 0) return DIG_OK;
    DIG_REQUIRE(L.total_ranges < (1ll << 31), "too many elements for one call");
    const int64_t* d_rp = (const int64_t*)(b + L.off_rowptr);
    const int64_t* d_rs = (const int64_t*)(b + L.off_rangestart);
    unsigned* rhist = (unsigned*)(b + L.off_rhist);
    unsigned* flags = (unsigned*)(b + L.off_small);
    uint64_t *k0 = (uint64_t*)(b + L.off_k0), *k1 = (uint64_t*)(b + L.off_k1);
    unsigned *v0 = (unsigned*)(b + L.off_v0), *v1 = (unsigned*)(b + L.off_v1);
    DIG_HIP_TRY(hipMemsetAsync(flags, 0, 256, s));
    const dim3 grid_r((unsigned)L.total_ranges), grid_rows((unsigned)rows);
    const int64_t* d_ts = (const int64_t*)(b + L.off_tilestart);
    // one pass: source (p or the keys / payloads of one buffer) -> the other buffer
    auto pass_from = [&](int pass, bool first, bool to0, int gate) {
        const uint64_t* ki = to0 ? k1 : k0;
        const unsigned* vi = to0 ? v1 : v0;
        if (first)
            hipLaunchKernelGGL((sort_range_hist_kernel<true>), grid_r, dim3(kSortBlock), 0, s, p, (const uint64_t*)nullptr, d_rp, d_rs, (int)rows, pass,
                               rhist, flags, gate);
        else
            hipLaunchKernelGGL((sort_range_hist_kernel<false>), grid_r, dim3(kSortBlock), 0, s, (const double*)nullptr, ki, d_rp, d_rs, (int)rows, pass,
                               rhist, flags, gate);
        hipLaunchKernelGGL(sort_range_scan_kernel, grid_rows, dim3(kSortBins), 0, s, rhist, d_rs, pass, (const unsigned*)flags, gate);
#define DIG_SCATTER(FIRSTV, PAYV, PIN, KIN, VIN)                                                                                                  \
    hipLaunchKernelGGL((sort_range_scatter_kernel<FIRSTV, PAYV>), grid_r, dim3(kScatBlock), 0, s, PIN, KIN, VIN, to0 ? k0 : k1, to0 ? v0 : v1, d_rp, \
                       d_rs, (int)rows, pass, (const unsigned*)rhist, (const unsigned*)flags, gate)
        if (first && pay) DIG_SCATTER(true, true, p, (const uint64_t*)nullptr, (const unsigned*)nullptr);
        else if (first) DIG_SCATTER(true, false, p, (const uint64_t*)nullptr, (const unsigned*)nullptr);
        else if (pay) DIG_SCATTER(false, true, (const double*)nullptr, ki, vi);
        else DIG_SCATTER(false, false, (const double*)nullptr, ki, vi);
#undef DIG_SCATTER
    };
    static const bool careful_only = getenv("DIG_SORT_FORM") && getenv("DIG_SORT_FORM")[0] == 'c';      // developer switch: seven passes always
    if (!careful_only) {
        // four passes over bits 27 .. 62 (+ the sign pass when a negative value exists): p -> 0 -> 1 -> 0 -> 1 (-> 0), then the fix-up
        // into the other buffer
        pass_from(3, true, true, 0);
        pass_from(4, false, false, 0);
        pass_from(5, false, true, 0);
        pass_from(6, false, false, 0);
        pass_from(7, false, true, 0);
        if (pay) hipLaunchKernelGGL(sort_fixup_kernel<true>, dim3((unsigned)L.total_tiles), dim3(kSortBlock), 0, s, k0, k1, v0, v1, d_rp, d_ts, (int)rows, flags);
        else hipLaunchKernelGGL(sort_fixup_kernel<false>, dim3((unsigned)L.total_tiles), dim3(kSortBlock), 0, s, k0, k1, v0, v1, d_rp, d_ts, (int)rows, flags);
    } else {
        hipLaunchKernelGGL(sort_giveup_kernel, dim3(1), dim3(1), 0, s, flags);
    }
    // the careful form: all eight passes from the start, only when the fix-up gave up (bit 1 of the flags)
    for (int pass = 0; pass < kSortPasses; ++pass) pass_from(pass, pass == 0, (pass & 1) == 0, 1);
    DIG_HIP_TRY(hipGetLastError());
    return DIG_OK;
}

}  // namespace

extern "C" {

int64_t dig_bh_ragged_workspace(const int64_t* row_ptr, int64_t rows)
{
    if (!row_ptr || rows <= 0) return 512;
    return sort_layout(row_ptr, rows).bytes + 256;
}

// Sorted values and the order (position in the row of every sorted element) of ragged rows of doubles: p_sorted / order may be NULL.
int dig_sort_rows(const double* p, const int64_t* row_ptr, int64_t rows, double* p_sorted, uint32_t* order, void* workspace,
                  int64_t workspace_bytes, void* stream)
{
    DIG_REQUIRE(rows >= 0 && (rows == 0 || row_ptr), "rows >= 0, row_ptr");
    if (rows == 0) return DIG_OK;
    for (int64_t r = 0; r < rows; ++r) DIG_REQUIRE(row_ptr[r + 1] >= row_ptr[r] && row_ptr[r + 1] - row_ptr[r] < (1ll << 30), "row lengths in [0, 2^30)");
    DIG_REQUIRE(row_ptr[rows] == row_ptr[0] || p, "p");
    SortPlan plan;
    const int rc = sort_rows(p + row_ptr[0], row_ptr, rows, workspace, workspace_bytes, (hipStream_t)stream, plan, true);
    if (rc != DIG_OK || plan.n == 0) return rc;
    char* b = plan.base;
    const SortLayout& L = plan.L;
    hipLaunchKernelGGL(sort_unpack_kernel, dim3((unsigned)grid_for(plan.n, 256)), dim3(256), 0, (hipStream_t)stream, (const uint64_t*)(b + L.off_k0),
                       (const uint64_t*)(b + L.off_k1), (const unsigned*)(b + L.off_v0), (const unsigned*)(b + L.off_v1),
                       (const unsigned*)(b + L.off_small), plan.n, p_sorted ? p_sorted + row_ptr[0] : nullptr, order ? order + row_ptr[0] : nullptr);
    DIG_HIP_TRY(hipGetLastError());
    return DIG_OK;
}

// Benjamini-Hochberg q-values of ragged rows of p-values, in place order (q[i] belongs to p[i]).  n_global / rank0 / carry (host
// arrays, may be NULL: the row is the whole list): the row is the ranks rank0 .. of a list of n_global values whose elements
// behind the row have the reverse running minimum `carry`.  row_min (device, may be NULL): the minimum of the row's
// p / (rank / n) -- what the ranks in front need as their carry.  sorted_out != 0: q leaves in sorted order instead.
//
// Two forms of the way back to the elements' places.  LOOKUP (the default): q is a step function of p with one step per record of
// the reverse running minimum, so only the KEYS are sorted (8 instead of 12 bytes per element and pass), the records go into a
// small table per row and every element finds its q by binary search with its own value -- p read and q written in order, instead
// of 266 M random 8-byte stores through a payload (4.5 of the payload form's 15.8 ms).  A few thousand records per row of 7.2 M
// p-values as a rule (null-dominated lists); a list with more records than half its length (q strictly increasing almost
// everywhere) is sent through the PAYLOAD form, which is also what DIG_BH_FORM=payload selects (developer switch, tests).
int dig_bh_qvalues_ragged(const double* p, const int64_t* row_ptr, int64_t rows, const double* n_global, const int64_t* rank0,
                          const double* carry, double* q, double* row_min, int sorted_out, void* workspace, int64_t workspace_bytes,
                          void* stream)
{
    DIG_REQUIRE(rows >= 0 && (rows == 0 || row_ptr), "rows >= 0, row_ptr");
    if (rows == 0) return DIG_OK;
    for (int64_t r = 0; r < rows; ++r) DIG_REQUIRE(row_ptr[r + 1] >= row_ptr[r] && row_ptr[r + 1] - row_ptr[r] < (1ll << 30), "row lengths in [0, 2^30)");
    DIG_REQUIRE(row_ptr[rows] == row_ptr[0] || (p && (q || row_min)), "p and q");
    static const bool payload_only = getenv("DIG_BH_FORM") && getenv("DIG_BH_FORM")[0] == 'p';
    hipStream_t s = (hipStream_t)stream;
    for (int attempt = payload_only ? 1 : 0; attempt < 2; ++attempt) {
        const bool lookup = attempt == 0 && q && !sorted_out;        // (no q, or q in sorted order: keys only anyway, no table)
        const bool pay = attempt == 1 && q && !sorted_out;
        SortPlan plan;
        const int rc = sort_rows(p + row_ptr[0], row_ptr, rows, workspace, workspace_bytes, s, plan, pay);
        if (rc != DIG_OK) return rc;
        char* b = plan.base;
        const SortLayout& L = plan.L;
        // per-row parameters
        std::string host((size_t)(3 * rows * 8), '\0');
        double* h_n = (double*)host.data();
        int64_t* h_r = (int64_t*)(host.data() + rows * 8);
        double* h_c = (double*)(host.data() + 2 * rows * 8);
        const double inf = __builtin_inf();
        for (int64_t r = 0; r < rows; ++r) {
            h_n[r] = n_global ? n_global[r] : (double)(row_ptr[r + 1] - row_ptr[r]);
            h_r[r] = rank0 ? rank0[r] : 0;
            h_c[r] = carry ? carry[r] : inf;
        }
        DIG_HIP_TRY(hipMemcpyAsync(b + L.off_nglob, h_n, rows * 8, hipMemcpyHostToDevice, s));
        DIG_HIP_TRY(hipMemcpyAsync(b + L.off_rank0, h_r, rows * 8, hipMemcpyHostToDevice, s));
        DIG_HIP_TRY(hipMemcpyAsync(b + L.off_carry, h_c, rows * 8, hipMemcpyHostToDevice, s));
        DIG_HIP_TRY(hipStreamSynchronize(s));
        if (plan.n == 0) {
            if (row_min) {
                std::string infs((size_t)(rows * 8), '\0');
                for (int64_t r = 0; r < rows; ++r) ((double*)infs.data())[r] = inf;
                DIG_HIP_TRY(hipMemcpyAsync(row_min, infs.data(), rows * 8, hipMemcpyHostToDevice, s));
                DIG_HIP_TRY(hipStreamSynchronize(s));
            }
            return DIG_OK;
        }
        const uint64_t *k0 = (const uint64_t*)(b + L.off_k0), *k1 = (const uint64_t*)(b + L.off_k1);
        const unsigned *v0 = (const unsigned*)(b + L.off_v0), *v1 = (const unsigned*)(b + L.off_v1);
        unsigned* flags = (unsigned*)(b + L.off_small);
        const int64_t *d_rp = (const int64_t*)(b + L.off_rowptr), *d_bs = (const int64_t*)(b + L.off_bhstart);
        const double *d_n = (const double*)(b + L.off_nglob), *d_c = (const double*)(b + L.off_carry);
        const int64_t* d_r = (const int64_t*)(b + L.off_rank0);
        double *cmin = (double*)(b + L.off_cmin), *suffix = (double*)(b + L.off_suffix), *rowmin_own = (double*)(b + L.off_rowmin);
        const dim3 grid_c((unsigned)L.bh_chunks), block_c(kBhrBlock);
        hipLaunchKernelGGL((bhr_chunk_kernel<0>), grid_c, block_c, 0, s, k0, k1, v0, v1, (const unsigned*)flags, d_rp, d_bs, (int)rows, d_n, d_r, d_c, cmin,
                           (const double*)nullptr, (double*)nullptr, 0, (unsigned*)nullptr, (unsigned short*)nullptr);
        hipLaunchKernelGGL(bhr_suffix_kernel, dim3((unsigned)rows), block_c, 0, s, cmin, d_bs, suffix, rowmin_own);
        if (row_min) DIG_HIP_TRY(hipMemcpyAsync(row_min, rowmin_own, rows * 8, hipMemcpyDeviceToDevice, s));
        if (!q) break;
        if (!lookup) {
            hipLaunchKernelGGL((bhr_chunk_kernel<1>), grid_c, block_c, 0, s, k0, k1, v0, v1, (const unsigned*)flags, d_rp, d_bs, (int)rows, d_n, d_r, d_c,
                               (double*)nullptr, (const double*)suffix, q + row_ptr[0], sorted_out ? 0 : 1, (unsigned*)nullptr, (unsigned short*)nullptr);
            break;
        }
        unsigned *rec_cnt = (unsigned*)(b + L.off_reccnt), *rec_off = (unsigned*)(b + L.off_recoff), *row_k = (unsigned*)(b + L.off_rowk);
        unsigned short* rec_mask = (unsigned short*)(b + L.off_recmask);
        hipLaunchKernelGGL((bhr_chunk_kernel<2>), grid_c, block_c, 0, s, k0, k1, v0, v1, (const unsigned*)flags, d_rp, d_bs, (int)rows, d_n, d_r, d_c,
                           (double*)nullptr, (const double*)suffix, (double*)nullptr, 0, rec_cnt, rec_mask);
        hipLaunchKernelGGL(bh_count_scan_kernel, dim3((unsigned)rows), block_c, 0, s, (const unsigned*)rec_cnt, d_bs, d_rp, rec_off, row_k, flags);
        unsigned h_flags = 0u;
        DIG_HIP_TRY(hipMemcpyAsync(&h_flags, flags, 4, hipMemcpyDeviceToHost, s));
        DIG_HIP_TRY(hipStreamSynchronize(s));
        if (h_flags & 4u) continue;                                 // more records than the table holds: the payload form
        // the tables live in the payload buffers (unused by a keys-only sort): keys in v0's, quotients in v1's
        uint64_t* tab_k = (uint64_t*)(b + L.off_v0);
        double* tab_q = (double*)(b + L.off_v1);
        hipLaunchKernelGGL(bh_table_kernel, grid_c, block_c, 0, s, k0, k1, (const unsigned*)flags, d_rp, d_bs, (int)rows, d_n, d_r,
                           (const unsigned short*)rec_mask, (const unsigned*)rec_off, tab_k, tab_q);
        hipLaunchKernelGGL(bh_lookup_kernel, grid_c, block_c, 0, s, p + row_ptr[0], d_rp, d_bs, (int)rows, (const unsigned*)row_k,
                           (const uint64_t*)tab_k, (const double*)tab_q, (const double*)rowmin_own, d_c, q + row_ptr[0]);
        break;
    }
    DIG_HIP_TRY(hipGetLastError());
    return DIG_OK;
}

}  // extern "C"
