// dig_tiles_rows.hip -- the row walk: tile probabilities for (2 U + 1)-base contexts on gfx950, round 6.
//
// Reference: sequence_tools.py:292-317 (base_probabilities_by_region) + nb_model.py:126-186 (apply_nb_to_region), DEFAULT
// signature n_up = n_down = 2 (nb_model.py:126,188): pt[c][tile] = sum over the tile's positions of S[c][context(position)],
// divided by the same sum over the region; a window with a non-ACGT base counts 0.  1 024 contexts: a gather-sum, 13.3 G table
// reads per 36 000 bins x 37 cohorts, all of them out of the LDS.  What the LDS can give: 256 bytes per clock and CU, a row
// narrower than 256 bytes shares a clock with other rows only when their banks differ.
//
// Round 4's kernel (base_tile_probs_ctx_kernel, dig_tiles.hip; still the general fallback) kept a table of 8 cohorts (64-byte
// rows), 8 lanes x 8 bytes per walker, 16-bit codes formed per region AND pass in a phase of their own, five passes for 37
// cohorts and six barriers per region and pass: 16 k cycles per region and pass of which the LDS was busy 6.3 k.  This one:
//   * a pass is SIXTEEN cohorts: rows of 128 bytes, a walker = 8 lanes, a lane = TWO cohorts of the row (ds_read_b128, two
//     v_add_f64 per read); the lanes of a walker are the lanes the hardware serves together (a b128 read is served in four
//     groups of sixteen lanes {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31}, ...: a group = two whole walkers = two rows, which
//     collide only when their row numbers have the same parity: 1.5 LDS clocks per group where four 64-byte rows took 2.1);
//     37 cohorts = 16 + 16 + 5, the last pass with 64-byte rows and 4-lane walkers; one launch per pass;
//   * no codes in LDS: the region's bases are staged ONCE per region and pass as 2-bit bases, leftmost base highest
//     (16 bases per dword) + one flag bit per non-ACGT base; a walker reads 16 bytes per trip of TP positions and every
//     context is ONE v_bfe_u32 of the 32-bit window (the row number IS the field: the table is staged in that digit order),
//     the row address one v_lshl_add_u32;
//   * tile sums stay in registers until the tile is done, wait in LDS ([tile][cohort], padded stride) for the region total,
//     which every output wave forms for its own cohort (no totals phase): TWO barriers per region and pass;
//   * 13 waves per workgroup for 200 tiles (104 walkers: two full rounds) -- the host picks the wave count by the tile count;
//   * regions the LDS budget does not cover (more than kRwMaxPos positions or more tiles than the sum buffer holds: binsize 1)
//     are marked n_valid = -2 and done by the general kernel in a launch behind this one (only_deferred).
// Sum order: a tile's positions in order (as the general kernel); the region total = the tile sums added lane-strided, then a
// butterfly: pt differs from the general kernel's by an ulp of the total (1e-16 relative), from the reference's
// normalise-then-sum by a few ulp (tests: 1e-12).
#include <stdlib.h>

#include <type_traits>

#include "dig_common.hpp"

namespace dig {

#ifdef DIG_TM_TIMING                            // developer build: cycles per phase (first and last wave of every workgroup), tools/penta_bench.py
__device__ unsigned long long g_rw_prof[8];
#define RW_MARK(k) do { if (lane == 0 && (wave == 0 || wave == n_waves - 1)) { const unsigned long long now_ = __builtin_readcyclecounter(); rw_acc[k] += now_ - rw_last; rw_last = now_; } } while (0)
#else
#define RW_MARK(k) do {} while (0)
#endif

constexpr int kRwMaxPos = 10240;                       // positions of a region whose bases are staged
constexpr int kRwEntries = (kRwMaxPos + 7 + 4 + 15) / 16 + 3;       // 16-base entries {bases, flags}; + 2 read past the end (a window = three words), + 1 spare
constexpr int kRwLds = 163840;                         // bytes of LDS per CU

struct RwRegion {                                      // wave-uniform description of a region
    int64_t first, n_pos, g0, tiles_valid, w0;
    int ne, sh0;
    bool deferred;
};

template <int U>
__device__ __forceinline__ RwRegion rw_region(int chrom, int64_t start, int64_t end, int64_t len, int64_t off, int binsize, unsigned bin_magic,
                                              int cap_tiles)
{
    RwRegion q;
    q.first = start == 0 ? U : start;
    const int64_t stop = end < len - U ? end : len - U;
    q.n_pos = stop > q.first ? stop - q.first : 0;
    q.g0 = off + q.first;
    // tiles of the region, without a 64-bit division (a hundred and fifty scalar instructions per wave and region on the CU's one
    // scalar unit): bin_magic = ceil(2^32 / binsize), exact while (n_pos + binsize) binsize < 2^32; only regions that fit matter
    q.tiles_valid = 0;
    if (q.n_pos > 0 && q.n_pos <= kRwMaxPos) {
        const unsigned n = (unsigned)q.n_pos + (unsigned)binsize - 1u;
        q.tiles_valid = binsize == 1 ? (int64_t)q.n_pos : (binsize > 32768 ? 1 : (int64_t)__umulhi(n, bin_magic));
    }
    q.deferred = q.n_pos > kRwMaxPos || q.tiles_valid > cap_tiles;
    const int64_t ga0 = q.g0 - U;                      // leftmost base of the first window
    q.w0 = (ga0 >> 3) + 1;                             // array word = genome word + 1 (leading pad word)
    q.sh0 = (int)(ga0 & 7);
    q.ne = (q.n_pos > 0 && !q.deferred) ? (int)((q.sh0 + q.n_pos + 2 * U - 1) >> 4) + 3 : 0;
    return q;
}

// sixteen bases (two packed words, nibble k of a word = base k) -> {2-bit bases, base k at bits 31 - 2 k .. 30 - 2 k with the two
// bits of a base exchanged (v_bfrev); flags: bit 31 - k set when base k is not A, C, G or T}
__device__ __forceinline__ uint2 rw_entry(uint32_t wa, uint32_t wb)
{
    auto squeeze = [](uint32_t w) {
        uint32_t x = w & 0x33333333u;
        x = (x | (x >> 2)) & 0x0F0F0F0Fu;
        x = (x | (x >> 4)) & 0x00FF00FFu;
        return (x | (x >> 8)) & 0xFFFFu;
    };
    auto flags = [](uint32_t w) {
        uint32_t f = ((w >> 2) | (w >> 3)) & 0x11111111u;
        f = (f | (f >> 3)) & 0x03030303u;
        f = (f | (f >> 6)) & 0x000F000Fu;
        return (f | (f >> 12)) & 0xFFu;
    };
    const uint32_t z = squeeze(wa) | (squeeze(wb) << 16);
    uint32_t f = 0u;
    if ((wa | wb) & 0xCCCCCCCCu) f = __builtin_bitreverse32(flags(wa) | (flags(wb) << 8));       // (rare: most words skip it)
    return make_uint2(__builtin_bitreverse32(z), f);
}

// sum of a value over the wave, the same in every lane: quads, half rows and rows by DPP (xor 1, xor 2, mirror of 8, mirror of
// 16), then the four row sums by v_readlane, added first to last
__device__ __forceinline__ double rw_wave_sum(double v)
{
    auto dpp_add = [](double x, auto ctrl) {
        constexpr int kCtrl = decltype(ctrl)::value;
        const long long b = __double_as_longlong(x);
        const int lo = __builtin_amdgcn_update_dpp(0, (int)b, kCtrl, 0xf, 0xf, false);
        const int hi = __builtin_amdgcn_update_dpp(0, (int)(b >> 32), kCtrl, 0xf, 0xf, false);
        return x + __longlong_as_double(((long long)hi << 32) | (unsigned)lo);
    };
    v = dpp_add(v, std::integral_constant<int, 0xB1>{});        // quad_perm [1, 0, 3, 2]
    v = dpp_add(v, std::integral_constant<int, 0x4E>{});        // quad_perm [2, 3, 0, 1]
    v = dpp_add(v, std::integral_constant<int, 0x141>{});       // row_half_mirror
    v = dpp_add(v, std::integral_constant<int, 0x140>{});       // row_mirror
    const long long b = __double_as_longlong(v);
    double rows[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int lo = __builtin_amdgcn_readlane((int)b, 16 * k), hi = __builtin_amdgcn_readlane((int)(b >> 32), 16 * k);
        rows[k] = __longlong_as_double(((long long)hi << 32) | (unsigned)lo);
    }
    return ((rows[0] + rows[1]) + rows[2]) + rows[3];
}

// lane -> slot: the sixteen lanes one clock of a ds_read_b128 serves get sixteen consecutive slots (two 8-lane walkers)
__device__ __forceinline__ int rw_slot(int lane, bool permute)
{
    if (!permute) return lane;
    const int l = lane & 31;
    int q;
    if (l < 4) q = l;
    else if (l < 12) q = l + 12;
    else if (l < 16) q = l - 8;
    else if (l < 20) q = l + 8;
    else if (l < 28) q = l - 12;
    else q = l;
    return q | (lane & 32);
}

// U: bases on either side; LW: lanes of a walker (cohorts of the pass = 2 LW); TP: positions per trip
template <int U, int LW, int TP, bool PERMUTE>
__global__ __launch_bounds__(1024) void base_tile_probs_rows_kernel(
    const uint32_t* __restrict__ words, int64_t n_words, const int64_t* __restrict__ chrom_off,
    const int64_t* __restrict__ chrom_len, const int32_t* __restrict__ reg_chrom, const int64_t* __restrict__ reg_start,
    const int64_t* __restrict__ reg_end, int64_t R, const double* __restrict__ s_prob, int c0, int cc, int binsize, unsigned bin_magic,
    int64_t n_tiles, double* __restrict__ pt, int64_t* __restrict__ first_pos, int32_t* __restrict__ n_valid, int write_meta)
{
    constexpr int W = 2 * U + 1;                       // window
    constexpr int K = 1 << (2 * W);                    // contexts
    constexpr int NC = 2 * LW;                         // cohorts of the pass
    constexpr int SS = NC + 1;                         // doubles per tile in the sum buffer (odd: a lane = a tile reads conflict-free)
    constexpr int kTabBytes = (K + 1) * NC * 8, kEntBytes = kRwEntries * 8;
    constexpr int kCapTiles = (kRwLds - 64 - (1 << (2 * 5)) * 128 - 128 - kEntBytes) / (17 * 8);      // the cap of the 16-cohort pass, for every pass
    constexpr int kSumDoubles = kCapTiles * SS;
    static_assert(kTabBytes + kEntBytes + kSumDoubles * 8 + 64 <= kRwLds, "LDS budget");
    static_assert(TP + 2 * U <= 16 || (TP > 12 && TP + 2 * U <= 32), "a trip's windows come out of a 16-base or a 32-base window");
    static_assert(kCapTiles <= 256, "the output phase holds four tile sums per lane");
    // ONE block of LDS, the table at address 0: a row address is then (window >> k) & mask | lane offset -- two instructions per
    // position; with the table anywhere else it is three
    constexpr int kTabDoubles = (K + 1) * NC;
    __shared__ alignas(16) double s_all[kTabDoubles + kSumDoubles + kRwEntries + 2];
    double* const s_tab = s_all;                       // row = context (digit order of the staged bases), row K: zeros
    double* const s_sum = s_all + kTabDoubles;
    uint2* const s_ent = reinterpret_cast<uint2*>(s_all + kTabDoubles + kSumDoubles);
    unsigned* const s_hasn = reinterpret_cast<unsigned*>(s_all + kTabDoubles + kSumDoubles + kRwEntries);      // [2]
    unsigned& s_ticket = s_hasn[2];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, n_waves = blockDim.x >> 6;
    const int slot = rw_slot(lane, PERMUTE), col = slot % LW;
    const double nan = __longlong_as_double(0x7ff8000000000000LL);
    const int64_t G = gridDim.x;

    // ---- the table of the pass: s_prob[c0 + co][ref] -> row (ref with the two bits of every base exchanged), column co.
    // A thread takes four consecutive ref of one cohort (32 contiguous bytes); sixteen consecutive lanes write one row's cohorts.
    for (int idx = tid; idx < K * NC / 4; idx += blockDim.x) {
        const int co = idx % NC, i0 = (idx / NC) * 4;
        double v[4] = {0.0, 0.0, 0.0, 0.0};
        if (co < cc) {
            const double* src = s_prob + (int64_t)(c0 + co) * K + i0;
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = src[j];
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int i = i0 + j, row = ((i & 0x155) << 1) | ((i >> 1) & 0x155);
            s_tab[row * NC + co] = v[j];
        }
    }
    for (int idx = tid; idx < NC; idx += blockDim.x) s_tab[K * NC + idx] = 0.0;
    if (tid == 0) s_hasn[0] = 0u, s_hasn[1] = 0u, s_ticket = (unsigned)n_waves;
    __syncthreads();

    // ---- region pipeline: description two regions ahead, packed words one region ahead (in registers during the walk)
    auto load_raw = [&](int64_t r, int& chrom, int64_t& start, int64_t& end) {
        chrom = 0, start = 0, end = 0;
        if (r < R) {
            chrom = reg_chrom[r];
            start = reg_start[r];
            end = reg_end[r];
        }
    };
    // (a thread stages entries tid and tid + blockDim.x: 643 entries at most, 384 threads at least)
    uint32_t wq[4] = {0u, 0u, 0u, 0u};
    auto request = [&](const RwRegion& q) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int e = tid + j * (int)blockDim.x;
            wq[2 * j] = 0u, wq[2 * j + 1] = 0u;
            if (e < q.ne) {
                const int64_t i = q.w0 + 2 * (int64_t)e;
                wq[2 * j] = words[i < n_words ? i : n_words - 1];
                wq[2 * j + 1] = words[i + 1 < n_words ? i + 1 : n_words - 1];
            }
        }
    };
    auto put_entries = [&](const RwRegion& q, int flag_slot) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int e = tid + j * (int)blockDim.x;
            if (e < q.ne) {
                const uint2 ent = rw_entry(wq[2 * j], wq[2 * j + 1]);
                s_ent[e] = ent;
                if (ent.y) s_hasn[flag_slot] = 1u;     // the region holds a base that is not A, C, G or T: its walk reads the flags
            }
        }
    };
    int chrom1;
    int64_t start1, end1;
    RwRegion nxt{};
    load_raw(blockIdx.x, chrom1, start1, end1);
    if ((int64_t)blockIdx.x < R) {
        nxt = rw_region<U>(chrom1, start1, end1, chrom_len[chrom1], chrom_off[chrom1], binsize, bin_magic, kCapTiles);
        request(nxt);
        put_entries(nxt, 0);
    }
    load_raw(blockIdx.x + G, chrom1, start1, end1);
    RwRegion cur = nxt;
    if ((int64_t)blockIdx.x + G < R) {
        nxt = rw_region<U>(chrom1, start1, end1, chrom_len[chrom1], chrom_off[chrom1], binsize, bin_magic, kCapTiles);
        request(nxt);
    }
    load_raw(blockIdx.x + 2 * G, chrom1, start1, end1);
    __syncthreads();                                    // table and the first region's bases are in LDS

#ifdef DIG_TM_TIMING
    unsigned long long rw_acc[4] = {0, 0, 0, 0}, rw_last = __builtin_readcyclecounter();
#endif
    int rk = 0;                                         // regions this workgroup has walked
    for (int64_t r = blockIdx.x; r < R; r += G, ++rk) {
        const RwRegion q = cur;
        const int tiles_valid = q.deferred ? 0 : (int)q.tiles_valid, n_pos = q.deferred ? 0 : (int)q.n_pos;
        const int tiles = (int)(tiles_valid < n_tiles ? tiles_valid : n_tiles);       // tiles written with values
        if (tid == 0 && write_meta) {
            if (q.deferred) {
                n_valid[r] = -2;                        // the general kernel's launch behind this one takes the region
            } else {
                first_pos[r] = q.first;
                n_valid[r] = tiles;
            }
        }
        // (requested now, used after the walk: the chromosome of the region after the next)
        int64_t len2 = 0, off2 = 0;
        if (r + 2 * G < R) {
            len2 = chrom_len[chrom1];
            off2 = chrom_off[chrom1];
        }
        // ---- the walk.  A wave takes 64 / LW consecutive tiles per ticket: its first ticket is its own number, the others come
        // from a counter in LDS (asked for a round ahead) -- the LDS serves the oldest wave first, so with a fixed deal the
        // youngest waves finish last and alone, at the rate of one wave.  A trip = TP positions out of one 32-bit window. ----
        if (tid == 0) s_hasn[(rk + 1) & 1] = 0u;       // (the flag of the region staged behind this walk)
        auto walk = [&](auto hasn_c) {
            constexpr bool HASN = decltype(hasn_c)::value;
            constexpr int kPerTicket = 64 / LW;
            constexpr int kRowLog = LW == 8 ? 7 : (LW == 4 ? 6 : 5);           // log2 of the bytes of a table row
            const int n_trips = (binsize + TP - 1) / TP;                        // of a full tile (wave-uniform)
            const int b_last = (q.ne - 3) << 4;                                 // (windows are read from entries b >> 4 and the two behind it)
            int ticket = wave;
            while (ticket * kPerTicket < tiles_valid) {
                int next = 0;
                if (lane == 0) next = (int)atomicAdd(&s_ticket, 1u);
                const int t = ticket * kPerTicket + slot / LW;
                const bool mine = t < tiles_valid;
                const int p0 = t * binsize;
                // positions of the tile; a walker without a tile in this ticket rides along: its address mask is 0 and its column
                // stands in the zero row, whatever window it decodes, and it never sends its wave to the careful branch
                int rem = mine ? (binsize < n_pos - p0 ? binsize : n_pos - p0) : (1 << 30);
                const uint32_t amask = mine ? (uint32_t)(K - 1) << kRowLog : 0u;
                const uint32_t abase = (mine ? 0u : (uint32_t)K << kRowLog) | ((uint32_t)col << 4);
                int b = mine ? q.sh0 + p0 : 0;                                  // staged base index of the trip's first window
                double acc0 = 0.0, acc1 = 0.0;
                // A trip's window is read a trip ahead (it returns in front of the row reads issued behind it); two register sets
                // take turns, so that no copy -- and no wait -- stands between the read and its use a trip later.
                // Trips of up to 12 positions come out of a 32-bit window (two staged words), trips of 13 .. 28 out of a 64-bit
                // window (three words): a tile of 50 positions is two trips of 25 instead of five of 10 -- the ~10 instructions around
                // a trip (window, shifts, the clean test, the loop) are a fifth of a 10-position trip's instructions.
                constexpr bool WIDE = TP > 12;
                constexpr int NZ = WIDE ? 3 : 2;                                // staged words per window
                constexpr int BATCH = WIDE ? 9 : TP;                            // rows in flight at once (36 registers)
                auto window = [&](int bb, uint32_t (&z)[6]) {
                    const int m = (bb < b_last ? bb : b_last) >> 4;
#pragma unroll
                    for (int k = 0; k < NZ; ++k) z[k] = s_ent[m + k].x;
                    if (HASN) {
#pragma unroll
                        for (int k = 0; k < NZ; ++k) z[3 + k] = s_ent[m + k].y;
                    }
                };
                auto read_row = [&](uint32_t x) {                             // x: the field of a position at the row bits
                    uint32_t off;                                               // (the compiler turns the | of disjoint bits into an add
                    asm("v_and_or_b32 %0, %1, %2, %3" : "=v"(off) : "v"(x), "v"(amask), "v"(abase));       //  and then has no fused form)
#if defined(DIG_RW_ABL) && DIG_RW_ABL == 1         // timing builds: every walker reads row 0 / paired walkers rows of different parity / of the same parity
                    off = abase;
#elif defined(DIG_RW_ABL) && DIG_RW_ABL == 2
                    off = (off & ~(1u << kRowLog)) | (((unsigned)(slot / LW) & 1u) << kRowLog);
#elif defined(DIG_RW_ABL) && DIG_RW_ABL == 3
                    off = off & ~(1u << kRowLog);
#endif
                    return *reinterpret_cast<const double2*>(reinterpret_cast<const char*>(s_all) + off);
                };
                // the row of position i of a trip whose bases sit at bits 63 - 2 j of (hi, lo) (32-bit windows: hi only)
                auto row_of = [&](uint32_t hi, uint32_t lo, int i) {
                    const int pos = (WIDE ? 64 : 32) - 2 * i - 2 * W - kRowLog;     // the field starts at bit pos + kRowLog of the window
                    uint32_t x;
                    if (!WIDE) x = pos >= 0 ? hi >> (pos >= 0 ? pos : 0) : hi << (pos < 0 ? -pos : 0);
                    else if (pos >= 32) x = hi >> (pos - 32 >= 0 ? pos - 32 : 0);
                    else if (pos >= 0) x = __builtin_amdgcn_alignbit(hi, lo, pos >= 0 && pos < 32 ? pos : 0);
                    else x = lo << (pos < 0 ? -pos : 0);
                    return read_row(x);
                };
                auto trip = [&](const uint32_t (&z)[6], uint32_t (&zn)[6]) {
                    const int o = b & 15;
                    const uint32_t hi = (uint32_t)(((((uint64_t)z[0] << 32) | z[1]) << (2 * o)) >> 32);       // base j of the trip at bits 31 - 2 j
                    const uint32_t lo = WIDE ? (uint32_t)(((((uint64_t)z[1] << 32) | z[2]) << (2 * o)) >> 32) : 0u;      // ... bases 16 .. 31
                    uint64_t fw = 0ull;                                         // flag of base j at bit 63 - j
                    if (HASN) fw = ((((uint64_t)(z[3] | (z[4] >> 16))) << 32) | (WIDE ? (uint64_t)z[5] : 0ull)) << o;
                    b += TP;
                    window(b, zn);                      // (also behind the last trip: a read that depends on nothing costs less than a branch around it)
                    const bool clean = (fw >> (64 - TP - 2 * U)) == 0ull && rem >= TP;
                    if (__all(clean)) {
#pragma unroll
                        for (int i0 = 0; i0 < TP; i0 += BATCH) {
                            double2 v[BATCH];
#pragma unroll
                            for (int i = 0; i < BATCH; ++i)
                                if (i0 + i < TP) v[i] = row_of(hi, lo, i0 + i);
#pragma unroll
                            for (int i = 0; i < BATCH; ++i)
                                if (i0 + i < TP) {
                                    acc0 += v[i].x;
                                    acc1 += v[i].y;
                                }
                        }
                    } else {                            // a short last trip or a window with a non-ACGT base: position by position,
                        uint64_t w2 = ((uint64_t)hi << 32) | lo, f2 = fw;     // what does not count adds nothing
#pragma unroll 1
                        for (int i = 0; i < TP; ++i, w2 <<= 2, f2 <<= 1) {
                            if (i < rem && (f2 >> (64 - W)) == 0ull) {
                                const double2 v = read_row((uint32_t)(w2 >> (64 - 2 * W - kRowLog)));
                                acc0 += v.x;
                                acc1 += v.y;
                            }
                        }
                    }
                    rem -= TP;
                };
                uint32_t za[6] = {0u, 0u, 0u, 0u, 0u, 0u}, zb[6] = {0u, 0u, 0u, 0u, 0u, 0u};
                window(b, za);
                for (int k = 0; k < n_trips; k += 2) {
                    trip(za, zb);
                    if (k + 1 >= n_trips) break;
                    trip(zb, za);
                }
                if (mine) {
                    s_sum[t * SS + 2 * col] = acc0;
                    s_sum[t * SS + 2 * col + 1] = acc1;
                }
                ticket = __builtin_amdgcn_readfirstlane(next);
            }
        };
        if (s_hasn[rk & 1]) walk(std::true_type{});
        else walk(std::false_type{});
        RW_MARK(0);
        __syncthreads();                                // sums complete; nobody reads the staged bases any more
        RW_MARK(1);
        // ---- the next region's bases -> LDS; the words of the region after it are requested (they travel during the output
        // phase and the next walk) ----
        cur = nxt;
        if (tid == 0) s_ticket = (unsigned)n_waves;     // (tickets 0 .. n_waves - 1 are the waves' own)
        put_entries(cur, (rk + 1) & 1);
        if (r + 2 * G < R) {
            nxt = rw_region<U>(chrom1, start1, end1, len2, off2, binsize, bin_magic, kCapTiles);
            request(nxt);
        }
        load_raw(r + 3 * G, chrom1, start1, end1);
        // ---- pt = sum / total: a wave takes a cohort plane, a lane = a tile ----
        // (a wave takes a cohort: its total by a lane-strided sum and a DPP reduction -- a butterfly of ds_bpermute would be six
        //  dependent LDS round trips --, the quotient as a multiplication by 1 / total, as in the trinucleotide matrix kernels)
        // (a lane's four sums -- tiles lane, lane + 64, ... : the sum buffer holds at most 202 -- are read once, all in flight)
#if defined(DIG_RW_OUT_ABL) && DIG_RW_OUT_ABL == 2
        if (false) {
#else
        if (!q.deferred) {
#endif
            const bool whole = tiles_valid == (int)n_tiles;     // (the usual case: every tile asked for exists and none beyond)
            for (int co = wave; co < cc; co += n_waves) {
                double sv[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int t = lane + 64 * k;
                    sv[k] = s_sum[(t < tiles_valid ? t : 0) * SS + co];
                }
                double part = 0.0;
#pragma unroll
                for (int k = 0; k < 4; ++k) part += lane + 64 * k < tiles_valid ? sv[k] : 0.0;
                const double inv = 1.0 / rw_wave_sum(part);
                double* plane = pt + ((int64_t)(c0 + co) * R + r) * n_tiles + lane;
#if defined(DIG_RW_OUT_ABL) && DIG_RW_OUT_ABL == 1         // timing builds: the output phase without its stores / without the whole phase
                if (sv[0] * inv == 12345.678) plane[0] = inv;
                continue;
#endif
                if (whole) {
#pragma unroll
                    for (int k = 0; k < 4; ++k)
                        if (lane + 64 * k < tiles_valid) plane[64 * k] = sv[k] * inv;
                } else {
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const int t = lane + 64 * k;
                        if (t < n_tiles) plane[64 * k] = t < tiles ? sv[k] * inv : nan;
                    }
                    for (int64_t t = lane + 256; t < n_tiles; t += 64) plane[t - lane] = nan;
                }
            }
        }
        RW_MARK(2);
        __syncthreads();                                // the sum buffer is free, the next region's bases are in LDS
        RW_MARK(3);
    }
#ifdef DIG_TM_TIMING
    if (lane == 0 && (wave == 0 || wave == n_waves - 1))
        for (int k = 0; k < 4; ++k) atomicAdd(&g_rw_prof[k + (wave == 0 ? 0 : 4)], rw_acc[k]);
#endif
}

template <int U, int LW, int TP>
static void launch_rows(bool permute, int grid, int block, hipStream_t stream, const uint32_t* words, int64_t n_words,
                        const int64_t* chrom_off, const int64_t* chrom_len, const int32_t* reg_chrom, const int64_t* reg_start,
                        const int64_t* reg_end, int64_t R, const double* s_prob, int c0, int cc, int binsize, int64_t n_tiles, double* pt,
                        int64_t* first_pos, int32_t* n_valid, int write_meta)
{
    const unsigned bin_magic = binsize >= 2 ? (unsigned)(((1ull << 32) + (unsigned)binsize - 1) / (unsigned)binsize) : 0u;
    if (permute)
        hipLaunchKernelGGL((base_tile_probs_rows_kernel<U, LW, TP, true>), dim3(grid), dim3(block), 0, stream, words, n_words, chrom_off,
                           chrom_len, reg_chrom, reg_start, reg_end, R, s_prob, c0, cc, binsize, bin_magic, n_tiles, pt, first_pos, n_valid, write_meta);
    else
        hipLaunchKernelGGL((base_tile_probs_rows_kernel<U, LW, TP, false>), dim3(grid), dim3(block), 0, stream, words, n_words, chrom_off,
                           chrom_len, reg_chrom, reg_start, reg_end, R, s_prob, c0, cc, binsize, bin_magic, n_tiles, pt, first_pos, n_valid, write_meta);
}

// The passes of one call: sixteen cohorts while more than eight are left, then one pass of eight (4-lane walkers) or four
// (2-lane walkers).  Every pass is a launch of its own (its table is staged once per workgroup).
int launch_tile_probs_rows(const uint32_t* words, int64_t n_words, const int64_t* chrom_off, const int64_t* chrom_len,
                           const int32_t* reg_chrom, const int64_t* reg_start, const int64_t* reg_end, int64_t R, const double* s_prob,
                           int64_t C, int n_up, int binsize, int64_t n_tiles, double* pt, int64_t* first_pos, int32_t* n_valid,
                           hipStream_t stream)
{
    static const bool permute = !(getenv("DIG_ROWS_PERMUTE") && getenv("DIG_ROWS_PERMUTE")[0] == '0');      // developer switch (A/B)
    static const int forced_waves = getenv("DIG_ROWS_WAVES") ? atoi(getenv("DIG_ROWS_WAVES")) : 0;
    // positions per trip: the one that wastes the fewest slots of a tile's last trip (ties: the longer trip)
    int tp = 12;
    {
        int best = -1;
        const int cand[4] = {25, 12, 10, 8};
        for (int k = 0; k < 4; ++k) {
            const int waste = (binsize + cand[k] - 1) / cand[k] * cand[k] - binsize;
            if (best < 0 || waste < best) best = waste, tp = cand[k];
        }
    }
    static const int forced_tp = getenv("DIG_ROWS_TP") ? atoi(getenv("DIG_ROWS_TP")) : 0;      // developer switch (A/B)
    if (forced_tp == 25 || forced_tp == 12 || forced_tp == 10 || forced_tp == 8) tp = forced_tp;
    const int grid = grid_for(R * 1024, 1024, 1);
    int c0 = 0;
    bool first = true;
    while (c0 < C || first) {
        const int left = (int)(C - c0);
        const int lw = left > 8 ? 8 : (left > 4 ? 4 : 2);
        const int cc = left < 2 * lw ? left : 2 * lw;
        // waves per workgroup
        int n_waves = 16;
        if (forced_waves >= 6 && forced_waves <= 16) {
            n_waves = forced_waves;
        } else {
            n_waves = 16;       // (tiles are dealt by tickets: more waves hide more latency, and a wave = a cohort in the output phase)
        }
        auto go = [&](auto u_c, auto lw_c) {
            constexpr int UU = decltype(u_c)::value, LL = decltype(lw_c)::value;
            if (tp == 25)
                launch_rows<UU, LL, 25>(permute, grid, 64 * n_waves, stream, words, n_words, chrom_off, chrom_len, reg_chrom, reg_start, reg_end,
                                        R, s_prob, c0, cc, binsize, n_tiles, pt, first_pos, n_valid, first ? 1 : 0);
            else if (tp == 12)
                launch_rows<UU, LL, 12>(permute, grid, 64 * n_waves, stream, words, n_words, chrom_off, chrom_len, reg_chrom, reg_start, reg_end,
                                        R, s_prob, c0, cc, binsize, n_tiles, pt, first_pos, n_valid, first ? 1 : 0);
            else if (tp == 10)
                launch_rows<UU, LL, 10>(permute, grid, 64 * n_waves, stream, words, n_words, chrom_off, chrom_len, reg_chrom, reg_start, reg_end,
                                        R, s_prob, c0, cc, binsize, n_tiles, pt, first_pos, n_valid, first ? 1 : 0);
            else
                launch_rows<UU, LL, 8>(permute, grid, 64 * n_waves, stream, words, n_words, chrom_off, chrom_len, reg_chrom, reg_start, reg_end,
                                       R, s_prob, c0, cc, binsize, n_tiles, pt, first_pos, n_valid, first ? 1 : 0);
        };
        auto go_u = [&](auto lw_c) {
            if (n_up == 2) go(std::integral_constant<int, 2>{}, lw_c);
            else go(std::integral_constant<int, 1>{}, lw_c);
        };
        if (lw == 8) go_u(std::integral_constant<int, 8>{});
        else if (lw == 4) go_u(std::integral_constant<int, 4>{});
        else go_u(std::integral_constant<int, 2>{});
        c0 += cc;
        first = false;
        if (cc == 0) break;
    }
    DIG_HIP_TRY(hipGetLastError());
    return DIG_OK;
}

}  // namespace dig

#ifdef DIG_TM_TIMING
extern "C" int dig_debug_rows_profile(unsigned long long* out8)
{
    DIG_HIP_TRY(hipDeviceSynchronize());
    DIG_HIP_TRY(hipMemcpyFromSymbol(out8, HIP_SYMBOL(dig::g_rw_prof), 8 * sizeof(unsigned long long)));
    unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    DIG_HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(dig::g_rw_prof), z, sizeof(z)));
    return DIG_OK;
}
#endif
