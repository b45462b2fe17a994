// dig_host.hip -- `_host` twins of the entry points whose device forms live in dig_pipeline.hip, dig_tiles.hip and
// dig_join.hip (one twin per entry point).  A twin takes host pointers, stages them through device buffers
// of its own, runs the device entry point on the null stream and copies the results back: the PCIe-inclusive path for
// callers without device memory of their own (small problems, tests, the reference-side binding of INTEGRATION.md).
#include <utility>
#include <vector>

#include "dig_common.hpp"

using namespace dig;

namespace {

struct Stage {                 // host <-> device staging of one call
    std::vector<std::pair<DevBuf*, std::pair<void*, size_t>>> outs;
    int up(DevBuf& b, const void* src, size_t bytes)
    {
        DIG_HIP_TRY(b.alloc(bytes));
        if (src && bytes) DIG_HIP_TRY(hipMemcpy(b.p, src, bytes, hipMemcpyHostToDevice));
        return DIG_OK;
    }
    int out(DevBuf& b, void* dst, size_t bytes)
    {
        DIG_HIP_TRY(b.alloc(bytes));
        outs.push_back({&b, {dst, bytes}});
        return DIG_OK;
    }
    int down()
    {
        DIG_HIP_TRY(hipDeviceSynchronize());
        for (auto& o : outs)
            if (o.second.second) DIG_HIP_TRY(hipMemcpy(o.second.first, o.first->p, o.second.second, hipMemcpyDeviceToHost));
        return DIG_OK;
    }
};
#define DIG_TRY(expr)            \
    do {                         \
        int _rc = (expr);        \
        if (_rc) return _rc;     \
    } while (0)

}  // namespace

extern "C" {

int dig_element_pipeline_host(const double* bin_mu, const double* bin_std, const int32_t* bin_y, const uint8_t* bin_flag,
                              const int32_t* bin_ctx, const int64_t* ov_ptr, const int32_t* ov_idx, const int32_t* L,
                              const uint8_t* strand_minus, const int32_t* gene_length, const double* d_pr,
                              const int32_t* obs_snv, const int32_t* obs_samples, const int32_t* obs_indel, const double* cj,
                              const double* cj_indel, double* MU, double* SIGMA, int32_t* R_OBS, int32_t* FLAG, double* P,
                              int32_t* R_SIZE, int32_t* ELT_SIZE, double* P_INDEL, double* out, int64_t N, int64_t E, int64_t C,
                              int device)
{
    DIG_REQUIRE(N >= 0 && E >= 0 && C >= 0, "N, E, C >= 0");
    if (E == 0 || C == 0) return DIG_OK;
    DIG_REQUIRE(bin_mu && bin_std && bin_y && bin_flag && bin_ctx && ov_ptr && ov_idx && L && strand_minus && d_pr,
                "non-null accumulation inputs");
    DIG_REQUIRE(obs_snv && obs_samples && obs_indel && cj && cj_indel, "non-null statistics inputs");
    DIG_REQUIRE(MU && SIGMA && R_OBS && FLAG && P && R_SIZE && ELT_SIZE && P_INDEL && out, "non-null outputs");
    DIG_HIP_TRY(hipSetDevice(device));
    const int64_t nnz = ov_ptr[E];
    DIG_REQUIRE(nnz >= 0, "ov_ptr[E] >= 0");
    for (int64_t q = 0; q < nnz; ++q) DIG_REQUIRE(ov_idx[q] >= 0 && ov_idx[q] < N, "ov_idx within [0, N)");
    const size_t nNC = (size_t)N * C, nEC = (size_t)E * C;
    Stage st;
    DevBuf d_mu, d_sd, d_y, d_fl, d_ctx, d_ptr, d_idx, d_L, d_sm, d_gl, d_dpr, d_o1, d_o2, d_o3, d_cj, d_cji;
    DevBuf o_mu, o_sg, o_ro, o_fg, o_p, o_rs, o_es, o_pi, o_out, d_ws;
    DIG_TRY(st.up(d_mu, bin_mu, nNC * 8));
    DIG_TRY(st.up(d_sd, bin_std, nNC * 8));
    DIG_TRY(st.up(d_y, bin_y, nNC * 4));
    DIG_TRY(st.up(d_fl, bin_flag, nNC));
    DIG_TRY(st.up(d_ctx, bin_ctx, (size_t)N * 64 * 4));
    DIG_TRY(st.up(d_ptr, ov_ptr, (size_t)(E + 1) * 8));
    DIG_TRY(st.up(d_idx, ov_idx, (size_t)(nnz > 0 ? nnz : 1) * 4));
    DIG_TRY(st.up(d_L, L, (size_t)E * 192 * 4));
    DIG_TRY(st.up(d_sm, strand_minus, (size_t)E));
    if (gene_length) DIG_TRY(st.up(d_gl, gene_length, (size_t)E * 4));
    DIG_TRY(st.up(d_dpr, d_pr, (size_t)C * 192 * 8));
    DIG_TRY(st.up(d_o1, obs_snv, nEC * 4));
    DIG_TRY(st.up(d_o2, obs_samples, nEC * 4));
    DIG_TRY(st.up(d_o3, obs_indel, nEC * 4));
    DIG_TRY(st.up(d_cj, cj, (size_t)C * 8));
    DIG_TRY(st.up(d_cji, cj_indel, (size_t)C * 8));
    DIG_TRY(st.out(o_mu, MU, nEC * 8));
    DIG_TRY(st.out(o_sg, SIGMA, nEC * 8));
    DIG_TRY(st.out(o_ro, R_OBS, nEC * 4));
    DIG_TRY(st.out(o_fg, FLAG, nEC * 4));
    DIG_TRY(st.out(o_p, P, nEC * 8));
    DIG_TRY(st.out(o_rs, R_SIZE, (size_t)E * 4));
    DIG_TRY(st.out(o_es, ELT_SIZE, (size_t)E * 4));
    DIG_TRY(st.out(o_pi, P_INDEL, (size_t)E * 8));
    DIG_TRY(st.out(o_out, out, nEC * 7 * 8));
    const int64_t wsb = dig_element_pipeline_workspace(E, C);
    DIG_REQUIRE(wsb > 0, "E * C must stay below 2^32 - 1");
    DIG_HIP_TRY(d_ws.alloc((size_t)wsb));
    int compact = 0;
    if (N >= 1) DIG_TRY(dig_element_pipeline_prepare(d_L.as<int32_t>(), E, C, d_ws.p, wsb, &compact, nullptr));
    DIG_TRY(dig_element_pipeline(d_mu.as<double>(), d_sd.as<double>(), d_y.as<int32_t>(), d_fl.as<uint8_t>(), d_ctx.as<int32_t>(),
                                 d_ptr.as<int64_t>(), d_idx.as<int32_t>(), d_L.as<int32_t>(), d_sm.as<uint8_t>(),
                                 gene_length ? d_gl.as<int32_t>() : nullptr, d_dpr.as<double>(), d_o1.as<int32_t>(),
                                 d_o2.as<int32_t>(), d_o3.as<int32_t>(), d_cj.as<double>(), d_cji.as<double>(), o_mu.as<double>(),
                                 o_sg.as<double>(), o_ro.as<int32_t>(), o_fg.as<int32_t>(), o_p.as<double>(), o_rs.as<int32_t>(),
                                 o_es.as<int32_t>(), o_pi.as<double>(), o_out.as<double>(), N, E, C, nullptr,
                                 DIG_PIPE_ALL | (compact ? DIG_PIPE_COMPACT_L : 0), d_ws.p, wsb, nullptr));
    return st.down();
}

int dig_base_tile_probs_host(const uint32_t* genome_words, int64_t n_words, const int64_t* chrom_off, const int64_t* chrom_len,
                             int n_chrom, const int32_t* reg_chrom, const int64_t* reg_start, const int64_t* reg_end, int64_t R,
                             const double* s_prob, int64_t C, int binsize, int64_t n_tiles, double* pt, int64_t* first_pos,
                             int32_t* n_valid, int device)
{
    DIG_REQUIRE(R >= 0 && C >= 0 && n_words >= 2 && n_chrom >= 0 && n_tiles >= 0, "non-negative sizes, n_words >= 2 (pad words)");
    if (R == 0) return DIG_OK;
    DIG_REQUIRE(genome_words && chrom_off && chrom_len && reg_chrom && reg_start && reg_end && first_pos && n_valid, "non-null pointers");
    DIG_REQUIRE(C == 0 || n_tiles == 0 || (s_prob && pt), "s_prob and pt");
    DIG_HIP_TRY(hipSetDevice(device));
    Stage st;
    DevBuf dw, doff, dlen, dc, ds, de, dsp, opt, ofp, onv;
    DIG_TRY(st.up(dw, genome_words, (size_t)n_words * 4));
    DIG_TRY(st.up(doff, chrom_off, (size_t)n_chrom * 8));
    DIG_TRY(st.up(dlen, chrom_len, (size_t)n_chrom * 8));
    DIG_TRY(st.up(dc, reg_chrom, (size_t)R * 4));
    DIG_TRY(st.up(ds, reg_start, (size_t)R * 8));
    DIG_TRY(st.up(de, reg_end, (size_t)R * 8));
    DIG_TRY(st.up(dsp, s_prob, (size_t)C * 64 * 8));
    DIG_TRY(st.out(opt, pt, (size_t)C * R * n_tiles * 8));
    DIG_TRY(st.out(ofp, first_pos, (size_t)R * 8));
    DIG_TRY(st.out(onv, n_valid, (size_t)R * 4));
    DIG_TRY(dig_base_tile_probs(dw.as<uint32_t>(), n_words, doff.as<int64_t>(), dlen.as<int64_t>(), n_chrom, dc.as<int32_t>(),
                                ds.as<int64_t>(), de.as<int64_t>(), R, dsp.as<double>(), C, binsize, n_tiles, opt.as<double>(),
                                ofp.as<int64_t>(), onv.as<int32_t>(), nullptr));
    return st.down();
}

int dig_tile_mut_counts_host(const int32_t* pair_mut, const int32_t* pair_reg, int64_t n_pairs, const int64_t* mut_start,
                             int64_t n_mut, const int32_t* mut_cohort, const int64_t* first_pos, const int32_t* n_valid,
                             int binsize, int64_t n_tiles, int64_t R, int64_t C, int32_t* k, int device)
{
    DIG_REQUIRE(n_pairs >= 0 && n_mut >= 0 && binsize >= 1 && n_tiles >= 0 && R >= 0 && C >= 0, "non-negative sizes, binsize >= 1");
    const size_t n = (size_t)C * R * n_tiles;
    if (n == 0) return DIG_OK;
    DIG_REQUIRE(k && first_pos && n_valid, "non-null outputs / region tables");
    DIG_REQUIRE(n_pairs == 0 || (pair_mut && pair_reg && mut_start && mut_cohort), "non-null pair / mutation arrays");
    for (int64_t i = 0; i < n_pairs; ++i)
        DIG_REQUIRE(pair_mut[i] >= 0 && pair_mut[i] < n_mut && pair_reg[i] >= 0 && pair_reg[i] < R, "pairs inside the mutation / region tables");
    DIG_HIP_TRY(hipSetDevice(device));
    Stage st;
    DevBuf dpm, dpr, dms, dmc, dfp, dnv, ok;
    DIG_TRY(st.up(dpm, pair_mut, (size_t)n_pairs * 4));
    DIG_TRY(st.up(dpr, pair_reg, (size_t)n_pairs * 4));
    DIG_TRY(st.up(dms, mut_start, (size_t)n_mut * 8));
    DIG_TRY(st.up(dmc, mut_cohort, (size_t)n_mut * 4));
    DIG_TRY(st.up(dfp, first_pos, (size_t)R * 8));
    DIG_TRY(st.up(dnv, n_valid, (size_t)R * 4));
    DIG_TRY(st.out(ok, k, n * 4));
    DIG_TRY(dig_tile_mut_counts(dpm.as<int32_t>(), dpr.as<int32_t>(), n_pairs, dms.as<int64_t>(), dmc.as<int32_t>(), dfp.as<int64_t>(),
                                dnv.as<int32_t>(), binsize, n_tiles, R, C, ok.as<int32_t>(), nullptr));
    return st.down();
}

int dig_overlap_join_count_host(const int64_t* blk_start_key, const int64_t* blk_runmax_key, const int64_t* blk_end, int64_t n_blk,
                                const int64_t* mut_chrom, const int64_t* mut_start, const int64_t* mut_end, int64_t n_mut,
                                int32_t* counts, int device)
{
    DIG_REQUIRE(n_blk >= 0 && n_mut >= 0, "sizes >= 0");
    if (n_mut == 0) return DIG_OK;
    DIG_REQUIRE(mut_chrom && mut_start && mut_end && counts, "non-null mutation arrays");
    DIG_REQUIRE(n_blk == 0 || (blk_start_key && blk_runmax_key && blk_end), "non-null block arrays");
    DIG_HIP_TRY(hipSetDevice(device));
    Stage st;
    DevBuf b1, b2, b3, m1, m2, m3, oc;
    DIG_TRY(st.up(b1, blk_start_key, (size_t)n_blk * 8));
    DIG_TRY(st.up(b2, blk_runmax_key, (size_t)n_blk * 8));
    DIG_TRY(st.up(b3, blk_end, (size_t)n_blk * 8));
    DIG_TRY(st.up(m1, mut_chrom, (size_t)n_mut * 8));
    DIG_TRY(st.up(m2, mut_start, (size_t)n_mut * 8));
    DIG_TRY(st.up(m3, mut_end, (size_t)n_mut * 8));
    DIG_TRY(st.out(oc, counts, (size_t)n_mut * 4));
    DIG_TRY(dig_overlap_join_count(b1.as<int64_t>(), b2.as<int64_t>(), b3.as<int64_t>(), n_blk, m1.as<int64_t>(), m2.as<int64_t>(),
                                   m3.as<int64_t>(), n_mut, oc.as<int32_t>(), nullptr));
    return st.down();
}

int dig_overlap_join_fill_host(const int64_t* blk_start_key, const int64_t* blk_runmax_key, const int64_t* blk_end, int64_t n_blk,
                               const int64_t* mut_chrom, const int64_t* mut_start, const int64_t* mut_end, int64_t n_mut,
                               const int64_t* offsets, int64_t n_pairs, int32_t* pair_mut, int32_t* pair_blk, int device)
{
    DIG_REQUIRE(n_blk >= 0 && n_mut >= 0 && n_pairs >= 0, "sizes >= 0");
    if (n_mut == 0 || n_blk == 0 || n_pairs == 0) return DIG_OK;
    DIG_REQUIRE(mut_chrom && mut_start && mut_end && offsets && pair_mut && pair_blk, "non-null arrays");
    DIG_REQUIRE(blk_start_key && blk_runmax_key && blk_end, "non-null block arrays");
    DIG_HIP_TRY(hipSetDevice(device));
    Stage st;
    DevBuf b1, b2, b3, m1, m2, m3, off, o1, o2;
    DIG_TRY(st.up(b1, blk_start_key, (size_t)n_blk * 8));
    DIG_TRY(st.up(b2, blk_runmax_key, (size_t)n_blk * 8));
    DIG_TRY(st.up(b3, blk_end, (size_t)n_blk * 8));
    DIG_TRY(st.up(m1, mut_chrom, (size_t)n_mut * 8));
    DIG_TRY(st.up(m2, mut_start, (size_t)n_mut * 8));
    DIG_TRY(st.up(m3, mut_end, (size_t)n_mut * 8));
    DIG_TRY(st.up(off, offsets, (size_t)n_mut * 8));
    DIG_TRY(st.out(o1, pair_mut, (size_t)n_pairs * 4));
    DIG_TRY(st.out(o2, pair_blk, (size_t)n_pairs * 4));
    DIG_TRY(dig_overlap_join_fill(b1.as<int64_t>(), b2.as<int64_t>(), b3.as<int64_t>(), n_blk, m1.as<int64_t>(), m2.as<int64_t>(),
                                  m3.as<int64_t>(), n_mut, off.as<int64_t>(), o1.as<int32_t>(), o2.as<int32_t>(), nullptr));
    return st.down();
}

int dig_base_tile_probs_ctx_host(const uint32_t* genome_words, int64_t n_words, const int64_t* chrom_off, const int64_t* chrom_len,
                                 int n_chrom, const int32_t* reg_chrom, const int64_t* reg_start, const int64_t* reg_end, int64_t R,
                                 const double* s_prob, int64_t C, int n_up, int binsize, int64_t n_tiles, double* pt,
                                 int64_t* first_pos, int32_t* n_valid, int device)
{
    DIG_REQUIRE(n_up == 1 || n_up == 2, "n_up = n_down = 1 or 2");
    DIG_REQUIRE(R >= 0 && C >= 0 && n_words >= 2 && n_chrom >= 0 && n_tiles >= 0, "non-negative sizes, n_words >= 2 (pad words)");
    if (R == 0) return DIG_OK;
    DIG_REQUIRE(genome_words && chrom_off && chrom_len && reg_chrom && reg_start && reg_end && first_pos && n_valid, "non-null pointers");
    DIG_REQUIRE(C == 0 || n_tiles == 0 || (s_prob && pt), "s_prob and pt");
    for (int64_t r = 0; r < R; ++r) {
        DIG_REQUIRE(reg_chrom[r] >= 0 && reg_chrom[r] < n_chrom && reg_start[r] >= 0 && reg_end[r] >= 0, "regions inside the genome table");
        DIG_REQUIRE(n_up == 1 || reg_end[r] - reg_start[r] <= 16384, "a region of the general-context form holds at most 16 384 positions");
        DIG_REQUIRE(n_up == 1 || reg_start[r] == 0 || reg_start[r] >= n_up, "a region that starts inside (0, n_up) would fetch from a negative position");
    }
    DIG_HIP_TRY(hipSetDevice(device));
    const int64_t K = n_up == 1 ? 64 : 1024;
    Stage st;
    DevBuf dw, doff, dlen, dc, ds, de, dsp, opt, ofp, onv;
    DIG_TRY(st.up(dw, genome_words, (size_t)n_words * 4));
    DIG_TRY(st.up(doff, chrom_off, (size_t)n_chrom * 8));
    DIG_TRY(st.up(dlen, chrom_len, (size_t)n_chrom * 8));
    DIG_TRY(st.up(dc, reg_chrom, (size_t)R * 4));
    DIG_TRY(st.up(ds, reg_start, (size_t)R * 8));
    DIG_TRY(st.up(de, reg_end, (size_t)R * 8));
    DIG_TRY(st.up(dsp, s_prob, (size_t)C * K * 8));
    DIG_TRY(st.out(opt, pt, (size_t)C * R * n_tiles * 8));
    DIG_TRY(st.out(ofp, first_pos, (size_t)R * 8));
    DIG_TRY(st.out(onv, n_valid, (size_t)R * 4));
    DIG_TRY(dig_base_tile_probs_ctx(dw.as<uint32_t>(), n_words, doff.as<int64_t>(), dlen.as<int64_t>(), n_chrom, dc.as<int32_t>(),
                                    ds.as<int64_t>(), de.as<int64_t>(), R, dsp.as<double>(), C, n_up, binsize, n_tiles,
                                    opt.as<double>(), ofp.as<int64_t>(), onv.as<int32_t>(), nullptr));
    return st.down();
}

int dig_gene_pipeline_host(const double* bin_mu, const double* bin_std, const int32_t* bin_y, const uint8_t* bin_flag,
                           const int32_t* bin_ctx, const int64_t* ov_ptr, const int32_t* ov_idx, const int32_t* L,
                           const uint8_t* strand_minus, const int32_t* gene_length, const double* d_pr, const int32_t* obs,
                           const int32_t* n_samp, const double* cj, const double* t_indel, int with_indel, double* MU, double* SIGMA,
                           int32_t* R_OBS, int32_t* FLAG, double* P, int32_t* R_SIZE, int32_t* ELT_SIZE, double* P_INDEL, double* out,
                           int64_t N, int64_t G, int64_t C, int device)
{
    DIG_REQUIRE(N >= 0 && G >= 0 && C >= 0, "N, G, C >= 0");
    if (G == 0 || C == 0) return DIG_OK;
    DIG_REQUIRE(bin_mu && bin_std && bin_y && bin_flag && bin_ctx && ov_ptr && ov_idx && L && strand_minus && d_pr, "non-null accumulation inputs");
    DIG_REQUIRE(obs && n_samp && cj && (!with_indel || t_indel), "non-null statistics inputs");
    DIG_REQUIRE(MU && SIGMA && R_OBS && FLAG && P && R_SIZE && ELT_SIZE && P_INDEL && out, "non-null outputs");
    DIG_HIP_TRY(hipSetDevice(device));
    const int64_t nnz = ov_ptr[G];
    DIG_REQUIRE(nnz >= 0, "ov_ptr[G] >= 0");
    for (int64_t q = 0; q < nnz; ++q) DIG_REQUIRE(ov_idx[q] >= 0 && ov_idx[q] < N, "ov_idx within [0, N)");
    const size_t nNC = (size_t)N * C, nGC = (size_t)G * C;
    Stage st;
    DevBuf d_mu, d_sd, d_y, d_fl, d_ctx, d_ptr, d_idx, d_L, d_sm, d_gl, d_dpr, d_ob, d_ns, d_cj, d_ti;
    DevBuf o_mu, o_sg, o_ro, o_fg, o_p, o_rs, o_es, o_pi, o_out, d_ws;
    DIG_TRY(st.up(d_mu, bin_mu, nNC * 8));
    DIG_TRY(st.up(d_sd, bin_std, nNC * 8));
    DIG_TRY(st.up(d_y, bin_y, nNC * 4));
    DIG_TRY(st.up(d_fl, bin_flag, nNC));
    DIG_TRY(st.up(d_ctx, bin_ctx, (size_t)N * 64 * 4));
    DIG_TRY(st.up(d_ptr, ov_ptr, (size_t)(G + 1) * 8));
    DIG_TRY(st.up(d_idx, ov_idx, (size_t)(nnz > 0 ? nnz : 1) * 4));
    DIG_TRY(st.up(d_L, L, (size_t)G * 4 * 192 * 4));
    DIG_TRY(st.up(d_sm, strand_minus, (size_t)G));
    if (gene_length) DIG_TRY(st.up(d_gl, gene_length, (size_t)G * 4));
    DIG_TRY(st.up(d_dpr, d_pr, (size_t)C * 192 * 8));
    DIG_TRY(st.up(d_ob, obs, nGC * 5 * 4));
    DIG_TRY(st.up(d_ns, n_samp, nGC * 6 * 4));
    DIG_TRY(st.up(d_cj, cj, (size_t)C * 8));
    if (t_indel) DIG_TRY(st.up(d_ti, t_indel, (size_t)C * 8));
    DIG_TRY(st.out(o_mu, MU, nGC * 8));
    DIG_TRY(st.out(o_sg, SIGMA, nGC * 8));
    DIG_TRY(st.out(o_ro, R_OBS, nGC * 4));
    DIG_TRY(st.out(o_fg, FLAG, nGC * 4));
    DIG_TRY(st.out(o_p, P, nGC * 4 * 8));
    DIG_TRY(st.out(o_rs, R_SIZE, (size_t)G * 4));
    DIG_TRY(st.out(o_es, ELT_SIZE, (size_t)G * 4));
    DIG_TRY(st.out(o_pi, P_INDEL, (size_t)G * 8));
    DIG_TRY(st.out(o_out, out, nGC * 22 * 8));
    const int64_t wsb = dig_accumulate_workspace(G, C);
    DIG_HIP_TRY(d_ws.alloc((size_t)wsb));
    DIG_TRY(dig_gene_pipeline(d_mu.as<double>(), d_sd.as<double>(), d_y.as<int32_t>(), d_fl.as<uint8_t>(), d_ctx.as<int32_t>(),
                              d_ptr.as<int64_t>(), d_idx.as<int32_t>(), d_L.as<int32_t>(), d_sm.as<uint8_t>(),
                              gene_length ? d_gl.as<int32_t>() : nullptr, d_dpr.as<double>(), d_ob.as<int32_t>(), d_ns.as<int32_t>(),
                              d_cj.as<double>(), t_indel ? d_ti.as<double>() : nullptr, with_indel, o_mu.as<double>(), o_sg.as<double>(),
                              o_ro.as<int32_t>(), o_fg.as<int32_t>(), o_p.as<double>(), o_rs.as<int32_t>(), o_es.as<int32_t>(),
                              o_pi.as<double>(), o_out.as<double>(), N, G, C, d_ws.p, wsb, nullptr));
    return st.down();
}

}  // extern "C"
