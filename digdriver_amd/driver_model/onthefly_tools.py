"""quickDriver: burden tests for ad-hoc elements without a pretrained element model.

Mirror of DIGDriver/driver_model/onthefly_tools.py.  The reference walks the elements in Python, fetching sequence
through pysam for every overlapped 10-kb window (onthefly_tools.py:109-164); here the window and block context
counts come from one dig_count_contexts launch each over the HBM-resident packed genome, the per-element sums from
dig_accumulate_elements, and the statistics from the same column functions as elementDriver.
"""
import os
import tempfile

import numpy as np
import pandas as pd

from .. import engine
from ..data_tools import mutation_tools
from ..io import mapfile
from ..sequence_model import genic_driver_tools, sequence_tools
from . import transfer_tools


def region_str_to_params(region_str):
    """onthefly_tools.py:19-27: 'chr1:100-200' -> ('1', 100, 200)."""
    name, _, span = region_str.partition(":")
    lo, _, hi = span.partition("-")
    return name.lstrip("chr"), int(lo), int(hi)          # (lstrip of the CHARACTERS c, h, r, as the reference does)


def DIG_onthefly(f_pretrained, f_mut, f_fasta, f_elts_bed=None, region_str=None, scale_factor=None,
                 scale_factor_indel=None, scale_type="genome", scale_by_expectation=True, max_muts_per_sample=3e9,
                 max_muts_per_elt_per_sample=3e9, skip_pvals=False, all_cosmic=None, strict_reference=True):
    """onthefly_tools.py:29-190.  `strict_reference=True` keeps the reference's double application of the indel scale
    factor (THETA_INDEL is multiplied by cj_indel at :151 and again by element_pvalue_indel, :181 ->
    transfer_tools.py:737); False applies it once."""
    assert f_elts_bed or region_str, "ERROR: you must provide --f-bed or --region_str."
    temp_name = None
    if region_str:                                   # one single-block '+' element named UserELT (:33-38)
        c, lo, hi = region_str_to_params(region_str)
        handle, temp_name = tempfile.mkstemp()
        with os.fdopen(handle, "w") as bed:
            bed.write("\t".join(str(x) for x in (c, lo, hi, "UserELT", 0, "+", 0, 0, ".", 1, "%d," % (hi - lo), "0,")))
        f_elts_bed = temp_name
    try:
        return _onthefly(f_pretrained, f_mut, f_fasta, f_elts_bed, scale_factor, scale_factor_indel, scale_type,
                         scale_by_expectation, max_muts_per_sample, max_muts_per_elt_per_sample, skip_pvals, all_cosmic,
                         strict_reference)
    finally:
        if temp_name:
            os.remove(temp_name)


def _onthefly(f_pretrained, f_mut, f_fasta, f_elts_bed, scale_factor, scale_factor_indel, scale_type,
              scale_by_expectation, max_muts_per_sample, max_muts_per_elt_per_sample, skip_pvals, all_cosmic,
              strict_reference):
    transfer_tools._say('Tabulating mutations')
    df_mut_tab, blacklist = mutation_tools.tabulate_mutations_in_element(
        f_mut, f_elts_bed, bed12=True, drop_duplicates=True, all_elements=True, max_muts_per_sample=max_muts_per_sample,
        max_muts_per_elt_per_sample=max_muts_per_elt_per_sample, return_blacklist=True)
    run = transfer_tools.CohortRun(f_mut, f_pretrained)          # the scale-factor rules are shared with elementDriver
    if scale_by_expectation:
        transfer_tools._say('scaling by expected number of mutations')
        coding = run.coding_rows()
        coding = coding.loc[~coding.SAMPLE.isin(blacklist)]
        cj = run.synonymous_scale(run.gene_model(), coding, dedup=True)                                        # :44-50
        cj_indel = run.uniform_indel_scale(run.gene_model(), coding, transfer_tools.cosmic_null_set(all_cosmic))   # :52-63
    elif scale_factor:
        cj, cj_indel = scale_factor, scale_factor_indel
    else:
        transfer_tools._say('Calculating scale factor')
        cj, cj_indel = run.genome_scale(scale_type)

    genome = sequence_tools.load_genome(f_fasta)
    # element block contexts (strand-aware), onthefly_tools.py:70-71
    L_contexts = sequence_tools.precount_region_contexts_parallel(f_elts_bed, genome, 10, 10000, sub_elts=True, n_up=1,
                                                                  n_down=1)
    tables = genic_driver_tools.RegionTables([mapfile.read_frame(f_pretrained, 'region_params')])
    window = tables.window
    d_pr = genic_driver_tools.sorted_d_pr(mapfile.read_frame(f_pretrained, 'sequence_model_192'))[None, :]

    df_elts = mutation_tools.bed12_boundaries(f_elts_bed)
    E = len(df_elts)
    blk_ptr = np.concatenate([[0], np.cumsum([len(b) for b in df_elts.BLOCK_STARTS])]).astype(np.int64)
    blk_start = np.array([s for b in df_elts.BLOCK_STARTS for s in b], np.int64)
    blk_end = np.array([e for b in df_elts.BLOCK_ENDS for e in b], np.int64)
    chrom = df_elts.CHROM.values.astype(np.int32)
    minus = np.array([(s == '-1' or s == '-') for s in df_elts.STRAND.astype(str)], np.uint8)     # :126
    # L = sum of the block rows of L_contexts (:130-132); integer counts stored as floats in the frame
    owner = np.repeat(np.arange(E), np.diff(blk_ptr))
    keys = ['chr{}:{}-{}'.format(c, s, e) for c, s, e in zip(chrom[owner], blk_start, blk_end)]
    Lb = L_contexts.loc[keys].values
    L = np.zeros((E, 192))
    np.add.at(L, owner, Lb)
    L = np.ascontiguousarray(np.rint(L)[:, None, :], np.int32)

    # overlapped windows (:116) and their context counts from sequence (:118-120), only for the windows touched
    ov_ptr, ov_idx = engine.ideal_overlaps(chrom, blk_ptr, blk_start, blk_end, window, tables.chrom, tables.start)
    rows, inv = np.unique(ov_idx, return_inverse=True)
    win = sequence_tools.count_contexts_by_regions(genome, ['chr' + str(c) for c in tables.chrom[rows]], tables.start[rows],
                                                   tables.start[rows] + window, n_up=1, n_down=1)
    acc = engine.accumulate_elements(tables.mu[rows], tables.std[rows], tables.y[rows], tables.flag[rows],
                                     np.ascontiguousarray(win.values, np.int32), ov_ptr, inv.astype(np.int32), L, minus, d_pr)
    mu, sigma = acc['MU'][:, 0], acc['SIGMA'][:, 0]
    from ..sequence_model import nb_model
    alpha, theta = nb_model.normal_params_to_gamma(mu, sigma)
    pretrain_df = pd.DataFrame({
        'ELT_SIZE': acc['ELT_SIZE'], 'FLAG': acc['FLAG'][:, 0], 'R_SIZE': acc['R_SIZE'], 'R_OBS': acc['R_OBS'][:, 0],
        'R_INDEL': acc['R_OBS'][:, 0], 'MU': mu, 'SIGMA': sigma, 'ALPHA': alpha, 'THETA': theta * cj,
        'MU_INDEL': mu, 'SIGMA_INDEL': sigma, 'ALPHA_INDEL': alpha,
        'THETA_INDEL': theta * cj_indel if strict_reference else theta,
        'Pi_SUM': acc['P'][:, 0, 0], 'Pi_INDEL': acc['P_INDEL']}, index=df_elts.ELT.values)

    T = transfer_tools
    df_model = T.element_expected_muts_nb(df_mut_tab.merge(pretrain_df, left_on='ELT', right_index=True))
    if skip_pvals:
        return df_model
    df_model = T.element_pvalue_burden_nb_by_sample(T.element_pvalue_burden_nb(df_model))
    return T.combine_snv_indel(T.element_pvalue_indel(df_model, cj_indel), 'PVAL_SNV_BURDEN')
