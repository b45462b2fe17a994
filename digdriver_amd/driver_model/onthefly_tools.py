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
    col_split = region_str.split(":")
    chrom = col_split[0].lstrip("chr")
    pos_split = col_split[1].split("-")
    return chrom, int(pos_split[0]), int(pos_split[1])


def DIG_onthefly(f_pretrained, f_mut, f_fasta, f_elts_bed=None, region_str=None, scale_factor=None,
                 scale_factor_indel=None, scale_type="genome", scale_by_expectation=True, max_muts_per_sample=3e9,
                 max_muts_per_elt_per_sample=3e9, skip_pvals=False, all_cosmic=None, strict_reference=True):
    """onthefly_tools.py:29-190.  `strict_reference=True` keeps the reference's double application of the indel scale
    factor (THETA_INDEL is multiplied by cj_indel at :151 and again by element_pvalue_indel, :181 ->
    transfer_tools.py:737); False applies it once."""
    assert f_elts_bed or region_str, "ERROR: you must provide --f-bed or --region_str."
    temp_name = None
    if region_str:
        temp_file, temp_name = tempfile.mkstemp()
        CHROM, START, END = region_str_to_params(region_str)
        os.write(temp_file, "{}\t{}\t{}\tUserELT\t0\t+\t0\t0\t.\t1\t{},\t0,".format(CHROM, START, END, END - START).encode())
        os.close(temp_file)
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
    print('Tabulating mutations')
    df_mut_tab, blacklist = mutation_tools.tabulate_mutations_in_element(
        f_mut, f_elts_bed, bed12=True, drop_duplicates=True, all_elements=True, max_muts_per_sample=max_muts_per_sample,
        max_muts_per_elt_per_sample=max_muts_per_elt_per_sample, return_blacklist=True)
    if scale_by_expectation:
        print('scaling by expected number of mutations')
        df_gene = transfer_tools.load_pretrained_model(f_pretrained)
        df_mut = transfer_tools.read_mutations_cds(f_mut)
        df_mut = df_mut[~df_mut.SAMPLE.isin(blacklist)]
        df_syn = df_mut[(df_mut.ANNOT == 'Synonymous') & (df_mut.GENE != 'TP53')].drop_duplicates()
        not_tp53 = df_gene[df_gene.index != 'TP53']
        cj = len(df_syn) / (not_tp53.MU * not_tp53.Pi_SYN).sum()
        if all_cosmic is None:
            all_cosmic = transfer_tools._read_gene_panel('CGC_ALL') + ['CDKN2A.p14arf', 'CDKN2A.p16INK4a']
        df_gene_null = df_gene[~df_gene.index.isin(all_cosmic)]
        df_mut_null = df_mut[~df_mut.index.isin(all_cosmic)]      # row-index filter = no-op, as in the reference (:59)
        exp_indel = (df_gene_null.Pi_INDEL * df_gene_null.ALPHA_INDEL * df_gene_null.THETA_INDEL).sum()
        cj_indel = len(df_mut_null[df_mut_null.ANNOT == 'INDEL']) / exp_indel
    elif scale_factor:
        cj, cj_indel = scale_factor, scale_factor_indel
    else:
        print('Calculating scale factor')
        cj, cj_indel = transfer_tools.calc_scale_factor_efficient(f_mut, f_pretrained, scale_type=scale_type)

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

    df_model = df_mut_tab.merge(pretrain_df, left_on='ELT', right_index=True)
    df_model = transfer_tools.element_expected_muts_nb(df_model)
    if not skip_pvals:
        df_model = transfer_tools.element_pvalue_burden_nb(df_model)
        df_model = transfer_tools.element_pvalue_burden_nb_by_sample(df_model)
        df_model = transfer_tools.element_pvalue_indel(df_model, cj_indel)
        df_model = transfer_tools.combine_snv_indel(df_model, 'PVAL_SNV_BURDEN')
    return df_model
