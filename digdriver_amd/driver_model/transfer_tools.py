"""Transfer a pretrained DIG model to a new cohort and test elements / genes for mutation burden.

Host mirror of DIGDriver/driver_model/transfer_tools.py: same function names, arguments, column
names and mutate-and-return-the-frame behaviour.  All p-value / expected-count arithmetic runs in
the HIP kernels of libdig_hip.so (``dig_element_stats`` for the element block and each gene
mutation class, ``dig_nb_midp_upper`` / ``dig_fisher`` for the single-column functions); pandas is
only used for the joins and the integer bookkeeping the reference also does in pandas.
"""
import os

import numpy as np
import pandas as pd

from .. import engine
from ..data_tools import mutation_tools
from ..io import mapfile
from ..sequence_model import nb_model

_DATA_DIR = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "data")


def _read_gene_panel(name):
    """Gene panels ship with the reference as package data (DIGDriver/data/genes_*.txt); a deployment
    drops them into digdriver_amd/data/ (see INTEGRATION.md).  Missing panel -> FileNotFoundError."""
    path = os.path.join(_DATA_DIR, 'genes_{}.txt'.format(name))
    return pd.read_table(path, names=['GENE']).GENE.to_list()


def load_pretrained_model(h5, key='genic_model', restrict_cols=True):
    """transfer_tools.py:11-76"""
    df_pretrain = mapfile.read_frame(h5, key)
    alpha, theta = nb_model.normal_params_to_gamma(df_pretrain.MU.values, df_pretrain.SIGMA.values)
    df_pretrain['ALPHA'] = alpha
    df_pretrain['THETA'] = theta

    def _indel_params():
        a, t = nb_model.normal_params_to_gamma(df_pretrain.MU_INDEL.values, df_pretrain.SIGMA_INDEL.values)
        df_pretrain['ALPHA_INDEL'] = a
        df_pretrain['THETA_INDEL'] = t

    if key == 'genic_model':
        df_pretrain.set_index(df_pretrain.GENE, inplace=True)
        df_pretrain.rename({'P_MIS': 'Pi_MIS', 'P_NONS': 'Pi_NONS', 'P_SILENT': 'Pi_SYN', 'P_SPLICE': 'Pi_SPL',
                            'P_TRUNC': 'Pi_TRUNC', 'P_INDEL': 'Pi_INDEL'}, axis=1, inplace=True)
        df_pretrain['Pi_NONSYN'] = df_pretrain.Pi_MIS + df_pretrain.Pi_TRUNC
        _indel_params()
    elif 'P_INDEL' in df_pretrain.columns:
        df_pretrain.set_index(df_pretrain.ELT, inplace=True)
        df_pretrain.rename({'P_SUM': 'Pi_SUM', 'P_INDEL': 'Pi_INDEL'}, axis=1, inplace=True)
        _indel_params()
    else:
        df_pretrain.set_index(df_pretrain.ELT, inplace=True)
        df_pretrain.rename({'P_SUM': 'Pi_SUM'}, axis=1, inplace=True)

    if restrict_cols:
        if key == 'genic_model':
            cols = ['CHROM', 'GENE_LENGTH', 'R_SIZE', 'R_OBS', 'R_INDEL', 'MU', 'SIGMA', 'ALPHA', 'THETA',
                    'MU_INDEL', 'SIGMA_INDEL', 'ALPHA_INDEL', 'THETA_INDEL', 'FLAG',
                    'Pi_SYN', 'Pi_MIS', 'Pi_NONS', 'Pi_SPL', 'Pi_TRUNC', 'Pi_NONSYN', 'Pi_INDEL']
        elif 'Pi_INDEL' in df_pretrain.columns:
            cols = ['ELT_SIZE', 'FLAG', 'R_SIZE', 'R_OBS', 'R_INDEL', 'MU', 'SIGMA', 'ALPHA', 'THETA',
                    'MU_INDEL', 'SIGMA_INDEL', 'ALPHA_INDEL', 'THETA_INDEL', 'Pi_SUM', 'Pi_INDEL']
        else:
            cols = ['R_OBS', 'MU', 'SIGMA', 'ALPHA', 'THETA', 'Pi_SUM']
        df_pretrain = df_pretrain[cols]
    return df_pretrain


def read_mutations_cds(f_mut, f_cds=None):
    """transfer_tools.py:78-92"""
    df_mut = mutation_tools.read_mutation_file(f_mut, drop_duplicates=False, drop_sex=True)
    df_mut_cds = df_mut[df_mut.GENE != '.']
    if f_cds:
        df_cds = pd.read_table(f_cds, names=['CHROM', 'START', 'END', 'GENE'], low_memory=False)
        df_mut_cds = mutation_tools.restrict_mutations_by_bed(df_mut_cds, df_cds, unique=True, replace_cols=True,
                                                              remove_X=False)
    return df_mut_cds


def calc_scale_factor(df_mut, h5_pretrain, scale_type='genome'):
    """transfer_tools.py:94-127"""
    df_dedup = mutation_tools.drop_duplicate_mutations(df_mut)
    attrs = mapfile.read_attrs(h5_pretrain)
    if scale_type == 'genome':
        idx = mapfile.read_array(h5_pretrain, 'idx')
        mapp = mapfile.read_array(h5_pretrain, 'mappability')
        idx_mapp = idx[mapp > attrs['mappability_threshold']]
        df_idx = pd.DataFrame(idx_mapp, columns=['CHROM', 'START', 'END'])
        df_inter = mutation_tools.restrict_mutations_by_bed(df_dedup, df_idx, remove_X=False)
        return len(df_inter) / attrs['N_MUT_TRAIN']
    if scale_type == 'exome':
        return len(df_dedup[df_dedup.ANNOT != 'Noncoding']) / attrs['N_MUT_CDS']
    if scale_type == 'sample':
        return len(df_dedup.SAMPLE.unique()) / attrs['N_SAMPLES']
    raise ValueError("scale_type {} is not recognized".format(scale_type))


def calc_scale_factor_efficient(f_mut, h5_pretrain, scale_type='genome'):
    """transfer_tools.py:129-159: (cj_snv, cj_indel) = (#SNV, #INDEL in unflagged bins) / sum Y_PRED[~FLAG].
    The masked column sum runs in dig_scale_suffstats."""
    if scale_type != 'genome':
        raise ValueError("scale_type {} is not recognized".format(scale_type))
    regions = mapfile.read_frame(h5_pretrain, 'region_params')
    regions_pass = regions[~regions.FLAG.astype(bool)]
    import tempfile
    fd, tmp = tempfile.mkstemp(suffix=".bed")
    os.close(fd)
    try:
        regions_pass[['CHROM', 'START', 'END']].to_csv(tmp, sep="\t", header=False, index=False)
        df_inter = mutation_tools.restrict_mutations_by_bed_efficient(f_mut, tmp, bed12=False, drop_duplicates=True)
    finally:
        os.remove(tmp)
    n_exp = float(engine.scale_suffstats(regions.Y_PRED.values[:, None], regions.FLAG.values.astype(np.uint8)[:, None])[0])
    n_snv = len(df_inter[df_inter.ANNOT != 'INDEL'])
    n_ind = len(df_inter[df_inter.ANNOT == 'INDEL'])
    return n_snv / n_exp, n_ind / n_exp


# ---------------------------------------------------------------------------------------------
# joins (integer bookkeeping, pandas like the reference)
# ---------------------------------------------------------------------------------------------
_GENE_COLS_LEFT = ['CHROM', 'GENE_LENGTH', 'R_SIZE', 'R_OBS', 'R_INDEL', 'MU', 'SIGMA', 'ALPHA', 'THETA',
                   'MU_INDEL', 'SIGMA_INDEL', 'ALPHA_INDEL', 'THETA_INDEL', 'FLAG',
                   'Pi_SYN', 'Pi_MIS', 'Pi_NONS', 'Pi_SPL', 'Pi_TRUNC', 'Pi_NONSYN', 'Pi_INDEL']


def transfer_gene_model(df_mut_cds, df_counts, df_pretrain, cj):
    """transfer_tools.py:196-270"""
    cols_right = ['OBS_SYN', 'OBS_MIS', 'OBS_NONS', 'OBS_SPL', 'OBS_INDEL']
    df_model = df_pretrain[_GENE_COLS_LEFT].merge(df_counts[cols_right], left_index=True, right_index=True, how='left')
    for c in ('OBS_MIS', 'OBS_NONS', 'OBS_SPL', 'OBS_SYN', 'OBS_INDEL'):
        df_model[c] = df_model[c].fillna(0)
    df_model['OBS_TRUNC'] = df_model.OBS_NONS + df_model.OBS_SPL
    df_model['OBS_NONSYN'] = df_model.OBS_MIS + df_model.OBS_TRUNC

    def _n_samp(mask):
        sub = df_mut_cds[mask]
        return sub.groupby(['GENE', 'SAMPLE']).size().reset_index(name='CNT').GENE.value_counts()

    ann = df_mut_cds.ANNOT
    sets = {'SYN': ann == 'Synonymous', 'MIS': ann == 'Missense', 'NONS': ann == 'Nonsense',
            'SPL': ann == 'Essential_Splice', 'TRUNC': ann.isin(['Nonsense', 'Essential_Splice']),
            'NONSYN': ann.isin(['Missense', 'Nonsense', 'Essential_Splice']), 'INDEL': ann == 'INDEL'}
    for name, mask in sets.items():
        col = 'N_SAMP_' + name
        df_model[col] = 0
        cnt = _n_samp(mask)
        cnt = cnt[cnt.index.isin(df_model.index)]
        df_model.loc[cnt.index, col] = cnt
    df_model.THETA = df_model.THETA * cj
    return df_model


def transfer_element_model_with_indels(df_mut_tab, df_pretrain, cj, use_chrom=False):
    """transfer_tools.py:272-302"""
    if use_chrom:
        cols_left = ['CHROM', 'R_OBS', 'MU', 'SIGMA', 'ALPHA', 'THETA', 'Pi_SUM']
    else:
        cols_left = ['ELT_SIZE', 'FLAG', 'R_SIZE', 'R_OBS', 'R_INDEL', 'MU', 'SIGMA', 'ALPHA', 'THETA',
                     'MU_INDEL', 'SIGMA_INDEL', 'ALPHA_INDEL', 'THETA_INDEL', 'Pi_SUM', 'Pi_INDEL']
    cols_right = ['OBS_SAMPLES', 'OBS_SNV', 'OBS_INDEL']
    df_model = df_pretrain[cols_left].merge(df_mut_tab[cols_right], left_index=True, right_index=True, how='left')
    for c in ('OBS_SNV', 'OBS_INDEL', 'OBS_SAMPLES'):
        df_model[c] = df_model[c].fillna(0)
    df_model.THETA = df_model.THETA * cj
    return df_model


def transfer_element_model(df_mut_tab, df_pretrain, cj, use_chrom=False):
    """transfer_tools.py:304-329"""
    cols_left = (['CHROM'] if use_chrom else []) + ['R_OBS', 'MU', 'SIGMA', 'ALPHA', 'THETA', 'Pi_SUM']
    df_model = df_pretrain[cols_left].merge(df_mut_tab[['OBS_SAMPLES', 'OBS_SNV']], left_index=True, right_index=True,
                                            how='left')
    for c in ('OBS_SNV', 'OBS_SAMPLES'):
        df_model[c] = df_model[c].fillna(0)
    df_model.THETA = df_model.THETA * cj
    return df_model


# ---------------------------------------------------------------------------------------------
# expected counts and burden p-values (HIP)
# ---------------------------------------------------------------------------------------------
def _col(df, name):
    return np.ascontiguousarray(df[name].values, dtype=np.float64)


def _p(df, pi_col):
    with np.errstate(all="ignore"):
        return 1 / (_col(df, 'THETA') * _col(df, pi_col) + 1)


def gene_expected_muts_nb(df_model):
    """transfer_tools.py:331-340"""
    for c in ('SYN', 'MIS', 'NONS', 'SPL', 'TRUNC', 'NONSYN'):
        df_model['EXP_' + c] = df_model.ALPHA * df_model.THETA * df_model['Pi_' + c]
    return df_model


def element_expected_muts_nb(df_model):
    """transfer_tools.py:342-345"""
    df_model['EXP_SNV'] = df_model.ALPHA * df_model.THETA * df_model.Pi_SUM
    return df_model


def gene_pvalue_burden_nb(df_model):
    """transfer_tools.py:394-456: six mid-p tests per gene (one launch over the stacked classes)."""
    classes = ('SYN', 'MIS', 'NONS', 'SPL', 'TRUNC', 'NONSYN')
    k = np.stack([_col(df_model, 'OBS_' + c) for c in classes])
    p = np.stack([_p(df_model, 'Pi_' + c) for c in classes])
    a = np.broadcast_to(_col(df_model, 'ALPHA'), k.shape)
    pv = nb_model.nb_pvalue_greater_midp(k, a, p)
    for i, c in enumerate(classes):
        df_model['PVAL_%s_BURDEN' % c] = pv[i]
    return df_model


def gene_pvalue_burden_nb_by_sample(df_model):
    """transfer_tools.py:484-592"""
    classes = ('SYN', 'MIS', 'NONS', 'SPL', 'TRUNC', 'NONSYN')
    k = np.stack([_col(df_model, 'N_SAMP_' + c) for c in classes])
    p = np.stack([_p(df_model, 'Pi_' + c) for c in classes])
    a = np.broadcast_to(_col(df_model, 'ALPHA'), k.shape)
    pv = nb_model.nb_pvalue_greater_midp(k, a, p)
    for i, c in enumerate(classes):
        df_model['PVAL_%s_BURDEN_SAMPLE' % c] = pv[i]
    return df_model


def element_pvalue_burden_nb(df_model):
    """transfer_tools.py:473-482"""
    df_model['PVAL_SNV_BURDEN'] = nb_model.nb_pvalue_greater_midp(_col(df_model, 'OBS_SNV'), _col(df_model, 'ALPHA'),
                                                                 _p(df_model, 'Pi_SUM'))
    return df_model


def element_pvalue_burden_nb_by_sample(df_model):
    """transfer_tools.py:594-615"""
    df_model['PVAL_SAMPLE_BURDEN'] = nb_model.nb_pvalue_greater_midp(_col(df_model, 'OBS_SAMPLES'),
                                                                    _col(df_model, 'ALPHA'), _p(df_model, 'Pi_SUM'))
    return df_model


def _indel_block(df_model, t_indel):
    df_model['THETA_INDEL'] = df_model.THETA_INDEL * t_indel
    df_model['EXP_INDEL'] = df_model.ALPHA_INDEL * df_model.THETA_INDEL * df_model.Pi_INDEL
    with np.errstate(all="ignore"):
        p = 1 / (_col(df_model, 'THETA_INDEL') * _col(df_model, 'Pi_INDEL') + 1)
    df_model['PVAL_INDEL_BURDEN'] = nb_model.nb_pvalue_greater_midp(_col(df_model, 'OBS_INDEL'),
                                                                   _col(df_model, 'ALPHA_INDEL'), p)
    return df_model


def gene_pvalue_indel(df_model, all_cosmic=None):
    """transfer_tools.py:709-729.  `all_cosmic` defaults to the packaged CGC panel + the two CDKN2A isoforms."""
    if all_cosmic is None:
        all_cosmic = _read_gene_panel('CGC_ALL') + ['CDKN2A.p14arf', 'CDKN2A.p16INK4a']
    null = df_model[~df_model.index.isin(all_cosmic)]
    exp_unif = (null.Pi_INDEL * null.ALPHA_INDEL * null.THETA_INDEL).sum()
    t_indel = null.OBS_INDEL.sum() / exp_unif
    return _indel_block(df_model, t_indel)


def element_pvalue_indel(df_model, t_indel):
    """transfer_tools.py:731-747"""
    return _indel_block(df_model, t_indel)


def combine_snv_indel(df_model, snv_col):
    """Fisher combination written inline in the reference (transfer_tools.py:860-861, 1086-1087)."""
    df_model['PVAL_MUT_BURDEN'] = nb_model.fisher_combine(_col(df_model, snv_col), _col(df_model, 'PVAL_INDEL_BURDEN'))
    return df_model


# ---------------------------------------------------------------------------------------------
# run_* drivers
# ---------------------------------------------------------------------------------------------
def run_gene_model(f_mut, f_h5_genemodel, scale_by_sample=False, pval_burden_nb=True, pval_burden_dnds=True,
                   pval_sel=True, max_muts_per_sample=3e9, max_muts_per_gene_per_sample=3e9, scale_factor=None,
                   scale_by_expectation=True, cgc_genes=False, all_cosmic=None):
    """transfer_tools.py:789-874"""
    df_pretrain = load_pretrained_model(f_h5_genemodel, restrict_cols=True)
    df_mut = read_mutations_cds(f_mut)
    if cgc_genes:
        genes = _read_gene_panel(cgc_genes)
        df_pretrain = df_pretrain[df_pretrain.index.isin(genes)]
        df_mut = df_mut[df_mut.GENE.isin(genes)]
    df_mut = mutation_tools.filter_hypermut_samples(df_mut, max_muts_per_sample)
    df_cnt = mutation_tools.mutations_per_gene(df_mut, max_muts_per_gene_per_sample=max_muts_per_gene_per_sample)
    if scale_by_expectation:
        print('scaling by expected synonymous mutations (excluding TP53)')
        not_tp53 = df_pretrain[df_pretrain.index != 'TP53']
        exp_mut = (not_tp53.MU * not_tp53.Pi_SYN).sum()
        cj = len(df_mut[(df_mut.GENE != 'TP53') & (df_mut.ANNOT == 'Synonymous')]) / exp_mut
    elif scale_factor:
        cj = scale_factor
    elif scale_by_sample:
        cj = calc_scale_factor(df_mut, f_h5_genemodel, scale_type='sample')
    else:
        cj = calc_scale_factor(df_mut, f_h5_genemodel, scale_type='exome')
    print("\tScaling factor is: {}".format(cj))
    df_model = transfer_gene_model(df_mut, df_cnt, df_pretrain, cj)
    df_model = gene_expected_muts_nb(df_model)
    if pval_burden_nb:
        print("\tCalculating burden p-values")
        df_model = gene_pvalue_burden_nb(df_model)
        df_model = gene_pvalue_burden_nb_by_sample(df_model)
    if df_model.OBS_INDEL.sum() != 0:
        print("\tCalculating indel burden p-values")
        df_model = gene_pvalue_indel(df_model, all_cosmic=all_cosmic)
        df_model = combine_snv_indel(df_model, 'PVAL_TRUNC_BURDEN')
    return df_model


def run_target_model(f_mut, f_h5_genemodel, scale_by_sample=False, panel="MSK_341", max_muts_per_sample=3e9,
                     max_muts_per_gene_per_sample=3e9, drop_synonymous=True, cgc_genes=False, scale_factor=None):
    """transfer_tools.py:876-967"""
    print(panel)
    genes1 = np.array(_read_gene_panel(panel))
    genes = np.array(_read_gene_panel(cgc_genes)) if cgc_genes else genes1
    df_mut = read_mutations_cds(f_mut)
    df_mut = df_mut[df_mut.GENE.isin(genes)]
    if drop_synonymous:
        df_mut = df_mut[df_mut.ANNOT != 'Synonymous']
    df_mut, sample_blacklist = mutation_tools.filter_hypermut_samples(df_mut, max_muts_per_sample, return_blacklist=True)
    df_cnt = mutation_tools.mutations_per_gene(df_mut, max_muts_per_gene_per_sample=max_muts_per_gene_per_sample)
    df_pretrain = load_pretrained_model(f_h5_genemodel)
    df_pretrain = df_pretrain.loc[df_pretrain.index.isin(genes), :]
    print(len(df_pretrain))
    df_dd = mutation_tools.read_mutation_file(f_mut, drop_duplicates=True)
    df_dd = df_dd[~df_dd.SAMPLE.isin(sample_blacklist)]
    df_dd = df_dd[(df_dd.ANNOT != 'Noncoding') & (df_dd.ANNOT != 'Synonymous') & (df_dd.ANNOT != 'Essential_Splice')]
    print(f_mut, df_dd.shape)
    df_dd = df_dd[df_dd.GENE.isin(genes1)]
    n_mut, n_sample = len(df_dd), len(df_dd.SAMPLE.unique())
    attrs = mapfile.read_attrs(f_h5_genemodel)
    if scale_factor:
        cj = scale_factor
    elif scale_by_sample:
        print(n_sample, attrs['N_SAMPLE_{}'.format(panel)])
        cj = n_sample / attrs['N_SAMPLE_{}'.format(panel)]
    else:
        cj = n_mut / attrs['N_MUT_{}'.format(panel)]
    print("\tScaling factor is: {}".format(cj))
    df_model = transfer_gene_model(df_mut, df_cnt, df_pretrain, cj)
    df_model = df_model.loc[df_model.index.isin(genes), :]
    df_model = gene_expected_muts_nb(df_model)
    df_model = gene_pvalue_burden_nb(df_model)
    df_model = gene_pvalue_burden_nb_by_sample(df_model)
    return df_model


def element_statistics_block(df_model, cj, cj_indel, skip_pvals=False):
    """The statistics block of run_element_region_model (transfer_tools.py:1069-1094) as ONE fused launch
    (dig_element_stats): EXP_SNV, both SNV tests and -- when the cohort has indels -- the indel test and the
    Fisher combination.  `df_model` comes from transfer_element_model_with_indels(..., cj): its THETA column is
    already scaled; the kernel redoes theta = sigma^2/mu * cj from MU/SIGMA with the same IEEE operations."""
    df_model = element_expected_muts_nb(df_model)
    if skip_pvals:
        return df_model
    have_indel = df_model.OBS_INDEL.sum() != 0
    obs = [np.ascontiguousarray(df_model[c].values, dtype=np.int32) for c in ('OBS_SNV', 'OBS_SAMPLES', 'OBS_INDEL')]
    res = engine.element_stats(_col(df_model, 'MU'), _col(df_model, 'SIGMA'), _col(df_model, 'Pi_SUM'),
                               _col(df_model, 'Pi_INDEL'), obs[0], obs[1], obs[2], np.array([float(cj)]),
                               np.array([float(cj_indel) if have_indel else 1.0]),
                               mu_indel=_col(df_model, 'MU_INDEL'), sigma_indel=_col(df_model, 'SIGMA_INDEL'))
    df_model['PVAL_SNV_BURDEN'] = res['PVAL_SNV_BURDEN'][:, 0]
    df_model['PVAL_SAMPLE_BURDEN'] = res['PVAL_SAMPLE_BURDEN'][:, 0]
    if have_indel:
        print("\tCalculating indel burden p-values")
        df_model['THETA_INDEL'] = res['THETA_INDEL'][:, 0]
        df_model['EXP_INDEL'] = res['EXP_INDEL'][:, 0]
        df_model['PVAL_INDEL_BURDEN'] = res['PVAL_INDEL_BURDEN'][:, 0]
        df_model['PVAL_MUT_BURDEN'] = res['PVAL_MUT_BURDEN'][:, 0]
    return df_model


def run_element_region_model(f_mut, f_bed, f_h5_pretrain, pretrain_key, scale_factor=None, scale_factor_indel=None,
                             scale_type="genome", scale_by_expectation=True, max_muts_per_sample=3e9,
                             max_muts_per_elt_per_sample=3e9, skip_pvals=False, all_cosmic=None, fused=False):
    """transfer_tools.py:969-1096.  `fused=True` computes the statistics block with the single fused kernel
    (element_statistics_block) instead of the reference's column-by-column sequence; results are identical."""
    df_pretrain = load_pretrained_model(f_h5_pretrain, key=pretrain_key, restrict_cols=True)
    print('Tabulating mutations')
    df_mut_tab, blacklist = mutation_tools.tabulate_mutations_in_element(
        f_mut, f_bed, bed12=True, drop_duplicates=True, max_muts_per_sample=max_muts_per_sample,
        max_muts_per_elt_per_sample=max_muts_per_elt_per_sample, return_blacklist=True)
    if scale_by_expectation:
        print('scaling by expected number of mutations')
        df_gene = load_pretrained_model(f_h5_pretrain)
        df_mut = read_mutations_cds(f_mut)
        df_mut = df_mut[~df_mut.SAMPLE.isin(blacklist)]
        df_syn = df_mut[(df_mut.ANNOT == 'Synonymous') & (df_mut.GENE != 'TP53')].drop_duplicates()
        not_tp53 = df_gene[df_gene.index != 'TP53']
        cj = len(df_syn) / (not_tp53.MU * not_tp53.Pi_SYN).sum()
        if all_cosmic is None:
            all_cosmic = _read_gene_panel('CGC_ALL') + ['CDKN2A.p14arf', 'CDKN2A.p16INK4a']
        df_gene_null = df_gene[~df_gene.index.isin(all_cosmic)]
        # the reference filters the mutation frame on its integer ROW INDEX here (transfer_tools.py:1014), i.e. the
        # CGC exclusion is a no-op for the observed indel count; reproduced as is
        df_mut_null = df_mut[~df_mut.index.isin(all_cosmic)]
        exp_indel = (df_gene_null.Pi_INDEL * df_gene_null.ALPHA_INDEL * df_gene_null.THETA_INDEL).sum()
        cj_indel = len(df_mut_null[df_mut_null.ANNOT == 'INDEL']) / exp_indel
    elif scale_type == 'PCAWG_cds':
        assert (pretrain_key == 'PCAWG_cds'), \
            "ERROR: can only scale by PCAWG_cds if the loaded reference model is PCAWG_cds. Specify <KEY> as \"PCAWG_cds\" and rerun."
        if all_cosmic is None:
            all_cosmic = _read_gene_panel('CGC_ALL') + ['CDKN2A.p14arf', 'CDKN2A.p16INK4a']
        df_pretrain['GENE'] = [elt.split('::')[2] for elt in df_pretrain.index]
        null = df_pretrain[~df_pretrain.GENE.isin(all_cosmic)]
        df_mut_tab['GENE'] = [elt.split('::')[2] for elt in df_mut_tab.index]
        tab_null = df_mut_tab[~df_mut_tab.GENE.isin(all_cosmic)]
        cj = tab_null.OBS_SNV.sum() / (null.MU * null.Pi_SUM).sum()
        cj_indel = tab_null.OBS_INDEL.sum() / (null.MU_INDEL * null.Pi_INDEL).sum()
    elif scale_factor:
        cj, cj_indel = scale_factor, scale_factor_indel
    else:
        print('Calculating scale factor')
        cj, cj_indel = calc_scale_factor_efficient(f_mut, f_h5_pretrain, scale_type=scale_type)
    print("\tScale factor is: {}".format(cj))
    print("\tINDEL scale factor is: {}".format(cj_indel))
    df_model = transfer_element_model_with_indels(df_mut_tab, df_pretrain, cj)
    print('Calculating statistics')
    if fused:
        return element_statistics_block(df_model, cj, cj_indel, skip_pvals=skip_pvals)
    df_model = element_expected_muts_nb(df_model)
    if not skip_pvals:
        df_model = element_pvalue_burden_nb(df_model)
        df_model = element_pvalue_burden_nb_by_sample(df_model)
        if df_model.OBS_INDEL.sum() != 0:
            print("\tCalculating indel burden p-values")
            df_model = element_pvalue_indel(df_model, cj_indel)
            df_model = combine_snv_indel(df_model, 'PVAL_SNV_BURDEN')
    return df_model


def run_sites_region_model(f_mut, f_sites, f_h5_pretrain, pretrain_key, scale_factor=None, scale_type="genome",
                           scale_by_expectation=True):
    """transfer_tools.py:1098-1173: the element model of a SITES set (pretrained from preprocess_sites) against the
    mutations that hit those sites exactly; SNVs only.  In the genome mode the reference assigns the (snv, indel) TUPLE
    of calc_scale_factor_efficient to cj (:1155) and fails at THETA * cj; the SNV factor is used here."""
    df_pretrain = load_pretrained_model(f_h5_pretrain, key=pretrain_key, restrict_cols=True)
    if scale_by_expectation:
        print('scaling by expected synonymous mutations (excluding TP53)')
        df_gene = load_pretrained_model(f_h5_pretrain)
        df_mut = mutation_tools.read_mutation_file(f_mut, drop_duplicates=False)
        not_tp53 = df_gene[df_gene.index != 'TP53']
        cj = len(df_mut[(df_mut.GENE != 'TP53') & (df_mut.ANNOT == 'Synonymous')]) / (not_tp53.MU * not_tp53.Pi_SYN).sum()
    elif scale_factor:
        cj = scale_factor
    elif scale_type == 'MSK_230':
        print('Scaling by samples in MSK 230 gene subset.')
        genes = _read_gene_panel('MSK_230')
        dd = mutation_tools.read_mutation_file(f_mut, drop_duplicates=True)
        dd = dd[(dd.ANNOT != 'Noncoding') & (dd.ANNOT != 'Synonymous') & (dd.ANNOT != 'Essential_Splice')]
        dd = dd[dd.GENE.isin(genes)]
        cj = len(dd.SAMPLE.unique()) / mapfile.read_attrs(f_h5_pretrain)['N_SAMPLE_MSK_230']
    else:
        print('Calculating scale factor')
        cj = calc_scale_factor_efficient(f_mut, f_h5_pretrain, scale_type=scale_type)[0]
    print("\tScale factor is: {}".format(cj))
    print('Tabulating mutations')
    df_mut_tab = mutation_tools.tabulate_sites_in_element(f_sites, f_mut)
    df_model = transfer_element_model(df_mut_tab, df_pretrain, cj, use_chrom=False)
    print('Calculating statistics')
    df_model = element_expected_muts_nb(df_model)
    df_model = element_pvalue_burden_nb(df_model)
    df_model = element_pvalue_burden_nb_by_sample(df_model)
    return df_model
