"""Transfer a pretrained DIG model to a new cohort and test elements / genes for mutation burden.

Host side of the reference's driver_model/transfer_tools.py behind the same function names, arguments, frame columns
and console messages, re-organised around the batched engine:

    CohortRun            one cohort against one pretrained map: mutation rows, gene model and map attributes are read
                         once and shared by whatever the driver needs (the reference re-reads the file per step)
    scale factors        every way the reference derives cj / cj_indel (expected synonymous count, uniform indel rate,
                         genome / exome / sample ratios, panels, PCAWG coding elements, manual) is a CohortRun method;
                         the four run_* drivers only choose one
    statistics           dig_element_stats for the element block (one launch), dig_nb_midp_upper over the stacked
                         mutation classes for genes, dig_fisher for the combination
pandas is used for the joins and the integer bookkeeping only.  There is no CPU path for the statistics.
"""
import importlib.util
import os

import numpy as np
import pandas as pd

from .. import engine
from ..data_tools import mutation_tools
from ..io import mapfile
from ..sequence_model import nb_model

GENE_CLASSES = ('SYN', 'MIS', 'NONS', 'SPL', 'TRUNC', 'NONSYN')
_ANNOTS = {'SYN': ('Synonymous',), 'MIS': ('Missense',), 'NONS': ('Nonsense',), 'SPL': ('Essential_Splice',),
           'TRUNC': ('Nonsense', 'Essential_Splice'), 'NONSYN': ('Missense', 'Nonsense', 'Essential_Splice'), 'INDEL': ('INDEL',)}
_RATE_COLS = ['MU', 'SIGMA', 'ALPHA', 'THETA']
_INDEL_RATE_COLS = [c + '_INDEL' for c in _RATE_COLS]
_GENE_COLS_LEFT = (['CHROM', 'GENE_LENGTH', 'R_SIZE', 'R_OBS', 'R_INDEL'] + _RATE_COLS + _INDEL_RATE_COLS + ['FLAG']
                   + ['Pi_' + c for c in GENE_CLASSES] + ['Pi_INDEL'])
_ELT_COLS_LEFT = ['ELT_SIZE', 'FLAG', 'R_SIZE', 'R_OBS', 'R_INDEL'] + _RATE_COLS + _INDEL_RATE_COLS + ['Pi_SUM', 'Pi_INDEL']
_COSMIC_EXTRA = ['CDKN2A.p14arf', 'CDKN2A.p16INK4a']          # added to the CGC panel wherever the reference uses it

# ---------------------------------------------------------------------------------------------
# gene panels (the reference ships them as package data: DIGDriver/data/genes_<name>.txt, transfer_tools.py:694,711,883)
# ---------------------------------------------------------------------------------------------
_PANEL_DIRS = []



def _say(*message):
    """Progress lines on stdout, worded as the reference words them (scripts that parse its log keep working)."""
    print(*message, flush=True)

def set_panel_dir(path):
    """Directory searched FIRST for genes_<name>.txt (the --panel-dir option of scripts/DigDriver.py)."""
    if path and path not in _PANEL_DIRS:
        _PANEL_DIRS.insert(0, os.fspath(path))


def _panel_search_path():
    here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    dirs = list(_PANEL_DIRS)
    if os.environ.get("DIG_DATA_DIR"):
        dirs.append(os.environ["DIG_DATA_DIR"])
    dirs.append(os.path.join(here, "data"))
    try:                                            # an installed reference package brings its panels along
        spec = importlib.util.find_spec("DIGDriver")
        if spec is not None and spec.submodule_search_locations:
            dirs.append(os.path.join(list(spec.submodule_search_locations)[0], "data"))
    except (ImportError, ValueError):
        pass
    return dirs


def gene_panel(name):
    """Gene symbols of panel `name` (CGC_ALL, MSK_341, ...): first genes_<name>.txt found in --panel-dir, $DIG_DATA_DIR,
    digdriver_amd/data/ or the data directory of an installed DIGDriver package."""
    fname = 'genes_{}.txt'.format(name)
    tried = _panel_search_path()
    for d in tried:
        path = os.path.join(d, fname)
        if os.path.exists(path):
            return pd.read_table(path, names=['GENE']).GENE.to_list()
    raise FileNotFoundError("gene panel file {} was not found (searched: {}).  The panels ship with the reference as "
                            "DIGDriver/data/; point --panel-dir or $DIG_DATA_DIR at a directory holding them, or pass "
                            "manual scale factors / --scale-type genome.".format(fname, ", ".join(tried)))


_read_gene_panel = gene_panel          # (older name, used by onthefly_tools and the tests)


def cosmic_null_set(all_cosmic=None):
    """Genes excluded from the uniform indel rate: the CGC panel plus the two CDKN2A isoforms (transfer_tools.py:711-713)."""
    return list(all_cosmic) if all_cosmic is not None else gene_panel('CGC_ALL') + _COSMIC_EXTRA


def load_pretrained_model(h5, key='genic_model', restrict_cols=True):
    """transfer_tools.py:11-76: the frame `key` of a pretrained map with Gamma parameters added, P_* renamed to Pi_* and,
    by default, only the columns the drivers use.  Three shapes: the gene model, element models with indel columns,
    element models without."""
    model = mapfile.read_frame(h5, key)
    model['ALPHA'], model['THETA'] = nb_model.normal_params_to_gamma(model.MU.values, model.SIGMA.values)
    is_gene = key == 'genic_model'
    has_indel = is_gene or 'P_INDEL' in model.columns
    model = model.set_index(model.GENE if is_gene else model.ELT)
    if is_gene:
        model = model.rename(columns={'P_SILENT': 'Pi_SYN', 'P_MIS': 'Pi_MIS', 'P_NONS': 'Pi_NONS', 'P_SPLICE': 'Pi_SPL',
                                      'P_TRUNC': 'Pi_TRUNC', 'P_INDEL': 'Pi_INDEL'})
        model['Pi_NONSYN'] = model.Pi_MIS + model.Pi_TRUNC
    else:
        model = model.rename(columns={'P_SUM': 'Pi_SUM', 'P_INDEL': 'Pi_INDEL'})
    if has_indel:
        model['ALPHA_INDEL'], model['THETA_INDEL'] = nb_model.normal_params_to_gamma(model.MU_INDEL.values,
                                                                                      model.SIGMA_INDEL.values)
    if not restrict_cols:
        return model
    if is_gene:
        return model[_GENE_COLS_LEFT]
    return model[_ELT_COLS_LEFT] if has_indel else model[['R_OBS'] + _RATE_COLS + ['Pi_SUM']]


def read_mutations_cds(f_mut, f_cds=None):
    """transfer_tools.py:78-92: the rows of a cohort file that carry a gene label, optionally restricted to a CDS bed."""
    rows = mutation_tools.read_mutation_file(f_mut, drop_duplicates=False, drop_sex=True)
    rows = rows.loc[rows.GENE != '.']
    if f_cds:
        cds = pd.read_table(f_cds, names=['CHROM', 'START', 'END', 'GENE'], low_memory=False)
        rows = mutation_tools.restrict_mutations_by_bed(rows, cds, unique=True, replace_cols=True, remove_X=False)
    return rows


# ---------------------------------------------------------------------------------------------
# one cohort against one pretrained map
# ---------------------------------------------------------------------------------------------
class CohortRun:
    """Inputs of a driver run, each read once on first use, and every rule the reference has for the cohort scale
    factors cj (SNV) and cj_indel."""

    def __init__(self, f_mut, f_map):
        self.f_mut, self.f_map = f_mut, f_map
        self._cache = {}

    def _once(self, name, make):
        if name not in self._cache:
            self._cache[name] = make()
        return self._cache[name]

    # ---- inputs ----
    @property
    def attrs(self):
        return self._once('attrs', lambda: mapfile.read_attrs(self.f_map))

    def coding_rows(self):
        return self._once('cds', lambda: read_mutations_cds(self.f_mut))

    def unique_rows(self):
        return self._once('dedup', lambda: mutation_tools.read_mutation_file(self.f_mut, drop_duplicates=True))

    def gene_model(self):
        return self._once('genes', lambda: load_pretrained_model(self.f_map))

    # ---- scale factors ----
    @staticmethod
    def synonymous_scale(gene_model, rows, dedup=False):
        """Observed over expected synonymous mutations, TP53 left out on both sides (transfer_tools.py:809-823,1002-1008)."""
        syn = rows.loc[(rows.ANNOT == 'Synonymous') & (rows.GENE != 'TP53')]
        if dedup:
            syn = syn.drop_duplicates()
        background = gene_model.loc[gene_model.index != 'TP53']
        return len(syn) / (background.MU * background.Pi_SYN).sum()

    @staticmethod
    def uniform_indel_scale(gene_model, rows, all_cosmic):
        """Observed indels over the indel expectation of the genes outside the CGC panel (:1010-1017).  The reference
        removes panel genes from the MUTATION frame with `.index.isin(all_cosmic)` -- a test on the integer row labels,
        which never matches a gene symbol: every indel row counts.  Kept (and pinned by tests/golden/run_element_*)."""
        null_genes = gene_model.loc[~gene_model.index.isin(all_cosmic)]
        counted = rows.loc[~rows.index.isin(all_cosmic)]
        expected = (null_genes.Pi_INDEL * null_genes.ALPHA_INDEL * null_genes.THETA_INDEL).sum()
        return len(counted.loc[counted.ANNOT == 'INDEL']) / expected

    def ratio_scale(self, rows, scale_type):
        return calc_scale_factor(rows, self.f_map, scale_type=scale_type)

    def panel_scale(self, panel, counted_genes, blacklist, by_sample):
        """Mutations (or samples) of the cohort inside a sequencing panel over the pretrained cohort's (:905-935): unique
        coding, non-synonymous, non-splice rows of samples that are not blacklisted."""
        rows = self.unique_rows()
        rows = rows.loc[~rows.SAMPLE.isin(blacklist)]
        rows = rows.loc[~rows.ANNOT.isin(['Noncoding', 'Synonymous', 'Essential_Splice'])]
        _say(self.f_mut, rows.shape)
        rows = rows.loc[rows.GENE.isin(counted_genes)]
        if by_sample:
            n_pre = self.attrs['N_SAMPLE_{}'.format(panel)]
            _say(rows.SAMPLE.nunique(), n_pre)
            return rows.SAMPLE.nunique() / n_pre
        return len(rows) / self.attrs['N_MUT_{}'.format(panel)]

    def genome_scale(self, scale_type):
        return calc_scale_factor_efficient(self.f_mut, self.f_map, scale_type=scale_type)


def calc_scale_factor(df_mut, h5_pretrain, scale_type='genome'):
    """transfer_tools.py:94-127: unique mutations of the cohort over the pretrained cohort's, counted three ways."""
    unique = mutation_tools.drop_duplicate_mutations(df_mut)
    attrs = mapfile.read_attrs(h5_pretrain)
    if scale_type == 'sample':
        return unique.SAMPLE.nunique() / attrs['N_SAMPLES']
    if scale_type == 'exome':
        return int((unique.ANNOT != 'Noncoding').sum()) / attrs['N_MUT_CDS']
    if scale_type != 'genome':
        raise ValueError("scale_type {} is not recognized".format(scale_type))
    bins = mapfile.read_array(h5_pretrain, 'idx')
    mappable = bins[mapfile.read_array(h5_pretrain, 'mappability') > attrs['mappability_threshold']]
    inside = mutation_tools.restrict_mutations_by_bed(unique, pd.DataFrame(mappable, columns=['CHROM', 'START', 'END']),
                                                      remove_X=False)
    return len(inside) / attrs['N_MUT_TRAIN']


def calc_scale_factor_efficient(f_mut, h5_pretrain, scale_type='genome'):
    """transfer_tools.py:129-159: (cj_snv, cj_indel) = (SNVs, indels of the cohort inside unflagged bins) over the summed
    predicted rate of those bins; the masked sum runs in dig_scale_suffstats."""
    if scale_type != 'genome':
        raise ValueError("scale_type {} is not recognized".format(scale_type))
    import tempfile
    bins = mapfile.read_frame(h5_pretrain, 'region_params')
    flagged = bins.FLAG.values.astype(bool)
    handle, f_bins = tempfile.mkstemp(suffix=".bed")
    os.close(handle)
    try:
        bins.loc[~flagged, ['CHROM', 'START', 'END']].to_csv(f_bins, sep="\t", header=False, index=False)
        inside = mutation_tools.restrict_mutations_by_bed_efficient(f_mut, f_bins, bed12=False, drop_duplicates=True)
    finally:
        os.remove(f_bins)
    expected = float(engine.scale_suffstats(bins.Y_PRED.values[:, None], flagged.astype(np.uint8)[:, None])[0])
    n_indel = int((inside.ANNOT == 'INDEL').sum())
    return (len(inside) - n_indel) / expected, n_indel / expected


# ---------------------------------------------------------------------------------------------
# joins (integer bookkeeping, pandas like the reference)
# ---------------------------------------------------------------------------------------------
def _attach_counts(model, counts, count_cols, cj):
    """Left-join observed counts onto model rows (missing = 0) and scale THETA by the cohort factor."""
    out = model.merge(counts[count_cols], left_index=True, right_index=True, how='left')
    out[count_cols] = out[count_cols].fillna(0)
    out['THETA'] = out.THETA * cj
    return out


def transfer_gene_model(df_mut_cds, df_counts, df_pretrain, cj):
    """transfer_tools.py:196-270: observed counts per mutation class and the number of distinct samples per class."""
    out = _attach_counts(df_pretrain[_GENE_COLS_LEFT], df_counts, ['OBS_SYN', 'OBS_MIS', 'OBS_NONS', 'OBS_SPL', 'OBS_INDEL'], 1.0)
    out['OBS_TRUNC'] = out.OBS_NONS + out.OBS_SPL
    out['OBS_NONSYN'] = out.OBS_MIS + out.OBS_TRUNC
    pairs = df_mut_cds[['GENE', 'SAMPLE', 'ANNOT']]
    for cls, annots in _ANNOTS.items():
        hit = pairs.loc[pairs.ANNOT.isin(annots), ['GENE', 'SAMPLE']].drop_duplicates()
        per_gene = hit.GENE.value_counts()
        out['N_SAMP_' + cls] = per_gene.reindex(out.index).fillna(0).astype(int)
    out['THETA'] = out.THETA * cj
    return out


def transfer_element_model_with_indels(df_mut_tab, df_pretrain, cj, use_chrom=False):
    """transfer_tools.py:272-302"""
    left = ['CHROM', 'R_OBS'] + _RATE_COLS + ['Pi_SUM'] if use_chrom else _ELT_COLS_LEFT
    return _attach_counts(df_pretrain[left], df_mut_tab, ['OBS_SAMPLES', 'OBS_SNV', 'OBS_INDEL'], cj)


def transfer_element_model(df_mut_tab, df_pretrain, cj, use_chrom=False):
    """transfer_tools.py:304-329"""
    left = (['CHROM'] if use_chrom else []) + ['R_OBS'] + _RATE_COLS + ['Pi_SUM']
    return _attach_counts(df_pretrain[left], df_mut_tab, ['OBS_SAMPLES', 'OBS_SNV'], cj)


# ---------------------------------------------------------------------------------------------
# expected counts and burden p-values (HIP)
# ---------------------------------------------------------------------------------------------
def _f64(df, name):
    return np.ascontiguousarray(df[name].values, dtype=np.float64)


def _success_prob(df, pi_col, theta_col='THETA'):
    with np.errstate(all="ignore"):
        return 1 / (_f64(df, theta_col) * _f64(df, pi_col) + 1)


def _midp_columns(df_model, count_prefix, out_pattern):
    """Six mid-p tests per gene in ONE launch over the stacked mutation classes."""
    k = np.stack([_f64(df_model, count_prefix + c) for c in GENE_CLASSES])
    p = np.stack([_success_prob(df_model, 'Pi_' + c) for c in GENE_CLASSES])
    pv = nb_model.nb_pvalue_greater_midp(k, np.broadcast_to(_f64(df_model, 'ALPHA'), k.shape), p)
    for row, c in zip(pv, GENE_CLASSES):
        df_model[out_pattern % c] = row
    return df_model


def gene_expected_muts_nb(df_model):
    """transfer_tools.py:331-340"""
    rate = df_model.ALPHA * df_model.THETA
    for c in GENE_CLASSES:
        df_model['EXP_' + c] = rate * df_model['Pi_' + c]
    return df_model


def element_expected_muts_nb(df_model):
    """transfer_tools.py:342-345"""
    df_model['EXP_SNV'] = df_model.ALPHA * df_model.THETA * df_model.Pi_SUM
    return df_model


def gene_pvalue_burden_nb(df_model):
    """transfer_tools.py:394-456"""
    return _midp_columns(df_model, 'OBS_', 'PVAL_%s_BURDEN')


def gene_pvalue_burden_nb_by_sample(df_model):
    """transfer_tools.py:484-592"""
    return _midp_columns(df_model, 'N_SAMP_', 'PVAL_%s_BURDEN_SAMPLE')


def element_pvalue_burden_nb(df_model):
    """transfer_tools.py:473-482"""
    df_model['PVAL_SNV_BURDEN'] = nb_model.nb_pvalue_greater_midp(_f64(df_model, 'OBS_SNV'), _f64(df_model, 'ALPHA'),
                                                                 _success_prob(df_model, 'Pi_SUM'))
    return df_model


def element_pvalue_burden_nb_by_sample(df_model):
    """transfer_tools.py:594-615"""
    df_model['PVAL_SAMPLE_BURDEN'] = nb_model.nb_pvalue_greater_midp(_f64(df_model, 'OBS_SAMPLES'), _f64(df_model, 'ALPHA'),
                                                                    _success_prob(df_model, 'Pi_SUM'))
    return df_model


def _indel_block(df_model, t_indel):
    df_model['THETA_INDEL'] = df_model.THETA_INDEL * t_indel
    df_model['EXP_INDEL'] = df_model.ALPHA_INDEL * df_model.THETA_INDEL * df_model.Pi_INDEL
    df_model['PVAL_INDEL_BURDEN'] = nb_model.nb_pvalue_greater_midp(_f64(df_model, 'OBS_INDEL'), _f64(df_model, 'ALPHA_INDEL'),
                                                                   _success_prob(df_model, 'Pi_INDEL', 'THETA_INDEL'))
    return df_model


def gene_pvalue_indel(df_model, all_cosmic=None):
    """transfer_tools.py:709-729: indel rate calibrated on the genes outside the CGC panel, then the indel burden test."""
    null = df_model.loc[~df_model.index.isin(cosmic_null_set(all_cosmic))]
    t_indel = null.OBS_INDEL.sum() / (null.Pi_INDEL * null.ALPHA_INDEL * null.THETA_INDEL).sum()
    return _indel_block(df_model, t_indel)


def element_pvalue_indel(df_model, t_indel):
    """transfer_tools.py:731-747"""
    return _indel_block(df_model, t_indel)


def combine_snv_indel(df_model, snv_col):
    """Fisher combination written inline in the reference (transfer_tools.py:860-861, 1086-1087)."""
    df_model['PVAL_MUT_BURDEN'] = nb_model.fisher_combine(_f64(df_model, snv_col), _f64(df_model, 'PVAL_INDEL_BURDEN'))
    return df_model


def element_statistics_block(df_model, cj, cj_indel, skip_pvals=False):
    """The statistics block of run_element_region_model (transfer_tools.py:1069-1094) as ONE fused launch
    (dig_element_stats): EXP_SNV, both SNV tests and -- when the cohort has indels -- the indel test and the
    Fisher combination.  `df_model` comes from transfer_element_model_with_indels(..., cj): its THETA column is
    already scaled; the kernel redoes theta = sigma^2/mu * cj from MU/SIGMA with the same IEEE operations."""
    df_model = element_expected_muts_nb(df_model)
    if skip_pvals:
        return df_model
    with_indels = df_model.OBS_INDEL.sum() != 0
    counts = [np.ascontiguousarray(df_model[c].values, dtype=np.int32) for c in ('OBS_SNV', 'OBS_SAMPLES', 'OBS_INDEL')]
    planes = engine.element_stats(_f64(df_model, 'MU'), _f64(df_model, 'SIGMA'), _f64(df_model, 'Pi_SUM'), _f64(df_model, 'Pi_INDEL'),
                                  *counts, np.array([float(cj)]), np.array([float(cj_indel) if with_indels else 1.0]),
                                  mu_indel=_f64(df_model, 'MU_INDEL'), sigma_indel=_f64(df_model, 'SIGMA_INDEL'))
    wanted = ['PVAL_SNV_BURDEN', 'PVAL_SAMPLE_BURDEN']
    if with_indels:
        _say("\tCalculating indel burden p-values")
        wanted += ['THETA_INDEL', 'EXP_INDEL', 'PVAL_INDEL_BURDEN', 'PVAL_MUT_BURDEN']
    for name in wanted:
        df_model[name] = planes[name][:, 0]
    return df_model


def gene_statistics_block(df_model, cj, indel=True, all_cosmic=None):
    """The statistics of the gene route -- gene_expected_muts_nb, gene_pvalue_burden_nb, gene_pvalue_burden_nb_by_sample,
    gene_pvalue_indel and the Fisher combination (transfer_tools.py:331-340,394-456,554-583,709-729,860-861) -- as ONE
    launch (dig_gene_stats) instead of four.  `df_model` comes from transfer_gene_model(..., cj) and `cj` is that same
    factor: the kernel forms theta = sigma^2 / mu * cj from MU / SIGMA with the IEEE operations that made the frame's
    THETA column.  Same columns in the same order as the column-by-column route."""
    with_indel = bool(indel and df_model.OBS_INDEL.sum() != 0)
    t_indel = None
    if with_indel:
        null = df_model.loc[~df_model.index.isin(cosmic_null_set(all_cosmic))]
        t_indel = null.OBS_INDEL.sum() / (null.Pi_INDEL * null.ALPHA_INDEL * null.THETA_INDEL).sum()      # :720-721
    pi = np.stack([_f64(df_model, 'Pi_' + c) for c in GENE_CLASSES], axis=1)[:, :, None]
    obs = np.stack([df_model[c].values for c in ('OBS_SYN', 'OBS_MIS', 'OBS_NONS', 'OBS_SPL', 'OBS_INDEL')], axis=1)[:, :, None]
    n_samp = np.stack([df_model['N_SAMP_' + c].values for c in GENE_CLASSES], axis=1)[:, :, None]
    planes = engine.gene_stats(_f64(df_model, 'MU'), _f64(df_model, 'SIGMA'), pi, _f64(df_model, 'Pi_INDEL'), obs.astype(np.int32),
                               n_samp.astype(np.int32), np.array([float(cj)]),
                               None if t_indel is None else np.array([float(t_indel)]),
                               mu_indel=_f64(df_model, 'MU_INDEL'), sigma_indel=_f64(df_model, 'SIGMA_INDEL'))
    for c in GENE_CLASSES:
        df_model['EXP_' + c] = planes['EXP_' + c][:, 0]
    for pattern in ('PVAL_%s_BURDEN', 'PVAL_%s_BURDEN_SAMPLE'):
        for c in GENE_CLASSES:
            df_model[pattern % c] = planes[pattern % c][:, 0]
    if with_indel:
        _say("\tCalculating indel burden p-values")
        for name in ('THETA_INDEL', 'EXP_INDEL', 'PVAL_INDEL_BURDEN', 'PVAL_MUT_BURDEN'):
            df_model[name] = planes[name][:, 0]
    return df_model


def _gene_statistics(df_model, burden=True, indel=True, all_cosmic=None, announce=True, fused_cj=None):
    """fused_cj: the cohort factor the frame was transferred with -> the one-launch form (gene_statistics_block); None: the
    reference's column-by-column sequence.  Same results (tests/test_gpu_host_mirror.py)."""
    if fused_cj is not None and burden:
        if announce:
            _say("\tCalculating burden p-values")
        return gene_statistics_block(df_model, fused_cj, indel=indel, all_cosmic=all_cosmic)
    df_model = gene_expected_muts_nb(df_model)
    if burden:
        if announce:
            _say("\tCalculating burden p-values")
        df_model = gene_pvalue_burden_nb_by_sample(gene_pvalue_burden_nb(df_model))
    if indel and df_model.OBS_INDEL.sum() != 0:
        _say("\tCalculating indel burden p-values")
        df_model = combine_snv_indel(gene_pvalue_indel(df_model, all_cosmic=all_cosmic), 'PVAL_TRUNC_BURDEN')
    return df_model


# ---------------------------------------------------------------------------------------------
# run_* drivers: choose the inputs and the scale-factor rule, then one shared statistics step
# ---------------------------------------------------------------------------------------------
def run_gene_model(f_mut, f_h5_genemodel, scale_by_sample=False, pval_burden_nb=True, pval_burden_dnds=True,
                   pval_sel=True, max_muts_per_sample=3e9, max_muts_per_gene_per_sample=3e9, scale_factor=None,
                   scale_by_expectation=True, cgc_genes=False, all_cosmic=None, fused=False):
    """transfer_tools.py:789-874.  `fused=True` computes the whole statistics block with the single dig_gene_stats launch
    (gene_statistics_block) instead of the reference's column-by-column sequence; same columns, same values."""
    run = CohortRun(f_mut, f_h5_genemodel)
    model, rows = run.gene_model(), run.coding_rows()
    if cgc_genes:
        keep = set(gene_panel(cgc_genes))
        model, rows = model.loc[model.index.isin(keep)], rows.loc[rows.GENE.isin(keep)]
    rows = mutation_tools.filter_hypermut_samples(rows, max_muts_per_sample)
    counts = mutation_tools.mutations_per_gene(rows, max_muts_per_gene_per_sample=max_muts_per_gene_per_sample)
    if scale_by_expectation:
        _say('scaling by expected synonymous mutations (excluding TP53)')
        cj = run.synonymous_scale(model, rows)
    elif scale_factor:
        cj = scale_factor
    else:
        cj = run.ratio_scale(rows, 'sample' if scale_by_sample else 'exome')
    _say("\tScaling factor is: {}".format(cj))
    return _gene_statistics(transfer_gene_model(rows, counts, model, cj), burden=pval_burden_nb, all_cosmic=all_cosmic,
                            fused_cj=cj if fused else None)


def run_target_model(f_mut, f_h5_genemodel, scale_by_sample=False, panel="MSK_341", max_muts_per_sample=3e9,
                     max_muts_per_gene_per_sample=3e9, drop_synonymous=True, cgc_genes=False, scale_factor=None):
    """transfer_tools.py:876-967: genes of a sequencing panel; the scale factor counts the cohort inside the panel."""
    _say(panel)
    panel_genes = gene_panel(panel)
    tested = gene_panel(cgc_genes) if cgc_genes else panel_genes
    run = CohortRun(f_mut, f_h5_genemodel)
    rows = run.coding_rows()
    rows = rows.loc[rows.GENE.isin(tested)]
    if drop_synonymous:
        rows = rows.loc[rows.ANNOT != 'Synonymous']
    rows, blacklist = mutation_tools.filter_hypermut_samples(rows, max_muts_per_sample, return_blacklist=True)
    counts = mutation_tools.mutations_per_gene(rows, max_muts_per_gene_per_sample=max_muts_per_gene_per_sample)
    model = run.gene_model()
    model = model.loc[model.index.isin(tested)]
    _say(len(model))
    cj = scale_factor if scale_factor else run.panel_scale(panel, panel_genes, blacklist, scale_by_sample)
    _say("\tScaling factor is: {}".format(cj))
    df_model = transfer_gene_model(rows, counts, model, cj)
    return _gene_statistics(df_model.loc[df_model.index.isin(tested)], indel=False, announce=False)


def run_element_region_model(f_mut, f_bed, f_h5_pretrain, pretrain_key, scale_factor=None, scale_factor_indel=None,
                             scale_type="genome", scale_by_expectation=True, max_muts_per_sample=3e9,
                             max_muts_per_elt_per_sample=3e9, skip_pvals=False, all_cosmic=None, fused=False):
    """transfer_tools.py:969-1096.  `fused=True` computes the statistics block with the single fused kernel
    (element_statistics_block) instead of the reference's column-by-column sequence; results are identical."""
    run = CohortRun(f_mut, f_h5_pretrain)
    # (the tabulation -- host work -- in front of the model: load_pretrained_model makes the process's first device call, and a
    # command line that started the HIP runtime on a thread of its own, _lib.prewarm_in_background, finds it ready by then)
    _say('Tabulating mutations')
    table, blacklist = mutation_tools.tabulate_mutations_in_element(
        f_mut, f_bed, bed12=True, drop_duplicates=True, max_muts_per_sample=max_muts_per_sample,
        max_muts_per_elt_per_sample=max_muts_per_elt_per_sample, return_blacklist=True)
    model = load_pretrained_model(f_h5_pretrain, key=pretrain_key, restrict_cols=True)
    if scale_by_expectation:
        _say('scaling by expected number of mutations')
        genes = run.gene_model()
        rows = run.coding_rows()
        rows = rows.loc[~rows.SAMPLE.isin(blacklist)]
        cj = run.synonymous_scale(genes, rows, dedup=True)
        cj_indel = run.uniform_indel_scale(genes, rows, cosmic_null_set(all_cosmic))
    elif scale_type == 'PCAWG_cds':
        assert (pretrain_key == 'PCAWG_cds'), \
            "ERROR: can only scale by PCAWG_cds if the loaded reference model is PCAWG_cds. Specify <KEY> as \"PCAWG_cds\" and rerun."
        # coding elements are named <..>::<..>::GENE: calibrate on the elements of genes outside the CGC panel (:1019-1038)
        cosmic = cosmic_null_set(all_cosmic)
        gene_of = lambda frame: pd.Index([name.split('::')[2] for name in frame.index])
        model['GENE'], table['GENE'] = gene_of(model), gene_of(table)
        null_model, null_table = model.loc[~model.GENE.isin(cosmic)], table.loc[~table.GENE.isin(cosmic)]
        cj = null_table.OBS_SNV.sum() / (null_model.MU * null_model.Pi_SUM).sum()
        cj_indel = null_table.OBS_INDEL.sum() / (null_model.MU_INDEL * null_model.Pi_INDEL).sum()
    elif scale_factor:
        cj, cj_indel = scale_factor, scale_factor_indel
    else:
        _say('Calculating scale factor')
        cj, cj_indel = run.genome_scale(scale_type)
    _say("\tScale factor is: {}".format(cj))
    _say("\tINDEL scale factor is: {}".format(cj_indel))
    df_model = transfer_element_model_with_indels(table, model, cj)
    _say('Calculating statistics')
    if fused:
        return element_statistics_block(df_model, cj, cj_indel, skip_pvals=skip_pvals)
    df_model = element_expected_muts_nb(df_model)
    if skip_pvals:
        return df_model
    df_model = element_pvalue_burden_nb_by_sample(element_pvalue_burden_nb(df_model))
    if df_model.OBS_INDEL.sum() != 0:
        _say("\tCalculating indel burden p-values")
        df_model = combine_snv_indel(element_pvalue_indel(df_model, cj_indel), 'PVAL_SNV_BURDEN')
    return df_model


def run_sites_region_model(f_mut, f_sites, f_h5_pretrain, pretrain_key, scale_factor=None, scale_type="genome",
                           scale_by_expectation=True):
    """transfer_tools.py:1098-1173: the element model of a SITES set (pretrained from preprocess_sites) against the
    mutations that hit those sites exactly; SNVs only.  In the genome mode the reference assigns the (snv, indel) TUPLE
    of calc_scale_factor_efficient to cj (:1155) and fails at THETA * cj; the SNV factor is used here."""
    run = CohortRun(f_mut, f_h5_pretrain)
    model = load_pretrained_model(f_h5_pretrain, key=pretrain_key, restrict_cols=True)
    if scale_by_expectation:
        _say('scaling by expected synonymous mutations (excluding TP53)')
        cj = run.synonymous_scale(run.gene_model(), mutation_tools.read_mutation_file(f_mut, drop_duplicates=False))
    elif scale_factor:
        cj = scale_factor
    elif scale_type == 'MSK_230':
        _say('Scaling by samples in MSK 230 gene subset.')
        cj = run.panel_scale('MSK_230', gene_panel('MSK_230'), (), by_sample=True)
    else:
        _say('Calculating scale factor')
        cj = run.genome_scale(scale_type)[0]
    _say("\tScale factor is: {}".format(cj))
    _say('Tabulating mutations')
    df_model = transfer_element_model(mutation_tools.tabulate_sites_in_element(f_sites, f_mut), model, cj, use_chrom=False)
    _say('Calculating statistics')
    return element_pvalue_burden_nb_by_sample(element_pvalue_burden_nb(element_expected_muts_nb(df_model)))
