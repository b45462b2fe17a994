"""Assemble the region_params frame from k-fold CNN+GP results.

Mirror of DIGDriver/region_model/region_model_tools.py:64-193.  Fold results are map-file containers
(digdriver_amd/io/mapfile.py) named gp_results_fold_{k}.* / sub_mapp_results_fold_{k}.* holding, per cohort,
    <cohort>/held-out/chr_locs [n,3], mappability, quantiles, y_true
    <cohort>/held-out/<run>/mean, <cohort>/held-out/<run>/std          (run = 0, 1, ...)
which is the reference's HDF5 group layout (gp_trainer.py:206-245) key for key.
"""
import glob
import os

import numpy as np
import pandas as pd

from ..io import mapfile


def _runs(path, cohort, key):
    runs = []
    r = 0
    while mapfile.has_key(path, "{}/{}/{}/mean".format(cohort, key, r)):
        runs.append(r)
        r += 1
    return runs


def _load_fold_avg(f, cancer, key='held-out', fold=None):
    """region_model_tools.py:64-100: mean over GP runs of one fold."""
    base = "{}/{}/".format(cancer, key)
    if not mapfile.has_key(f, base + "chr_locs"):
        raise KeyError('Cannot compute pretrained model with no saved held-out set in {}'.format(f))
    chr_locs = mapfile.read_array(f, base + "chr_locs")
    runs = _runs(f, cancer, key)
    means = np.mean([mapfile.read_array(f, base + "{}/mean".format(r)) for r in runs], axis=0)
    stds = np.mean([mapfile.read_array(f, base + "{}/std".format(r)) for r in runs], axis=0)
    return pd.DataFrame({'CHROM': chr_locs[:, 0], 'START': chr_locs[:, 1], 'END': chr_locs[:, 2],
                         'Y_TRUE': mapfile.read_array(f, base + "y_true"), 'Y_PRED': means, 'STD': stds,
                         'MAPP': mapfile.read_array(f, base + "mappability"),
                         'QUANT': mapfile.read_array(f, base + "quantiles")})


_TYPES = {'CHROM': int, 'START': int, 'END': int, 'Y_TRUE': int, 'Y_PRED': float, 'STD': float, 'MAPP': float,
          'QUANT': float}


def _finish(df, flag, sort, drop_pos_cols):
    df = df.astype(_TYPES)
    df['FLAG'] = flag
    df['Region'] = ['chr{}:{}-{}'.format(c, s, e) for c, s, e in zip(df.CHROM, df.START, df.END)]
    if sort:
        df = df.sort_values(by=['CHROM', 'START'])
    if drop_pos_cols:
        df = df.drop(['CHROM', 'START', 'END'], axis=1)
    return df.set_index('Region')


def _fold_files(kfold_path, stem):
    return sorted(glob.glob(os.path.join(str(kfold_path), stem + "*")))


def kfold_supmap_results(kfold_path, cancer_str, key='held-out', drop_pos_cols=False, sort=True):
    """region_model_tools.py:102-125: supra-mappability folds are concatenated, FLAG = False."""
    frames = [_load_fold_avg(f, cancer_str, key) for f in _fold_files(kfold_path, "gp_results_fold")]
    return _finish(pd.concat(frames), False, sort, drop_pos_cols)


def kfold_submap_results(kfold_path, cancer_str, key='held-out', drop_pos_cols=False, sort=True):
    """region_model_tools.py:127-167: the sub-mappability set is predicted by every fold -> averaged ACROSS
    folds (Y_PRED and STD), FLAG = True."""
    frames = [_load_fold_avg(f, cancer_str, key) for f in _fold_files(kfold_path, "sub_mapp_results_fold")]
    first = frames[0]
    df = pd.DataFrame({'CHROM': first.CHROM.values, 'START': first.START.values, 'END': first.END.values,
                       'Y_TRUE': first.Y_TRUE.values,
                       'Y_PRED': np.mean([f.Y_PRED.values for f in frames], axis=0),
                       'STD': np.mean([f.STD.values for f in frames], axis=0),
                       'MAPP': first.MAPP.values, 'QUANT': first.QUANT.values})
    return _finish(df, True, sort, drop_pos_cols)


def kfold_results(kfold_path, cohort_name, key='held-out'):
    """region_model_tools.py:169-193: merge both sets; duplicated bins are an error."""
    try:
        df_sup = kfold_supmap_results(kfold_path, cohort_name, key=key)
        df_sub = kfold_submap_results(kfold_path, cohort_name, key=key)
    except Exception as exc:
        raise Exception('ERROR: failed to load kfold {}. You should rerun the CNN+GP kfold.'.format(kfold_path)) from exc
    df = pd.concat([df_sup, df_sub]).sort_values(by=['CHROM', 'START'])
    assert len(df) == len(df.drop_duplicates(['CHROM', 'START', 'END'])), \
        "Oh snap! There are duplicate entries in the folds. You should rerun this kfold."
    return df
