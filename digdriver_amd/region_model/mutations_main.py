#!/usr/bin/env python
"""Single-split CNN + GP training of the region model on MI355X: one held-out set, `nn_reruns` re-initialisations.

Mirror of DIGDriver/region_model/mutations_main.py (OutputGenerator :98-281, main :284-446) for its live route -- the
SimpleMultiTaskResNet CNN with sparse-GP calibration.  Per rerun: a fresh train / validation split of the bins that are
not held out, the best epoch by mean validation r^2 with more than one live feature per label, the held-out prediction,
`run_gaussian` GP fits written to `gp_results_run{r}.h5` in GPTrainer.save_results' layout, and the mean over the
runs appended to `<label>.Pretrained.h5:region_params` (OutputGenerator.store_pretrained) -- the frame
`DigPretrain.py` / `DigDriver.py` read.  `<label>_gp_runs_summary.csv`, `<label>_pretrained_accuracy.txt`,
`run_params.txt`, `run_accuracies.csv` (-st) and `best_model_{r}.pt` / `ho_indices_{r}.npy` (-sm) as the reference.

Not built (the reference's experiment zoo): the attention maps (-a / -at / -ac), the fully connected
and autoregressive networks (-n fc, -as), TensorBoard logging.  The reference's store_pretrained calls
DataFrame.append, which pandas 2 removed (mutations_main.py:161); the frame it would have built is built here.

The data container is the one kfold_mutations_main reads; the track matrix stays in HBM, batches are gathered on the
device.  One process drives one GPU (-g is accepted and ignored).
"""
import argparse
import copy
import os
from datetime import datetime

import numpy as np
import pandas as pd
import torch
from torch import nn, optim

from ..io import mapfile
from .data_aux import dataset_generator as dg
from .kfold_mutations_main import run_gp_fold
from .nets.cnn_predictors import SimpleMultiTaskResNet
from .predict import predict, r2_score
from .trainers import gp_trainer
from .trainers.nn_trainer import NNTrainer, adam_for as _adam


def _say(*message):
    """Progress lines on stdout, worded as the reference words them."""
    print(*message, flush=True)


def get_cmd_arguments(text=None):
    ap = argparse.ArgumentParser(description='single-split CNN + GP region model (MI355X build)')
    ap.add_argument('-c', '--cancer-id', required=True, nargs='*', type=str, dest='label_ids',
                    help='label arrays in the data container; the best model is selected on their mean validation r2')
    ap.add_argument('-d', '--data', required=True, type=str, dest='data_file', help='training data container')
    ap.add_argument('-o', '--out-dir', required=True, type=str, dest='out_dir', help='output directory')
    ap.add_argument('-u', '--held-out', type=str, dest='heldout_file', default=None, help='file of predefined held-out windows')
    ap.add_argument('-t', '--tracks', type=str, dest='track_file', default=None, help='track selection file')
    ap.add_argument('-s', '--split', type=str, dest='split_method', default='random', help='random / chr')
    ap.add_argument('-m', '--mappability', type=float, dest='mappability', default=0.7, help='mappability lower bound')
    ap.add_argument('-cq', '--count-quantile', type=float, dest='count_quantile', default=0.995, help='count quantile cap')
    ap.add_argument('-a', '--attention', action='store_true', dest='get_attention', help='(not built)')
    ap.add_argument('-at', '--attended-tracks', action='store_true', dest='get_attended_tracks', help='(not built)')
    ap.add_argument('-ac', '--attended-columns', action='store_true', dest='get_attended_cols', help='(not built)')
    ap.add_argument('-gp', '--gaussian', type=int, dest='run_gaussian', default=0, help='number of GP fits per rerun')
    ap.add_argument('-n', '--network', type=str, dest='net', default='cnn', help="'cnn' ('fc' is not built)")
    ap.add_argument('-as', '--autoregressive-size', type=int, dest='autoregressive_size', default=0, help='(not built)')
    ap.add_argument('-vr', '--val-ratio', type=float, dest='val_ratio', default=0.2, help='validation share')
    ap.add_argument('-hr', '--heldout-ratio', type=float, dest='heldout_ratio', default=0.2, help='held-out share (taken first)')
    ap.add_argument('-e', '--epochs', type=int, dest='epochs', default=20, help='epochs')
    ap.add_argument('-b', '--batch', type=int, dest='bs', default=128, help='batch size')
    ap.add_argument('-nd', '--n-inducing', type=int, dest='n_inducing', default=400, help='GP inducing points')
    ap.add_argument('-nt', '--n-iter', type=int, dest='n_iter', default=50, help='GP iterations')
    ap.add_argument('-gr', '--gp-reruns', type=int, dest='gp_reruns', default=3, help='GP retries per inducing-point count')
    ap.add_argument('-gd', '--gp-delta', type=float, dest='gp_delta', default=0.03, help='tolerated GP-vs-CNN r2 drop')
    ap.add_argument('-re', '--nn-reruns', type=int, dest='nn_reruns', default=1, help='CNN re-initialisations')
    ap.add_argument('-mr', '--max-nn-reruns', type=int, dest='max_nn_reruns', default=2, help='CNN retrainings when the GP fails')
    ap.add_argument('-sm', '--save-model', action='store_true', dest='save_model', help='save the best model of every rerun')
    ap.add_argument('-st', '--save-training', action='store_true', dest='save_training', help='save predictions and accuracies')
    ap.add_argument('-g', '--gpus', type=str, dest='gpus', default='all', help='accepted for compatibility; one process = one GPU')
    ap.add_argument('--seed', type=int, default=0, help='seed of the splits, shuffles and initialisations')
    return ap.parse_args(text.split()) if text else ap.parse_args()


class SplitData:
    """BaseDatasetGenerator + DatasetGenerator (dataset_generator.py:16-50,84-193) over a device-resident matrix: the
    bin filters, then the held-out set (a file of windows, a random share, or the tail of every chromosome), then a
    train / validation split per call of get_datasets."""

    def __init__(self, args, device):
        _say('Loading data and labels from file {}...'.format(args.data_file))
        self.locs = np.asarray(mapfile.read_array(args.data_file, 'idx'))
        self.mapp = np.asarray(mapfile.read_array(args.data_file, 'mappability'), float)
        self.labels = [np.asarray(mapfile.read_array(args.data_file, l), float) for l in args.label_ids]
        self.quantiles = dg.rank_quantiles(self.labels[0])
        idxs, _ = dg.select_bins(self.mapp, self.labels[0], args.mappability, args.count_quantile)
        x = dg.load_track_matrix(args.data_file, device, log=_say)       # slab by slab: int16 in HBM when exact, else float32
        tracks = None
        if args.track_file is not None:
            with open(args.track_file) as f:
                tracks = dg.load_track_selection(f.readlines())
        self.store = dg.BinTrackStore(x, tracks)
        self.rng = np.random.default_rng(args.seed)
        self.val_ratio, self.split_method = args.val_ratio, args.split_method
        if args.heldout_file is not None:
            _say('Using predefined held-out samples from {}'.format(args.heldout_file))
            self.idxs, self.heldout_idxs = self.extract_heldout_set(np.asarray(idxs), args.heldout_file)
        else:
            self.idxs, self.heldout_idxs = self.split(np.asarray(idxs), args.heldout_ratio, 'held-out')
        _say('Input data is of size: {}'.format(self.store.shape(len(self.idxs))))
    def split(self, idxs, ratio, what):
        if self.split_method == 'random':                               # split_randomly, dataset_generator.py:82-88
            idxs = self.rng.permutation(idxs)
            cut = int((1 - ratio) * len(idxs))
            _say('Splitting {} data at random to {} and {} samples'.format(what, cut, len(idxs) - cut))
            return idxs[:cut], idxs[cut:]
        if self.split_method == 'chr':
            # split_by_chromosome (:90-101): the last `ratio` of every chromosome's bins.  (The reference returns the
            # POSITIONS inside `idxs` instead of the bins at those positions -- the same thing only when no bin was
            # filtered out; the bins are returned here.)
            _say('Splitting {} data by chromosome...'.format(what))
            chrom = self.locs[idxs, 0]
            head, tail = [], []
            for c in np.unique(chrom):
                rows = idxs[chrom == c]
                cut = int((1 - ratio) * len(rows))
                head.extend(rows[:cut])
                tail.extend(rows[cut:])
            return np.sort(np.asarray(head, int)), np.sort(np.asarray(tail, int))
        raise Exception("Expected split_method to be 'random' or 'chr', but found {}".format(self.split_method))

    def extract_heldout_set(self, idxs, path):
        """dataset_generator.py:134-158: a tab-separated table with a header whose first four columns are CHROM, START,
        END, Y_TRUE; every row must name one window of the filtered set with that count."""
        rows = [l.split('\t') for l in open(path).read().split('\n')[1:] if l.strip()]
        key = {(int(c), int(s)): i for i, (c, s) in enumerate(zip(self.locs[:, 0], self.locs[:, 1]))}
        inside = set(int(i) for i in idxs)
        held = []
        for r in rows:
            where = (int(r[0]), int(r[1]))
            assert where in key, 'Found 0 matches for location {}'.format(r)
            i = key[where]
            assert float(r[3]) == float(self.labels[0][i]), \
                'Mismatch of ground truth mutation count. Expected {}, but found {}.'.format(r[3], self.labels[0][i])
            assert i in inside, "Expected the following to be in the data set, but wasn't found \n{}".format(r)
            inside.discard(i)
            held.append(i)
        _say('Heldout {} windows.'.format(len(held)))
        return np.asarray([i for i in idxs if int(i) in inside], int), np.asarray(held, int)

    def get_datasets(self):
        return self.split(self.idxs, self.val_ratio, 'validation')

    def meta(self, rows):
        return self.locs[rows], self.mapp[rows], self.quantiles[rows]


class OutputGenerator:
    """mutations_main.py:98-281: held-out prediction, the GP runs of a rerun and the accumulated pretrained frame."""
    pretrained_cols = ['CHROM', 'START', 'END', 'Y_TRUE', 'Y_PRED', 'STD', 'FLAG', 'MAPP', 'QUANT', 'FOLD']
    nn_acc_col, val_acc_col, ho_acc_col = 'nn_acc', 'val_acc', 'test_acc'

    def __init__(self, args, device, out_dir):
        self.args, self.device, self.out_dir = args, device, out_dir
        self.pretrained_path = os.path.join(out_dir, '{}.Pretrained.h5')
        self.acc_path = os.path.join(out_dir, '{}_pretrained_accuracy.txt')
        folds = np.arange(args.k) if hasattr(args, 'k') else np.arange(args.nn_reruns)
        runs = [str(r) for r in range(args.run_gaussian)]
        index = pd.MultiIndex.from_product([folds, runs], names=['fold', 'gp_run'])
        cols = [self.nn_acc_col, self.val_acc_col, self.ho_acc_col]
        self.score_dict = {l: pd.DataFrame(index=index, columns=cols, dtype=float) for l in args.label_ids}
        self.pretrained_dict = {l: pd.DataFrame(columns=self.pretrained_cols, dtype=float) for l in args.label_ids}

    def predict(self, model, store, rows, labels):
        """:121-146 -> (preds [C][n], true [C][n], features [C][n, 16], r2 [C])."""
        preds, feats, accs = predict(model, store, rows, labels=labels, batch_size=max(self.args.bs, 512))
        C = len(self.args.label_ids)
        return [preds[c] for c in range(C)], [labels[c][rows] for c in range(C)], [feats[c] for c in range(C)], list(accs)

    def store_pretrained(self, lbl, chr_locs, mapps, quants, y_true, means, stds, fold, is_flagged=False):
        """:148-172: the rerun's held-out windows join the label's frame, which is written sorted by position."""
        n = len(y_true)
        block = pd.DataFrame(np.column_stack([np.asarray(chr_locs, float).reshape(n, 3), y_true, means, stds,
                                              np.full(n, 1.0 if is_flagged else 0.0), mapps, quants, np.full(n, float(fold))]),
                             columns=self.pretrained_cols)
        have = self.pretrained_dict[lbl]
        df = block if have.empty else pd.concat([have, block], ignore_index=True)
        mapfile.write_frame(self.pretrained_path.format(lbl), 'region_params',
                            df.sort_values(by=['CHROM', 'START']).reset_index(drop=True))
        ok = df.FLAG.values == 0
        acc = r2_score(df.Y_TRUE.values[ok], df.Y_PRED.values[ok])
        _say('Overall unflagged pretrained accuracy after fold {} is: {}'.format(fold + 1, acc))
        with open(self.acc_path.format(lbl), 'w') as f:
            f.write(str(acc))
        self.pretrained_dict[lbl] = df

    def run_gp(self, f_name, train, val, ho, nn_scores, fold, prefix=''):
        """:202-247: the GP fits of all labels for one rerun (file `f_name`), their scores, the pretrained frame.
        False when a label's GP cannot be fitted (the caller retrains the CNN)."""
        path = os.path.join(self.out_dir, f_name)
        try:
            run_gp_fold(self.args, self.device, path, self.args.label_ids, train, val, ho, nn_scores,
                        seed=self.args.seed + 31 * fold)
        except AssertionError as exc:
            _say('GP run failed: {}'.format(exc))
            return False
        for l, lbl in enumerate(self.args.label_ids):
            scores = self.score_dict[lbl]
            for j in range(self.args.run_gaussian):
                run = '{}/{}/{}'.format(lbl, '{}', j)
                val_r2 = float(mapfile.read_attrs(path, run.format('val'))['R2'])
                ho_r2 = float(mapfile.read_attrs(path, run.format('held-out'))['R2'])
                scores.loc[(fold, (prefix + '_' if prefix else '') + str(j)), :] = [float(nn_scores[l]), val_r2, ho_r2]
            scores.to_csv(os.path.join(self.out_dir, '{}_gp_runs_summary.csv'.format(lbl)))
            chr_locs, mapps, quants, y_true, means, stds = gp_trainer.compute_pretrained(path, lbl, self.args.run_gaussian)
            _say('Fold {} pretrained model R2: {}'.format(str(fold + 1) + ('_' + prefix if prefix else ''), r2_score(y_true, means)))
            _say('GP fold results summary:')
            _say(scores)
            self.store_pretrained(lbl, chr_locs, mapps, quants, y_true, means, stds, fold, is_flagged=len(prefix) > 0)
        return True

    def save_prediction(self, f_name, locs, rows, preds, true, feats=None):
        """:266-276"""
        path = os.path.join(self.out_dir, f_name)
        with mapfile.batch(path):
            mapfile.write_array(path, 'chr_locs', np.asarray(locs))
            mapfile.write_array(path, 'idxs', np.asarray(rows))
            mapfile.write_array(path, 'pred_lbls', np.asarray(preds))
            mapfile.write_array(path, 'true_lbls', np.asarray(true))
            for k, v in (feats or {}).items():
                mapfile.write_array(path, k + '_feats', np.asarray(v))


def main(input_args=None):
    args = get_cmd_arguments() if input_args is None else input_args
    if args.get_attention or args.get_attended_tracks or args.get_attended_cols or args.net != 'cnn' or args.autoregressive_size:
        raise SystemExit("attention maps, the 'fc' network and autoregressive features are not built (CNN + GP route only)")
    labels_str = '-'.join(args.label_ids)
    out_dir = os.path.join(args.out_dir, labels_str, str(datetime.now()))
    _say('Generating prediction for cancer types: {}'.format(args.label_ids))
    torch.cuda.set_device(int(os.environ.get('LOCAL_RANK', '0')))
    device = torch.device('cuda', torch.cuda.current_device())
    out_pred = OutputGenerator(args, device, out_dir)
    if args.save_model or args.save_training or args.run_gaussian:
        _say("Saving results under: '{}'".format(out_dir))
        os.makedirs(out_dir)
        with open(os.path.join(out_dir, 'run_params.txt'), 'w') as f:
            for k, v in vars(args).items():
                f.write('{}: {}\n'.format(k, v))
    data = SplitData(args, device)
    ho_rows = data.heldout_idxs
    C = len(args.label_ids)
    accs_df = pd.DataFrame()
    r, re, gp_succeed = 0, 0, False
    while r < args.nn_reruns and re < args.max_nn_reruns:
        train_rows, val_rows = data.get_datasets()                      # a new split per (re)run, mutations_main.py:324-326
        shape = data.store.shape(len(train_rows))
        _say('Using {} predictors for prediction.'.format(shape[2]))
        _say('Setting model and optimizers for run {}/{}...'.format(r + 1, args.nn_reruns))
        torch.manual_seed(args.seed + 1000 * r + 10 * re)
        model = SimpleMultiTaskResNet(shape, C)
        trainer = NNTrainer(model, _adam(model, device), nn.MSELoss(), args.bs, args.label_ids,
                            data.store, train_rows, val_rows, data.labels, device, seed=args.seed + 7919 * r + re)
        best = dict(accs=np.zeros(C))
        for epoch in range(1, args.epochs + 1):
            _say('Running epoch {}/{}'.format(epoch, args.epochs))
            _, train_accs, tr_feat, _, tr_true = trainer.train(epoch, r)
            _, val_accs, va_feat, _, va_true, _ = trainer.test(epoch, r)
            live = [int((np.abs(f).mean(axis=0) > 0).sum()) for f in tr_feat]
            if np.mean(val_accs) > np.mean(best['accs']) and all(n > 1 for n in live):      # :365-376
                _say('Changing run model since best R2 was {} compared to previous {}'.format(np.mean(val_accs), np.mean(best['accs'])))
                best = dict(accs=np.asarray(val_accs), train_accs=train_accs, model=copy.deepcopy(trainer.model),
                            train=dict(feat=tr_feat, lbls=tr_true, meta=data.meta(trainer.last_train_rows)),
                            val=dict(feat=va_feat, lbls=va_true, meta=data.meta(val_rows)))
        if 'model' not in best:                                          # no epoch qualified: treated like a failed GP
            re += 1
            _say('No epoch with a positive validation r2 and live features! Rerunning NN, attempt {}/{}'.format(re + 1, args.max_nn_reruns))
            continue
        _say('Best validation accuracy for run {}/{} was: {}.'.format(r + 1, args.nn_reruns, np.mean(best['accs'])))
        _say('Running best model over {} held-out set samples...'.format(len(ho_rows)))
        ho_preds, ho_true, ho_feat, ho_accs = out_pred.predict(best['model'], data.store, ho_rows, data.labels)
        _say('Model held-out accuracy: {}'.format(ho_accs))
        for j, l in enumerate(args.label_ids):
            accs_df.loc[r, 'Train_{}'.format(l)] = best['train_accs'][j]
            accs_df.loc[r, 'Va_{}'.format(l)] = best['accs'][j]
            accs_df.loc[r, 'Held-out_{}'.format(l)] = ho_accs[j]
        if args.save_model:
            _say('Saving model and held-out indices from run {} to {}...'.format(r, out_dir))
            np.save(os.path.join(out_dir, 'ho_indices_{}'.format(r)), ho_rows)
            torch.save(best['model'].state_dict(), os.path.join(out_dir, 'best_model_{}.pt'.format(r)))
        if args.save_training:
            out_pred.save_prediction('preds_{}.h5'.format(r), data.locs[ho_rows], ho_rows, ho_preds, ho_true,
                                     dict(train=best['train']['feat'], val=best['val']['feat'], ho=ho_feat))
        if args.run_gaussian > 0:
            ho = dict(feat=ho_feat, lbls=ho_true, meta=data.meta(ho_rows))
            gp_succeed = out_pred.run_gp('gp_results_run{}.h5'.format(r), best['train'], best['val'], ho, best['accs'], r)
        if args.run_gaussian > 0 and not gp_succeed:
            re += 1
            _say('GP run failed! Rerunning NN, attempt {}/{}'.format(re + 1, args.max_nn_reruns))
        else:
            r, re = r + 1, 0
    assert args.run_gaussian < 1 or gp_succeed, 'GP failed at run {} after {} NN reruns'.format(r, re)
    if args.save_training:
        accs_df.to_csv(os.path.join(out_dir, 'run_accuracies.csv'))
    _say('Results summary for {} runs:\n {}'.format(args.nn_reruns, accs_df.describe()))
    _say('Done!')
    return out_dir


if __name__ == '__main__':
    t0 = datetime.now()
    main()
    _say('Time elapsed: {}'.format(datetime.now() - t0))