#!/usr/bin/env python
"""k-fold CNN + GP training of the region model on MI355X.

Mirror of DIGDriver/region_model/kfold_mutations_main.py:102-229 (+ OutputGenerator.run_gp, mutations_main.py:
174-247 and GPTrainer.save_results, gp_trainer.py:206-245): for every fold train the CNN (best epoch by validation
r^2 with more than one live feature), predict the held-out fold, fit `run_gaussian` sparse GPs on the CNN features
and write `gp_results_fold_{k}` -- and `sub_mapp_results_fold_{k}` with -u -- in the layout
region_model_tools.kfold_results / `DigPretrain.py regionModel` read.

The data container (map file, digdriver_amd/io/mapfile.py) holds `x_data` [N, L, T], `idx` [N, 3], `mappability`
[N] and one label array per cancer id, as the reference's training h5 does.  The track matrix is uploaded once and
stays in HBM (int16 when its values are the reference's round(x, 2) * 100 integers); batches are gathered on the
device.  Launch with torch.distributed.run for data-parallel training (one process per GPU, RCCL).
"""
import argparse
import copy
import os
from datetime import datetime

import numpy as np
import torch
from torch import nn, optim

from .. import parallel
from ..io import mapfile
from .data_aux import dataset_generator as dg
from .nets.cnn_predictors import SimpleMultiTaskResNet
from .predict import predict, predict_sharded, r2_score
from .trainers import gp_trainer
from .trainers.nn_trainer import NNTrainer, adam_for as _adam


def get_cmd_arguments(text=None):
    ap = argparse.ArgumentParser(description='k-fold CNN + GP region model (MI355X build)')
    ap.add_argument('-c', '--cancer-id', required=True, nargs='*', type=str, dest='label_ids', help='label arrays in the data container')
    ap.add_argument('-d', '--data', required=True, type=str, dest='data_file', help='training data container')
    ap.add_argument('-o', '--out-dir', required=True, type=str, dest='out_dir', help='output directory')
    ap.add_argument('-t', '--tracks', type=str, dest='track_file', default=None, help='track selection file')
    ap.add_argument('-s', '--split', type=str, dest='split_method', default='random', help='random (chr is not built)')
    ap.add_argument('-m', '--mappability', type=float, dest='mappability', default=0.5, help='mappability lower bound')
    ap.add_argument('-cq', '--count-quantile', type=float, dest='count_quantile', default=0.999, help='count quantile cap')
    ap.add_argument('-gp', '--gaussian', type=int, dest='run_gaussian', default=5, help='number of GP fits per fold')
    ap.add_argument('-k', type=int, dest='k', default=5, help='number of folds')
    ap.add_argument('-gr', '--gp-reruns', type=int, dest='gp_reruns', default=3, help='GP retries per inducing-point count')
    ap.add_argument('-gd', '--gp-delta', type=float, dest='gp_delta', default=0.03, help='tolerated GP-vs-CNN r2 drop')
    ap.add_argument('-re', '--nn-reruns', type=int, dest='nn_reruns', default=1, help='CNN re-initialisations per fold')
    ap.add_argument('-mr', '--max-nn-reruns', type=int, dest='max_nn_reruns', default=3, help='CNN retrainings when the GP fails')
    ap.add_argument('-vr', '--val-ratio', type=float, dest='val_ratio', default=0.2, help='validation share of the training folds')
    ap.add_argument('-e', '--epochs', type=int, dest='epochs', default=20, help='epochs')
    ap.add_argument('-b', '--batch', type=int, dest='bs', default=128, help='batch size (global)')
    ap.add_argument('-nd', '--n-inducing', type=int, dest='n_inducing', default=400, help='GP inducing points')
    ap.add_argument('-nt', '--n-iter', type=int, dest='n_iter', default=50, help='GP iterations')
    ap.add_argument('-sm', '--save-model', action='store_true', dest='save_model', help='save the best model of every fold')
    ap.add_argument('-u', '--sub_mapp', action='store_true', dest='sub_mapp', help='also predict the sub-threshold bins')
    ap.add_argument('--seed', type=int, default=0, help='seed of the fold split, shuffles and initialisations')
    return ap.parse_args(text.split()) if text else ap.parse_args()


class KFoldData:
    """BaseDatasetGenerator + KFoldDatasetGenerator (dataset_generator.py:16-50,195-273) over a device-resident matrix."""

    def __init__(self, args, device):
        print('Loading data and labels from file {}...'.format(args.data_file))
        self.locs = mapfile.read_array(args.data_file, 'idx')
        self.mapp = np.asarray(mapfile.read_array(args.data_file, 'mappability'), float)
        self.labels = [np.asarray(mapfile.read_array(args.data_file, l), float) for l in args.label_ids]
        self.quantiles = dg.rank_quantiles(self.labels[0])
        self.idxs, self.below_mapp = dg.select_bins(self.mapp, self.labels[0], args.mappability, args.count_quantile)
        x = dg.load_track_matrix(args.data_file, device, log=print)      # slab by slab: int16 in HBM when exact, else float32
        tracks = None
        if args.track_file is not None:
            with open(args.track_file) as f:
                tracks = dg.load_track_selection(f.readlines())
        self.store = dg.BinTrackStore(x, tracks)
        self.folds = dg.split_folds(self.idxs, args.k, seed=args.seed)
        self.val_ratio, self.rng = args.val_ratio, np.random.default_rng(args.seed + 1)
        print('Input data is of size: {}'.format(self.store.shape(len(self.idxs))))

    def get_datasets(self, fold):
        ho = np.asarray(self.folds[fold])
        train = np.concatenate([f for i, f in enumerate(self.folds) if i != fold])
        train = self.rng.permutation(train)
        split = int((1 - self.val_ratio) * len(train))
        return train[:split], train[split:], ho

    def meta(self, rows):
        return self.locs[rows], self.mapp[rows], self.quantiles[rows]


def _write_set(path, base, feats, y_true, meta):
    mapfile.write_array(path, base + 'nn_features', np.asarray(feats, np.float32))
    mapfile.write_array(path, base + 'y_true', np.asarray(y_true))
    mapfile.write_array(path, base + 'chr_locs', np.asarray(meta[0]))
    mapfile.write_array(path, base + 'mappability', np.asarray(meta[1]))
    mapfile.write_array(path, base + 'quantiles', np.asarray(meta[2]))


def run_gp_fold(args, device, path, label_ids, train, val, ho, nn_scores, seed, held_key='held-out'):
    """OutputGenerator.run_gp (mutations_main.py:202-247) for one fold: `train` / `val` / `ho` are dicts with
    feat [C][n,16], lbls [C][n], meta (chr_locs, mappability, quantiles).  Returns the per-label fold r^2."""
    scores = []
    with mapfile.batch(path):                      # an HDF5 results file is written once, when the fold is complete
        for l, lbl in enumerate(label_ids):
            print('Running gaussian process model for {}...'.format(lbl))
            tup = lambda d: (np.asarray(d['feat'][l]), np.asarray(d['lbls'][l])) + tuple(d['meta'])
            results, means, stds = gp_trainer.run_gp(device, tup(train), tup(val), tup(ho), n_runs=args.run_gaussian,
                                                     n_iter=args.n_iter, n_inducing=args.n_inducing, gp_reruns=args.gp_reruns,
                                                     gp_delta=args.gp_delta, nn_r2=float(nn_scores[l]), seed=seed + 17 * l)
            _write_set(path, '{}/train/'.format(lbl), train['feat'][l], train['lbls'][l], train['meta'])
            _write_set(path, '{}/val/'.format(lbl), val['feat'][l], val['lbls'][l], val['meta'])
            _write_set(path, '{}/{}/'.format(lbl, held_key), ho['feat'][l], ho['lbls'][l], ho['meta'])
            for j, res in enumerate(results):                                   # GPTrainer.save_results, gp_trainer.py:224-245
                for grp, r in ((held_key, res), ('val', res.get('val'))):
                    if r is None:
                        continue
                    base = '{}/{}/{}/'.format(lbl, grp, j)
                    mapfile.write_array(path, base + 'mean', r['gp_mean'])
                    mapfile.write_array(path, base + 'std', r['gp_std'])
                    mapfile.write_array(path, base + 'params', r['params'])
                    mapfile.write_attrs(path, base.rstrip('/'), R2=float(r['r2']), loss=float(r['loss']))
            fold_r2 = r2_score(ho['lbls'][l], means)
            print('Fold pretrained model R2: {}'.format(fold_r2))
            scores.append(fold_r2)
    return scores


def main(input_args=None):
    args = get_cmd_arguments() if input_args is None else input_args
    if args.split_method != 'random':
        raise SystemExit("only the random split is built")
    import torch.distributed as dist
    world = int(os.environ.get('WORLD_SIZE', '1'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    torch.cuda.set_device(local_rank)
    device = torch.device('cuda', local_rank)
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        # rank 0 fits the GPs of a fold while the other ranks wait in a broadcast: no collective may time out meanwhile
        from datetime import timedelta
        dist.init_process_group('nccl', device_id=device, timeout=timedelta(hours=12))
    rank = dist.get_rank() if world > 1 else 0
    labels_str = '-'.join(args.label_ids)
    out_dir = os.path.join(args.out_dir, 'kfold', labels_str)
    if rank == 0:
        os.makedirs(out_dir, exist_ok=True)
        with open(os.path.join(out_dir, 'run_params.txt'), 'w') as f:
            for k, v in vars(args).items():
                f.write('{}: {}\n'.format(k, v))
    data = KFoldData(args, device)
    shape = data.store.shape(len(data.idxs))
    print('Running {}-fold prediction...'.format(args.k))
    k, retries, summary = 0, 0, []
    while k < args.k and retries < args.max_nn_reruns:
        train_rows, val_rows, ho_rows = data.get_datasets(k)
        best = dict(acc=-np.inf)
        for r in range(args.nn_reruns):
            print('Setting model and optimizers for run {}/{} and fold {}/{}...'.format(r + 1, args.nn_reruns, k + 1, args.k))
            torch.manual_seed(args.seed + 1000 * k + 10 * retries + r)           # same initial weights on every rank
            model = SimpleMultiTaskResNet(shape, len(args.label_ids))
            trainer = NNTrainer(model, _adam(model, device), nn.MSELoss(), args.bs,
                                args.label_ids, data.store, train_rows, val_rows, data.labels, device,
                                seed=args.seed + 7919 * k + r)
            for epoch in range(1, args.epochs + 1):
                print('Running epoch {}/{}'.format(epoch, args.epochs))
                _, _, tr_feat, _, tr_true = trainer.train(epoch, r)
                _, val_accs, va_feat, _, va_true, _ = trainer.test(epoch, r)
                live = int((np.abs(tr_feat[0]).mean(axis=0) > 0).sum())
                print('#non-zero features: {}'.format(live))
                # the decision is rank 0's (the ranks hold the same weights, running statistics and gathered features, so
                # they agree anyway; broadcasting it rules out a rank leaving the collective sequence on a rounding tie)
                if parallel.broadcast_flag(val_accs[0] > best['acc'] and live > 1, device):   # kfold_mutations_main.py:170-177
                    best = dict(acc=val_accs[0], accs=val_accs, model=copy.deepcopy(trainer.model),
                                train=dict(feat=tr_feat, lbls=tr_true, rows=trainer.last_train_rows),
                                val=dict(feat=va_feat, lbls=va_true))
        if 'model' not in best:
            retries += 1
            continue
        print('Best overall validation accuracy was: {}.'.format(best['acc']))
        ok = True
        ho_feat = sub_feat = None
        sub_rows = np.asarray(data.below_mapp)
        if args.run_gaussian > 0:
            # held-out (and sub-threshold) bins: every rank forwards its share of the list, predictions and features are
            # all-gathered (the reference scatters each batch over the GPUs with nn.DataParallel, kfold_mutations_main.py:143)
            ho_pred, ho_feat, ho_acc = predict_sharded(best["model"], data.store, ho_rows, labels=data.labels)
            if args.sub_mapp and len(sub_rows):
                _, sub_feat, _ = predict_sharded(best["model"], data.store, sub_rows)
        if rank == 0:
            if args.save_model:
                torch.save(best['model'].state_dict(), os.path.join(out_dir, 'best_model_fold_{}.pt'.format(k)))
                np.save(os.path.join(out_dir, 'val_indices_fold_{}'.format(k)), val_rows)
            if args.run_gaussian > 0:
                print('Model held-out accuracy: {}'.format(ho_acc))
                C = len(args.label_ids)
                ho = dict(feat=[ho_feat[c] for c in range(C)], lbls=[data.labels[c][ho_rows] for c in range(C)],
                          meta=data.meta(ho_rows))
                best['train']['meta'] = data.meta(best['train']['rows'])       # features are in visiting order
                best['val']['meta'] = data.meta(val_rows)
                try:
                    scores = run_gp_fold(args, device, os.path.join(out_dir, 'gp_results_fold_{}.h5'.format(k)), args.label_ids,
                                         best['train'], best['val'], ho, best['accs'], seed=args.seed + 31 * k)
                    summary.append(scores)
                    if sub_feat is not None:
                        sub = dict(feat=[sub_feat[c] for c in range(C)], lbls=[data.labels[c][sub_rows] for c in range(C)],
                                   meta=data.meta(sub_rows))
                        run_gp_fold(args, device, os.path.join(out_dir, 'sub_mapp_results_fold_{}.h5'.format(k)), args.label_ids,
                                    best['train'], best['val'], sub, [-np.inf] * C, seed=args.seed + 31 * k + 5)
                except AssertionError as exc:
                    print('GP run failed: {}'.format(exc))
                    ok = False
        if world > 1:
            flag = torch.tensor([1 if ok else 0], device=device)
            dist.broadcast(flag, src=0)
            ok = bool(flag.item())
        if ok:
            k, retries = k + 1, 0
        else:
            retries += 1
            print('GP run failed! Rerunning NN, attempt {}/{}'.format(retries + 1, args.max_nn_reruns))
    assert k == args.k, 'GP failed at fold {} after {} NN reruns'.format(k, retries)
    if rank == 0 and summary:
        np.savetxt(os.path.join(out_dir, 'fold_r2.txt'), np.asarray(summary))
    print('Done!')
    return out_dir


if __name__ == '__main__':
    t0 = datetime.now()
    main()
    print('Time elapsed: {}'.format(datetime.now() - t0))
