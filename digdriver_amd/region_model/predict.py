"""Batched CNN inference over a bin set: predictions + 16-d features per cohort.

Mirror of OutputGenerator.predict (DIGDriver/region_model/mutations_main.py:121-146) and of
NNTrainer.test's feature collection (trainers/nn_trainer.py:93-141): forward in eval mode, collect
outputs, labels and penultimate features, Pearson r^2 with NaN -> 0 (gp_trainer.py:23-25).
"""
import numpy as np
import torch


def r2_score(y_true, y_pred):
    """gp_trainer.py:23-25: squared Pearson correlation, 0 when undefined."""
    y_true, y_pred = np.asarray(y_true, float), np.asarray(y_pred, float)
    if len(y_true) < 2 or y_true.std() == 0 or y_pred.std() == 0:
        return 0.0
    r = np.corrcoef(y_true, y_pred)[0, 1]
    return float(r * r) if np.isfinite(r) else 0.0


@torch.no_grad()
def r2_rows(y_true, y_pred):
    """r2_score for every row of two [C, n] device tensors at once (float64 on the device, no host round trip): squared
    Pearson correlation, 0 where it is undefined (fewer than two values, a constant row)."""
    t, q = y_true.double(), y_pred.double()
    if t.shape[1] < 2:
        return torch.zeros(t.shape[0], dtype=torch.float64, device=t.device)
    tm, qm = t - t.mean(dim=1, keepdim=True), q - q.mean(dim=1, keepdim=True)
    den = (tm * tm).sum(dim=1) * (qm * qm).sum(dim=1)
    r = (tm * qm).sum(dim=1) / torch.sqrt(den)
    r2 = r * r
    return torch.where(torch.isfinite(r2) & (den > 0), r2, torch.zeros_like(r2))


def tune_gemms(device):
    """PyTorch's TunableOp for the CNN's GEMMs (NNTrainer and predict): every GEMM shape of the step is timed over hipBLASLt's / rocBLAS's solutions the first
    time it is met (the trainer's two eager batches, the predictor's first batch; a few seconds per process, results kept in `DIG_NN_TUNE_FILE` or
    ~/.cache/digdriver_amd/tunableop.csv for the next one) instead of taking the library's heuristic choice -- the step's
    weight-gradient products (a few output tiles, K = 1 664 .. 12 544 rows) are where the heuristic is off: 5.56 -> 4.87 ms per
    batch of 128.  Same arithmetic, another summation order inside a GEMM (fp32 rounding differences of 1e-7 relative).  DIG_NN_TUNE=0
    switches it off (the test-suite does, but for one test)."""
    import os
    if torch.device(device).type != "cuda" or os.environ.get("DIG_NN_TUNE", "1") == "0":
        return False
    try:
        import torch.cuda.tunable as tunable
        if not tunable.is_enabled():
            path = os.environ.get("DIG_NN_TUNE_FILE") or os.path.join(os.path.expanduser("~"), ".cache", "digdriver_amd", "tunableop.csv")
            try:
                os.makedirs(os.path.dirname(path), exist_ok=True)
                tunable.set_filename(path, insert_device_ordinal=True)
            except OSError:
                pass                                           # (no place to keep the results: tuned again next time)
            tunable.enable(True)
            tunable.tuning_enable(True)
        return True
    except Exception as e:                                     # (a build of torch without it: the heuristic choice, as before)
        print("GEMM tuning not available (%s)" % e)
        return False



@torch.no_grad()
def predict(model, store, bin_rows, labels=None, batch_size=2048, fold_bn=True, dtype=torch.float32):
    """Returns (preds [C, n], features [C, n, 16], r2 [C] or None).  `store` is a BinTrackStore; the batch
    arrives channels-first straight from dig_gather_bins."""
    net = model.fold_batchnorm() if fold_bn and not getattr(model, "_folded", False) else model
    net = net.eval()
    if dtype != next(net.parameters()).dtype:          # reduced-precision inference: a converted copy, never the caller's model
        import copy
        net = (copy.deepcopy(net) if net is model else net).to(dtype)
        net._gw = None
    dev = next(net.parameters()).device
    tune_gemms(dev)
    bin_rows = np.asarray(bin_rows)
    preds, feats = [], []
    use_gemm = getattr(net, "_folded", False) and not net.get_attention_maps
    for s in range(0, len(bin_rows), batch_size):
        # GEMM path consumes the row-major (channels-last) batch directly; the conv path wants channels-first
        xb = store.batch(bin_rows[s:s + batch_size], channels_first=not use_gemm,
                         out_dtype="bf16" if dtype == torch.bfloat16 else "f32")
        if xb.device != dev:
            xb = xb.to(dev)
        out, fv, _ = net.forward_gemm(xb.to(dtype)) if use_gemm else net.forward_channels_first(xb.to(dtype))
        preds.append(torch.stack(out).float())
        feats.append(torch.stack(fv).float())
    if not preds:                                      # an empty bin list (a rank without bins in a sharded run)
        C = getattr(net, "task_num", None) or len(getattr(net, "heads", [])) or 1
        preds, feats = [torch.zeros((C, 0), device=dev)], [torch.zeros((C, 0, 16), device=dev)]
    preds = torch.cat(preds, dim=1).cpu().numpy()
    feats = torch.cat(feats, dim=1).cpu().numpy()
    r2 = None
    if labels is not None:
        r2 = np.array([r2_score(np.asarray(labels[c])[bin_rows], preds[c]) for c in range(preds.shape[0])])
    return preds, feats, r2


def predict_sharded(model, store, bin_rows, labels=None, batch_size=2048, group=None, rank=None, world=None, **kw):
    """OutputGenerator.predict (mutations_main.py:121-146) over a process group: the reference scatters every batch over
    the GPUs of one process (nn.DataParallel, kfold_mutations_main.py:143); here every rank forwards its own share of the
    bin list and the predictions + 16-d features are all-gathered (rank order) and put back into the order of `bin_rows`,
    so every rank returns what the single-process call returns: (preds [C, n], features [C, n, 16], r2 [C] or None).
    Which bins a rank takes: those whose rows of the track matrix it holds when `store` is a shard (store.row_ranges),
    else a contiguous piece of the list (parallel.shard_rows)."""
    import torch.distributed as dist
    from .. import parallel
    on = dist.is_available() and dist.is_initialized()
    world = world if world is not None else (dist.get_world_size(group) if on else 1)
    rank = rank if rank is not None else (dist.get_rank(group) if on else 0)
    bin_rows = np.asarray(bin_rows)
    n = len(bin_rows)
    ranges = getattr(store, "row_ranges", None)
    if ranges is not None:
        his = np.array([hi for _, hi in ranges])
        owner = np.searchsorted(his, bin_rows, side="right")
    else:
        owner = np.zeros(n, np.int64)
        for r in range(world):
            lo, hi = parallel.shard_rows(n, r, world)
            owner[lo:hi] = r
    pos = [np.flatnonzero(owner == r) for r in range(world)]
    p, f, _ = predict(model, store, bin_rows[pos[rank]], labels=None, batch_size=batch_size, **kw)
    C = p.shape[0]
    if world > 1 or parallel.collectives_on(group):
        dev = parallel.comm_device(next(model.parameters()).device, group)
        flat = torch.cat([torch.as_tensor(p.T, dtype=torch.float32), torch.as_tensor(f.transpose(1, 0, 2).reshape(len(pos[rank]), C * 16),
                                                                                  dtype=torch.float32)], dim=1).to(dev)
        everything = parallel.all_gather_rows(flat.contiguous(), group).cpu().numpy()       # rows in rank order
        order = np.concatenate(pos)
        full = np.empty_like(everything)
        full[order] = everything
        p = np.ascontiguousarray(full[:, :C].T)
        f = np.ascontiguousarray(full[:, C:].reshape(n, C, 16).transpose(1, 0, 2))
    r2 = None
    if labels is not None:
        r2 = np.array([r2_score(np.asarray(labels[c])[bin_rows], p[c]) for c in range(C)])
    return p, f, r2
