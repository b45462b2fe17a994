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
    preds = torch.cat(preds, dim=1).cpu().numpy()
    feats = torch.cat(feats, dim=1).cpu().numpy()
    r2 = None
    if labels is not None:
        r2 = np.array([r2_score(np.asarray(labels[c])[bin_rows], preds[c]) for c in range(preds.shape[0])])
    return preds, feats, r2
