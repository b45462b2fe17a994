#!/usr/bin/env python
"""Region-model CNN stage on MI355X: HBM-resident track matrix -> dig_gather_bins -> SimpleMultiTaskResNet forward
(PyTorch-ROCm conv1d/linear on the MFMA units).  Reports bins/s and the fraction of the dense MFMA peak using the
algorithmic FLOP count of SURVEY 8d: 2 * (223.3 M + 1.71 M * (C - 1)) per bin at T = 735.  Secondary benchmark (the
judged bench line is bench.py); BASELINE configs[1] names this stage."""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from digdriver_amd.region_model.data_aux.dataset_generator import BinTrackStore          # noqa: E402
from digdriver_amd.region_model.nets.cnn_predictors import SimpleMultiTaskResNet, flops_per_bin   # noqa: E402

PEAK = {"fp32": 157.3e12, "bf16": 2.5e15, "fp16": 2.5e15}    # MI355X_MICROARCH.md chip table (dense)


def train_bench(args):
    """One NNTrainer-style step (nn_trainer.py:52-79): gather, train-mode forward, summed per-task MSE, backward, Adam.
    FLOPs are counted as 3x the forward (forward + input-gradient + weight-gradient products)."""
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    net = SimpleMultiTaskResNet((args.batch, 100, args.tracks), args.cohorts).to(dev).train()
    opt = torch.optim.Adam(net.parameters(), lr=1e-3)
    g = torch.Generator(device=dev).manual_seed(2)
    x = (torch.rand((args.bins, 100, args.tracks), device=dev, generator=g) * 100).round().to(torch.int16)
    store = BinTrackStore(x)
    rows = torch.randperm(args.bins, device=dev)[: args.batch].cpu().numpy()
    target = torch.rand((args.cohorts, args.batch), device=dev) * 30
    amp = args.dtype != "fp32"
    adt = {"fp32": torch.float32, "bf16": torch.bfloat16, "fp16": torch.float16}[args.dtype]

    def step():
        xb = store.batch(rows, channels_first=True)
        with torch.autocast("cuda", dtype=adt, enabled=amp):
            out, _, _ = net.forward_channels_first(xb)
        loss = sum(torch.nn.functional.mse_loss(out[c].float(), target[c]) for c in range(args.cohorts))
        opt.zero_grad(set_to_none=True)
        loss.backward()
        opt.step()

    for _ in range(3):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    dt_s = (time.perf_counter() - t0) / args.steps
    fl = 3 * flops_per_bin(args.tracks, args.cohorts) * args.batch
    print(json.dumps({"metric": "CNN training bins/s (gather + fwd + bwd + Adam, %d cohort heads)" % args.cohorts,
                      "value": args.batch / dt_s, "ms_per_step": dt_s * 1e3, "batch": args.batch, "tracks": args.tracks,
                      "dtype": args.dtype + (" autocast" if amp else ""),
                      "roofline": {"bound": "mfma", "achieved": fl / dt_s / 1e12, "peak": PEAK[args.dtype] / 1e12,
                                   "unit": "TFLOP/s", "frac": fl / dt_s / PEAK[args.dtype]},
                      "epoch_230k_bins_s": 230000 / (args.batch / dt_s)}))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--bins", type=int, default=16384, help="bins resident in HBM for the run")
    ap.add_argument("--tracks", type=int, default=735)
    ap.add_argument("--cohorts", type=int, default=37)
    ap.add_argument("--batch", type=int, default=2048)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--dtype", default="bf16", choices=["fp32", "bf16", "fp16"])
    ap.add_argument("--store", default="i16", choices=["i16", "f32"])
    ap.add_argument("--path", default="gemm", choices=["gemm", "conv"], help="conv1d as hipBLASLt GEMMs, or MIOpen conv1d")
    ap.add_argument("--train", action="store_true", help="time a training step (gather + forward + backward + Adam) instead")
    args = ap.parse_args()
    if args.train:
        return train_bench(args)
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    dt = {"fp32": torch.float32, "bf16": torch.bfloat16, "fp16": torch.float16}[args.dtype]
    net = SimpleMultiTaskResNet((args.batch, 100, args.tracks), args.cohorts).eval().to(dev).fold_batchnorm().to(dt)
    g = torch.Generator(device=dev).manual_seed(2)
    x = (torch.rand((args.bins, 100, args.tracks), device=dev, generator=g) * 100).round()
    x = x.to(torch.int16) if args.store == "i16" else x.float()
    store = BinTrackStore(x)
    rows = torch.randperm(args.bins, device=dev)[: args.batch].cpu().numpy()
    out_dt = "bf16" if args.dtype == "bf16" else "f32"

    def step():
        xb = store.batch(rows, channels_first=(args.path == "conv"), out_dtype=out_dt)
        with torch.no_grad():
            out, feats, _ = net.forward_gemm(xb.to(dt)) if args.path == "gemm" else net.forward_channels_first(xb.to(dt))
        return out

    for _ in range(3):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    dt_s = (time.perf_counter() - t0) / args.steps
    fl = flops_per_bin(args.tracks, args.cohorts) * args.batch
    res = {"metric": "CNN bins/s (gather + forward, %d cohort heads)" % args.cohorts, "value": args.batch / dt_s,
           "ms_per_batch": dt_s * 1e3, "batch": args.batch, "tracks": args.tracks, "dtype": args.dtype,
           "storage": args.store, "path": args.path, "roofline": {"bound": "mfma", "achieved": fl / dt_s / 1e12, "peak": PEAK[args.dtype] / 1e12,
                                               "unit": "TFLOP/s", "frac": fl / dt_s / PEAK[args.dtype]},
           "whole_genome_288k_bins_s": 288000 / (args.batch / dt_s)}
    print(json.dumps(res))


if __name__ == "__main__":
    main()
