#!/usr/bin/env python
"""Developer probe: one epoch of the REAL NNTrainer (gather-fed, T = 735, 37 heads, batch 128, fp32) -- the leg bench.py's aux_rooflines
times -- as a target for rocprofv3 --kernel-trace --stats."""
import contextlib, io, os, sys, time, json
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from digdriver_amd.region_model.nets.cnn_predictors import SimpleMultiTaskResNet, flops_per_bin
from digdriver_amd.region_model.data_aux.dataset_generator import BinTrackStore
from digdriver_amd.region_model.trainers.nn_trainer import NNTrainer, adam_for
dev = torch.device("cuda:0")
N, L, T, C, bs = 64 * 128 + 256, 100, 735, 37, int(os.environ.get("BATCH", 128))
g = torch.Generator(device=dev).manual_seed(2)
x16 = (torch.rand((N, L, T), device=dev, generator=g) * 100).round().to(torch.int16)
store = BinTrackStore(x16)
torch.manual_seed(0)
net = SimpleMultiTaskResNet((bs, L, T), C)
opt = adam_for(net, dev)
lab = [np.random.default_rng(9 + c).gamma(9.0, 3.0, N) for c in range(C)]
n_tr = 64 * 128
tr = NNTrainer(net, opt, torch.nn.MSELoss(), bs, list(range(C)), store, np.arange(n_tr), np.arange(n_tr, n_tr + 256), lab, dev, seed=1)
with contextlib.redirect_stdout(io.StringIO()):
    tr.train(0)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    tr.train(1)
    torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / (n_tr // bs)
fl = 3.0 * float(flops_per_bin(T, C)) * bs
print(json.dumps({"ms_per_step": dt * 1e3, "batch": bs, "tflops": fl / dt / 1e12, "frac_fp32_matrix": fl / dt / 157.3e12}))
