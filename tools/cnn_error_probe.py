#!/usr/bin/env python
"""Developer tool: max error of the GPU CNN forward (conv path and folded-GEMM path) against the reference-generated goldens."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from test_region_and_sequence_models import _golden_net, _golden_net_735   # noqa: E402

dev = torch.device("cuda:0")
for name, (net, d) in (("T=32", _golden_net()), ("T=735", _golden_net_735())):
    x = torch.tensor(d["x"].astype(np.float32), device=dev)
    net = net.to(dev)
    with torch.no_grad():
        o1, f1, _ = net(x)
        o2, f2, _ = net.fold_batchnorm().forward_gemm(x)
        o3, f3, _ = net.fold_batchnorm().forward_channels_first(x.transpose(1, 2).contiguous())
    for tag, o, f in (("module", o1, f1), ("folded gemm", o2, f2), ("folded conv", o3, f3)):
        eo = np.abs(torch.stack(o).cpu().numpy() - d["outputs"])
        ef = np.abs(torch.stack(f).cpu().numpy() - d["features"])
        print(name, tag, "outputs max abs %.3g (scale %.3g) max rel %.3g | features max abs %.3g (scale %.3g)" % (
            eo.max(), np.abs(d["outputs"]).max(), (eo / np.abs(d["outputs"])).max(), ef.max(), np.abs(d["features"]).max()))
