#!/opt/conda/bin/python3.9
"""Golden for GPTrainer.compute_pretrained (DIGDriver/region_model/trainers/gp_trainer.py:247-261): the REFERENCE's own
method, run with h5py on the committed k-fold result files (tests/golden/kfold_genuine, written by make_h5_fixture.py).
gpytorch / torch are not needed by the method and are absent from this interpreter: empty stand-in modules let the
reference module import.  Run here: /opt/conda/bin/python3.9 tests/golden/make_pretrained_golden.py"""
import json
import os
import sys
import types

sys.dont_write_bytecode = True            # importing the reference must not leave .pyc files under /root/reference

import h5py
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
for name in ("torch", "gpytorch", "gpytorch.models", "gpytorch.means", "gpytorch.kernels", "gpytorch.distributions",
             "gpytorch.likelihoods", "gpytorch.mlls", "gpytorch.settings", "sklearn", "sklearn.preprocessing", "scipy", "scipy.stats"):
    if name not in sys.modules:
        try:
            __import__(name)
        except Exception:
            sys.modules[name] = types.ModuleType(name)
gp = sys.modules["gpytorch"]
gp.models = sys.modules["gpytorch.models"]
gp.models.ApproximateGP = gp.models.ExactGP = object
for attr in ("means", "kernels", "distributions", "likelihoods", "mlls", "settings"):
    setattr(gp, attr, sys.modules["gpytorch." + attr])
if not hasattr(sys.modules["sklearn.preprocessing"], "StandardScaler"):
    sys.modules["sklearn.preprocessing"].StandardScaler = object
if not hasattr(sys.modules["scipy.stats"], "pearsonr"):
    sys.modules["scipy.stats"].pearsonr = None
import importlib.util
spec = importlib.util.spec_from_file_location("ref_gp_trainer", "/root/reference/DIGDriver/region_model/trainers/gp_trainer.py")
ref = importlib.util.module_from_spec(spec)
spec.loader.exec_module(ref)

out = {}
for fn in ("gp_results_fold_0.h5", "gp_results_fold_1.h5", "sub_mapp_results_fold_0.h5"):
    with h5py.File(os.path.join(HERE, "kfold_genuine", fn), "r") as f:
        cohort = list(f.keys())[0]
        for runs in (1, 3):
            chr_locs, mapps, quants, y_true, means, stds = ref.GPTrainer.compute_pretrained(None, f[cohort], runs)
            out["%s:%d" % (fn, runs)] = {"cohort": cohort, "chr_locs": np.asarray(chr_locs).tolist(), "mapps": np.asarray(mapps).tolist(),
                                        "quants": np.asarray(quants).tolist(), "y_true": np.asarray(y_true).tolist(),
                                        "means": np.asarray(means).tolist(), "stds": np.asarray(stds).tolist()}
with open(os.path.join(HERE, "compute_pretrained_golden.json"), "w") as f:
    json.dump(out, f)
print("wrote", len(out), "cases")
