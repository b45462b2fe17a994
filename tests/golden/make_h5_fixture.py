#!/opt/conda/bin/python3.9
"""Generates the HDF5 fixtures that pin digdriver_amd/io/{h5lite,pandas_fixed,mapfile}.py to files written by the very
libraries the reference uses: h5py 3.3.0 (HDF5 1.10.6) for datasets / attributes / element groups, pandas 2.3.3 +
PyTables 3.6.1 for the `DataFrame.to_hdf` frames.  Run with the image's second interpreter:

    /opt/conda/bin/python3.9 tests/golden/make_h5_fixture.py

That interpreter's PyTables predates its numpy (it imports `numpy.typeDict`, gone since numpy 1.24) and its pandas asks
for PyTables >= 3.8 by version string; the two lines below let the installed libraries run unmodified.  PyTables 3.6.1
WRITES correctly with them (h5dump of the result is what DESIGN.md quotes); it cannot READ string arrays back under
this numpy, which is why the reader is checked against h5py and h5dump instead.

Outputs (all small, data only):
    pretrained_genuine.h5.gz    the layout of DigPretrain.py:82-96,156-177,207-208,234,266: idx / mappability (gzip),
                                root attributes, frames region_params, sequence_model_192 / _64, genic_model, an element frame
    element_data_genuine.h5     the layout of sequence_tools.py:460-478,639-641
    h5_fixture_expected.json    the same content as plain JSON
"""
import json
import os
import sys

sys.dont_write_bytecode = True            # nothing may be written under /root/reference

import numpy
numpy.typeDict = numpy.sctypeDict          # see the docstring
import tables                               # noqa: E402
tables.__version__ = "3.8.0"                # see the docstring
import h5py                                 # noqa: E402
import numpy as np                          # noqa: E402
import pandas as pd                         # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
rng = np.random.default_rng(20261003)


def frame_json(df):
    return {"columns": [str(c) for c in df.columns], "index": [x if isinstance(x, str) else (int(x) if float(x).is_integer() else float(x)) for x in df.index.tolist()],
            "index_name": df.index.name, "dtypes": [str(t) for t in df.dtypes],
            "data": {str(c): [v.item() if hasattr(v, "item") else v for v in df[c].tolist()] for c in df.columns}}


def main():
    exp = {}
    # ---- pretrained map ---------------------------------------------------------------------------------------------
    path = os.path.join(HERE, "pretrained_genuine.h5")
    if os.path.exists(path):
        os.remove(path)
    n = 37
    chrom = np.repeat([1, 2], [20, 17])
    start = np.concatenate([np.arange(20), np.arange(17)]) * 10000
    idx = np.stack([chrom, start, start + 10000], axis=1).astype(np.int32)
    mapp = rng.uniform(0.3, 1.0, n).astype(np.float32)
    with h5py.File(path, "w") as f:                                     # DigPretrain.py:82-89
        f.create_dataset("idx", data=idx, dtype=np.int32, compression="gzip")
        f.create_dataset("mappability", data=mapp, dtype=np.float32, compression="gzip")
        f.attrs["cohort_name"] = "Synthetic-Cohort".encode("utf-8")
    rp = pd.DataFrame({"CHROM": chrom.astype(np.int64), "START": start.astype(np.int64), "END": (start + 10000).astype(np.int64),
                       "Y_TRUE": rng.poisson(30, n).astype(np.int64), "Y_PRED": rng.gamma(9.0, 3.0, n),
                       "STD": rng.gamma(4.0, 1.0, n), "MAPP": mapp.astype(np.float64), "QUANT": rng.uniform(size=n),
                       "FLAG": rng.uniform(size=n) < 0.2},
                      index=["chr{}:{}-{}".format(c, s, s + 10000) for c, s in zip(chrom, start)])
    rp.to_hdf(path, key="region_params", mode="a")                       # :96
    with h5py.File(path, "a") as f:                                     # :104,156-177
        f.attrs["N_SAMPLES"] = 812
        f.attrs["N_MUT_TOTAL"] = np.int64(1234567)
        f.attrs["N_MUT_TRAIN"] = int(rp.Y_TRUE.sum())
        f.attrs["N_MUT_CDS"] = 4321
        f.attrs["mappability_threshold"] = 0.5
    bases = "ACGT"
    rows = [(a + ">" + b, x + a + y) for a in "CT" for b in bases if b != a for x in bases for y in bases]
    rows = [(m, c) for m, c in rows] + [({"C": "G", "T": "A"}[m[0]] + ">" + {"A": "T", "C": "G", "G": "C", "T": "A"}[m[2]], c[::-1].translate(str.maketrans("ACGT", "TGCA")))
                                        for m, c in rows]
    cnt = rng.integers(1, 5000, len(rows))
    sm192 = pd.DataFrame({"MUT_TYPE": [m for m, _ in rows], "CONTEXT": [c for _, c in rows], "COUNT": cnt.astype(np.int64),
                          "FREQ": cnt / rng.integers(10 ** 6, 10 ** 7, len(rows))})
    sm192.to_hdf(path, key="sequence_model_192", mode="a")              # :207
    sm64 = sm192.pivot_table(index="CONTEXT", values=["COUNT", "FREQ"], aggfunc="sum")
    sm64.to_hdf(path, key="sequence_model_64", mode="a")                # :208
    genes = ["GENE%02d" % i for i in range(9)]
    genic = pd.DataFrame({"CHROM": rng.integers(1, 3, 9).astype(np.int64), "GENE_LENGTH": rng.integers(300, 9000, 9).astype(np.int64),
                          "R_SIZE": rng.integers(10000, 40000, 9).astype(np.int64), "R_OBS": rng.integers(5, 90, 9).astype(np.int64),
                          "R_INDEL": rng.integers(0, 9, 9).astype(np.int64), "MU": rng.gamma(9., 3., 9), "SIGMA": rng.gamma(4., 1., 9),
                          "FLAG": rng.uniform(size=9) < 0.3, "STRAND": np.where(rng.uniform(size=9) < 0.5, "+", "-"),
                          "P_SILENT": rng.uniform(1e-4, 1e-3, 9), "P_MIS": rng.uniform(1e-4, 1e-3, 9), "P_NONS": rng.uniform(1e-5, 1e-4, 9),
                          "P_SPLICE": rng.uniform(1e-6, 1e-5, 9), "P_TRUNC": rng.uniform(1e-5, 1e-4, 9), "P_INDEL": rng.uniform(1e-3, 1e-2, 9)},
                         index=pd.Index(genes, name="GENE"))
    genic.to_hdf(path, key="genic_model", mode="a")                     # :234
    elts = pd.DataFrame({"ELT_SIZE": rng.integers(100, 3000, 6).astype(np.int64), "FLAG": rng.uniform(size=6) < 0.3,
                         "R_SIZE": rng.integers(10000, 30000, 6).astype(np.int64), "R_OBS": rng.integers(5, 90, 6).astype(np.int64),
                         "R_INDEL": rng.integers(0, 9, 6).astype(np.int64), "MU": rng.gamma(9., 3., 6), "SIGMA": rng.gamma(4., 1., 6),
                         "MU_INDEL": rng.gamma(9., 3., 6), "SIGMA_INDEL": rng.gamma(4., 1., 6), "P_SUM": rng.uniform(1e-4, 1e-2, 6),
                         "P_INDEL": rng.uniform(1e-3, 1e-1, 6)}, index=pd.Index(["enh_%d" % i for i in range(6)], name="ELT"))
    elts.to_hdf(path, key="enhancers", mode="a")                        # :266
    exp["pretrained"] = {"idx": idx.tolist(), "mappability": [float(x) for x in mapp], "attrs": {
        "cohort_name": "Synthetic-Cohort", "N_SAMPLES": 812, "N_MUT_TOTAL": 1234567, "N_MUT_TRAIN": int(rp.Y_TRUE.sum()),
        "N_MUT_CDS": 4321, "mappability_threshold": 0.5},
        "frames": {"region_params": frame_json(rp), "sequence_model_192": frame_json(sm192), "sequence_model_64": frame_json(sm64),
                   "genic_model": frame_json(genic), "enhancers": frame_json(elts)}}
    # ---- element data -----------------------------------------------------------------------------------------------
    path2 = os.path.join(HERE, "element_data_genuine.h5")
    if os.path.exists(path2):
        os.remove(path2)
    subst = sorted(c + ">" + c[0] + m[2] + c[2] for m, c in rows)
    win_vals = rng.integers(0, 400, (n, 64))
    E = {}
    with h5py.File(path2, "w") as f:                                    # sequence_tools.py:460-478
        f.create_dataset("substitution_idx", data=np.array([t.encode("ascii") for t in subst]))
        f.create_dataset("window_10000/full_window_si_values", data=win_vals, dtype=int)
        f.create_dataset("window_10000/full_window_si_index", data=idx)
        for i in range(11):                                             # :639-641
            name = "enh_%d" % i
            L = rng.integers(0, 30, 192).astype(float)
            ov = [(int(chrom[i]), int(start[i]), int(start[i]) + 10000), (int(chrom[i]), int(start[i]) + 10000, int(start[i]) + 20000)][: 1 + i % 2]
            rc = np.array([np.repeat(win_vals[i + j], 3) for j in range(len(ov))]).sum(axis=0)
            f.create_dataset("window_10000/enhancers/%s/L_counts" % name, data=L)
            f.create_dataset("window_10000/enhancers/%s/region_counts" % name, data=rc)
            f["window_10000/enhancers/%s" % name].attrs.create("overlaps", ov)
            E[name] = {"L_counts": L.tolist(), "region_counts": rc.tolist(), "overlaps": [list(o) for o in ov]}
    exp["element_data"] = {"substitution_idx": subst, "full_window_si_values": win_vals.tolist(), "full_window_si_index": idx.tolist(),
                           "elements": E}
    # ---- k-fold GP result files (GPTrainer.save_results, gp_trainer.py:206-245) + the REFERENCE's own assembly ------
    kdir = os.path.join(HERE, "kfold_genuine")
    os.makedirs(kdir, exist_ok=True)
    for fn in os.listdir(kdir):
        os.remove(os.path.join(kdir, fn))
    cohort = "Synthetic-Cohort"
    sup = np.arange(n)[mapp >= 0.5]
    sub = np.arange(n)[mapp < 0.5]
    folds = np.array_split(rng.permutation(sup), 2)

    def fold_file(name, rows, with_train):
        with h5py.File(os.path.join(kdir, name), "w") as f:
            g = f.create_group(cohort)
            sets = [("held-out", rows)] + ([("train", sup[:5]), ("val", sup[5:9])] if with_train else [])
            for key, r in sets:
                grp = g.create_group(key)
                grp.create_dataset("nn_features", data=rng.normal(size=(len(r), 16)).astype(np.float32))
                grp.create_dataset("y_true", data=rp.Y_TRUE.values[r].astype(float))
                grp.create_dataset("chr_locs", data=idx[r])
                grp.create_dataset("mappability", data=mapp[r])
                grp.create_dataset("quantiles", data=rp.QUANT.values[r])
                if key == "train":
                    continue
                for run in range(3):
                    rg = grp.create_group(str(run))
                    rg.create_dataset("mean", data=rp.Y_PRED.values[r] + rng.normal(0, 0.5, len(r)))
                    rg.create_dataset("std", data=rp.STD.values[r] * rng.uniform(0.9, 1.1, len(r)))
                    rg.create_dataset("params", data=rng.uniform(size=3))
                    rg.attrs["R2"] = float(rng.uniform(0.5, 0.9))
                    rg.attrs["loss"] = float(rng.uniform(0.1, 2))

    for k, rows in enumerate(folds):
        fold_file("gp_results_fold_%d.h5" % k, rows, True)
        fold_file("sub_mapp_results_fold_%d.h5" % k, sub, False)
    import importlib.util
    import pathlib
    spec = importlib.util.spec_from_file_location("ref_rmt", "/root/reference/DIGDriver/region_model/region_model_tools.py")
    ref = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ref)
    want = ref.kfold_results(pathlib.Path(kdir), cohort)             # the reference function itself, on these files
    exp["kfold_results"] = frame_json(want)
    with open(os.path.join(HERE, "h5_fixture_expected.json"), "w") as f:
        json.dump(exp, f)
    # PyTables allocates its VLArray chunks generously (2 MB of mostly zeros): the fixture is committed gzipped
    import gzip
    import shutil
    with open(path, "rb") as src, gzip.GzipFile(path + ".gz", "wb", mtime=0) as dst:
        shutil.copyfileobj(src, dst)
    os.remove(path)
    print("wrote", path + ".gz", os.path.getsize(path + ".gz"), path2, os.path.getsize(path2))


if __name__ == "__main__":
    main()
