#!/usr/bin/env python
"""A/B kernel builds on the bench workload (developer tool, not the judged bench).

    python tools/variant_bench.py base.so new.so ...      (paths under tools/variants/, or absolute)

Each library runs in its own process (DIG_HIP_LIB).  Per library: HIP-event times of the statistics stage alone
(`stages=4`), the accumulate stages (`stages=3`) and the whole dig_element_pipeline, interleaved over several rounds;
outputs are dumped by the first library and compared by the others (max relative difference per plane)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def child(ref_path, write_ref):
    import numpy as np
    import torch
    sys.path.insert(0, ROOT)
    from bench import make_workload
    from digdriver_amd import engine
    dev = torch.device("cuda:0")
    E, C = int(os.environ.get("KB_E", 120091)), int(os.environ.get("KB_C", 37))
    w = make_workload(288000, E, C, seed=3)
    td = {k: torch.as_tensor(v, device=dev) for k, v in w.items() if isinstance(v, np.ndarray)}
    oa = engine.alloc_accumulate_outputs(E, C, 1, dev)
    st = torch.empty((7, E, C), dtype=torch.float64, device=dev)
    plan = engine.PipelinePlan(td["bin_mu"], td["bin_std"], td["bin_y"], td["bin_flag"], td["bin_ctx"], td["ov_ptr"], td["ov_idx"],
                               td["L"], td["strand_minus"], td["d_pr"], td["obs_snv"], td["obs_samples"], td["obs_indel"],
                               out_acc=oa, out_stats=st, records_out=os.environ.get("DIG_PLAN_RECORDS", "0") == "1")
    s = torch.cuda.current_stream(dev)

    def timed(stages, n):
        if stages == 4:      # the statistics stage bracketed inside split calls, like bench.py samples it
            evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
            for a, b in evs:
                plan.run(td["cj"], td["cj_indel"], stages=1, stream=s)
                plan.run(td["cj"], td["cj_indel"], stages=2, stream=s)
                a.record(s)
                plan.run(td["cj"], td["cj_indel"], stages=4, stream=s)
                b.record(s)
            torch.cuda.synchronize()
            return sum(a.elapsed_time(b) for a, b in evs) / n * 1e3
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(s)
        for _ in range(n):
            plan.run(td["cj"], td["cj_indel"], stages=stages, stream=s)
        b.record(s)
        torch.cuda.synchronize()
        return a.elapsed_time(b) / n * 1e3

    plan.run(td["cj"], td["cj_indel"], stages=3, stream=s)      # (P in memory before anything else: the -DDIG_FUSE_ABL timing build reads it back)
    for _ in range(300):
        plan.run(td["cj"], td["cj_indel"], stages=7, stream=s)
    torch.cuda.synchronize()
    res = {"stats": [], "acc": [], "pipe": []}
    for _ in range(5):
        res["stats"].append(timed(4, 40))
        res["acc"].append(timed(3, 100))
        res["pipe"].append(timed(7, 100))
    out = {k: round(min(v), 2) for k, v in res.items()}
    out.update({k + "_med": round(sorted(v)[len(v) // 2], 2) for k, v in res.items()})
    off = (engine._lib.workspace_bytes("accumulate", E, C) + 255) // 256 * 256
    hdr = plan.ws[off:off + 16].view(torch.int32).cpu().numpy()
    out["finished_from_overflow_segment"], out["finished_from_lds_queue"] = int(hdr[2]), int(hdr[3])
    if plan.records_out:
        plan.unpack()
        torch.cuda.synchronize()
    got = st.cpu().numpy()
    extra = np.concatenate([oa["MU"].cpu().numpy().ravel(), oa["SIGMA"].cpu().numpy().ravel(), oa["R_OBS"].cpu().numpy().ravel().astype(np.float64),
                            oa["FLAG"].cpu().numpy().ravel().astype(np.float64), oa["P"].cpu().numpy().ravel(), oa["P_INDEL"].cpu().numpy().ravel(),
                            oa["R_SIZE"].cpu().numpy().ravel().astype(np.float64), oa["ELT_SIZE"].cpu().numpy().ravel().astype(np.float64)])
    if write_ref:
        np.save(ref_path + ".extra.npy", extra)
    else:
        out["rate_outputs_bit_equal"] = bool(np.array_equal(extra, np.load(ref_path + ".extra.npy"), equal_nan=True))
    if write_ref:
        np.save(ref_path, got)
    else:
        ref = np.load(ref_path)
        m = np.isfinite(ref) & (np.abs(ref) > 1e-250)
        rel = np.zeros_like(ref)
        rel[m] = np.abs(got[m] - ref[m]) / np.abs(ref[m])
        out["max_rel_vs_first"] = [float(rel[j].max()) for j in range(7)]
        out["nan_mismatch"] = int((np.isnan(got) != np.isnan(ref)).sum())
        out["bit_equal_planes"] = [bool(np.array_equal(got[j], ref[j], equal_nan=True)) for j in range(7)]
    print("RESULT " + json.dumps(out), flush=True)


def main():
    if sys.argv[1] == "--child":
        return child(sys.argv[2], sys.argv[3] == "1")
    ref = "/tmp/variant_ref.npy"
    for i, spec in enumerate(sys.argv[1:]):          # lib.so[:ENV=VAL,ENV=VAL]
        lib, _, envs = spec.partition(":")
        path = lib if os.path.isabs(lib) else os.path.join(ROOT, "tools", "variants", lib)
        env = dict(os.environ, DIG_HIP_LIB=path)
        env.update(dict(kv.split("=") for kv in envs.split(",") if kv))
        p = subprocess.run([sys.executable, __file__, "--child", ref, "1" if i == 0 else "0"], env=env, capture_output=True, text=True)
        line = [l for l in p.stdout.split("\n") if l.startswith("RESULT ")]
        print("%-44s %s" % (spec, line[0][7:] if line else "FAILED\n" + p.stdout[-1500:] + p.stderr[-3000:]), flush=True)


if __name__ == "__main__":
    main()
