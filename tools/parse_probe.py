#!/usr/bin/env python
"""Where does the time of parsing C mutation files side by side go?  (developer probe for the e2e stage `parse_mutation_files`;
needs no GPU, but the host it is meant for is the GPU box with its 256 cores)"""
import os, sys, time, types, threading
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import e2e_bench
from digdriver_amd.data_tools import tabulate_gpu
from digdriver_amd.driver_model import cohort_batch
import pyarrow as pa, pyarrow.csv as pcsv

work = "/tmp/dig_parse_probe"
os.makedirs(work, exist_ok=True)
args = types.SimpleNamespace(bins=288_000, elements=120_091, cohorts=int(os.environ.get("COHORTS", 37)), mut_rows=300_000, seed=3)
paths = e2e_bench.write_inputs(args, work)
files = paths["mut"]
print("cores", os.cpu_count(), "arrow cpu threads", pa.cpu_count(), "io threads", pa.io_thread_count(), "files", len(files))

def staged(path, acc, use_threads=True):
    names = ['CHROM', 'START', 'END', 'REF', 'ALT', 'SAMPLE', 'GENE', 'ANNOT']
    t0 = time.perf_counter()
    text = {k: pa.string() for k in ('CHROM', 'REF', 'ALT', 'SAMPLE', 'GENE', 'ANNOT')}
    all_names = names + ['X0', 'X1']
    tb = pcsv.read_csv(path, read_options=pcsv.ReadOptions(column_names=all_names, use_threads=use_threads, block_size=8 << 20),
                       parse_options=pcsv.ParseOptions(delimiter="\t"),
                       convert_options=pcsv.ConvertOptions(column_types=dict(text, START=pa.int64(), END=pa.int64()), include_columns=names, strings_can_be_null=False))
    t1 = time.perf_counter()
    cols = {}
    for col in ('CHROM', 'REF', 'ALT', 'SAMPLE', 'GENE', 'ANNOT'):
        d = tb[col].combine_chunks().dictionary_encode()
        cols[col] = (d.indices.to_numpy(zero_copy_only=False).astype(np.int64), d.dictionary.to_pylist())
    t2 = time.perf_counter()
    acc["read_csv"] = acc.get("read_csv", 0) + t1 - t0
    acc["dictionary_encode"] = acc.get("dictionary_encode", 0) + t2 - t1

for use_threads in (True, False):
    for workers in (1, 8, len(files)):
        acc = {}
        lock = threading.Lock()
        def job(p):
            a = {}
            staged(p, a, use_threads)
            with lock:
                for k, v in a.items(): acc[k] = acc.get(k, 0) + v
        t0 = time.perf_counter()
        if workers == 1:
            for p in files: job(p)
        else:
            from concurrent.futures import ThreadPoolExecutor
            with ThreadPoolExecutor(workers) as pool: list(pool.map(job, files))
        dt = time.perf_counter() - t0
        print("arrow use_threads=%s workers=%2d: wall %.3f s | summed per file: %s" % (use_threads, workers, dt, {k: round(v, 2) for k, v in acc.items()}))
for workers in (1, 8, len(files)):
    t0 = time.perf_counter()
    enc = cohort_batch._encode_all_mutations(files, workers)
    print("_encode_all_mutations workers=%2d: %.3f s" % (workers, time.perf_counter() - t0))
# the numpy part alone on already parsed arrays
e = enc[0]
t0 = time.perf_counter()
for _ in range(5):
    tabulate_gpu._host_record(e["chrom"], e["start"], e["end"], e["uid"] % 7, e["uid"] % 5, e["sample"], e["sample_names"], e["gene"], e["indel"], 0)
print("_host_record alone: %.3f s per file" % ((time.perf_counter() - t0) / 5))
