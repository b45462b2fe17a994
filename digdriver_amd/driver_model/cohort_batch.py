"""elementDriver for MANY cohorts in one pass over device-resident tables.

The reference runs one cohort per process: `DigPretrain.py elementModel` (genic_driver_tools.nonc_model) writes the
element frame, `DigDriver.py elementDriver` (transfer_tools.run_element_region_model, transfer_tools.py:969-1096) joins
the cohort's observed counts and computes the statistics block.  With C pretrained maps on one bin grid the same
arithmetic is a batch along the cohort axis:

    observed counts   dig_overlap_join_* + tabulate_gpu.tabulate_cohorts   -> OBS_* [E, C]     (mutation_tools.py:155-230)
    scale factors     genome mode, transfer_tools.py:129-159: unique mutations in unflagged bins / sum Y_PRED[~FLAG]
    element pipeline  dig_element_pipeline                                 -> every column of the result frames

and the result is one frame per cohort with the reference's column names, identical to what the per-cohort route
gives (tests/test_gpu_cohort_batch.py).
"""
import numpy as np
import pandas as pd

from .. import engine
from ..data_tools import tabulate_gpu
from ..io import mapfile
from ..sequence_model import genic_driver_tools, nb_model

_MUT_COLS10 = ['CHROM', 'START', 'END', 'REF', 'ALT', 'SAMPLE', 'GENE', 'ANNOT', 'MUT_TYPE', 'CONTEXT']


def _read_raw_mutations(f_mut):
    df = pd.read_csv(f_mut, sep="\t", header=None, low_memory=False, dtype={0: str})
    df = df.iloc[:, :8].copy()
    df.columns = _MUT_COLS10[:8]
    return df


def genome_scale_factors(tables, cohorts, device, dev_tables=None):
    """calc_scale_factor_efficient, genome mode (transfer_tools.py:129-159), for all cohorts: (#unique SNVs, #unique
    indels overlapping an unflagged bin) / sum of Y_PRED over unflagged bins.  `cohorts` = encode_mutations() dicts.
    The reference re-reads the intersected rows with read_mutation_file(drop_duplicates=True, unique_indels=True)
    (mutation_tools.py:22-43,45-104): two de-duplications one after the other.  Rows with the same mutation have the same
    coordinates, hence the same bins and the same "an overlapped bin is unflagged" answer, so the de-duplications commute with
    the intersection: they are per-row flags of the FILE (dedup_flags / dig_mutation_file_flags_host, formed where the file
    is parsed), and the device only joins, marks the rows that touch an unflagged bin and counts -- no sort (round 4 ran three
    torch.unique row sorts over the 11 M rows of 37 cohorts here: genome_scale_factors_by_sorting, kept as the cross-check)."""
    import torch
    C = len(cohorts)
    bins = tabulate_gpu.ElementBlocks(tables.chrom, tables.start, tables.start + tables.window, np.arange(len(tables.start)),
                                      len(tables.start), device)
    mu_d, flag = (dev_tables[0], dev_tables[3]) if dev_tables is not None else \
        (torch.as_tensor(tables.mu, device=device), torch.as_tensor(tables.flag, device=device))
    cat = lambda k: torch.cat([c[k] for c in cohorts])
    chrom, cohort = cat("chrom"), cat("cohort")
    pm, pb = tabulate_gpu.overlap_pairs(bins, chrom, cat("start"), cat("end"))
    pm = pm.long()
    ok_pair = flag[bins.elt[pb.long()], cohort[pm]] == 0              # the overlapped bin is unflagged in that cohort
    ok = torch.zeros(chrom.numel(), dtype=torch.bool, device=device)
    ok[pm[ok_pair]] = True                                             # the row touches an unflagged bin
    snv = ok & (cat("first_row") != 0) & (cat("indel") == 0)
    ind = ok & (cat("first_indel") != 0)
    # rows are concatenated cohort by cohort: a cohort's count is a difference of two entries of the running sum (an index_add
    # of 11 M flags onto 37 addresses serialises on its atomics: 0.2 s)
    ends = torch.as_tensor(np.cumsum([c["chrom"].numel() for c in cohorts]), device=device)
    def per_cohort(mask):
        run = torch.cat([torch.zeros(1, dtype=torch.int64, device=device), torch.cumsum(mask.to(torch.int64), 0)])
        at = run[ends]
        return (at - torch.cat([at.new_zeros(1), at[:-1]])).double()
    n_snv, n_ind = per_cohort(snv), per_cohort(ind)
    exp_sum = engine.scale_suffstats(mu_d, flag)
    return n_snv / exp_sum, n_ind / exp_sum


def genome_scale_factors_by_sorting(tables, cohorts, device):
    """genome_scale_factors as round 4 formed it: the de-duplications as three torch.unique row sorts over the intersected
    rows (the cross-check of the flag form; tests/test_gpu_host_mirror.py)."""
    import torch
    C = len(cohorts)
    bins = tabulate_gpu.ElementBlocks(tables.chrom, tables.start, tables.start + tables.window, np.arange(len(tables.start)),
                                      len(tables.start), device)
    flag = torch.as_tensor(tables.flag, device=device)
    cat = lambda k: torch.cat([c[k] for c in cohorts])
    pm, pb = tabulate_gpu.overlap_pairs(bins, cat("chrom"), cat("start"), cat("end"))
    pm = pm.long()
    cohort = cat("cohort")
    ok = flag[bins.elt[pb.long()], cohort[pm]] == 0                    # the overlapped bin is unflagged in that cohort
    # the reference re-reads the intersected rows with read_mutation_file(drop_duplicates=True, unique_indels=True)
    # (mutation_tools.py:22-43,45-104), i.e. two de-duplications one after the other:
    uid_off = np.concatenate([[0], np.cumsum([int(c["uid"].max().item()) + 1 if c["uid"].numel() else 0 for c in cohorts])])
    uid = torch.cat([c["uid"] + int(o) for c, o in zip(cohorts, uid_off[:-1])])
    smp_off = np.concatenate([[0], np.cumsum([len(c["sample_names"]) for c in cohorts])])
    smp = torch.cat([c["sample"] + int(o) for c, o in zip(cohorts, smp_off[:-1])])
    indel = cat("indel")
    gene_off = np.concatenate([[0], np.cumsum([int(c["gene"].max().item()) + 1 if c["gene"].numel() else 0 for c in cohorts])])
    gene = torch.cat([c["gene"] + int(o) for c, o in zip(cohorts, gene_off[:-1])])
    pm_ok = pm[ok]
    # step 1, drop_duplicate_mutations (mutation_tools.py:106-108): first row of every (cohort, uid, sample)
    key, inv = torch.unique(torch.stack([cohort[pm_ok], uid[pm_ok], smp[pm_ok]], dim=1), dim=0, return_inverse=True)
    first = torch.full((key.shape[0],), torch.iinfo(torch.int64).max, dtype=torch.int64, device=device)
    first.scatter_reduce_(0, inv, pm_ok, reduce="amin", include_self=True)      # file order = row index
    is_ind = indel[first] != 0
    n_snv = torch.zeros(C, dtype=torch.float64, device=device).index_add_(0, key[~is_ind, 0], torch.ones(int((~is_ind).sum()), dtype=torch.float64, device=device))
    # step 2, get_unique_indels (mutation_tools.py:110-117): indels recurring in several samples count once per
    # (CHROM, START, END, REF, ALT, GENE) -- GENE is that of the row step 1 kept
    ind_key = torch.unique(torch.stack([key[is_ind, 0], key[is_ind, 1], gene[first[is_ind]]], dim=1), dim=0)
    n_ind = torch.zeros(C, dtype=torch.float64, device=device).index_add_(0, ind_key[:, 0], torch.ones(ind_key.shape[0], dtype=torch.float64, device=device))
    exp_sum = engine.scale_suffstats(torch.as_tensor(tables.mu, device=device), flag)
    return n_snv / exp_sum, n_ind / exp_sum


class _Stages:
    """Wall-clock per stage of run_element_cohorts (tools/e2e_bench.py): the device is drained at every boundary."""

    def __init__(self, sink, dev):
        import time
        self.sink, self.dev, self.clock, self.t = sink, dev, time.perf_counter, time.perf_counter()

    def mark(self, name):
        if self.sink is None:
            return
        import torch
        torch.cuda.synchronize(self.dev)
        now = self.clock()
        self.sink[name] = self.sink.get(name, 0.0) + (now - self.t)
        self.t = now


def run_element_cohorts(f_muts, f_pretrained, f_element_data, save_key, scale_factors=None, max_muts_per_sample=3e9,
                        max_muts_per_elt_per_sample=3e9, device=0, timings=None, read_workers=None, on_frame=None,
                        output_form="records"):
    """One result frame per cohort (index ELT, the columns of run_element_region_model) for the mutation files
    `f_muts[c]` against the pretrained maps `f_pretrained[c]` (one bin grid) and the element set `save_key` of
    `f_element_data`.  scale_factors: (cj [C], cj_indel [C]) or None for the genome mode.
    timings: dict that receives seconds per stage; read_workers: threads that parse the C mutation files and read the C
    maps side by side (default: min(C, cores) from 8 cohorts on, else 1 = in this process).
    on_frame(c, frame, columns, index): called as soon as cohort c's frame exists -- the frame, the dict of column arrays it was
    built from and its row index (run_and_write_element_cohorts hands them to a writer thread there, so that the result files
    are written while the next frames are assembled).
    output_form: "records" (default) or "planes" -- the layout the statistics stage writes on the device (see below; same frames).
    The three inputs are independent until the kernels need them, so they are read SIDE BY SIDE (round 5; one after the other
    they were 0.8 of the 1.5 s of a 37-cohort run): the maps (a thread per map), the element container, and the mutation
    files -- each parsed by the library's own parser and uploaded by its own thread as soon as it is parsed."""
    import os
    import time
    import torch
    from concurrent.futures import ThreadPoolExecutor
    dev = torch.device("cuda", device) if isinstance(device, int) else torch.device(device)
    assert len(f_muts) == len(f_pretrained) and len(f_muts) > 0
    C = len(f_muts)
    if read_workers is None:
        read_workers = min(C, os.cpu_count() or 1) if C >= 8 else 1
    st_ = _Stages(timings, dev)
    first = list(f_pretrained)[0]
    idx0 = mapfile.read_array(first, 'idx')
    w = int(idx0[0, 2] - idx0[0, 1])                        # the grid's window (RegionTables checks that every map has it)
    spans = {}

    def timed(name, fn):
        def run():
            t0 = time.perf_counter()
            out = fn()
            spans[name] = time.perf_counter() - t0
            return out
        return run

    def read_elements():
        elts_ = genic_driver_tools._element_set(f_element_data, w, save_key)
        si_index = mapfile.read_array(f_element_data, 'window_{}/full_window_si_index'.format(w))
        si_values = mapfile.read_array(f_element_data, 'window_{}/full_window_si_values'.format(w))
        return elts_, si_index, si_values

    if read_workers > 1:
        with ThreadPoolExecutor(max_workers=3) as pool:
            fut_maps = pool.submit(timed("read_maps", lambda: genic_driver_tools._load_cohorts(list(f_pretrained), workers=read_workers)))
            fut_elts = pool.submit(timed("read_element_data", read_elements))
            fut_muts = pool.submit(timed("parse_and_upload_mutation_files", lambda: _encode_all_mutations(f_muts, read_workers, device=dev)))
            files, tables, d_pr = fut_maps.result()
            elts, si_index, si_values = fut_elts.result()
            cohorts = fut_muts.result()
    else:
        files, tables, d_pr = timed("read_maps", lambda: genic_driver_tools._load_cohorts(list(f_pretrained), workers=1))()
        elts, si_index, si_values = timed("read_element_data", read_elements)()
        cohorts = timed("parse_and_upload_mutation_files", lambda: _encode_all_mutations(f_muts, 1, device=dev))()
    assert tables.window == w, "the maps' bin grid and the first map's idx disagree"
    E = len(elts['names'])
    ctx = tables.aligned_context(si_index, si_values)
    st_.mark("read_parse_upload")
    if timings is not None:
        timings["inside_read_parse_upload"] = {k: round(v, 4) for k, v in spans.items()}
    ov_ptr, ov_idx = engine.ideal_overlaps(elts['chrom'], elts['blk_ptr'], elts['blk_start'], elts['blk_end'], w,
                                           tables.chrom, tables.start)
    st_.mark("bin_overlaps")
    print('Tabulating mutations')
    owner = np.repeat(np.arange(E), np.diff(elts['blk_ptr']))
    blocks = tabulate_gpu.ElementBlocks(elts['chrom'][owner], elts['blk_start'], elts['blk_end'], owner, E, dev)
    obs_snv, obs_smp, obs_ind, _ = tabulate_gpu.tabulate_cohorts(blocks, cohorts, drop_duplicates=True,
                                                                 max_muts_per_sample=max_muts_per_sample,
                                                                 max_muts_per_elt_per_sample=max_muts_per_elt_per_sample)
    st_.mark("join_tabulate")
    dev_tables = tables.on_device(dev)                      # [N, C] on the device (uploaded cohort-major, transposed there)
    if scale_factors is None:
        print('Calculating scale factor')
        cj, cji = genome_scale_factors(tables, cohorts, dev, dev_tables)
    else:
        cj = torch.as_tensor(np.asarray(scale_factors[0], float), device=dev).reshape(C)
        cji = torch.as_tensor(np.asarray(scale_factors[1], float), device=dev).reshape(C)
    st_.mark("scale_factors")
    print('Calculating statistics')
    t = lambda a: torch.as_tensor(np.ascontiguousarray(a), device=dev)
    args = (*dev_tables, t(ctx), t(ov_ptr), t(ov_idx), t(elts['L']), t(elts['strand_minus']), t(d_pr))
    st_.mark("h2d")
    # The pass.  The statistics stage writes tile-blocked RECORDS (DIG_PIPE_RECORDS: one aligned run per 64-pair tile instead of
    # eleven store streams) and dig_element_records_unpack turns them into what the frames below want -- every plane cohort-major
    # ([C, E]: a cohort's column is one contiguous row; read with a stride of C values, the 24 columns of 37 frames were 0.35 s of
    # cache misses on the host) -- in ONE kernel; the plane form (`output_form="planes"`: eleven [E, C] arrays) needs a transpose per
    # plane behind the pass.  Same bits.  The general (192-substitution) form of the accumulation, as the per-cohort route runs it:
    # the same P to the last bit.
    host = lambda x: x.cpu().numpy()
    by_cohort = lambda x: host(x.transpose(-1, -2).contiguous())
    plan = engine.PipelinePlan(*args, obs_snv, obs_smp, obs_ind, compact=False, pack_bins=True, records_out=(output_form == "records"))
    if plan.records_out:
        acc, _ = plan.run(cj, cji)
        st_cm = torch.empty((len(engine.ES_PLANES), plan.C, plan.E), dtype=torch.float64, device=dev)
        rates = {"MU": torch.empty((plan.C, plan.E), dtype=torch.float64, device=dev),
                 "SIGMA": torch.empty((plan.C, plan.E), dtype=torch.float64, device=dev),
                 "R_OBS": torch.empty((plan.C, plan.E), dtype=torch.int32, device=dev),
                 "FLAG": torch.empty((plan.C, plan.E), dtype=torch.int32, device=dev)}
        plan.unpack(cohort_major=True, stats=st_cm, rates=rates)
        alpha, theta = nb_model.normal_params_to_gamma(rates["MU"], rates["SIGMA"])
        st_.mark("kernels")
        A = {k: host(v) for k, v in acc.items() if k not in ('P', 'MU', 'SIGMA', 'R_OBS', 'FLAG')}
        A.update({k: host(v) for k, v in rates.items()})
        S, al, th = host(st_cm), host(alpha), host(theta)
    else:
        acc, st = plan.run(cj, cji)
        alpha, theta = nb_model.normal_params_to_gamma(acc["MU"], acc["SIGMA"])
        st_.mark("kernels")
        A = {k: (by_cohort(v) if v.dim() == 2 else host(v)) for k, v in acc.items() if k != 'P'}            # (P [E, 1, C]: its one class below)
        S, al, th = by_cohort(st), by_cohort(alpha), by_cohort(theta)
    P0 = host(acc['P'][:, 0, :].transpose(0, 1).contiguous())
    cj_h, cji_h = host(cj), host(cji)
    o_snv, o_smp, o_ind = by_cohort(obs_snv), by_cohort(obs_smp), by_cohort(obs_ind)
    st_.mark("d2h")
    frames = []
    index = pd.Index(elts['names'], name='ELT')              # one object for all cohorts
    for c in range(C):
        have_indel = o_ind[c].sum() != 0
        d = {'ELT_SIZE': A['ELT_SIZE'], 'FLAG': A['FLAG'][c].astype(bool), 'R_SIZE': A['R_SIZE'], 'R_OBS': A['R_OBS'][c],
             'R_INDEL': A['R_OBS'][c], 'MU': A['MU'][c], 'SIGMA': A['SIGMA'][c], 'ALPHA': al[c],
             'THETA': th[c] * cj_h[c], 'MU_INDEL': A['MU'][c], 'SIGMA_INDEL': A['SIGMA'][c], 'ALPHA_INDEL': al[c],
             'THETA_INDEL': S[3][c] if have_indel else th[c], 'Pi_SUM': P0[c], 'Pi_INDEL': A['P_INDEL'],
             'OBS_SAMPLES': o_smp[c].astype(float), 'OBS_SNV': o_snv[c].astype(float), 'OBS_INDEL': o_ind[c].astype(float),
             'EXP_SNV': S[0][c], 'PVAL_SNV_BURDEN': S[1][c], 'PVAL_SAMPLE_BURDEN': S[2][c]}
        if have_indel:                                       # transfer_tools.py:1079-1087
            d.update({'EXP_INDEL': S[4][c], 'PVAL_INDEL_BURDEN': S[5][c], 'PVAL_MUT_BURDEN': S[6][c]})
        frames.append(pd.DataFrame(d, index=index))
        if on_frame is not None:
            on_frame(c, frames[-1], d, index)
    st_.mark("frames")
    return frames


def run_and_write_element_cohorts(f_muts, f_pretrained, f_element_data, save_key, outdir, prefixes, write_threads=None, **kw):
    """run_element_cohorts + write_results with the two overlapped: cohort c's <outdir>/<prefix>.results.txt is written by a
    worker thread (the native writer holds no interpreter lock) while the frames of the cohorts after it are assembled.
    Returns (frames, paths); the files are complete when the call returns."""
    import os
    from concurrent.futures import ThreadPoolExecutor
    os.makedirs(outdir, exist_ok=True)
    C = len(f_muts)
    assert len(prefixes) == C
    paths = [os.path.join(outdir, pfx + '.results.txt') for pfx in prefixes]
    cores = os.cpu_count() or 1
    threads = int(write_threads) if write_threads else max(1, min(8, cores // max(C, 1)))      # threads the writer uses per file
    pending = []
    with ThreadPoolExecutor(max_workers=max(1, min(C, cores))) as pool:
        labels = []

        def on_frame(c, frame, cols, index):
            # the writer takes the frame's COLUMN ARRAYS (the dict the frame was built from, the OBS_* columns as integers:
            # DigDriver.py:108-112) and the row labels encoded once: no pandas object is touched by a writer thread, whose
            # Python part is then a few microseconds (frame.assign + the column extraction of 37 writer threads held the
            # interpreter lock for 0.3 s of the 0.4 s the frames took)
            if not labels:
                labels.append(mapfile.encode_labels(index))
            out_cols = [(k, np.asarray(v).astype(np.int64) if k in ('OBS_SAMPLES', 'OBS_SNV', 'OBS_INDEL') else np.asarray(v)) for k, v in cols.items()]

            def write():
                # (row labels or a column the native writer does not cover -- a quote, a tab or a newline in an element name, an
                #  object column: encode_labels gives None / write_columns_tsv raises ValueError -- go the route write_results takes
                #  for them: pandas.  ADVICE r5: the run used to abort here, behind all the device work.)
                if labels[0] is not None:
                    try:
                        return mapfile.write_columns_tsv(paths[c], index.name, labels[0], out_cols, threads)
                    except ValueError:
                        pass
                return _write_one(frame, paths[c], threads)
            pending.append(pool.submit(write))
        frames = run_element_cohorts(f_muts, f_pretrained, f_element_data, save_key, on_frame=on_frame, **kw)
        for f in pending:
            f.result()
    return frames, paths


def _write_one(df, path, threads):
    """One result file as DigDriver.py writes it (DigDriver.py:108-118: the OBS_* columns as integers)."""
    ints = {col: df[col].astype(int) for col in ('OBS_SAMPLES', 'OBS_SNV', 'OBS_INDEL') if col in df.columns}
    mapfile.write_results_tsv(df.assign(**ints), path, threads=threads)
    return path


def _encode_all_mutations(f_muts, workers=None, device=None):
    """The C mutation files as the host arrays of tabulate_gpu.encode_mutation_file, parsed and encoded side by side: a
    cohort's file is 10^5 - 10^6 rows (pandas: 0.45 s per 300 000; C of them one after the other were the largest stage of a
    many-cohort run).  Threads, not processes: pyarrow's reader and numpy's sorts release the interpreter lock, and worker
    processes would have to be spawned -- this process may hold the GPU -- which re-imports the caller's main module."""
    import os
    n = len(f_muts)
    workers = min(n, os.cpu_count() or 1) if workers is None else int(workers)
    # device: every file's arrays are uploaded by the thread that parsed it, as soon as it is parsed (round 4 uploaded the 37 x 8
    # arrays one after the other behind the last parse: 0.13 s of small pageable copies)
    def job(cf):
        enc = tabulate_gpu.encode_mutation_file(cf[1], cohort_id=cf[0])
        return enc if device is None else tabulate_gpu.to_device(enc, device)
    if workers <= 1 or n == 1:
        return [job(cf) for cf in enumerate(f_muts)]
    from concurrent.futures import ThreadPoolExecutor
    with ThreadPoolExecutor(max_workers=workers) as pool:
        return list(pool.map(job, list(enumerate(f_muts))))


def write_results(frames, outdir, prefixes, workers=None):
    """<outdir>/<prefix>.results.txt per cohort, as DigDriver.py writes it (DigDriver.py:108-118: the OBS_* columns as
    integers, tab-separated, the index column first) -- through the native writer (mapfile.write_results_tsv: the bytes
    of DataFrame.to_csv), several files at a time (the native call does not hold the interpreter lock)."""
    import os
    os.makedirs(outdir, exist_ok=True)
    paths = [os.path.join(outdir, pfx + '.results.txt') for pfx in prefixes]

    jobs = list(zip(frames, paths))
    cores = os.cpu_count() or 1
    workers = min(len(jobs), cores) if workers is None else int(workers)
    threads = max(1, min(8, cores // max(workers, 1)))      # (the writer keeps one buffer per thread: files side by side do not contend)

    def one(job):
        return _write_one(job[0], job[1], threads)

    if jobs:
        one(jobs[0])                                         # (fills the label cache)
    if workers <= 1:
        for job in jobs[1:]:
            one(job)
    elif len(jobs) > 1:
        from concurrent.futures import ThreadPoolExecutor
        with ThreadPoolExecutor(max_workers=workers) as pool:
            list(pool.map(one, jobs[1:]))
    return paths
