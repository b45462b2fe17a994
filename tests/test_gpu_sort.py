"""csrc/dig_sort.hip: the library's batched radix sort of p-values and the Benjamini-Hochberg pass behind it (nb_model.get_q_vals,
nb_model.py:340-342 = statsmodels' fdrcorrection) against numpy: sorted values and order bit for bit, q-values bit for bit with
the host form (whose operations are statsmodels' own), ragged rows, rows that are ranges of a longer list (the sample sort of
parallel.ShardedTiles)."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _rows(rng, lengths, negatives=False):
    out = []
    for n in lengths:
        kind = rng.integers(0, 4)
        if kind == 0:
            p = rng.random(n)
        elif kind == 1:
            p = 10.0 ** rng.uniform(-320, 0, n)                      # every binade down to the subnormals
        elif kind == 2:
            p = rng.choice(rng.random(max(1, n // 50 + 1)), n)       # many ties
        else:
            p = np.minimum(1.0, rng.exponential(0.05, n))            # most of the mass in two binades, some at exactly 1
        if n > 10:
            p[rng.integers(0, n, 3)] = 0.0
            p[rng.integers(0, n, 2)] = 1.0
        if negatives and n > 4:
            p[rng.integers(0, n, max(1, n // 7))] *= -1.0
        out.append(p)
    return out


def _sort(p, rp, want_sorted=True, want_order=True):
    import torch
    from digdriver_amd import _lib
    dev = torch.device("cuda:0")
    t = torch.as_tensor(p, device=dev)
    ps = torch.full_like(t, -7.0) if want_sorted else None
    od = torch.full((t.numel(),), 0x7fffffff, dtype=torch.int32, device=dev) if want_order else None
    wsb = int(_lib.load().dig_bh_ragged_workspace(_lib.host_ptr(rp), rp.size - 1))
    ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
    _lib.call("dig_sort_rows", _lib.dev_ptr(t), _lib.host_ptr(rp), rp.size - 1, _lib.dev_ptr(ps), _lib.dev_ptr(od), _lib.dev_ptr(ws), wsb,
              _lib.stream_ptr())
    torch.cuda.synchronize()
    return (ps.cpu().numpy() if want_sorted else None), (od.cpu().numpy().astype(np.int64) if want_order else None)


@pytest.mark.parametrize("negatives", [False, True])
def test_radix_sort_of_ragged_rows_equals_numpy_stable_sort(negatives):
    """Sorted values and the order: bit for bit numpy's stable sort (NaN last), for rows of 0 ... 1.3 M elements -- tile edges
    (4 095 / 4 096 / 4 097), a single element, an empty row in the middle; with negative values the eighth pass (bit 63) runs."""
    rng = np.random.default_rng(5 + negatives)
    lengths = [0, 1, 2, 63, 64, 65, 4095, 4096, 4097, 0, 8193, 100_003, 1_300_001, 17]
    rows = _rows(rng, lengths, negatives)
    rows[10][[5, 77, 8000]] = np.nan                                 # NaNs sort last, among themselves in their order
    rp = np.concatenate([[0], np.cumsum(lengths)]).astype(np.int64)
    p = np.concatenate(rows)
    ps, od = _sort(p, rp)
    for r, row in enumerate(rows):
        want_order = np.argsort(row, kind="stable")
        got = ps[rp[r]:rp[r + 1]]
        assert np.array_equal(got[~np.isnan(got)], row[want_order][~np.isnan(row[want_order])]), r      # (-0.0 == +0.0: one key)
        assert np.isnan(got).sum() == np.isnan(row).sum() and (not np.isnan(got).any() or np.isnan(got[-np.isnan(row).sum():]).all())
        assert np.array_equal(od[rp[r]:rp[r + 1]], want_order), r


@pytest.mark.parametrize("form", ["default", "careful"])
def test_sort_fix_up_and_the_careful_passes_behind_it(form):
    """The sort is four passes over the upper 36 bits + a fix-up of the runs that share them (csrc/dig_sort.hip): (a) ties far longer
    than the fix-up's reach (5 000 copies of one value) stay a run and are right; (b) 1 000 values that share their upper 36 bits
    and differ below make the fix-up give up, and the seven careful passes behind it sort the lists from the start; (c) the same
    lists with DIG_SORT_FORM=careful (own process): the careful passes alone.  Values and order against numpy's stable sort."""
    import subprocess
    import sys
    import tempfile
    from conftest import ROOT
    rng = np.random.default_rng(41)
    a = rng.random(70_001)
    b = np.concatenate([rng.random(30_000), np.full(5_000, 0.75)])                                 # (a)
    c = np.concatenate([rng.random(20_000), 0.5 + np.arange(1_000) * 2.0 ** -40, [np.nan, 0.0]])      # (b): upper 36 bits shared, low bits differ
    rng.shuffle(b)
    rng.shuffle(c)
    rows = [a, b, c, a[:300]]
    rp = np.concatenate([[0], np.cumsum([len(x) for x in rows])]).astype(np.int64)
    p = np.concatenate(rows)
    if form == "default":
        ps, od = _sort(p, rp)
    else:
        code = ("import sys, numpy as np; sys.path.insert(0, %r); sys.path.insert(0, %r); import test_gpu_sort as t; d = np.load(sys.argv[1]); "
                "ps, od = t._sort(d['p'], d['rp']); np.savez(sys.argv[2], ps=ps, od=od)") % (ROOT, os.path.join(ROOT, "tests"))
        with tempfile.TemporaryDirectory() as tmp:
            np.savez(os.path.join(tmp, "in.npz"), p=p, rp=rp)
            subprocess.check_call([sys.executable, "-c", code, os.path.join(tmp, "in.npz"), os.path.join(tmp, "out.npz")],
                                  env=dict(os.environ, DIG_SORT_FORM="careful"))
            got = np.load(os.path.join(tmp, "out.npz"))
            ps, od = got["ps"], got["od"]
    for r, row in enumerate(rows):
        want = np.argsort(row, kind="stable")
        assert np.array_equal(od[rp[r]:rp[r + 1]], want), r
        assert np.array_equal(ps[rp[r]:rp[r + 1]], row[want], equal_nan=True), r


@pytest.mark.parametrize("longest", [255, 256, 257, 600])
def test_sort_runs_at_the_edge_of_the_fix_up(longest):
    """Runs of values that share their upper 36 bits and differ below, of every length up to `longest`, among exact ties by the
    thousand (p = 1: tiles without a mutation) and spread over tile borders: up to 256 the fix-up orders them, one longer run sends
    the lists through the careful passes; the far-end test of the run (csrc/dig_sort.hip) must call neither too early."""
    rng = np.random.default_rng(longest)
    parts = [np.ones(9_000), rng.random(3_000)]
    for j, m in enumerate([2, 3, 17, 100, longest - 1, longest, longest]):
        base = 0.25 + 0.01 * j
        parts.append(base + rng.permutation(m) * 2.0 ** -42)
    row = np.concatenate(parts)
    rng.shuffle(row)
    rows = [row, np.concatenate([np.ones(20_000), 0.5 + rng.permutation(longest) * 2.0 ** -45]), row[:4097]]
    rp = np.concatenate([[0], np.cumsum([len(x) for x in rows])]).astype(np.int64)
    ps, od = _sort(np.concatenate(rows), rp)
    for r, x in enumerate(rows):
        want = np.argsort(x, kind="stable")
        assert np.array_equal(od[rp[r]:rp[r + 1]], want), r
        assert np.array_equal(ps[rp[r]:rp[r + 1]], x[want]), r


def test_bh_qvalues_of_ragged_rows_equal_the_host_form_bit_for_bit():
    """dig_bh_qvalues_ragged against nb_model.get_q_vals' host form (numpy, statsmodels' own operations) for every row: in
    place, bit for bit; a row with a NaN is all NaN (statsmodels); ties, zeros, subnormals; and the uniform [rows, n] form
    get_q_vals_rows that the per-base route calls."""
    import torch
    from digdriver_amd.sequence_model import nb_model
    rng = np.random.default_rng(11)
    lengths = [5, 0, 4096, 77_777, 1, 300_001, 12_289]
    rows = _rows(rng, lengths)
    rows[3][[9, 70_000]] = np.nan
    rp = np.concatenate([[0], np.cumsum(lengths)]).astype(np.int64)
    p = torch.as_tensor(np.concatenate(rows), device="cuda:0")
    q, rmin = nb_model.bh_ragged(p, rp, want_row_min=True)
    q, rmin = q.cpu().numpy(), rmin.cpu().numpy()
    for r, row in enumerate(rows):
        want = nb_model.get_q_vals(row)
        assert np.array_equal(q[rp[r]:rp[r + 1]], want, equal_nan=True), r
        if len(row) and not np.isnan(row).any():
            srt = np.sort(row)
            assert rmin[r] == (srt / (np.arange(1, len(row) + 1) / float(len(row)))).min()
    m = rng.random((37, 50_021)) ** 3
    got = nb_model.get_q_vals_rows(torch.as_tensor(m, device="cuda:0")).cpu().numpy()
    for c in (0, 17, 36):
        assert np.array_equal(got[c], nb_model.get_q_vals(m[c]))


def test_ranges_of_a_longer_list_finish_with_rank_offset_and_carry():
    """What a rank of the sample sort does: a list cut into three ranges of its value order; every range as a row of its own with
    rank0 = the number of smaller elements, n_global = the length of the whole list and carry = the minimum of the row_min of
    the ranges behind it, gives the q-values of the whole list bit for bit."""
    import torch
    from digdriver_amd.sequence_model import nb_model
    rng = np.random.default_rng(3)
    p = np.minimum(1.0, rng.exponential(0.2, 250_000))
    want = nb_model.get_q_vals(p)
    cuts = np.quantile(p, [0.3, 0.8])
    part = np.searchsorted(cuts, p, side="right")                  # 0 / 1 / 2 by value (equal values stay together)
    pieces = [p[part == k] for k in range(3)]
    rp = np.concatenate([[0], np.cumsum([len(x) for x in pieces])]).astype(np.int64)
    rank0 = rp[:-1].copy()
    n_glob = np.full(3, float(len(p)))
    t = torch.as_tensor(np.concatenate(pieces), device="cuda:0")
    _, rmin = nb_model.bh_ragged(t, rp, n_global=n_glob, rank0=rank0, want_q=False, want_row_min=True)
    rmin = rmin.cpu().numpy()
    carry = np.array([min(rmin[1], rmin[2]), rmin[2], np.inf])
    q, _ = nb_model.bh_ragged(t, rp, n_global=n_glob, rank0=rank0, carry=carry)
    q = q.cpu().numpy()
    for k in range(3):
        assert np.array_equal(q[rp[k]:rp[k + 1]], want[part == k]), k


@pytest.mark.parametrize("form", ["lookup", "payload"])
def test_q_values_by_lookup_and_through_the_payload_are_the_host_form(form):
    """The two ways back to the elements' places (csrc/dig_sort.hip): LOOKUP -- keys-only sort, a table of the records of the reverse
    running minimum per row, every element finds its q by its own value -- and PAYLOAD (DIG_BH_FORM=payload, own process): both the
    host form bit for bit.  Rows: a null-dominated list (a few records), a list whose q strictly increases (every element a record: more
    than the table holds -> the call falls back to the payload form by itself), heavy ties, a single value repeated, odd row starts
    (the tables are 8-byte entries over 4-byte slots), a list with +inf, an empty row, a row with a NaN."""
    import subprocess
    import sys
    import tempfile
    from conftest import ROOT
    from digdriver_amd.sequence_model import nb_model
    rng = np.random.default_rng(23)
    n = 60_001
    rows = [rng.random(n),                                                  # null
            ((np.arange(n) + 1.0) / n) ** 2,                                # every element a record
            rng.choice(rng.random(50), 33_333),                             # ties
            np.full(4_097, 0.3),
            np.concatenate([rng.random(7) ** 3, [np.inf, np.inf]]),
            np.zeros(0),
            np.concatenate([rng.random(5_000), [np.nan]]),
            np.where(rng.random(40_000) < 0.02, rng.random(40_000) ** 6 * 1e-3, rng.random(40_000))]      # a signal among nulls
    for x in rows:
        rng.shuffle(x)
    rp = np.concatenate([[0], np.cumsum([len(x) for x in rows])]).astype(np.int64)
    p = np.concatenate(rows)
    code = ("import sys, numpy as np, torch; sys.path.insert(0, %r); from digdriver_amd.sequence_model import nb_model; d = np.load(sys.argv[1]); "
            "q, m = nb_model.bh_ragged(torch.as_tensor(d['p'], device='cuda:0'), d['rp'], want_row_min=True); "
            "np.savez(sys.argv[2], q=q.cpu().numpy(), m=m.cpu().numpy())") % ROOT
    with tempfile.TemporaryDirectory() as tmp:
        np.savez(os.path.join(tmp, "in.npz"), p=p, rp=rp)
        env = dict(os.environ)
        env.pop("DIG_BH_FORM", None)
        if form == "payload":
            env["DIG_BH_FORM"] = "payload"
        subprocess.check_call([sys.executable, "-c", code, os.path.join(tmp, "in.npz"), os.path.join(tmp, "out.npz")], env=env)
        got = np.load(os.path.join(tmp, "out.npz"))
    for r, x in enumerate(rows):
        want = nb_model.get_q_vals(x) if len(x) else x
        assert np.array_equal(got["q"][rp[r]:rp[r + 1]], want, equal_nan=True), r
    # the lists without the one that overflows the table: the lookup form proper (no fallback), same answer
    import torch
    keep = [0, 2, 3, 4, 7]
    rp2 = np.concatenate([[0], np.cumsum([len(rows[r]) for r in keep])]).astype(np.int64)
    q2, _ = nb_model.bh_ragged(torch.as_tensor(np.concatenate([rows[r] for r in keep]), device="cuda:0"), rp2)
    q2 = q2.cpu().numpy()
    for j, r in enumerate(keep):
        assert np.array_equal(q2[rp2[j]:rp2[j + 1]], nb_model.get_q_vals(rows[r])), r


def test_q_values_of_random_lists_and_ranges_fuzz():
    """Forty random calls of dig_bh_qvalues_ragged (the default lookup form, its fallback included where a list overflows its table):
    one to nine lists of 0 .. 30 000 values from mixtures of uniform values, powers, heavy ties, exact zeros and ones, subnormals and
    +inf; half of the calls as RANGES of longer lists (rank0, n_global, carry) -- against the host form's operations in numpy, bit for
    bit, q-values and row minima."""
    import torch
    from digdriver_amd.sequence_model import nb_model
    rng = np.random.default_rng(2024)

    def values(n):
        kind = rng.integers(0, 6)
        if kind == 0:
            x = rng.random(n)
        elif kind == 1:
            x = rng.random(n) ** rng.integers(2, 9)
        elif kind == 2:
            x = rng.choice(rng.random(max(1, n // rng.integers(2, 200) + 1)), n)
        elif kind == 3:
            x = ((np.arange(n) + 1.0) / max(n, 1)) ** 2                       # every element a record
            rng.shuffle(x)
        elif kind == 4:
            x = np.where(rng.random(n) < 0.03, rng.random(n) ** 8 * 1e-4, 0.5 + 0.1 * rng.random(n))
        else:
            x = 10.0 ** rng.uniform(-320, 0, n)
        if n > 6:
            x[rng.integers(0, n, 2)] = 0.0
            x[rng.integers(0, n, 2)] = 1.0
            if rng.random() < 0.3:
                x[rng.integers(0, n)] = np.inf
        return x

    for call in range(40):
        rows = [values(int(rng.integers(0, 30_000)) if rng.random() < 0.9 else int(rng.integers(0, 4))) for _ in range(int(rng.integers(1, 10)))]
        rp = np.concatenate([[0], np.cumsum([len(x) for x in rows])]).astype(np.int64)
        as_ranges = call % 2 == 1
        n_glob = np.array([len(x) + (int(rng.integers(0, 5000)) if as_ranges else 0) for x in rows], dtype=np.float64)
        rank0 = np.array([int(rng.integers(0, int(n_glob[r]) - len(x) + 1)) if as_ranges else 0 for r, x in enumerate(rows)], dtype=np.int64)
        carry = np.array([(rng.random() * 2.0 if rng.random() < 0.7 else np.inf) if as_ranges else np.inf for _ in rows])
        p = torch.as_tensor(np.concatenate(rows) if rp[-1] else np.zeros(0), device="cuda:0")
        q, rmin = nb_model.bh_ragged(p, rp, n_global=n_glob, rank0=rank0, carry=carry, want_row_min=True)
        q, rmin = q.cpu().numpy(), rmin.cpu().numpy()
        for r, x in enumerate(rows):
            if len(x) == 0:
                assert rmin[r] == np.inf
                continue
            order = np.argsort(x, kind="stable")
            with np.errstate(all="ignore"):
                v = x[order] / ((rank0[r] + np.arange(1, len(x) + 1)) / n_glob[r])
            want_min = v.min()
            run = np.minimum.accumulate(np.minimum(v, carry[r])[::-1])[::-1]
            want = np.empty_like(run)
            want[order] = np.minimum(run, 1.0)
            assert np.array_equal(q[rp[r]:rp[r + 1]], want), (call, r)
            assert rmin[r] == want_min, (call, r)
