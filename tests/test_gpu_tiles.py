"""Per-base / tiled route (BASELINE configs[4]): dig_base_tile_probs + dig_tile_mut_counts +
dig_tiled_nb_test against the REFERENCE's own nb_model output (tests/golden/tiled_golden.json.gz, made by running
nb_model.py:188-234 with in-memory fasta / tabix stand-ins), tiles of 1 and of 50 positions, and at the full size of
configs[4] through properties that do not need an oracle."""
import gzip
import json
import os

import numpy as np
import pandas as pd
import pytest

from conftest import GOLDEN, rel_close

pytestmark = pytest.mark.gpu


def _golden():
    return json.loads(gzip.open(os.path.join(GOLDEN, "tiled_golden.json.gz")).read())


@pytest.mark.parametrize("binsize", [1, 50])
def test_nb_model_matches_reference(binsize):
    """nb_model (host mirror, one cohort) == the reference's frame: integer columns exact, Pi / EXP to 1e-12, PVAL within
    the tolerance contract.  The genome holds N runs and soft-masked stretches, one bin starts at 0 and one is cut off by
    the chromosome end; some mutation rows hit a position twice, some are longer than one base."""
    from digdriver_amd.data_tools.genome import PackedGenome
    from digdriver_amd.sequence_model import nb_model
    g = _golden()
    genome = PackedGenome.from_sequences(g["genome"])
    for coh in g["cohorts"]:
        muts = pd.DataFrame(coh["rows"], columns=["CHROM", "START", "END", "REF", "ALT", "ID"])
        muts["CHROM"] = muts.CHROM.astype(str)
        df = nb_model.nb_model(coh["d_pr"], np.array(g["idx"]), coh["mu"], coh["sigma"], muts, genome, n_up=1, n_down=1,
                               binsize=binsize)
        run = coh["runs"][str(binsize)]
        assert len(df) == len(run["PVAL"])
        assert np.array_equal(df.OBS.values, np.array(run["OBS"])) and np.array_equal(df.POS.values, np.array(run["POS"]))
        assert np.array_equal(df.CHROM.values, np.array(run["CHROM"]))
        assert [df.REGION.iloc[0], df.REGION.iloc[-1]] == run["REGION_first_last"]
        np.testing.assert_allclose(df.Pi.values, run["Pi"], rtol=1e-12, atol=0)
        np.testing.assert_allclose(df.EXP.values, run["EXP"], rtol=1e-12, atol=0)
        rel_close(df.PVAL.values, np.array(run["PVAL"]), rtol=1e-6)
        assert int(df.OBS.sum()) > 0 and (df.OBS.values > 1).any()


def test_two_cohorts_in_one_call_equal_single_cohort_calls():
    from digdriver_amd import engine
    from digdriver_amd.data_tools.genome import PackedGenome
    g = _golden()
    genome = PackedGenome.from_sequences(g["genome"])
    ctx = g["contexts"]
    idx = np.array(g["idx"])
    chroms = [str(c) for c in idx[:, 0]]
    S = np.array([[coh["d_pr"][c] for c in ctx] for coh in g["cohorts"]])
    mu = np.array([coh["mu"] for coh in g["cohorts"]])
    sg = np.array([coh["sigma"] for coh in g["cohorts"]])
    rows = [(r[0], r[1], r[2], c) for c, coh in enumerate(g["cohorts"]) for r in coh["rows"]]
    mc, ms, me, co = (np.array([r[j] for r in rows]) for j in range(4))
    both = engine.tiled_nb_model(genome, chroms, idx[:, 1], idx[:, 2], S, mu, sg, mc.astype(str), ms, me, co.astype(np.int32), binsize=50)
    for c in range(2):
        sel = co == c
        one = engine.tiled_nb_model(genome, chroms, idx[:, 1], idx[:, 2], S[c:c + 1], mu[c:c + 1], sg[c:c + 1], mc[sel].astype(str),
                                    ms[sel], me[sel], np.zeros(sel.sum(), np.int32), binsize=50)
        for key in ("pt", "k", "pval", "exp"):
            a, b = both[key][c].cpu().numpy(), one[key][0].cpu().numpy()
            assert np.array_equal(a, b, equal_nan=True), (c, key)
    nv = both["n_valid"].cpu().numpy()
    assert nv.tolist() == [10, 10, 10, 10, 10, 2, 10, 10, 9]            # 499 / 500 / 99 (chr1 end) / 436 (chr2 end) positions
    assert np.isnan(both["pt"][0, 5, 2:].cpu().numpy()).all()          # tiles a short region does not have


@pytest.mark.timeout(900)
@pytest.mark.parametrize("n_up", [1, 2])
def test_full_size_configs4_properties(n_up):
    """BASELINE configs[4] on one GPU at full size: 288 000 bins of 10 kb x 37 cohorts x 200 tiles of 50 positions =
    2.13 G tile tests from a 2.88 Gb genome -- with trinucleotide tables and with the reference's DEFAULT penta-nucleotide
    tables (n_up = n_down = 2: nb_model.py:126,188; 1 024 contexts).  Oracle-free properties: tile probabilities of a bin
    sum to 1, tile counts sum to the mutations inside the bins' positions, pt agrees with an independent torch evaluation on
    sampled bins, p-values are finite probabilities, and a sample of tiles agrees with the elementwise entry point dig_nb_exact."""
    import torch
    from digdriver_amd import engine
    from digdriver_amd.data_tools.genome import PackedGenome
    from digdriver_amd.sequence_model import nb_model
    dev = torch.device("cuda:0")
    R, C, W, B = 288_000, 37, 10_000, 50
    n_chrom = 24
    per = R // n_chrom
    gen = torch.Generator(device=dev).manual_seed(4)
    lengths = np.full(n_chrom, per * W, np.int64)
    n_words = int(lengths.sum() // 8) + 2
    words = torch.randint(0, 2 ** 31 - 1, (n_words,), generator=gen, device=dev, dtype=torch.int64).to(torch.int32) & 0x33333333
    words[0] = 0x44444444
    words[-1] = 0x44444444
    genome = PackedGenome(["chr%d" % i for i in range(n_chrom)], np.arange(n_chrom) * per * W, lengths,
                          np.zeros(2, np.uint32))
    genome._dev[(dev.type, dev.index)] = (words, torch.as_tensor(genome.offsets, device=dev), torch.as_tensor(genome.lengths, device=dev))
    chroms = np.repeat(["chr%d" % i for i in range(n_chrom)], per)
    starts = np.tile(np.arange(per) * W, n_chrom).astype(np.int64)
    ends = starts + W
    S = torch.rand((C, 4 ** (2 * n_up + 1)), generator=gen, device=dev, dtype=torch.float64) * 1e-2
    mu = torch.rand((C, R), generator=gen, device=dev, dtype=torch.float64) * 40 + 5
    sg = torch.rand((C, R), generator=gen, device=dev, dtype=torch.float64) * 6 + 1
    M = 4_000_000
    m_chrom = torch.randint(0, n_chrom, (M,), generator=gen, device=dev)
    m_start = torch.randint(0, per * W, (M,), generator=gen, device=dev)
    m_coh = torch.randint(0, C, (M,), generator=gen, device=dev, dtype=torch.int64).to(torch.int32)
    pt, first, nval = engine.base_tile_probs(genome, chroms, starts, ends, S, B, device=dev)
    assert pt.shape == (C, R, 200)
    k = engine.tile_mut_counts(genome, chroms, starts, ends, first, nval, m_chrom, m_start, m_start + 1, m_coh, C, B, 200)
    pval, ex = engine.tiled_nb_test(pt, k, mu, sg)
    torch.cuda.synchronize()
    assert int(nval.min()) == 200 and int(nval.max()) == 200
    sums = pt.sum(dim=2)
    assert float((sums - 1).abs().max()) < 1e-12
    # every mutation lies in exactly one bin; those within n_up of a chromosome's end have no window
    pos_in_chrom = m_start
    inside = (pos_in_chrom >= n_up) & (pos_in_chrom <= per * W - 1 - n_up)
    assert int(k.sum()) == int(inside.sum())
    per_cohort = torch.bincount(m_coh[inside].long(), minlength=C)
    assert torch.equal(k.sum(dim=(1, 2)), per_cohort)
    # independent evaluation of pt on sampled bins: per-position contexts with torch
    code = torch.stack([(words >> (4 * j)) & 15 for j in range(8)], dim=1).reshape(-1)[8:]      # bases from word 1 on
    for r in (0, 1, per - 1, per, 17 * per + 123, R - 1):
        ci, s = r // per, (r % per) * W
        f = max(s, n_up)
        stop = min(s + W, per * W - n_up)
        g0 = ci * per * W
        ctx = torch.zeros(stop - f, dtype=torch.int64, device=dev)
        for o in range(-n_up, n_up + 1):                                    # index in itertools.product('ACGT', repeat = 2 n_up + 1) order
            ctx = ctx * 4 + code[g0 + f + o:g0 + stop + o].long()
        probs = S[:, ctx]                                                   # [C, n_pos]
        tot = probs.sum(dim=1, keepdim=True)
        n_pos = probs.shape[1]
        pad = (-n_pos) % B
        tiles = torch.nn.functional.pad(probs, (0, pad)).reshape(C, -1, B).sum(dim=2) / tot
        got = pt[:, r, :tiles.shape[1]]
        assert float(((got - tiles).abs() / tiles).max()) < 1e-12, r
    assert bool(torch.isfinite(pval).all()) and float(pval.min()) >= 0.0 and float(pval.max()) <= 1.0
    assert float((ex - pt * mu[:, :, None]).abs().max()) == 0.0
    # the same arithmetic through the elementwise entry point on a sample (incl. the tiles holding the largest counts)
    flat = torch.cat([torch.randint(0, pval.numel(), (200_000,), generator=gen, device=dev), (k.reshape(-1) >= 2).nonzero().reshape(-1)[:5000],
                      k.reshape(-1).argmax().reshape(1)])
    cc, rr = flat // (R * 200), (flat // 200) % R
    alpha, theta = nb_model.normal_params_to_gamma(mu[cc, rr], sg[cc, rr])
    p = 1.0 / (pt.reshape(-1)[flat] * theta + 1.0)
    want = nb_model.nb_pvalue_exact(k.reshape(-1)[flat].double(), alpha, p)
    assert torch.equal(pval.reshape(-1)[flat], want)


def test_q_values_on_the_device_equal_the_host_form():
    """get_q_vals on a CUDA tensor (whole-genome tile sets) == the host form, which is pinned to statsmodels
    (tests/golden/qvals_golden.json): same operations in the same order, same bits; ties and NaNs included."""
    import torch
    from digdriver_amd.sequence_model import nb_model
    rng = np.random.default_rng(8)
    p = rng.uniform(size=200_003) ** 3
    p[rng.integers(0, p.size, 500)] = p[rng.integers(0, p.size, 500)]            # ties
    p[:7] = [0.0, 1.0, 1e-300, 0.5, 0.5, 1.0, 0.0]
    want = nb_model.get_q_vals(p)
    got = nb_model.get_q_vals(torch.as_tensor(p, device="cuda:0")).cpu().numpy()
    assert np.array_equal(got, want)
    q2 = nb_model.get_q_vals(torch.as_tensor(p.reshape(-1, 1), device="cuda:0"))
    assert q2.shape == (p.size, 1) and np.array_equal(q2.cpu().numpy()[:, 0], want)


def _tile_problem(seed=3, C=3):
    """A small genome (three chromosomes of different lengths, N runs, one too short for a whole bin), 1 kb bins in genome
    order with a ragged last bin per chromosome, mutations of C cohorts incl. rows on a chromosome without bins."""
    from digdriver_amd.data_tools.genome import PackedGenome
    rng = np.random.default_rng(seed)
    seqs = {"chr1": "".join(rng.choice(list("ACGTN"), 23_017, p=[.24, .25, .25, .24, .02])),
            "chr2": "".join(rng.choice(list("ACGT"), 9_400)), "chr3": "".join(rng.choice(list("ACGT"), 731)),
            "chrX": "".join(rng.choice(list("ACGT"), 500))}
    seqs["chr1"] = seqs["chr1"][:5000] + "N" * 1500 + seqs["chr1"][6500:]
    genome = PackedGenome.from_sequences(seqs)
    chroms, starts, ends = [], [], []
    for name in ("chr1", "chr2", "chr3"):
        for s in range(0, len(seqs[name]), 1000):
            chroms.append(name)
            starts.append(s)
            ends.append(min(s + 1000, len(seqs[name]) + 40))        # the last bin pokes over the chromosome end
    R = len(chroms)
    S = rng.uniform(1e-3, 1e-2, (C, 64))
    mu, sg = rng.uniform(3, 40, (C, R)), rng.uniform(1, 6, (C, R))
    M = 6000
    mc = rng.choice(["chr1", "chr2", "chr3", "chrX"], M, p=[.6, .3, .05, .05])
    ms = np.array([rng.integers(0, len(seqs[c])) for c in mc])
    me = ms + rng.integers(1, 4, M)
    co = rng.integers(0, C, M).astype(np.int32)
    return seqs, genome, np.array(chroms), np.array(starts), np.array(ends), S, mu, sg, mc, ms, me, co


@pytest.mark.parametrize("n_up", [1, 2])
def test_sharded_tiles_equal_unsharded_bit_for_bit(n_up):
    """parallel.ShardedTiles (BASELINE configs[4]: the per-base route sharded by bins): the ranks of a 2-, 3- and 8-rank
    plan, walked one after the other on one device -- each on its own slab of the genome, with shifted coordinates and its
    own mutations -- reproduce the single-device result bit for bit (pt, k, exp, pval, first positions), and so do the
    Benjamini-Hochberg q-values formed from the rank-ordered concatenation of the ranks' p-values.  n_up = 2: the reference's
    default penta-nucleotide tables (a slab's margin must then cover two bases on either side of its first and last bin)."""
    import torch
    from digdriver_amd import engine, parallel
    from digdriver_amd.sequence_model import nb_model
    dev = torch.device("cuda:0")
    seqs, genome, chroms, starts, ends, S, mu, sg, mc, ms, me, co = _tile_problem()
    if n_up == 2:
        S = np.random.default_rng(8).uniform(1e-3, 1e-2, (S.shape[0], 1024))
    C, R = S.shape[0], len(chroms)
    whole = engine.tiled_nb_model(genome, chroms, starts, ends, S, mu, sg, mc[mc != "chrX"], ms[mc != "chrX"], me[mc != "chrX"],
                                  co[mc != "chrX"], binsize=50, device=0)
    assert int(whole["k"].sum()) > 1000
    T = whole["pt"].shape[2]
    valid = torch.arange(T, device=dev)[None, :] < whole["n_valid"][:, None]
    for world in ((2, 3, 8, 40) if n_up == 1 else (3, 8)):    # 40 ranks for 35 bins: some ranks hold no bin at all
        ranks = [parallel.ShardedTiles(genome, chroms, starts, ends, S, mu, sg, mc, ms, me, co, 50, dev, r, world) for r in range(world)]
        res = [r.run() for r in ranks]
        assert sum(len(r.genome.words) for r in ranks if r.hi > r.lo) < len(genome.words) + 6 * world
        for key in ("pt", "k", "exp", "pval"):
            got = torch.cat([x[key] for x in res], dim=1)
            assert got.shape == whole[key].shape
            a, b = got.double(), whole[key].double()
            assert torch.equal(torch.nan_to_num(a, nan=-7.0), torch.nan_to_num(b, nan=-7.0)), (world, key)
        assert torch.equal(torch.cat([x["first_pos"] for x in res]), whole["first_pos"])
        assert torch.equal(torch.cat([x["n_valid"] for x in res]), whole["n_valid"])
        for r_, x in zip(ranks, res):                           # (the all-N bin's NaN p-values would make every q NaN, as in statsmodels)
            x["pval"] = torch.nan_to_num(x["pval"], nan=0.5)
        whole_p = torch.nan_to_num(whole["pval"], nan=0.5)
        for c in range(C):
            parts = [r.valid_pvalues(c)[0] for r in ranks]
            allp = torch.cat(parts)
            assert torch.equal(allp, whole_p[c][valid])                             # = the PVAL column of the reference's frame
            want = nb_model.get_q_vals(whole_p[c][valid])
            assert bool(torch.isfinite(want).all())
            off, got = 0, []
            for r, p in zip(ranks, parts):
                q = r.q_values(c, gathered=allp, offset=off)
                got.append(q[r.valid_pvalues(c)[1]])
                off += p.numel()
            assert torch.equal(torch.cat(got), want), (world, c)
        with pytest.raises(ValueError):
            ranks[0].q_values(0, gathered=allp)                                    # the caller's exchange needs the rank's offset
    # all cohorts at once (one segmented sort, one batched pass): the bits of the per-cohort calls, with batches of one, two and all cohorts
    one = parallel.ShardedTiles(genome, chroms, starts, ends, S, mu, sg, mc, ms, me, co, 50, dev, 0, 1)
    res1 = one.run()
    res1["pval"] = torch.nan_to_num(res1["pval"], nan=0.5)
    per_cohort = torch.stack([one.q_values(c) for c in range(C)])
    for cap in (1, 2 * int(valid.sum()), 1 << 28):
        got = one.q_values_all(max_elements=cap)
        assert torch.equal(torch.nan_to_num(got, nan=-7.0), torch.nan_to_num(per_cohort, nan=-7.0)), cap


@pytest.mark.parametrize("binsize", [50, 1])
def test_nb_model_penta_default_mode_matches_reference(binsize):
    """The per-base route in the reference's DEFAULT mode (n_up = n_down = 2: penta-nucleotide contexts, 1 024-entry S_prob;
    nb_model.py:126,188, sequence_tools.py:292): nb_model called WITHOUT n_up / n_down against the reference's own frames
    (tests/golden/tiled_penta_golden.json.gz: default call and binsize = 1)."""
    import itertools
    from digdriver_amd.data_tools.genome import PackedGenome
    from digdriver_amd.sequence_model import nb_model
    g = json.loads(gzip.open(os.path.join(GOLDEN, "tiled_penta_golden.json.gz")).read())
    genome = PackedGenome.from_sequences(g["genome"])
    ctx = ["".join(t) for t in itertools.product("ACGT", repeat=5)]
    for coh in g["cohorts"]:
        muts = pd.DataFrame(coh["rows"], columns=["CHROM", "START", "END", "REF", "ALT", "ID"])
        muts["CHROM"] = muts.CHROM.astype(str)
        d_pr = dict(zip(ctx, coh["d_pr"]))
        kw = {} if binsize == 50 else {"binsize": 1}
        df = nb_model.nb_model(d_pr, np.array(g["idx"]), coh["mu"], coh["sigma"], muts, genome, **kw)       # the defaults
        run = coh["runs"][str(binsize)]
        assert len(df) == len(run["PVAL"])
        assert np.array_equal(df.OBS.values, np.array(run["OBS"])) and np.array_equal(df.POS.values, np.array(run["POS"]))
        assert [df.REGION.iloc[0], df.REGION.iloc[-1]] == run["REGION_first_last"]
        np.testing.assert_allclose(df.Pi.values, run["Pi"], rtol=1e-12, atol=0)
        np.testing.assert_allclose(df.EXP.values, run["EXP"], rtol=1e-12, atol=0)
        rel_close(df.PVAL.values, np.array(run["PVAL"]), rtol=1e-6)
        assert int(df.OBS.sum()) > 0


def test_general_context_kernel_many_cohorts_against_oracle_and_trinucleotide_kernels():
    """base_tile_probs_ctx_kernel: 37 cohorts (three LDS chunks of the table), ragged last tiles, N runs, a region at a
    chromosome's start and one over its end, fewer tiles asked for than a region has -- penta-nucleotide tables against the
    oracle; and the same kernel at n_up = 1 (DIG_TILES_FORM=general, own process) against the trinucleotide kernels."""
    import subprocess
    import sys
    import torch
    from digdriver_amd import engine
    from digdriver_amd.data_tools.genome import PackedGenome
    from oracle import dig_oracle as O
    rng = np.random.default_rng(17)
    seqs = {"chr1": "".join(rng.choice(list("ACGTN"), 4211, p=[.24, .25, .25, .24, .02])), "chr2": "".join(rng.choice(list("ACGT"), 1803))}
    genome = PackedGenome.from_sequences(seqs)
    chroms = ["chr1"] * 5 + ["chr2"] * 2
    starts = np.array([0, 1000, 2000, 3000, 4000, 0, 1000], np.int64)
    ends = np.array([1000, 2000, 3000, 4000, 5000, 1000, 2000], np.int64)          # two regions poke over the chromosome end
    C = 37
    S5 = rng.uniform(1e-4, 1e-2, (C, 1024))
    for binsize, n_tiles in ((50, None), (7, None), (50, 11)):
        pt, first, nval = engine.base_tile_probs(genome, chroms, starts, ends, S5, binsize, n_tiles=n_tiles, device=0)
        pt, first, nval = pt.cpu().numpy(), first.cpu().numpy(), nval.cpu().numpy()
        for r in range(len(chroms)):
            for c in (0, 15, 16, 36):
                probs, poss = O.base_probabilities_by_region(seqs[chroms[r]], S5[c], int(starts[r]), int(ends[r]), n_up=2)
                assert first[r] == poss[0]
                want = np.array([probs[i:i + binsize].sum() for i in range(0, len(probs), binsize)])
                assert nval[r] == min(len(want), pt.shape[2])
                np.testing.assert_allclose(pt[c, r, :nval[r]], want[:nval[r]], rtol=1e-12)
                assert np.isnan(pt[c, r, nval[r]:]).all()
    code = ("import sys, numpy as np, torch; sys.path.insert(0, %r); from digdriver_amd import engine; "
            "from digdriver_amd.data_tools.genome import PackedGenome; d = np.load(sys.argv[1], allow_pickle=True); "
            "g = PackedGenome.from_sequences(d['seqs'].item()); "
            "pt, f, n = engine.base_tile_probs(g, list(d['chroms']), d['starts'], d['ends'], d['S'], 50, device=0); "
            "np.savez(sys.argv[2], pt=pt.cpu().numpy(), f=f.cpu().numpy(), n=n.cpu().numpy())") % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    import tempfile
    with tempfile.TemporaryDirectory() as tmp:
        S3 = rng.uniform(1e-4, 1e-2, (5, 64))
        np.savez(os.path.join(tmp, "in.npz"), seqs=np.array(seqs, dtype=object), chroms=np.array(chroms), starts=starts, ends=ends, S=S3)
        out = {}
        for form in ("mfma", "general", "rows"):
            env = dict(os.environ, DIG_TILES_FORM=form)
            subprocess.check_call([sys.executable, "-c", code, os.path.join(tmp, "in.npz"), os.path.join(tmp, form + ".npz")], env=env)
            out[form] = np.load(os.path.join(tmp, form + ".npz"))
        for form in ("general", "rows"):                 # (rows: the row walk of dig_tiles_rows.hip at n_up = 1)
            assert np.array_equal(out["mfma"]["f"], out[form]["f"]) and np.array_equal(out["mfma"]["n"], out[form]["n"])
            np.testing.assert_allclose(out[form]["pt"], out["mfma"]["pt"], rtol=1e-12, equal_nan=True)


def test_row_walk_on_ten_kb_bins_against_the_general_kernel_and_the_oracle():
    """base_tile_probs_rows_kernel (dig_tiles_rows.hip: what dig_base_tile_probs_ctx runs at n_up = 2) on 10-kb bins, the
    shape of BASELINE configs[4]: more regions than workgroups, 37 cohorts (passes of 16 + 16 + 5), N runs and single N,
    a bin at a chromosome's start and one over its end, a 12-kb region (more positions than the walk stages: left to the general
    kernel by the n_valid = -2 mark), binsize 50 / 64 / 60 (trips of 10 / 8 / 12 positions; fewer tiles asked for than a region has) and 1 (every region deferred);
    against the general kernel (DIG_TILES_FORM=general, own process) and, for a sample, the oracle."""
    import subprocess
    import sys
    import tempfile
    import torch
    from digdriver_amd import engine
    from digdriver_amd.data_tools.genome import PackedGenome
    from oracle import dig_oracle as O
    rng = np.random.default_rng(23)
    n1, n2 = 3_000_000 + 4321, 612_345
    a = rng.choice(list("ACGT"), n1)
    a[rng.integers(0, n1, 300)] = "N"
    for s0 in rng.integers(0, n1 - 3000, 12):
        a[s0:s0 + int(rng.integers(1, 2500))] = "N"
    a[20_000:30_000] = "N"                                       # a bin without any window
    seqs = {"chr1": "".join(a), "chr2": "".join(rng.choice(list("ACGT"), n2))}
    genome = PackedGenome.from_sequences(seqs)
    chroms = ["chr1"] * 301 + ["chr2"] * 62 + ["chr1"]
    starts = np.concatenate([np.arange(301) * 10_000, np.arange(62) * 10_000, [1_000_000]]).astype(np.int64)
    ends = starts + 10_000
    ends[-1] = starts[-1] + 12_000                               # deferred: 12 000 positions
    C = 37
    S5 = rng.uniform(1e-4, 1e-2, (C, 1024))
    code = ("import sys, numpy as np, torch; sys.path.insert(0, %r); from digdriver_amd import engine; "
            "from digdriver_amd.data_tools.genome import PackedGenome; d = np.load(sys.argv[1], allow_pickle=True); "
            "g = PackedGenome.from_sequences(d['seqs'].item()); "
            "pt, f, n = engine.base_tile_probs(g, list(d['chroms']), d['starts'], d['ends'], d['S'], int(sys.argv[3]), "
            "n_tiles=(int(sys.argv[4]) or None), device=0); "
            "np.savez(sys.argv[2], pt=pt.cpu().numpy(), f=f.cpu().numpy(), n=n.cpu().numpy())") % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with tempfile.TemporaryDirectory() as tmp:
        np.savez(os.path.join(tmp, "in.npz"), seqs=np.array(seqs, dtype=object), chroms=np.array(chroms), starts=starts, ends=ends, S=S5)
        for binsize, n_tiles, sel in ((50, 0, slice(None)), (64, 0, slice(None)), (60, 100, slice(None)), (1, 0, slice(0, 3))):
            sub = os.path.join(tmp, "sub.npz")
            np.savez(sub, seqs=np.array(seqs, dtype=object), chroms=np.array(chroms)[sel], starts=starts[sel], ends=ends[sel], S=S5)
            env = dict(os.environ, DIG_TILES_FORM="general")
            subprocess.check_call([sys.executable, "-c", code, sub, os.path.join(tmp, "general.npz"), str(binsize), str(n_tiles)], env=env)
            want = np.load(os.path.join(tmp, "general.npz"))
            pt, first, nval = engine.base_tile_probs(genome, list(np.array(chroms)[sel]), starts[sel], ends[sel], S5, binsize,
                                                     n_tiles=n_tiles or None, device=0)
            pt, first, nval = pt.cpu().numpy(), first.cpu().numpy(), nval.cpu().numpy()
            assert np.array_equal(first, want["f"]) and np.array_equal(nval, want["n"]) and nval.min() >= 0
            assert np.array_equal(np.isnan(pt), np.isnan(want["pt"]))
            np.testing.assert_allclose(pt, want["pt"], rtol=1e-13, equal_nan=True)
            if binsize == 50:
                assert np.isnan(pt[:, 2]).all() and nval[2] == 200                      # the all-N bin: 0 / 0
                for r in (0, 1, 7, 300, 301, 362, 363):
                    for c in (0, 15, 16, 31, 32, 36):
                        probs, poss = O.base_probabilities_by_region(seqs[chroms[r]], S5[c], int(starts[r]), int(ends[r]), n_up=2)
                        assert first[r] == poss[0]
                        ref = np.array([probs[i:i + binsize].sum() for i in range(0, len(probs), binsize)])
                        assert nval[r] == min(len(ref), pt.shape[2])
                        np.testing.assert_allclose(pt[c, r, :nval[r]], ref[:nval[r]], rtol=1e-12)
                        assert np.isnan(pt[c, r, nval[r]:]).all()


def test_two_role_matrix_kernel_gives_the_bits_of_the_one_role_kernel():
    """base_tile_probs_roles_kernel (walker waves + multiplier waves, what dig_base_tile_probs launches for 32 cohorts and more)
    against base_tile_probs_mfma_kernel (DIG_TILES_FORM=one-role, own process): 37 and 48 cohorts, a single region and more
    regions than workgroups, ragged last tiles, N runs, regions at a chromosome's start and over its end, fewer tiles asked
    for than a region has (positions behind the last tile), binsize 7 -- every value bit for bit, NaN for NaN."""
    import subprocess
    import sys
    import tempfile
    rng = np.random.default_rng(23)
    seqs = {"chr1": "".join(rng.choice(list("ACGTN"), 60211, p=[.24, .25, .25, .24, .02])), "chr2": "".join(rng.choice(list("ACGT"), 1803))}
    n1 = 600                                                   # more regions than the 256 workgroups of the two-role kernel
    chroms = ["chr1"] * n1 + ["chr2"] * 2
    starts = np.r_[np.arange(n1, dtype=np.int64) * 100, [0, 1000]]
    ends = np.r_[starts[:n1] + rng.integers(1, 1000, n1), [1000, 2000]]
    ends[n1 - 1] = 70000                                       # over the chromosome's end
    code = ("import sys, numpy as np, torch; sys.path.insert(0, %r); from digdriver_amd import engine; "
            "from digdriver_amd.data_tools.genome import PackedGenome; d = np.load(sys.argv[1], allow_pickle=True); "
            "g = PackedGenome.from_sequences(d['seqs'].item()); out = {}\n"
            "for k, (C, binsize, n_tiles, nreg) in enumerate(((37, 50, None, None), (48, 50, 11, None), (37, 7, 120, None), (33, 50, None, 1), (64, 25, None, 300))):\n"
            "    n = nreg or len(d['chroms'])\n"
            "    pt, f, nv = engine.base_tile_probs(g, list(d['chroms'][:n]), d['starts'][:n], d['ends'][:n], d['S'][:C], binsize, n_tiles=n_tiles, device=0)\n"
            "    out['pt%%d' %% k], out['f%%d' %% k], out['n%%d' %% k] = pt.cpu().numpy(), f.cpu().numpy(), nv.cpu().numpy()\n"
            "np.savez(sys.argv[2], **out)") % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with tempfile.TemporaryDirectory() as tmp:
        np.savez(os.path.join(tmp, "in.npz"), seqs=np.array(seqs, dtype=object), chroms=np.array(chroms), starts=starts, ends=ends,
                 S=rng.uniform(1e-4, 1e-2, (64, 64)))
        out = {}
        for form in ("one-role", "two-role"):
            env = dict(os.environ, DIG_TILES_FORM=form)
            subprocess.check_call([sys.executable, "-c", code, os.path.join(tmp, "in.npz"), os.path.join(tmp, form + ".npz")], env=env)
            out[form] = np.load(os.path.join(tmp, form + ".npz"))
        assert len(out["one-role"].files) == 15
        for key in out["one-role"].files:
            a, b = out["one-role"][key], out["two-role"][key]
            assert a.shape == b.shape and a.tobytes() == b.tobytes(), key
        assert np.isfinite(out["two-role"]["pt0"]).sum() > 1000


def test_collapsed_contexts_k96_per_base_route_and_counts_match_reference():
    """collapse=True (the north star's "96-trinucleotide-context" wording; no live caller of the reference passes it): the
    per-base route nb_model(..., collapse=True) for trinucleotide (32-context S_prob) and penta-nucleotide (512) tables and
    count_contexts_by_regions(collapse=True) against the reference's own outputs (tests/golden/collapse_golden.json.gz)."""
    from digdriver_amd.data_tools.genome import PackedGenome
    from digdriver_amd.sequence_model import nb_model, sequence_tools
    g = json.loads(gzip.open(os.path.join(GOLDEN, "collapse_golden.json.gz")).read())
    genome = PackedGenome.from_sequences(g["genome"])
    regs = g["count_regions"]
    cc = sequence_tools.count_contexts_by_regions(genome, [r[0] for r in regs], [r[1] for r in regs], [r[2] for r in regs],
                                                  n_up=1, n_down=1, collapse=True)
    assert list(cc.columns) == g["count_columns"] and list(cc.index) == g["count_index"]
    assert np.array_equal(cc.values, np.array(g["count_values"]))
    muts = pd.DataFrame(g["rows"], columns=["CHROM", "START", "END", "REF", "ALT", "ID"])
    muts["CHROM"] = muts.CHROM.astype(str)
    for n_up in (1, 2):
        run = g["runs"][str(n_up)]
        df = nb_model.nb_model(dict(zip(run["keys"], run["d_pr"])), np.array(g["idx"]), g["mu"], g["sigma"], muts, genome, n_up=n_up,
                               n_down=n_up, binsize=25, collapse=True)
        assert len(df) == len(run["PVAL"])
        assert np.array_equal(df.OBS.values, np.array(run["OBS"])) and np.array_equal(df.POS.values, np.array(run["POS"]))
        np.testing.assert_allclose(df.Pi.values, run["Pi"], rtol=1e-12, atol=0)
        np.testing.assert_allclose(df.EXP.values, run["EXP"], rtol=1e-12, atol=0)
        rel_close(df.PVAL.values, np.array(run["PVAL"]), rtol=1e-6)


def test_bh_pass_of_the_library_equals_the_host_form_at_every_size():
    """dig_bh_qvalues_sorted (round 5: p / (rank / n), reverse running minimum and cap in one pass behind torch's sort -- torch.cummin
    was 97 % of a cohort's q-values) against the host form of get_q_vals = statsmodels' operations, bit for bit: sizes around the
    thread, workgroup and chunk boundaries of the kernel, ties at 0 and 1, and a NaN, which makes every q-value NaN as it does
    in statsmodels."""
    import torch
    from digdriver_amd.sequence_model import nb_model
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(3)
    for n in (1, 2, 15, 16, 17, 255, 256, 257, 4095, 4096, 4097, 8193, 100_003, 1_048_577, 7_200_001):
        p = rng.uniform(0, 1, n) ** 3
        if n > 10:
            p[rng.integers(0, n, 5)] = 1.0
            p[rng.integers(0, n, 5)] = 0.0
        want = nb_model.get_q_vals(p)
        got = nb_model.get_q_vals(torch.as_tensor(p, device=dev)).cpu().numpy()
        assert np.array_equal(want, got), n
        if n > 3:
            p[n // 2] = np.nan
            want = nb_model.get_q_vals(p)
            got = nb_model.get_q_vals(torch.as_tensor(p, device=dev)).cpu().numpy()
            assert np.isnan(want).all() and np.isnan(got).all(), n
