"""GPU interval join + integer tabulation against the host implementation (which is pinned to the reference's
integer semantics in tests/test_host_tools.py): observed counts must be bit-exact."""
import numpy as np
import pandas as pd
import pytest

pytestmark = pytest.mark.gpu


def _cohort(rng, n_mut, elements, n_samples, dup_frac=0.05):
    rows = []
    for _ in range(n_mut):
        c, s, e, _n = elements[rng.integers(0, len(elements))]
        p = int(rng.integers(max(s - 300, 0), e + 300))
        if rng.uniform() < 0.15:
            rows.append((str(c), p, p + int(rng.integers(1, 6)), "ACG", "A", "S%d" % rng.integers(0, n_samples), "G1", "INDEL", "DEL", "."))
        else:
            rows.append((str(c), p, p + 1, "A", rng.choice(list("CGT")), "S%d" % rng.integers(0, n_samples), "G1", "Noncoding", "A>T", "CAG"))
    for i in rng.integers(0, len(rows), int(dup_frac * len(rows))):
        r = list(rows[i]); r[6] = "G2"; rows.append(tuple(r))                      # same mutation, second annotation
    rows.append(("X", 10, 11, "A", "T", "S0", ".", "Noncoding", "A>T", "CAG"))
    return pd.DataFrame(rows, columns=['CHROM', 'START', 'END', 'REF', 'ALT', 'SAMPLE', 'GENE', 'ANNOT', 'MUT_TYPE', 'CONTEXT'])


def test_overlap_kernel_against_host_pairs():
    import torch
    from digdriver_amd import _lib
    from digdriver_amd.data_tools import mutation_tools as mt, tabulate_gpu as tg
    _lib.require_device()
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(0)
    nb, nm = 500, 20000
    bc = rng.integers(1, 6, nb); bs = rng.integers(0, 200000, nb); be = bs + rng.integers(0, 3000, nb)   # incl. zero-length
    mc = rng.integers(1, 7, nm); ms = rng.integers(0, 205000, nm); me = ms + rng.integers(0, 4, nm)
    blocks = tg.ElementBlocks(bc, bs, be, np.arange(nb), nb, dev)
    t = lambda a: torch.as_tensor(a.astype(np.int64), device=dev)
    pm, pb = tg.overlap_pairs(blocks, t(mc), t(ms), t(me))
    order = np.lexsort((bs, bc))
    got = set(zip(pm.cpu().numpy().tolist(), order[pb.cpu().numpy()].tolist()))
    mi, bi = mt._overlap_pairs(mc.astype(str), ms, me, bc.astype(str), bs, be)
    assert got == set(zip(mi.tolist(), bi.tolist()))
    assert torch.all(pm[1:] >= pm[:-1])                                              # mutation-major order


@pytest.mark.parametrize("caps", [(1e9, 3e9), (40, 2)])
def test_tabulate_cohorts_bit_exact(tmp_path, caps):
    import torch
    from digdriver_amd.data_tools import mutation_tools as mt, tabulate_gpu as tg
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(3)
    E = 200
    elements, lines = [], []
    for i in range(E):
        c = int(rng.integers(1, 5)); s = int(rng.integers(1000, 400000)); nb = int(rng.integers(1, 4))
        sizes = rng.integers(50, 600, nb); rel = np.concatenate([[0], np.cumsum(sizes + rng.integers(10, 400, nb))[:-1]])
        elements.append((c, s, s + int(rel[-1] + sizes[-1]), "E%03d" % i))
        lines.append("%d\t%d\t%d\tE%03d\t0\t%s\t%d\t%d\t.\t%d\t%s,\t%s,\n" % (
            c, s, s + rel[-1] + sizes[-1], i, "+-"[i % 2], s, s, nb, ",".join(map(str, sizes)), ",".join(map(str, rel))))
    bed = tmp_path / "e.bed"
    bed.write_text("".join(lines))
    names = ["E%03d" % i for i in range(E)]
    blocks, names = tg.ElementBlocks.from_bed12(str(bed), dev, names=names)
    cohorts, want = [], []
    for c in range(3):
        df = _cohort(rng, 3000 + 500 * c, elements, 25 + 5 * c)
        f = tmp_path / ("m%d.tsv" % c)
        df.to_csv(f, sep="\t", header=False, index=False)
        cohorts.append(tg.encode_mutations(df, dev, cohort_id=c))
        tab, black = mt.tabulate_mutations_in_element(str(f), str(bed), bed12=True, drop_duplicates=True,
                                                      max_muts_per_sample=caps[0], max_muts_per_elt_per_sample=caps[1],
                                                      return_blacklist=True)
        want.append((tab.reindex(names).fillna(0).astype(np.int64), sorted(black)))
    snv, smp, ind, blacklists = tg.tabulate_cohorts(blocks, cohorts, max_muts_per_sample=caps[0],
                                                    max_muts_per_elt_per_sample=caps[1])
    assert snv.shape == (E, 3) and snv.dtype == torch.int32
    for c in range(3):
        tab, black = want[c]
        assert np.array_equal(snv[:, c].cpu().numpy(), tab.OBS_SNV.values), c
        assert np.array_equal(ind[:, c].cpu().numpy(), tab.OBS_INDEL.values), c
        assert np.array_equal(smp[:, c].cpu().numpy(), tab.OBS_SAMPLES.values), c
        assert sorted(blacklists[c]) == black
    assert int(snv.sum()) > 0 and int(ind.sum()) > 0
