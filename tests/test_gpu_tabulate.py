"""GPU interval join + integer tabulation against the ORACLE (oracle.interval_join_pairs /
tabulate_elements: plain loops restating bedtools' overlap and mutation_tools.py:155-230, pinned on CPU to frames the
reference's own functions returned, tests/test_oracle_golden.py) and against those reference frames themselves
(tests/golden/tabulate_golden.json.gz).  Observed counts are integers: everything here is bit-exact."""
import gzip
import json
import os

import numpy as np
import pandas as pd
import pytest

from conftest import GOLDEN
from oracle import dig_oracle as O

pytestmark = pytest.mark.gpu

_COLS = ['CHROM', 'START', 'END', 'REF', 'ALT', 'SAMPLE', 'GENE', 'ANNOT', 'MUT_TYPE', 'CONTEXT']


def _elements(rng, n, chroms=("1", "2", "3", "chr4", "X"), zero_len_every=0):
    """bed12 rows: 1-4 blocks; every 5th element starts inside an earlier one (nested / overlapping blocks of different
    elements); optionally zero-length blocks."""
    bed, spans = [], []
    for i in range(n):
        c = chroms[int(rng.integers(0, len(chroms)))]
        nb = int(rng.integers(1, 5))
        if i % 5 == 4 and spans:
            c, s0, e0 = spans[int(rng.integers(0, len(spans)))]
            start = int(rng.integers(s0, max(s0 + 1, e0 - 20)))
        else:
            start = int(rng.integers(1000, 300000))
        sizes = rng.integers(30, 500, nb)
        if zero_len_every and i % zero_len_every == 1:
            sizes[int(rng.integers(0, nb))] = 0
        rel = np.concatenate([[0], np.cumsum(sizes + rng.integers(1, 200, nb))[:-1]])
        end = start + int(rel[-1] + sizes[-1])
        spans.append((c, start, max(end, start + 1)))
        trail = "," if i % 2 else ""
        bed.append([c, str(start), str(end), "E%04d" % i, "0", "+-"[i % 2], str(start), str(start), ".", str(nb),
                    ",".join(map(str, sizes)) + trail, ",".join(map(str, rel)) + trail])
    return bed


def _mutations(rng, n, blocks, n_samples, chrom_extra="9"):
    rows = []
    for _ in range(n):
        c, s, e = blocks[int(rng.integers(0, len(blocks)))][:3]
        p = int(rng.integers(max(s - 100, 0), e + 100))
        smp = "S%02d" % int(min(rng.geometric(0.15), n_samples))
        if rng.uniform() < 0.15:
            ln = int(rng.integers(2, 250))                                 # long enough to span two blocks of an element
            rows.append([c, p, p + ln, "ACG", "A", smp, "G1", "INDEL", "DEL", "."])
        else:
            rows.append([c, p, p + 1, "A", "CGT"[int(rng.integers(0, 3))], smp, "G1", "Noncoding", "A>T", "CAG"])
    for i in rng.integers(0, len(rows), max(1, n // 15)):                  # same mutation, second annotation
        r = list(rows[int(i)]); r[6] = "G2"; rows.append(r)
    rows.append([chrom_extra, 10, 11, "A", "T", "S01", ".", "Noncoding", "A>T", "CAG"])
    return rows


def test_overlap_kernel_against_oracle_pairs():
    """dig_overlap_join_count / dig_overlap_join_fill through the C ABI against oracle.interval_join_pairs: every
    (mutation, block) pair once, mutation-major; nested and overlapping blocks, zero-length blocks and zero-length
    mutations, mutations on a chromosome without blocks."""
    import torch
    from digdriver_amd import _lib
    from digdriver_amd.data_tools import tabulate_gpu as tg
    _lib.require_device()
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(0)
    nb, nm = 700, 30000
    bc = rng.integers(1, 6, nb); bs = rng.integers(0, 200000, nb); be = bs + rng.integers(0, 3000, nb)   # incl. zero-length
    bs[-100:] = bs[:100] + rng.integers(0, 50, 100); bc[-100:] = bc[:100]                                # nested in others
    be[-100:] = np.minimum(be[:100], bs[-100:] + rng.integers(0, 60, 100)); be = np.maximum(be, bs)
    mc = rng.integers(1, 7, nm); ms = rng.integers(0, 205000, nm); me = ms + rng.integers(0, 4, nm)
    ms[:50] = bs[:50]; me[:50] = bs[:50] + 1; mc[:50] = bc[:50]                                          # first base of a block
    ms[50:100] = be[:50]; me[50:100] = be[:50] + 1; mc[50:100] = bc[:50]                                 # first base after it
    blocks = tg.ElementBlocks(bc, bs, be, np.arange(nb), nb, dev)
    t = lambda a: torch.as_tensor(a.astype(np.int64), device=dev)
    pm, pb = tg.overlap_pairs(blocks, t(mc), t(ms), t(me))
    order = np.lexsort((bs, bc))                                                     # ElementBlocks' block order
    got = sorted(zip(pm.cpu().numpy().tolist(), order[pb.cpu().numpy()].tolist()))
    mi, bi = O.interval_join_pairs(mc, ms, me, bc, bs, be)
    want = sorted(zip(mi.tolist(), bi.tolist()))
    assert len(want) > 20000 and got == want
    assert torch.all(pm[1:] >= pm[:-1])                                              # mutation-major order


@pytest.mark.parametrize("caps", [(1e9, 3e9), (40, 2), (15, 1)])
@pytest.mark.parametrize("dedup", [True, False])
def test_tabulate_cohorts_against_oracle(tmp_path, caps, dedup):
    """tabulate_cohorts (HIP join + device tabulation, three cohorts at once) against oracle.tabulate_elements per
    cohort: nested blocks, zero-length blocks, indels across two blocks of one element, doubly annotated mutations,
    the same rows present in two cohorts (they count in each), text chromosome labels ('chr4' != '4', 'X' joined)."""
    import torch
    from digdriver_amd.data_tools import tabulate_gpu as tg
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(3)
    E = 240
    bed = _elements(rng, E, zero_len_every=17)
    blocks_rows = O.bed12_blocks(bed)
    assert any(b[1] == b[2] for b in blocks_rows)
    f_bed = tmp_path / "e.bed"
    f_bed.write_text("".join("\t".join(r) + "\n" for r in bed))
    names = [r[3] for r in bed]
    blocks, names = tg.ElementBlocks.from_bed12(str(f_bed), dev, names=names)
    cohorts, rows_of = [], []
    for c in range(3):
        rows = _mutations(rng, 4000 + 700 * c, blocks_rows, 25 + 5 * c, chrom_extra=("4" if c else "9"))
        if c == 2:
            rows += rows_of[0][:500]                                                  # shared with cohort 0
        rows_of.append(rows)
        cohorts.append(tg.encode_mutations(pd.DataFrame(rows, columns=_COLS), dev, cohort_id=c, chrom_ids=blocks.chrom_ids))
    snv, smp, ind, blacklists = tg.tabulate_cohorts(blocks, cohorts, drop_duplicates=dedup, max_muts_per_sample=caps[0],
                                                    max_muts_per_elt_per_sample=caps[1])
    assert snv.shape == (E, 3) and snv.dtype == torch.int32
    hit_black = False
    for c in range(3):
        _, per_elt, black = O.tabulate_elements(rows_of[c], blocks_rows, drop_duplicates=dedup, max_muts_per_sample=caps[0],
                                                max_muts_per_elt_per_sample=caps[1])
        want = np.array([per_elt.get(n, (0, 0, 0)) for n in names], np.int64)
        assert np.array_equal(smp[:, c].cpu().numpy(), want[:, 0]), c
        assert np.array_equal(snv[:, c].cpu().numpy(), want[:, 1]), c
        assert np.array_equal(ind[:, c].cpu().numpy(), want[:, 2]), c
        assert sorted(blacklists[c]) == black
        hit_black |= bool(black)
    assert int(snv.sum()) > 0 and int(ind.sum()) > 0
    assert hit_black == (caps[0] < 1e9)


def test_tabulate_cohorts_against_reference_frames(tmp_path):
    """The same device route on the files of tests/golden/tabulate_golden.json.gz against the frames the reference's
    tabulate_mutations_in_element returned for them (all_elements=True rows, every cap / duplicate setting)."""
    import torch
    from digdriver_amd.data_tools import tabulate_gpu as tg
    dev = torch.device("cuda:0")
    with gzip.open(os.path.join(GOLDEN, "tabulate_golden.json.gz"), "rt") as f:
        g = json.load(f)
    f_bed = tmp_path / "e.bed"
    f_bed.write_text("".join("\t".join(r) + "\n" for r in g["bed_rows"]))
    blocks, names = tg.ElementBlocks.from_bed12(str(f_bed), dev)
    df = pd.DataFrame([[r[0], int(r[1]), int(r[2])] + r[3:] for r in g["mut_rows"]], columns=_COLS)
    coh = [tg.encode_mutations(df, dev, cohort_id=0, chrom_ids=blocks.chrom_ids)]
    n = 0
    for case in g["cases"]:
        if not case["all_elements"]:
            continue
        snv, smp, ind, black = tg.tabulate_cohorts(blocks, coh, drop_duplicates=case["drop_duplicates"],
                                                   max_muts_per_sample=case["max_muts_per_sample"],
                                                   max_muts_per_elt_per_sample=case["max_muts_per_elt_per_sample"])
        assert names == case["index"]
        assert snv[:, 0].cpu().tolist() == case["OBS_SNV"]
        assert ind[:, 0].cpu().tolist() == case["OBS_INDEL"]
        assert smp[:, 0].cpu().tolist() == case["OBS_SAMPLES"]
        assert sorted(black[0]) == case["blacklist"]
        n += 1
    assert n == 6
