"""tabulate_mutations_in_element (mutation_tools.py:155-189): the integer-coded route (_summary_by_codes) against the frame route
that mirrors the reference line by line -- same table, dtypes, row order and blacklist -- on seeded files with nested and
overlapping elements, indels across two blocks, doubly annotated rows, hypermutated samples and capped pairs.  (Both routes are
pinned to the reference's own outputs by tests/test_oracle_golden.py's tabulate golden through the public function.)"""
import numpy as np
import pandas as pd
import pytest

from digdriver_amd.data_tools import mutation_tools as mt


def _files(tmp_path, seed, numeric_names=False):
    rng = np.random.default_rng(seed)
    n_elt, n_mut = 400, 6000
    rows = []
    for e in range(n_elt):
        chrom = rng.integers(1, 4)
        start = int(rng.integers(0, 200_000))
        nb = int(rng.integers(1, 4))
        sizes = rng.integers(20, 300, nb)
        gaps = rng.integers(0, 200, nb)
        starts = np.concatenate([[0], np.cumsum(sizes[:-1] + gaps[:-1])])
        name = str(1000 + e // 2) if numeric_names else "ELT%04d" % (e // 2 if e % 7 == 0 else e)      # some names on two rows
        trail = "," if e % 3 else ""
        rows.append("%d\t%d\t%d\t%s\t0\t%s\t%d\t%d\t.\t%d\t%s%s\t%s%s\n" % (
            chrom, start, start + starts[-1] + sizes[-1], name, "+-"[e % 2], start, start, nb,
            ",".join(map(str, sizes)), trail, ",".join(map(str, starts)), trail))
    f_bed = tmp_path / "e.bed"
    f_bed.write_text("".join(rows))
    chrom = rng.integers(1, 4, n_mut)
    pos = rng.integers(0, 201_000, n_mut)
    indel = rng.uniform(size=n_mut) < 0.15
    ln = np.where(indel, rng.integers(2, 400, n_mut), 1)
    samp = np.where(rng.uniform(size=n_mut) < 0.3, "HYPER", np.char.add("S", rng.integers(0, 40, n_mut).astype(str)))
    df = pd.DataFrame({0: chrom, 1: pos, 2: pos + ln, 3: np.where(indel, "ACG", "A"), 4: np.where(indel, "A", "T"), 5: samp, 6: ".",
                       7: np.where(indel, "INDEL", "Noncoding"), 8: "x", 9: "y"})
    dup = df.iloc[rng.integers(0, n_mut, 300)].copy()
    dup[6] = "G2"
    f_mut = tmp_path / "m.txt"
    pd.concat([df, dup]).to_csv(f_mut, sep="\t", header=False, index=False)
    return str(f_mut), str(f_bed)


@pytest.mark.parametrize("kw", [
    dict(drop_duplicates=True, max_muts_per_sample=1e9, max_muts_per_elt_per_sample=3e9),
    dict(drop_duplicates=False, max_muts_per_sample=200, max_muts_per_elt_per_sample=2),
    dict(drop_duplicates=True, max_muts_per_sample=200, max_muts_per_elt_per_sample=1, all_elements=True),
])
@pytest.mark.parametrize("numeric_names", [False, True])
def test_integer_coded_route_equals_the_frame_route(tmp_path, monkeypatch, kw, numeric_names):
    f_mut, f_bed = _files(tmp_path, 5, numeric_names)
    fast, bl_fast = mt.tabulate_mutations_in_element(f_mut, f_bed, bed12=True, return_blacklist=True, **kw)
    assert mt._summary_by_codes(f_mut, f_bed, True, kw["drop_duplicates"], kw["max_muts_per_sample"], kw["max_muts_per_elt_per_sample"]) is not None
    monkeypatch.setattr(mt, "_summary_by_codes", lambda *a: None)
    frame, bl_frame = mt.tabulate_mutations_in_element(f_mut, f_bed, bed12=True, return_blacklist=True, **kw)
    pd.testing.assert_frame_equal(fast, frame)
    assert list(bl_fast) == list(bl_frame)
    if kw["max_muts_per_sample"] < 1e9:
        assert "HYPER" in list(bl_fast)
    assert fast.OBS_SNV.sum() > 0 and fast.OBS_INDEL.sum() > 0


def test_nothing_to_count_and_short_files_take_the_frame_route(tmp_path):
    f_mut, f_bed = _files(tmp_path, 6)
    far = tmp_path / "far.txt"
    far.write_text("9\t5\t6\tA\tT\tS1\t.\tNoncoding\tx\ty\n")
    t = mt.tabulate_mutations_in_element(str(far), f_bed, bed12=True)
    assert len(t) == 0 and list(t.columns) == ['OBS_SAMPLES', 'OBS_SNV', 'OBS_INDEL']
    t = mt.tabulate_mutations_in_element(str(far), f_bed, bed12=True, all_elements=True)
    assert len(t) > 0 and float(t.OBS_SNV.sum()) == 0.0
