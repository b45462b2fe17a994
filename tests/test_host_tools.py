"""CPU-only tests of the host-side mirror: mutation-file reader, integer tabulation (the interval join that
replaces `bedtools intersect`), the map-file container.  Integer outputs must be bit-exact against the
goldens produced by the reference (tests/golden/mutation_tools_golden.json)."""
import json
import os

import numpy as np
import pandas as pd
import pytest

from conftest import GOLDEN
from digdriver_amd.data_tools import mutation_tools as mt
from digdriver_amd.io import mapfile


def test_read_mutation_file_and_counts_match_reference():
    g = json.load(open(os.path.join(GOLDEN, "mutation_tools_golden.json")))
    path = os.path.join(GOLDEN, "mutations_small.tsv")
    rd = mt.read_mutation_file(path, drop_sex=True, drop_duplicates=True, unique_indels=True)
    rd2 = mt.read_mutation_file(path, drop_sex=True, drop_duplicates=False, unique_indels=True)
    assert len(rd) == g["n_read_dedup"] and len(rd2) == g["n_read_nodedup"]
    assert rd.astype(str).values.tolist()[:50] == g["read_dedup_rows"]
    chk = int(pd.util.hash_pandas_object(rd.reset_index(drop=True).astype(str), index=False).sum() % (2 ** 61))
    assert chk == g["read_dedup_checksum"]
    filt, black = mt.filter_hypermut_samples(rd2, 95, return_blacklist=True)
    assert sorted(black) == g["blacklist"] and len(filt) == g["n_filtered"]
    cnt = mt.mutations_per_gene(rd2[rd2.GENE != '.'], max_muts_per_gene_per_sample=2)
    assert list(cnt.index) == g["cnt_index"]
    assert sorted(cnt.columns) == sorted(g["cnt_cols"])
    assert np.array_equal(cnt[g["cnt_cols"]].values, np.array(g["cnt_vals"]))


def test_bed12_boundaries_matches_reference():
    g = json.load(open(os.path.join(GOLDEN, "mutation_tools_golden.json")))["bed12"]
    bb = mt.bed12_boundaries(os.path.join(GOLDEN, "elements_small.bed"))
    assert [int(c) for c in bb.CHROM] == g["CHROM"] and list(bb.ELT) == g["ELT"] and list(bb.STRAND) == g["STRAND"]
    assert [list(map(int, x)) for x in bb.BLOCK_STARTS] == g["BLOCK_STARTS"]
    assert [list(map(int, x)) for x in bb.BLOCK_ENDS] == g["BLOCK_ENDS"]


def _brute_pairs(mc, ms, me, bc, bs, be):
    out = []
    for i in range(len(mc)):
        for j in range(len(bc)):
            if str(mc[i]) == str(bc[j]) and ms[i] < be[j] and bs[j] < me[i]:
                out.append((i, j))
    return out


def test_interval_join_against_brute_force():
    rng = np.random.default_rng(1)
    for trial in range(5):
        nb, nm = 60, 400
        bc = rng.integers(1, 4, nb).astype(str)
        bs = rng.integers(0, 5000, nb)
        be = bs + rng.integers(1, 800, nb)          # overlapping and nested blocks included
        mc = rng.integers(1, 5, nm).astype(str)
        ms = rng.integers(0, 6000, nm)
        me = ms + rng.integers(1, 4, nm)
        mi, bi = mt._overlap_pairs(mc, ms, me, bc, bs, be)
        assert sorted(zip(mi.tolist(), bi.tolist())) == sorted(_brute_pairs(mc, ms, me, bc, bs, be))
    # boundaries: half-open on both sides
    mi, bi = mt._overlap_pairs(["1"] * 4, [99, 100, 199, 200], [100, 101, 200, 201], ["1"], [100], [200])
    assert mi.tolist() == [1, 2]


def test_tabulation_integer_semantics(tmp_path):
    # two elements; E1 has two blocks; S1 hits E1 three times (one a duplicated annotation row), S2 once + an indel
    bed = tmp_path / "e.bed"
    bed.write_text("1\t100\t400\tE1\t0\t+\t100\t100\t.\t2\t50,100,\t0,200,\n"
                   "2\t1000\t1100\tE2\t0\t-\t1000\t1000\t.\t1\t100,\t0,\n")
    rows = [
        ("1", 110, 111, "A", "T", "S1", "G1", "Noncoding", "A>T", "CAG"),
        ("1", 110, 111, "A", "T", "S1", "G2", "Noncoding", "A>T", "CAG"),     # same mutation, second gene annotation
        ("1", 320, 321, "C", "G", "S1", "G1", "Noncoding", "C>G", "ACT"),
        ("1", 160, 161, "C", "G", "S1", "G1", "Noncoding", "C>G", "ACT"),     # between the blocks: no hit
        ("1", 399, 400, "C", "G", "S2", "G1", "Noncoding", "C>G", "ACT"),
        ("1", 300, 305, "CAAAA", "C", "S2", "G1", "INDEL", "DEL", "."),
        ("2", 1099, 1100, "G", "A", "S3", ".", "Noncoding", "G>A", "AGT"),
        ("2", 1100, 1101, "G", "A", "S3", ".", "Noncoding", "G>A", "AGT"),     # END of block is exclusive
        ("X", 5, 6, "G", "A", "S3", ".", "Noncoding", "G>A", "AGT"),
    ]
    mut = tmp_path / "m.tsv"
    pd.DataFrame(rows).to_csv(mut, sep="\t", header=False, index=False)
    tab, black = mt.tabulate_mutations_in_element(str(mut), str(bed), bed12=True, drop_duplicates=True,
                                                  return_blacklist=True)
    assert tab.loc["E1"].tolist() == [2, 3, 1]      # OBS_SAMPLES, OBS_SNV, OBS_INDEL
    assert tab.loc["E2"].tolist() == [1, 1, 0]
    assert len(black) == 0
    # without de-duplication the doubly annotated SNV counts twice
    tab2 = mt.tabulate_mutations_in_element(str(mut), str(bed), bed12=True, drop_duplicates=False)
    assert tab2.loc["E1", "OBS_SNV"] == 4
    # hypermutator blacklist and per-element cap
    tab3, black3 = mt.tabulate_mutations_in_element(str(mut), str(bed), bed12=True, drop_duplicates=True,
                                                    max_muts_per_sample=1, max_muts_per_elt_per_sample=1,
                                                    return_blacklist=True)
    assert sorted(black3) == ["S1", "S2"] and "E1" not in tab3.index and tab3.loc["E2", "OBS_SNV"] == 1
    # all_elements adds zero rows
    tab4 = mt.tabulate_mutations_in_element(str(mut), str(bed), bed12=True, drop_duplicates=True, all_elements=True,
                                            max_muts_per_sample=1)
    assert tab4.loc["E1"].tolist() == [0, 0, 0]
    # empty intersection
    mut2 = tmp_path / "m2.tsv"
    pd.DataFrame(rows[-1:]).to_csv(mut2, sep="\t", header=False, index=False)
    tab5 = mt.tabulate_mutations_in_element(str(mut2), str(bed), bed12=True)
    assert len(tab5) == 0 and list(tab5.columns) == ['OBS_SAMPLES', 'OBS_SNV', 'OBS_INDEL']


def test_mapfile_directory_roundtrip(tmp_path):
    path = str(tmp_path / "cohort.map")
    df = pd.DataFrame({"CHROM": [1, 1, 2], "START": [0, 10000, 0], "END": [10000, 20000, 10000],
                       "Y_PRED": [1.5, 2.5, 3.5], "FLAG": [False, True, False], "NAME": ["a", "b", "c"]},
                      index=["chr1:0-10000", "chr1:10000-20000", "chr2:0-10000"])
    mapfile.write_frame(path, "region_params", df)
    mapfile.write_array(path, "idx", df[["CHROM", "START", "END"]].values.astype(np.int32))
    mapfile.write_attrs(path, cohort_name="X", N_SAMPLES=np.int64(12))
    back = mapfile.read_frame(path, "region_params")
    assert list(back.columns) == list(df.columns) and list(back.index) == list(df.index)
    assert back.FLAG.dtype == bool and np.array_equal(back.Y_PRED.values, df.Y_PRED.values)
    assert mapfile.read_array(path, "idx").dtype == np.int32
    assert mapfile.read_attrs(path) == {"cohort_name": "X", "N_SAMPLES": 12}
    assert mapfile.has_key(path, "region_params") and not mapfile.has_key(path, "nope")
    with pytest.raises(KeyError):
        mapfile.read_frame(path, "nope")


def test_tabulate_sites_matches_reference(tmp_path):
    """tabulate_sites_in_element against the reference's own output (tests/golden/sites_golden.json)."""
    import json
    from digdriver_amd.data_tools import mutation_tools as mt
    g = json.load(open(os.path.join(GOLDEN, "sites_golden.json")))
    fs, fm = tmp_path / "sites.tsv", tmp_path / "m.tsv"
    pd.DataFrame(g["sites_rows"]).to_csv(fs, sep="\t", header=False, index=False)
    pd.DataFrame(g["mut_rows"]).to_csv(fm, sep="\t", header=False, index=False)
    tab = mt.tabulate_sites_in_element(str(fs), str(fm))
    assert [str(i) for i in tab.index] == g["tab_index"]
    assert tab.OBS_SAMPLES.astype(int).tolist() == g["tab_obs_samples"]
    assert tab.OBS_SNV.astype(int).tolist() == g["tab_obs_snv"]


def test_get_q_vals_equals_statsmodels():
    """nb_model.get_q_vals (reference: nb_model.py:340-342 = statsmodels fdrcorrection, method 'indep') against values
    produced by statsmodels 0.12.2 under the image's second interpreter (ties, a zero, a one, a denormal-scale value)."""
    import json
    import os
    import numpy as np
    from conftest import GOLDEN
    from digdriver_amd.sequence_model import nb_model
    g = json.load(open(os.path.join(GOLDEN, "qvals_golden.json")))
    assert np.array_equal(nb_model.get_q_vals(np.array(g["p"])), np.array(g["q"]))
    assert nb_model.get_q_vals([]).size == 0


def test_tabulation_matches_the_reference_functions(tmp_path):
    """mutation_tools.tabulate_* of this package against the frames the reference's functions returned on the same
    files (tests/golden/tabulate_golden.json.gz: only pybedtools' intersect was stood in for): row order, column order,
    index, counts and blacklist."""
    import gzip
    with gzip.open(os.path.join(GOLDEN, "tabulate_golden.json.gz"), "rt") as f:
        g = json.load(f)
    f_mut, f_bed = tmp_path / "m.tsv", tmp_path / "e.bed"
    f_mut.write_text("".join("\t".join(r) + "\n" for r in g["mut_rows"]))
    f_bed.write_text("".join("\t".join(r) + "\n" for r in g["bed_rows"]))
    for dd in (False, True):
        cnt = mt.tabulate_muts_per_sample_per_element(str(f_mut), str(f_bed), bed12=True, drop_duplicates=dd)
        want = g["per_pair_dedup" if dd else "per_pair"]
        assert list(cnt.columns) == ['ELT', 'SAMPLE', 'OBS_SNV', 'OBS_INDEL', 'OBS_MUT']
        for col in cnt.columns:
            assert cnt[col].tolist() == want[col], (dd, col)
    for case in g["cases"]:
        tab, black = mt.tabulate_mutations_in_element(
            str(f_mut), str(f_bed), bed12=True, drop_duplicates=case["drop_duplicates"], all_elements=case["all_elements"],
            max_muts_per_sample=case["max_muts_per_sample"], max_muts_per_elt_per_sample=case["max_muts_per_elt_per_sample"],
            return_blacklist=True)
        assert sorted(str(b) for b in black) == case["blacklist"]
        assert [str(i) for i in tab.index] == case["index"] and list(tab.columns) == case["columns"]
        for col in ("OBS_SAMPLES", "OBS_SNV", "OBS_INDEL"):
            assert tab[col].astype(int).tolist() == case[col], (case, col)


def test_native_mutation_file_parser_gives_the_arrays_of_the_python_routes(tmp_path):
    """dig_mutation_file_*_host (include/dig_hip.h, ABI 7) against the two Python routes of tabulate_gpu.encode_mutation_file /
    encode_mutations_host on the reference-format golden file and on a file with the corner cases of the format: 'chr' prefixes,
    contigs that are dropped, CRLF line ends, empty lines, repeated and out-of-order rows, extra columns.  Content the parser
    does not cover is handed back to the Python route."""
    from digdriver_amd.data_tools import tabulate_gpu as tg

    def same(a, b):
        assert a["sample_names"] == b["sample_names"]
        for k in b:
            if k != "sample_names":
                assert a[k].dtype == b[k].dtype and np.array_equal(a[k], b[k]), k

    small = os.path.join(GOLDEN, "mutations_small.tsv")
    nat = tg._encode_mutation_file_native(small, 5)
    assert nat is not None and len(nat["chrom"]) > 1000
    same(nat, tg.encode_mutation_file(small, 5, native=False))
    df = pd.read_csv(small, sep="\t", header=None, low_memory=False, dtype={0: str}).iloc[:, :8]
    df.columns = ['CHROM', 'START', 'END', 'REF', 'ALT', 'SAMPLE', 'GENE', 'ANNOT']
    same(nat, tg.encode_mutations_host(df, 5))

    rng = np.random.default_rng(11)
    rows = []
    for i in range(4000):
        c = rng.choice(["1", "chr2", "22", "chr22", "X", "chrY", "MT", "23", "chr0", "GL000191.1", "7"])
        s = int(rng.integers(0, 5000))
        ln = int(rng.integers(1, 4))
        ref, alt = rng.choice(["A", "C", "G", "T", "AT", "-"]), rng.choice(["A", "C", "G", "T", "-", "GG"])
        rows.append("\t".join([c, str(s), str(s + ln), ref, alt, "S%d" % rng.integers(0, 40), rng.choice([".", "TP53", "KRAS"]),
                               rng.choice(["SNV", "INDEL", "Missense", "Noncoding"]), "x", "ACG"]))
    rows += rows[:300]                                      # exact duplicates
    text = "\r\n".join(rows[:2000]) + "\r\n\r\n" + "\n".join(rows[2000:]) + "\n"
    f = tmp_path / "corner.annot.txt"
    f.write_bytes(text.encode())
    nat = tg._encode_mutation_file_native(str(f), 0)
    assert nat is not None
    same(nat, tg.encode_mutation_file(str(f), 0, native=False))
    assert set(np.unique(nat["chrom"])) <= {1, 2, 7, 22} and nat["indel"].sum() > 0
    # not covered: left to the Python route
    for bad in ('1\t5\t6\tA\tC\t"S1"\t.\tSNV\n', '1\t5\t6\tA\tC\tS1\t.\tSNV\n2\t7\t8\tA\tC\tS1\t.\n', '1\t5.0\t6\tA\tC\tS1\t.\tSNV\n'):
        g = tmp_path / "bad.txt"
        g.write_text(bad)
        assert tg._encode_mutation_file_native(str(g), 0) is None
    nothing = tmp_path / "nothing.txt"
    nothing.write_text("")
    assert tg._encode_mutation_file_native(str(nothing), 0) is None       # an empty file: the Python route raises as before
    with pytest.raises(ValueError):
        tg.encode_mutation_file(str(nothing), 0)
    empty = tmp_path / "empty.txt"
    empty.write_text("X\t1\t2\tA\tC\tS1\t.\tSNV\n")
    nat = tg._encode_mutation_file_native(str(empty), 0)
    assert nat is not None and len(nat["chrom"]) == 0 and nat["sample_names"] == []
    # labels pandas reads as missing, other column counts, a header line and the 'chr' rule go the same way in all routes
    na = tmp_path / "na.txt"
    na.write_text("1\t5\t6\tA\tC\tS1\t.\tSNV\n1\t7\t8\tA\tC\tNA\t.\tSNV\n2\t9\t10\tA\tG\tS2\t\tINDEL\n")
    assert tg._encode_mutation_file_native(str(na), 0) is None                 # 'NA' and '' in label columns
    via_file = tg.encode_mutation_file(str(na), 0)
    frame = pd.read_csv(str(na), sep="\t", names=['CHROM', 'START', 'END', 'REF', 'ALT', 'SAMPLE', 'GENE', 'ANNOT'], dtype={'CHROM': str})
    same(via_file, tg.encode_mutations_host(frame, 0))
    assert via_file["sample_names"] == ["S1", "S2"] and len(via_file["chrom"]) == 2     # the row without a SAMPLE is dropped (the reference's groupby)
    nine = tmp_path / "nine.txt"                                                # 9 columns: ANNOT is column 6, there is no GENE (mutation_tools.py:68-70)
    nine.write_text("1\t5\t6\tA\tC\tS1\tINDEL\tC>A\tACG\n3\t7\t8\tA\tC\tS2\tSNV\tC>A\tACG\n")
    assert tg._encode_mutation_file_native(str(nine), 0) is None
    enc9 = tg.encode_mutation_file(str(nine), 0)
    assert enc9["indel"].tolist() == [1, 0] and enc9["gene"].tolist() == [0, 0]
    head = tmp_path / "header.txt"
    head.write_text("CHROM\tSTART\tEND\tREF\tALT\tSAMPLE\tGENE\tANNOT\n1\t5\t6\tA\tC\tS1\t.\tSNV\n")
    assert tg._encode_mutation_file_native(str(head), 0) is None
    with pytest.raises(ValueError):                                             # what pandas says about START = 'START' under dtype int
        tg.encode_mutation_file(str(head), 0)
    chrs = tmp_path / "chr.txt"
    chrs.write_text("".join("%s\t5\t6\tA\tC\tS1\t.\tSNV\n" % c for c in ("chr1", "chrchr1", "1chr", "2", "chr22", "chr23")))
    want = [1, 2, 22]
    assert tg._encode_mutation_file_native(str(chrs), 0)["chrom"].tolist() == want
    assert tg.encode_mutation_file(str(chrs), 0, native=False)["chrom"].tolist() == want
    frame = pd.read_csv(str(chrs), sep="\t", names=['CHROM', 'START', 'END', 'REF', 'ALT', 'SAMPLE', 'GENE', 'ANNOT'], dtype={'CHROM': str})
    assert tg.encode_mutations_host(frame, 0)["chrom"].tolist() == want


def test_region_tables_take_any_row_order_and_refuse_another_grid():
    """RegionTables (the [N, C] tables of C maps on one (CHROM, START)-ordered grid): a map stored in another row order, or in the
    same shuffled order as the first, gives the tables of the sorted maps; a map on another grid is refused."""
    from digdriver_amd.sequence_model import genic_driver_tools as g
    rng = np.random.default_rng(4)
    N = 3000
    chrom = np.repeat(np.arange(1, 4), N // 3)
    start = np.tile(np.arange(N // 3) * 10_000, 3)

    def frame(seed):
        r = np.random.default_rng(seed)
        return pd.DataFrame({"CHROM": chrom, "START": start, "END": start + 10_000, "Y_TRUE": r.integers(0, 50, N),
                             "Y_PRED": r.gamma(5, 2, N), "STD": r.gamma(2, 1, N), "FLAG": r.random(N) < 0.1})
    frames = [frame(s) for s in range(5)]
    want = g.RegionTables(frames)
    assert want.mu.shape == (N, 5) and want.mu.flags.c_contiguous and want.y.dtype == np.int32 and want.flag.dtype == np.uint8
    for c, f in enumerate(frames):
        assert np.array_equal(want.mu[:, c], f.Y_PRED.values) and np.array_equal(want.y[:, c], f.Y_TRUE.values)
        assert np.array_equal(want.flag[:, c], f.FLAG.values.astype(np.uint8)) and np.array_equal(want.std[:, c], f.STD.values)
    perm = rng.permutation(N)
    same_shuffle = g.RegionTables([f.iloc[perm].reset_index(drop=True) for f in frames])
    mixed = g.RegionTables([frames[0]] + [f.iloc[rng.permutation(N)].reset_index(drop=True) for f in frames[1:]])
    first_shuffled = g.RegionTables([frames[0].iloc[perm].reset_index(drop=True)] + frames[1:])
    for got in (same_shuffle, mixed, first_shuffled):
        for name in ("chrom", "start", "mu", "std", "y", "flag"):
            assert np.array_equal(getattr(got, name), getattr(want, name)), name
        assert got.window == want.window == 10_000
    other = frames[1].copy()
    other.loc[7, "START"] += 5
    with pytest.raises(ValueError):
        g.RegionTables([frames[0], other])


def test_two_bit_genome_against_a_brute_force_walk_and_its_disk_cache(tmp_path):
    """PackedGenome.two_bit (the form dig_count_contexts2 reads): 2-bit codes, runs of letters other than ACGT and
    their bucket index against a per-base walk -- runs at the chromosome ends, across the alignment padding, of length 1, letters
    in lower case -- and the cache next to the FASTA: the second process-worth of calls reads <fasta>.dig2.npz instead of
    converting again, and a rewritten FASTA invalidates it."""
    from digdriver_amd.data_tools.genome import PackedGenome
    rng = np.random.default_rng(11)
    seqs = {}
    for name, L in (("1", 40_013), ("2", 9_999), ("X", 517)):
        s = np.frombuffer(b"ACGTacgt", np.uint8)[rng.integers(0, 8, L)].copy()
        for _ in range(25):
            p = int(rng.integers(0, L))
            s[p:min(L, p + int(rng.integers(1, 300)))] = ord("N")
        s[:5] = ord("N")
        s[-1:] = ord("R")
        s[100] = ord("n")
        seqs[name] = s.tobytes()
    g = PackedGenome.from_sequences(seqs)
    w2, ns, ne, bucket = g.two_bit()
    total = (g.words.size - 2) * 8
    flat = np.ones(total, bool)                                      # "other letter" per array base (padding counts as such)
    code = np.zeros(total, np.uint32)
    lut = {ord(c): i for i, c in enumerate("ACGT")}
    lut.update({ord(c): i for i, c in enumerate("acgt")})
    for name, off in zip(g.names, g.offsets):
        b = np.frombuffer(seqs[name], np.uint8)
        known = np.isin(b, list(lut))
        flat[off:off + len(b)] = ~known
        code[off:off + len(b)] = np.where(known, np.vectorize(lambda x: lut.get(x, 0))(b), 0)
    edge = np.diff(np.concatenate([[0], flat.astype(np.int8), [0]]))
    assert np.array_equal(ns, np.flatnonzero(edge == 1) + g.PAD2_BASES) and np.array_equal(ne, np.flatnonzero(edge == -1) + g.PAD2_BASES)
    got = (w2[4:4 + (total + 15) // 16, None] >> (2 * np.arange(16, dtype=np.uint32))[None, :]) & 3
    assert np.array_equal(got.reshape(-1)[:total], code)
    want_bucket = np.searchsorted(ne, np.arange(len(bucket), dtype=np.int64) << g.BUCKET_SHIFT, side="right")
    assert np.array_equal(bucket, want_bucket)
    # the disk cache
    fa = tmp_path / "g.fa"
    fa.write_bytes(b"".join(b">" + n.encode() + b"\n" + s + b"\n" for n, s in seqs.items()))
    g1 = PackedGenome.from_fasta(str(fa))
    r1 = g1.two_bit()
    assert os.path.exists(str(fa) + ".dig2.npz")
    g2 = PackedGenome.from_fasta(str(fa))
    calls = []
    orig = np.savez
    np.savez = lambda *a, **k: calls.append(a[0])                    # a second conversion would write the cache again
    try:
        r2 = g2.two_bit()
    finally:
        np.savez = orig
    assert not calls and all(np.array_equal(a, b) for a, b in zip(r1, r2)) and all(np.array_equal(a, b) for a, b in zip(r1, (w2, ns, ne, bucket)))
    import time
    time.sleep(0.05)
    fa.write_bytes(b">1\nACGTNNACGT\n")
    os.utime(str(fa), None)
    g3 = PackedGenome.from_fasta(str(fa))
    assert len(g3.two_bit()[1]) == 2 and g3.lengths.tolist() == [10]  # the run of two N and the alignment padding behind the chromosome
