"""Packed reference genome for the context-counting kernel (dig_count_contexts).

The reference reads sequence through pysam.FastaFile per region (sequence_tools.py:21-29, onthefly_tools.py:120).
Here the FASTA is parsed once into the 4-bit layout of include/dig_hip.h (A=0 C=1 G=2 T=3, anything else 4; eight
bases per 32-bit word, base 0 in the low nibble; chromosomes word-aligned; one all-N pad word at either end), cached
as an .npz next to the FASTA if the directory is writable, and uploaded to HBM once (hg19: 1.55 GB).
"""
import gzip
import os

import numpy as np

_CODE = np.full(256, 4, np.uint8)
for _i, _ch in enumerate("ACGT"):
    _CODE[ord(_ch)] = _i
    _CODE[ord(_ch.lower())] = _i


class PackedGenome:
    def __init__(self, names, offsets, lengths, words):
        self.names = list(names)
        self.index = {n: i for i, n in enumerate(self.names)}
        self.offsets = np.asarray(offsets, np.int64)     # in bases, counted from word 1, multiples of 8
        self.lengths = np.asarray(lengths, np.int64)
        self.words = np.ascontiguousarray(words, np.uint32)
        self._dev = {}

    # ---- construction ----------------------------------------------------------------------
    @classmethod
    def from_sequences(cls, seqs):
        """seqs: dict name -> str/bytes (any case)."""
        names, offsets, lengths, total = [], [], [], 0
        for n, s in seqs.items():
            names.append(n)
            offsets.append(total)
            lengths.append(len(s))
            total += (len(s) + 7) // 8 * 8
        nib = np.full(total + 16, 4, np.uint8)            # + one pad word (8 nibbles) at either end
        for n, o in zip(names, offsets):
            s = seqs[n]
            b = np.frombuffer(s.encode() if isinstance(s, str) else bytes(s), np.uint8)
            nib[8 + o: 8 + o + len(b)] = _CODE[b]
        n8 = nib.reshape(-1, 8).astype(np.uint32)
        words = np.zeros(n8.shape[0], np.uint32)
        for k in range(8):
            words |= n8[:, k] << np.uint32(4 * k)
        return cls(names, offsets, lengths, words)

    @classmethod
    def from_fasta(cls, f_fasta, cache=True):
        f_cache = f_fasta + ".dig4.npz"
        if cache and os.path.exists(f_cache) and os.path.getmtime(f_cache) >= os.path.getmtime(f_fasta):
            d = np.load(f_cache, allow_pickle=False)
            g = cls([str(n) for n in d["names"]], d["offsets"], d["lengths"], d["words"])
            g._cache2 = f_fasta + ".dig2.npz"
            return g
        seqs, name, parts = {}, None, []
        opener = gzip.open if f_fasta.endswith(".gz") else open
        with opener(f_fasta, "rb") as f:
            for line in f:
                if line.startswith(b">"):
                    if name is not None:
                        seqs[name] = b"".join(parts)
                    name, parts = line[1:].split()[0].decode(), []
                else:
                    parts.append(line.strip())
        if name is not None:
            seqs[name] = b"".join(parts)
        g = cls.from_sequences(seqs)
        if cache:
            try:
                np.savez(f_cache, names=np.array(g.names), offsets=g.offsets, lengths=g.lengths, words=g.words)
                g._cache2 = f_fasta + ".dig2.npz"
                if os.path.exists(g._cache2):
                    os.remove(g._cache2)                  # (a 2-bit cache of an older FASTA)
            except OSError:
                pass
        return g

    # ---- lookup ------------------------------------------------------------------------------
    def chrom_index(self, chroms):
        """Chromosome labels -> indices; accepts '1' or 'chr1' whichever the FASTA uses."""
        out = np.empty(len(chroms), np.int32)
        for i, c in enumerate(chroms):
            c = str(c)
            if c in self.index:
                out[i] = self.index[c]
            elif "chr" + c in self.index:
                out[i] = self.index["chr" + c]
            elif c.startswith("chr") and c[3:] in self.index:
                out[i] = self.index[c[3:]]
            else:
                raise KeyError("chromosome %r is not in the genome" % c)
        return out

    def slab(self, chroms, starts, ends, margin_words=1):
        """The part of the genome a set of regions needs -- one rank's bin range of the per-base route (the reference
        fetches every bin's sequence from the FASTA on its own, sequence_tools.py:21-29; regions are independent).
        Per chromosome the regions touch: the words from the one holding min(start) - 1 to the one holding max(end) + 1, plus
        `margin_words` on either side (clipped to the chromosome).  Returns (sub, shift): `sub` is a PackedGenome with the
        same chromosome names (untouched chromosomes have length 0) in which a true position p of chromosome c sits at
        p - shift[c]; a region shifted that way gets the same positions, contexts and chromosome-end clipping from the
        kernels as the true region on the whole genome (shift is 0 or leaves >= 8 bases in front of the first region; the
        slab ends >= 8 bases behind the last region or at the true chromosome end)."""
        ci = self.chrom_index(chroms)
        st, en = np.asarray(starts, np.int64).ravel(), np.asarray(ends, np.int64).ravel()
        n_chrom = len(self.names)
        shift, new_len = np.zeros(n_chrom, np.int64), np.zeros(n_chrom, np.int64)
        pieces, offsets, total = [np.full(1, 0x44444444, np.uint32)], np.zeros(n_chrom, np.int64), 0
        for c in range(n_chrom):
            offsets[c] = total
            sel = ci == c
            if not sel.any():
                continue
            length = int(self.lengths[c])
            lo = max(0, (int(st[sel].min()) - 1) // 8 * 8 - 8 * margin_words)
            hi = min(length, (int(en[sel].max()) + 2 + 7) // 8 * 8 + 8 * margin_words)
            if hi <= lo:
                continue
            w0 = 1 + (int(self.offsets[c]) + lo) // 8
            w1 = 1 + (int(self.offsets[c]) + hi + 7) // 8
            pieces.append(self.words[w0:w1])
            shift[c], new_len[c] = lo, hi - lo
            total += (w1 - w0) * 8
        pieces.append(np.full(1, 0x44444444, np.uint32))
        return PackedGenome(self.names, offsets, new_len, np.concatenate(pieces)), shift

    # ---- the 2-bit form (dig_count_contexts2, include/dig_hip.h) ------------------------------------------
    PAD2_BASES = 64            # bases in front of the chromosome data of the 2-bit array
    BUCKET_SHIFT = 12

    def two_bit(self, cache_path=None):
        """(words2, nint_start, nint_end, nint_bucket): the genome at 2 bits per base (every letter other than ACGT stored as
        A), the maximal runs of such letters as sorted intervals of ARRAY bases (array base = 64 + offset + position; the
        alignment padding between chromosomes counts as such letters) and the bucket index of the interval list.
        Derived from the 4-bit words once per genome: byte-wise through 256-entry tables (a byte of the 4-bit form is two
        bases: their 2-bit codes and their two "other letter" flags), 64 M bases at a time -- about 1 s per 100 Mb -- and kept
        next to the 4-bit cache as <fasta>.dig2.npz (from_fasta passes the path), so a process pays for it once per FASTA."""
        if getattr(self, "_two_bit", None) is not None:
            return self._two_bit
        cache_path = cache_path or getattr(self, "_cache2", None)
        if cache_path and os.path.exists(cache_path):
            try:
                d = np.load(cache_path, allow_pickle=False)
                if int(d["n_words4"]) == int(self.words.size):
                    self._two_bit = (d["words2"], d["nint_start"], d["nint_end"], d["nint_bucket"])
                    return self._two_bit
            except (OSError, KeyError, ValueError):
                pass
        body = self.words[1:-1]                                   # without the two pad words
        total = int(body.size) * 8
        words2 = np.zeros(4 + (total + 15) // 16 + 24, np.uint32)
        b = np.arange(256, dtype=np.uint32)
        lo, hi = b & 15, b >> 4
        lut_code = (np.where(lo > 3, 0, lo & 3) | (np.where(hi > 3, 0, hi & 3) << 2)).astype(np.uint16)    # 4 bits: two bases
        lut_lo, lut_hi = (lo > 3).astype(np.int8), (hi > 3).astype(np.int8)
        by = body.view(np.uint8)                                  # little endian: byte k of a word holds bases 2 k, 2 k + 1
        starts, ends = [], []
        chunk = 1 << 25                                           # bytes per pass (a multiple of 8: whole 2-bit words)
        prev = np.int8(0)                                         # flag of the base in front of the chunk
        for b0 in range(0, by.size, chunk):
            seg = by[b0:b0 + chunk]
            c = lut_code[seg]
            if len(c) % 8:
                c = np.concatenate([c, np.zeros(8 - len(c) % 8, np.uint16)])
            c = c.reshape(-1, 4)
            w16 = (c[:, 0] | (c[:, 1] << 4) | (c[:, 2] << 8) | (c[:, 3] << 12)).astype(np.uint32)          # one 4-bit word = 8 bases
            packed = w16[0::2] | (w16[1::2] << 16)
            words2[4 + b0 // 8: 4 + b0 // 8 + len(packed)] = packed
            fl, fh = lut_lo[seg], lut_hi[seg]                     # flags of bases 2 i and 2 i + 1 of the chunk
            before = np.empty_like(fl)                            # flag of the base in front of base 2 i
            before[0] = prev
            before[1:] = fh[:-1]
            base0 = 2 * b0
            d_lo, d_hi = fl - before, fh - fl                     # +1: a run starts at that base, -1: it ended in front of it
            s_lo, s_hi = np.flatnonzero(d_lo == 1), np.flatnonzero(d_hi == 1)
            e_lo, e_hi = np.flatnonzero(d_lo == -1), np.flatnonzero(d_hi == -1)
            starts.append(np.sort(np.concatenate([base0 + 2 * s_lo, base0 + 2 * s_hi + 1]).astype(np.int64)))
            ends.append(np.sort(np.concatenate([base0 + 2 * e_lo, base0 + 2 * e_hi + 1]).astype(np.int64)))
            prev = fh[-1]
        ns = np.concatenate(starts) if starts else np.zeros(0, np.int64)
        ne = np.concatenate(ends) if ends else np.zeros(0, np.int64)
        if prev:                                                  # the genome ends inside a run
            ne = np.concatenate([ne, [total]])
        ns, ne = ns + self.PAD2_BASES, ne + self.PAD2_BASES
        n_buckets = ((total + self.PAD2_BASES) >> self.BUCKET_SHIFT) + 2
        bucket = np.searchsorted(ne, np.arange(n_buckets, dtype=np.int64) << self.BUCKET_SHIFT, side="right").astype(np.int32)
        self._two_bit = (words2, ns, ne, bucket)
        if cache_path:
            try:
                np.savez(cache_path, words2=words2, nint_start=ns, nint_end=ne, nint_bucket=bucket, n_words4=np.int64(self.words.size))
            except OSError:
                pass
        return self._two_bit

    def on_device2(self, device):
        """Device tensors of the 2-bit form: (words2, nint_start, nint_end, nint_bucket, offsets, lengths)."""
        import torch
        dev = torch.device(device)
        key = ("2bit", dev.type, dev.index)
        if key not in self._dev:
            w2, ns, ne, bk = self.two_bit()
            self._dev[key] = (torch.as_tensor(w2.view(np.int32), device=dev), torch.as_tensor(ns, device=dev), torch.as_tensor(ne, device=dev),
                              torch.as_tensor(bk, device=dev), torch.as_tensor(self.offsets, device=dev), torch.as_tensor(self.lengths, device=dev))
        return self._dev[key]

    def on_device(self, device):
        import torch
        dev = torch.device(device)
        key = (dev.type, dev.index)
        if key not in self._dev:
            self._dev[key] = (torch.as_tensor(self.words.view(np.int32), device=dev),
                              torch.as_tensor(self.offsets, device=dev), torch.as_tensor(self.lengths, device=dev))
        return self._dev[key]
