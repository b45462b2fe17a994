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
            return cls([str(n) for n in d["names"]], d["offsets"], d["lengths"], d["words"])
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

    def two_bit(self):
        """(words2, nint_start, nint_end, nint_bucket): the genome at 2 bits per base (every letter other than ACGT stored as
        A), the maximal runs of such letters as sorted intervals of ARRAY bases (array base = 64 + offset + position; the
        alignment padding between chromosomes counts as such letters) and the bucket index of the interval list.
        Derived from the 4-bit words once, 32 M bases at a time."""
        if getattr(self, "_two_bit", None) is not None:
            return self._two_bit
        body = self.words[1:-1]                                   # without the two pad words
        total = int(body.size) * 8
        words2 = np.zeros(4 + (total + 15) // 16 + 24, np.uint32)
        shifts4 = (4 * np.arange(8, dtype=np.uint32))[None, :]
        shifts2 = (2 * np.arange(16, dtype=np.uint32))[None, :]
        starts, ends = [], []
        chunk = 1 << 22                                           # 4-bit words per pass (an even number: whole 2-bit words)
        for w0 in range(0, body.size, chunk):
            nib = ((body[w0:w0 + chunk, None] >> shifts4) & np.uint32(15)).astype(np.uint8).reshape(-1)
            base0 = w0 * 8                                        # offset + position of nib[0]
            isn = nib > 3
            code = np.where(isn, 0, nib & 3).astype(np.uint32)
            if len(code) % 16:
                code = np.concatenate([code, np.zeros(16 - len(code) % 16, np.uint32)])
            packed = (code.reshape(-1, 16) << shifts2).sum(axis=1, dtype=np.uint64).astype(np.uint32)
            words2[4 + base0 // 16: 4 + base0 // 16 + len(packed)] = packed
            edge = np.diff(np.concatenate([[0], isn.astype(np.int8), [0]]))       # runs cut at the chunk's ends are joined below
            starts.append(base0 + np.flatnonzero(edge == 1).astype(np.int64))
            ends.append(base0 + np.flatnonzero(edge == -1).astype(np.int64))
        ns = np.concatenate(starts) if starts else np.zeros(0, np.int64)
        ne = np.concatenate(ends) if ends else np.zeros(0, np.int64)
        if len(ns) > 1:                                           # a run that crosses a chunk boundary: end == next start
            glue = ne[:-1] == ns[1:]
            ns, ne = ns[np.concatenate([[True], ~glue])], ne[np.concatenate([~glue, [True]])]
        ns, ne = ns + self.PAD2_BASES, ne + self.PAD2_BASES
        n_buckets = ((total + self.PAD2_BASES) >> self.BUCKET_SHIFT) + 2
        bucket = np.searchsorted(ne, np.arange(n_buckets, dtype=np.int64) << self.BUCKET_SHIFT, side="right").astype(np.int32)
        self._two_bit = (words2, ns, ne, bucket)
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
