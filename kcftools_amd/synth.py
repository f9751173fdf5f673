"""Seeded synthetic inputs for the getVariations hot path (numpy; small / medium scale).

There is no `kmc` binary and no real data in the build image, so reference FASTA text,
a mutated "sample" genome and a KMC database in the on-disk layout the reference reads
(KMC.java:107-189, docs/formats/kmc.md) are generated here.
The large bench-scale databases are written by the native generator
(kcftools_amd/csrc/tools/kcfx_synth.cpp); this module is its independent small-scale twin
and the two are cross-checked in tests/.

Nothing here is on the product compute path.
"""
from __future__ import annotations

import os
import struct
from dataclasses import dataclass

import numpy as np

_CODE = np.full(256, 255, dtype=np.uint8)
for _i, _c in enumerate("ACGT"):
    _CODE[ord(_c)] = _i
    _CODE[ord(_c.lower())] = _i
_ACGT = np.frombuffer(b"ACGT", dtype=np.uint8)


# ---------------------------------------------------------------------------------------
# reference / sample text
# ---------------------------------------------------------------------------------------
def random_reference(length: int, seed: int, n_run_every: int = 100_000, max_n_run: int = 2000,
                     lower_frac: float = 0.10, iupac_rate: float = 1e-4) -> bytes:
    """Uniform i.i.d. ACGT with N-runs, lower-case (soft-masked) blocks and rare IUPAC codes."""
    rng = np.random.Generator(np.random.PCG64(seed))
    seq = _ACGT[rng.integers(0, 4, size=length, dtype=np.uint8)].copy()
    # soft-masked blocks
    n_blocks = max(1, int(length * lower_frac / 500)) if lower_frac > 0 else 0
    for _ in range(n_blocks):
        s = int(rng.integers(0, max(1, length)))
        e = min(length, s + int(rng.integers(50, 950)))
        seq[s:e] |= 0x20
    # N runs
    if n_run_every > 0:
        pos = int(rng.integers(0, n_run_every))
        while pos < length:
            run = int(rng.integers(1, max_n_run + 1))
            seq[pos:pos + run] = ord("N")
            pos += int(rng.integers(n_run_every // 2, n_run_every * 3 // 2 + 1))
    # IUPAC codes
    n_iupac = int(length * iupac_rate)
    if n_iupac:
        idx = rng.integers(0, length, size=n_iupac)
        codes = np.frombuffer(b"YRWSMKHBVDyrn", dtype=np.uint8)
        seq[idx] = codes[rng.integers(0, len(codes), size=n_iupac)]
    return seq.tobytes()


def mutate_sample(ref: bytes, seed: int, sub_rate: float = 1e-2, indel_rate: float = 1e-3,
                  big_del_per_mb: float = 5.0, big_del_len: int = 20_000) -> bytes:
    """Sample genome = reference with substitutions, 1-10 bp indels and large deletions."""
    rng = np.random.Generator(np.random.PCG64(seed))
    a = np.frombuffer(ref, dtype=np.uint8).copy()
    a = np.where(a >= 0x60, a - 0x20, a).astype(np.uint8)  # upper-case
    n = len(a)
    valid = _CODE[a] < 4
    # substitutions
    sub = (rng.random(n) < sub_rate) & valid
    shift = rng.integers(1, 4, size=int(sub.sum()), dtype=np.uint8)
    a[sub] = _ACGT[(_CODE[a[sub]] + shift) & 3]
    keep = np.ones(n, dtype=bool)
    # big deletions
    n_big = int(round(n / 1e6 * big_del_per_mb))
    for _ in range(n_big):
        s = int(rng.integers(0, max(1, n - 1)))
        keep[s:s + min(big_del_len, max(1, n // 20))] = False
    # small deletions
    n_del = int(n * indel_rate / 2)
    for s in rng.integers(0, n, size=n_del):
        keep[s:s + int(rng.integers(1, 11))] = False
    out = a[keep]
    # small insertions
    n_ins = int(n * indel_rate / 2)
    if n_ins and len(out) > 0:
        pos = np.sort(rng.integers(0, len(out), size=n_ins))
        pieces = []
        last = 0
        for p in pos:
            pieces.append(out[last:p])
            pieces.append(_ACGT[rng.integers(0, 4, size=int(rng.integers(1, 11)), dtype=np.uint8)])
            last = p
        pieces.append(out[last:])
        out = np.concatenate(pieces)
    return out.tobytes()


def write_fasta(path: str, contigs: list[tuple[str, bytes]], line_bases: int = 60) -> None:
    """FASTA with uniform line width and a trailing newline (FastaIndex.java:169 steps past the last line end)."""
    with open(path, "wb") as f:
        for name, seq in contigs:
            f.write(b">" + name.encode() + b"\n")
            a = np.frombuffer(seq, dtype=np.uint8)
            n = len(a)
            full = n // line_bases
            if full:
                body = np.empty((full, line_bases + 1), dtype=np.uint8)
                body[:, :line_bases] = a[:full * line_bases].reshape(full, line_bases)
                body[:, line_bases] = 10
                f.write(body.tobytes())
            if n % line_bases:
                f.write(a[full * line_bases:].tobytes() + b"\n")


# ---------------------------------------------------------------------------------------
# k-mers (left-aligned 2-bit words, the reference's Kmer.kmerLong layout, Kmer.java:232-252)
# ---------------------------------------------------------------------------------------
def kmers_of_text(seq: bytes, k: int, canonical: bool) -> np.ndarray:
    """All valid k-mers of `seq` (case-insensitive ACGT windows) as left-aligned uint64, in order."""
    a = np.frombuffer(seq, dtype=np.uint8)
    code = _CODE[a]
    n = len(a)
    if n < k:
        return np.zeros(0, dtype=np.uint64)
    bad = (code > 3).astype(np.int64)
    cs = np.concatenate([[0], np.cumsum(bad)])
    ok = (cs[k:] - cs[:-k]) == 0
    c = (code & 3).astype(np.uint64)
    w = np.zeros(n - k + 1, dtype=np.uint64)
    for j in range(k):
        w |= c[j:n - k + 1 + j] << np.uint64(62 - 2 * j)
    w = w[ok]
    if canonical:
        w = np.minimum(w, revcomp(w, k))
    return w


def revcomp(w: np.ndarray, k: int) -> np.ndarray:
    w = np.asarray(w, dtype=np.uint64)
    out = np.zeros_like(w)
    for i in range(k):
        b = (w >> np.uint64(62 - 2 * i)) & np.uint64(3)
        out |= (np.uint64(3) - b) << np.uint64(62 - 2 * (k - 1 - i))
    return out


def kmer_to_str(w: int, k: int) -> str:
    return "".join("ACGT"[(int(w) >> (62 - 2 * i)) & 3] for i in range(k))


def str_to_kmer(s: str) -> int:
    w = 0
    for i, ch in enumerate(s):
        w |= "ACGT".index(ch) << (62 - 2 * i)
    return w


# ---------------------------------------------------------------------------------------
# KMC signature (independent, vectorised restatement of Signature.java:23-95)
# ---------------------------------------------------------------------------------------
def norm_table(m: int) -> np.ndarray:
    special = 1 << (2 * m)
    x = np.arange(special, dtype=np.int64)

    def allowed(s: np.ndarray) -> np.ndarray:
        ok = ((s & 0x3F) != 0x3F) & ((s & 0x3F) != 0x3B) & ((s & 0x3C) != 0x3C)
        t = s.copy()
        for _ in range(m - 3):
            ok &= (t & 0xF) != 0
            t >>= 2
        ok &= (t != 0) & (t != 4) & ((t & 0xF) != 0)
        return ok

    rev = np.zeros_like(x)
    t = x.copy()
    for _ in range(m):
        rev = (rev << 2) | (3 - (t & 3))
        t >>= 2
    sv = np.where(allowed(x), x, special)
    rv = np.where(allowed(rev), rev, special)
    return np.minimum(sv, rv).astype(np.int32)


def signatures(kmers: np.ndarray, k: int, m: int, norm: np.ndarray) -> np.ndarray:
    """min over the k-m+1 m-mers of norm[m-mer] (Kmer.java:105-118)."""
    kmers = np.asarray(kmers, dtype=np.uint64)
    mask = np.uint64((1 << (2 * m)) - 1)
    best = np.full(len(kmers), 1 << (2 * m), dtype=np.int32)
    for i in range(k - m + 1):
        mm = (kmers >> np.uint64(64 - 2 * (i + m))) & mask
        best = np.minimum(best, norm[mm.astype(np.int64)])
    return best


# ---------------------------------------------------------------------------------------
# KMC database writer (layout: SURVEY.md Appendix A; KMC.java:107-189)
# ---------------------------------------------------------------------------------------
@dataclass
class KmcParams:
    k: int = 31
    p: int = 7                 # lut_prefix_length; (k - p) % 4 must be 0
    counter_size: int = 2
    sig_len: int = 9
    n_bins: int = 512
    both_strands: bool = True  # header byte 0 => canonical (KMC.java:133)
    min_count: int = 1
    max_count: int = 65535
    mode: int = 0


def signature_map(sig_len: int, n_bins: int) -> np.ndarray:
    """Any map sig -> bin is valid for the reader (Appendix A); use a multiplicative scramble."""
    s = np.arange((1 << (2 * sig_len)) + 1, dtype=np.uint64)
    return (((s * np.uint64(2654435761)) >> np.uint64(7)) % np.uint64(n_bins)).astype(np.uint32)


def write_kmc(prefix: str, kmers: np.ndarray, counts: np.ndarray, prm: KmcParams) -> None:
    """Write <prefix>.kmc_pre / .kmc_suf.  `kmers`: distinct left-aligned uint64 (already canonical
    when prm.both_strands); `counts`: same length."""
    k, p, cs, m = prm.k, prm.p, prm.counter_size, prm.sig_len
    assert (k - p) % 4 == 0 and 1 <= k <= 32
    kmers = np.asarray(kmers, dtype=np.uint64)
    counts = np.asarray(counts, dtype=np.uint64)
    assert len(np.unique(kmers)) == len(kmers), "KMC databases hold distinct k-mers"
    n = len(kmers)
    norm = norm_table(m)
    smap = signature_map(m, prm.n_bins)
    bins = smap[signatures(kmers, k, m, norm)].astype(np.uint64) if n else np.zeros(0, dtype=np.uint64)
    order = np.lexsort((kmers, bins))
    kmers, counts, bins = kmers[order], counts[order], bins[order]
    lut_size = 1 << (2 * p)
    pre = (kmers >> np.uint64(64 - 2 * p)) if p else np.zeros(n, dtype=np.uint64)
    combined = bins * np.uint64(lut_size) + pre
    lut = np.searchsorted(combined, np.arange(prm.n_bins * lut_size, dtype=np.uint64), side="left").astype(np.uint64)

    sb = (k - p) // 4
    rec = np.zeros((n, sb + cs), dtype=np.uint8)
    suf = (kmers << np.uint64(2 * p)) if p else kmers
    for j in range(sb):
        rec[:, j] = ((suf >> np.uint64(56 - 8 * j)) & np.uint64(0xFF)).astype(np.uint8)
    for j in range(cs):
        rec[:, sb + j] = ((counts >> np.uint64(8 * j)) & np.uint64(0xFF)).astype(np.uint8)

    os.makedirs(os.path.dirname(os.path.abspath(prefix)), exist_ok=True)
    with open(prefix + ".kmc_suf", "wb") as f:
        f.write(b"KMCS")
        f.write(rec.tobytes())
        f.write(b"KMCS")
    header = struct.pack("<7IQB3x6II", k, prm.mode, cs, p, m, prm.min_count, prm.max_count & 0xFFFFFFFF, n,
                         0 if prm.both_strands else 1, 0, 0, 0, 0, 0, 0, 0x200)
    assert len(header) == 68
    with open(prefix + ".kmc_pre", "wb") as f:
        f.write(b"KMCP")
        f.write(lut.astype("<u8").tobytes())
        f.write(struct.pack("<Q", n))          # guard (KMC.java:153-156 skips it)
        f.write(smap.astype("<u4").tobytes())
        f.write(header)
        f.write(struct.pack("<I", len(header)))
        f.write(b"KMCP")


def build_sample_db(sample_text: bytes, prm: KmcParams, seed: int, noise_kmers: int = 0,
                    drop_below: int = 0) -> tuple[np.ndarray, np.ndarray]:
    """(kmers, counts) of a sample genome: all its (canonical) k-mers with occurrence counts capped at the
    counter maximum, plus `noise_kmers` uniformly random k-mers with small counts."""
    w = kmers_of_text(sample_text, prm.k, prm.both_strands)
    uniq, cnt = np.unique(w, return_counts=True)
    rng = np.random.Generator(np.random.PCG64(seed))
    # give the counts a depth-like spread so that -c thresholds and KD are exercised
    depth = rng.poisson(12, size=len(uniq)).astype(np.uint64) + 1
    cnt = cnt.astype(np.uint64) * depth
    if noise_kmers:
        mask = np.uint64(0xFFFFFFFFFFFFFFFF) << np.uint64(64 - 2 * prm.k) if prm.k < 32 else np.uint64(0xFFFFFFFFFFFFFFFF)
        nz = rng.integers(0, np.iinfo(np.uint64).max, size=noise_kmers, dtype=np.uint64, endpoint=True)
        nz &= mask
        if prm.both_strands:
            nz = np.minimum(nz, revcomp(nz, prm.k))
        nz = np.setdiff1d(np.unique(nz), uniq)
        uniq = np.concatenate([uniq, nz])
        cnt = np.concatenate([cnt, rng.integers(1, 4, size=len(nz), dtype=np.uint64)])
    cmax = min(prm.max_count, (1 << (8 * prm.counter_size)) - 1) if prm.counter_size else 0
    cnt = np.minimum(cnt, cmax)
    if drop_below:
        keep = cnt >= drop_below
        uniq, cnt = uniq[keep], cnt[keep]
    return uniq, cnt


def tiling_windows(seq_len: int, window: int, k: int, step: int = 0) -> tuple[np.ndarray, np.ndarray]:
    """GetVariants.java:293-320 (independent restatement used by the generators)."""
    starts, ends = [], []
    if step > 0:
        pos = 0
        while pos < seq_len:
            e = min(pos + window, seq_len)
            if e - pos >= k:
                starts.append(pos)
                ends.append(e)
            pos += step
    else:
        assert window > k - 1
        last_end = 0
        while last_end < seq_len:
            s = max(0, last_end - k + 1)
            e = min(s + window, seq_len)
            if e - s >= k:
                starts.append(s)
                ends.append(e)
            last_end = e
    return np.asarray(starts, dtype=np.int64), np.asarray(ends, dtype=np.int64)


def random_gtf(path: str, contigs: list[tuple[str, int]], seed: int, genes_per_mb: float = 200.0,
               with_gene_lines: bool = True) -> None:
    """Synthetic GTF (SURVEY 8d config 3 style): genes with 1-3 transcripts (overlapping isoforms sharing exons),
    1-8 exons of 50-900 bp, both strands; optionally without `gene` lines (GTF.java:61-68 synthesises them); plus
    feature types the parser must skip (CDS, start_codon) and a comment / blank line."""
    rng = np.random.Generator(np.random.PCG64(seed))
    lines = ["# synthetic annotation", ""]
    gid = 0
    for name, length in contigs:
        n_genes = max(1, int(length / 1e6 * genes_per_mb))
        starts = np.sort(rng.integers(1, max(2, length - 6000), size=n_genes))
        for gs in starts:
            gid += 1
            strand = "+-"[int(rng.integers(0, 2))]
            gene = f"G{gid:05d}"
            kind = "pseudogene" if rng.random() < 0.1 else "gene"
            n_tx = int(rng.integers(1, 4))
            # shared exon pool
            n_ex = int(rng.integers(1, 9))
            pos = int(gs)
            pool = []
            for _ in range(n_ex):
                el = int(rng.integers(50, 900))
                if pos + el >= length:
                    break
                pool.append((pos, pos + el - 1))
                pos += el + int(rng.integers(30, 600))
            if not pool:
                continue
            gene_end = pool[-1][1]
            recs = []
            for t in range(n_tx):
                tx = f"{gene}.t{t + 1}"
                keep = [e for e in pool if rng.random() < 0.8] or [pool[0]]
                # isoform-specific boundary shifts create overlapping, non-identical exons
                ex = [(max(1, s - int(rng.integers(0, 20)) * (rng.random() < 0.3)), min(length, e + int(rng.integers(0, 20)) * (rng.random() < 0.3))) for s, e in keep]
                ttype = ["transcript", "mRNA", "lnc_RNA"][int(rng.integers(0, 3))]
                recs.append((tx, ttype, ex))
                gene_end = max(gene_end, max(e for _, e in ex))
            if with_gene_lines:
                lines.append(f'{name}\tsynth\t{kind}\t{pool[0][0]}\t{gene_end}\t.\t{strand}\t.\tgene_id "{gene}";')
            for tx, ttype, ex in recs:
                ts, te = min(s for s, _ in ex), max(e for _, e in ex)
                lines.append(f'{name}\tsynth\t{ttype}\t{ts}\t{te}\t.\t{strand}\t.\tgene_id "{gene}"; transcript_id "{tx}";')
                for s, e in ex:
                    lines.append(f'{name}\tsynth\texon\t{s}\t{e}\t.\t{strand}\t.\tgene_id "{gene}"; transcript_id "{tx}";')
                    if rng.random() < 0.3:
                        lines.append(f'{name}\tsynth\tCDS\t{s}\t{e}\t.\t{strand}\t0\tgene_id "{gene}"; transcript_id "{tx}";')
    with open(path, "w") as f:
        f.write("\n".join(lines) + "\n")
