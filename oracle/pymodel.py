"""Second, independent restatement of the getVariations hot path (pure Python / numpy).

TEST INFRASTRUCTURE ONLY -- same rules as oracle/kcf_oracle.h.  PARITY UNPINNED (no runnable
reference, no reference tests).  This model deliberately does NOT follow the reference's data
structures: database lookups are dict membership over k-mer *strings* (instead of signature ->
bin -> LUT -> bisection), gaps are found by run-length grouping (instead of the streaming state
machine), EFFLEN by a regular expression.  It must agree with oracle/kcf_oracle.c, whose code
follows the Java line by line; disagreement means one of the two mis-reads the reference.

Semantics restated (paths under /root/reference/src/main/java/nl/wur/bis/kcftools/):
  Fasta.java:90-127,140-167   valid k-mer list, effective length
  Kmer.java:72-79,300-338     canonical = min(fwd, revcomp), A<C<G<T
  KMC.java:107-189,427-450    file layout, enumeration of (k-mer, count)
  GetVariants.java:202-273    obs / var / inner / left / right / count sum
  Data.java:87,95-107         mean count, score
"""
from __future__ import annotations

import itertools
import re
import struct

import numpy as np

_COMP = str.maketrans("ACGT", "TGCA")


def revcomp_str(s: str) -> str:
    return s.translate(_COMP)[::-1]


def read_kmc(prefix: str) -> tuple[dict, dict]:
    """Parse <prefix>.kmc_pre/.kmc_suf -> ({kmer string: count}, header dict)."""
    pre = open(prefix + ".kmc_pre", "rb").read()
    suf = open(prefix + ".kmc_suf", "rb").read()
    size = len(pre)
    (hdr_off,) = struct.unpack_from("<I", pre, size - 8)
    pos = size - hdr_off - 8
    k, mode, cs, p, m, cmin, cmax = struct.unpack_from("<7I", pre, pos)
    (total,) = struct.unpack_from("<Q", pre, pos + 28)
    both = pre[pos + 36] == 0
    (version,) = struct.unpack_from("<I", pre, pos + 64)
    if version != 0x200:
        raise ValueError("KMC version is not 0x200")
    map_len = 4 ** m + 1
    map_start = pos - 4 * map_len
    lut_size = 4 ** p
    n_bins = (map_start - 8 - 4) // (lut_size * 8)
    lut = np.frombuffer(pre, dtype="<u8", count=n_bins * lut_size, offset=4).astype(np.int64)
    sb = (k - p) // 4
    rs = sb + cs
    rec = np.frombuffer(suf, dtype=np.uint8, count=total * rs, offset=4).reshape(total, rs)
    # record j belongs to the last LUT slot i with lut[i] <= j ; prefix = i mod 4^p
    slot = np.searchsorted(lut, np.arange(total), side="right") - 1
    prefix_int = slot % lut_size
    table = {}
    for j in range(total):
        pi = int(prefix_int[j])
        s = "".join("ACGT"[(pi >> (2 * (p - 1 - b))) & 3] for b in range(p))
        for b in range(k - p):
            s += "ACGT"[(int(rec[j, b // 4]) >> (6 - 2 * (b % 4))) & 3]
        c = 0
        for b in range(cs):
            c |= int(rec[j, sb + b]) << (8 * b)
        if c >= 1 << 31:
            c -= 1 << 32                      # Java int
        table[s] = c
    hdr = dict(k=k, mode=mode, counter_size=cs, p=p, sig_len=m, min_count=cmin, max_count=cmax,
               total=total, both_strands=both, n_bins=n_bins)
    return table, hdr


def window_counts(text: str, table: dict, k: int, both_strands: bool) -> list[int]:
    """Counts of the valid k-mers of `text`, in order (invalid positions vanish)."""
    up = text.upper()
    out = []
    for i in range(len(up) - k + 1):
        s = up[i:i + k]
        if s.strip("ACGT"):
            continue
        if both_strands:
            r = revcomp_str(s)
            if r < s:
                s = r
        out.append(table.get(s, 0))
    return out


def distance(k: int, g: int) -> int:
    d = g - (k - 1)
    return d if d > 0 else abs(d + 1)


def reduce_counts(counts: list[int], k: int, min_count: int) -> dict:
    hits = [c >= min_count for c in counts]
    runs = [(h, len(list(g))) for h, g in itertools.groupby(hits)]
    obs = sum(hits)
    res = dict(total_kmers=len(counts), obs=obs, var=0, inner=0, left=0, right=0,
               kmer_count_sum=sum(c for c, h in zip(counts, hits) if h))
    for idx, (h, n) in enumerate(runs):
        if h:
            continue
        res["var"] += 1
        if idx == len(runs) - 1:
            res["right"] += n                  # reaches the end (also the all-miss case)
        elif idx == 0:
            res["left"] += n
        else:
            res["inner"] += distance(k, n)
    return res


def eff_length(text: str, k: int) -> int:
    return sum(len(r) for r in re.findall(r"[ACGTacgt]+", text) if len(r) >= k)


def process_window(text: str, table: dict, k: int, both_strands: bool, min_count: int = 1) -> dict:
    res = reduce_counts(window_counts(text, table, k, both_strands), k, min_count)
    res["eff_length"] = eff_length(text, k)
    return res


def score(res: dict, wi: float = 0.3, wt: float = 0.3, wr: float = 0.4) -> tuple[float, float]:
    obs, total, eff = res["obs"], res["total_kmers"], res["eff_length"]
    mean = res["kmer_count_sum"] / obs if obs > 0 else 0.0
    if obs == 0 or total == 0 or eff == 0:
        return 0.0, mean
    s = ((wr * (obs / total)) + (wi * (1.0 - res["inner"] / eff)) + (wt * (1.0 - (res["left"] + res["right"]) / eff))) * 100.0
    return s, mean
