"""Python restatement of the host-side rows around the hot path: the .faidx FASTA index,
GTF gene / transcript windows and the KCF text format.  TEST INFRASTRUCTURE ONLY; PARITY UNPINNED (the reference
cannot run here and has no tests) -- every function cites the Java lines it follows
(paths under /root/reference/src/main/java/nl/wur/bis/kcftools/).
"""
from __future__ import annotations

import datetime
import os
import struct
from decimal import ROUND_HALF_UP, Decimal

KCF_VERSION = "0.4.0"   # pom.xml:8 -> version.properties -> Configs.getVersion (Utils/Configs.java:49-57)

# Utils/Configs.java:14-37 (Java text blocks: common indentation stripped, no trailing newline)
KCF_INFO_LINES = [
    '<ID=EFFLEN,Type=Integer,Description="Effective length of the window">',
    '<ID=IS,Type=Float,Description="Minimum score for the window">',
    '<ID=XS,Type=Float,Description="Maximum score for the window">',
    '<ID=MS,Type=Float,Description="Mean score for the window">',
    '<ID=IO,Type=Integer,Description="Minimum observed kmers in the window">',
    '<ID=XO,Type=Integer,Description="Maximum observed kmers in the window">',
    '<ID=MO,Type=Integer,Description="Mean observed kmers in the window">',
    '<ID=IV,Type=Integer,Description="Minimum variations in the window">',
    '<ID=XV,Type=Integer,Description="Maximum variations in the window">',
    '<ID=MV,Type=Integer,Description="Mean variations in the window">',
]
KCF_FORMAT_LINES = [
    '<ID=IB,Type=Integer,Description="IBS number">',
    '<ID=VA,Type=Integer,Description="Variations">',
    '<ID=OB,Type=Integer,Description="Observed kmers">',
    '<ID=ID,Type=Integer,Description="Inner Distance">',
    '<ID=LD,Type=Integer,Description="Kmer Variation Distance at the leftTail">',
    '<ID=RD,Type=Integer,Description="Kmer Variation Distance at the rightTail">',
    '<ID=KD,Type=Float,Description="Mean Kmer Depth">',
    '<ID=SC,Type=Float,Description="Score">',
]


# ---------------------------------------------------------------------------------------------------------
# Java number formatting
# ---------------------------------------------------------------------------------------------------------
def java_fmt_2f(x: float) -> str:
    """String.format("%.2f", double) under Locale.US (KCFTOOLS.java:31): Java rounds HALF_UP on the digits of the
    shortest repr of the double (FormattedFloatingDecimal), not on its exact binary value: 2.125 -> 2.13, 1.005 -> 1.01."""
    if x != x:
        return "NaN"
    if x in (float("inf"), float("-inf")):
        return "Infinity" if x > 0 else "-Infinity"
    d = Decimal(repr(float(x)))
    q = d.quantize(Decimal("0.01"), rounding=ROUND_HALF_UP)
    s = f"{abs(q):.2f}"
    neg = (struct.pack(">d", float(x))[0] & 0x80) != 0
    return ("-" + s) if neg else s


def _java_sci(digits: str, exp10: int, neg: bool) -> str:
    """Double/Float.toString layout from the shortest digit string `digits` with value 0.digits x 10^exp10."""
    sign = "-" if neg else ""
    e = exp10 - 1                      # exponent of d.ddd form
    if -3 <= e < 7:
        if e >= 0:
            ip = digits[: e + 1].ljust(e + 1, "0")
            fp = digits[e + 1:] or "0"
        else:
            ip = "0"
            fp = "0" * (-e - 1) + digits
        return f"{sign}{ip}.{fp}"
    m = digits[0] + "." + (digits[1:] or "0")
    return f"{sign}{m}E{e}"


def java_double_to_string(x: float) -> str:
    """String.valueOf(double) (KCFHeader.java:140-166 weights): shortest digits, decimal for 1e-3 <= |x| < 1e7."""
    if x == 0:
        return "-0.0" if struct.pack(">d", x)[0] & 0x80 else "0.0"
    d = Decimal(repr(abs(float(x))))
    sign, digs, exp = d.as_tuple()
    digits = "".join(map(str, digs)).rstrip("0") or "0"
    exp10 = len(digs) + exp
    return _java_sci(digits, exp10, x < 0)


def java_float_to_string(x: float) -> str:
    """Float.toString(float) (Window.java:151 prints the float meanVariations by concatenation)."""
    import numpy as np
    f = np.float32(x)
    if f == 0:
        return "0.0"
    s = np.format_float_scientific(abs(f), unique=True, trim="-")     # shortest digits for float32
    mant, e = s.split("e")
    digits = mant.replace(".", "").rstrip("0") or "0"
    return _java_sci(digits, int(e) + 1, f < 0)


# ---------------------------------------------------------------------------------------------------------
# FASTA index (Data/FastaIndex.java)
# ---------------------------------------------------------------------------------------------------------
VALID_BASES = set("ACGTYRWSMKHBVDNacgtyrwsmkhbvdn")    # FastaIndex.java:21-24


def generate_faidx(fasta_path: str) -> list[tuple[str, int, int, int, int]]:
    """FastaIndex.generateIndexFile (:239-299): (name, length, offset, lineBases, lineWidth) per record; offsets count
    one byte per line separator; lineBases/lineWidth come from the record's first sequence line."""
    entries = []
    offset = 0
    cur = None
    seq_len = line_bases = line_width = start = 0
    with open(fasta_path, "r", newline="") as f:
        data = f.read()
    lines = data.split("\n")
    if lines and lines[-1] == "":
        lines.pop()                        # BufferedReader.readLine yields no empty last line
    for line in lines:
        line = line.rstrip("\r")           # readLine strips \r\n too (offsets then drift, as in the reference)
        if offset == 0 and not line.startswith(">"):
            raise ValueError("Invalid fasta file: " + fasta_path)                      # :261-263
        if line.startswith(">"):
            if cur is not None:
                entries.append((cur, seq_len, start, line_bases, line_width))
            cur = line[1:].split(" ")[0]                                                # :268
            offset += len(line) + 1
            start = offset
            seq_len = 0
            if any(e[0] == cur for e in entries):
                raise ValueError("Duplicate sequence name in fasta file: " + cur)     # :273-274
        else:
            for ch in line:
                if ch not in VALID_BASES:
                    raise ValueError(f"Invalid character '{ch}' in fasta file")       # :279-286
            if seq_len == 0:
                line_bases = len(line)
                line_width = len(line) + 1
            seq_len += len(line)
            offset += len(line) + 1
    if cur is not None:
        entries.append((cur, seq_len, start, line_bases, line_width))
    return entries


def faidx_text(entries) -> str:
    return "".join(f"{n}\t{l}\t{o}\t{lb}\t{lw}\n" for n, l, o, lb, lw in entries)


def get_sequence(data: bytes, entry, start: int, length: int) -> bytes:
    """FastaIndex.getSequence (:122-182): pure arithmetic on (offset, lineBases, lineWidth)."""
    name, seqlen, offset, line_bases, line_width = entry
    end = start + length
    if start < 0 or end > seqlen or start >= end:
        raise ValueError(f"Invalid range: {start}-{end} for sequence: {name}")        # :131-134
    out = bytearray()
    pos = offset + (start // line_bases) * line_width + start % line_bases             # :148-154
    in_line = start % line_bases
    todo = length
    while todo > 0:
        n = min(todo, line_bases - in_line)                                            # :160-161
        out += data[pos:pos + n]
        pos += n + (line_width - line_bases)                                           # :169
        todo -= n
        in_line = 0
    return bytes(out)


# ---------------------------------------------------------------------------------------------------------
# GTF (Data/GTF.java)
# ---------------------------------------------------------------------------------------------------------
class Gtf:
    GENE = ("gene", "pseudogene")
    TRANSCRIPT = ("transcript", "mRNA", "RNA", "lnc_RNA", "rRNA", "tRNA", "snRNA", "snoRNA")

    def __init__(self, path: str):
        self.children: dict[str, list[str]] = {}      # vertex -> children in edge insertion order (jgrapht)
        self.feature: dict[str, list] = {}            # id -> [chrom, start, end, strand, type]
        exon_counts: dict[str, int] = {}

        def vertex(v):
            self.children.setdefault(v, [])

        def edge(a, b):
            vertex(a)
            vertex(b)
            if b not in self.children[a]:
                self.children[a].append(b)

        for line in open(path):
            line = line.rstrip("\n")
            if line.startswith("#") or not line.strip():
                continue                                                       # GTF.java:35
            f = line.split("\t")
            if len(f) < 9:
                raise ValueError("Malformed line: " + line)                   # :37-40 (Logger.error exits)
            attrs = {}
            for a in f[8].split(";"):                                          # :156-163
                pair = a.strip().replace('"', "").split(" ")
                if len(pair) == 2:
                    attrs[pair[0]] = pair[1]
            typ, chrom = f[2], f[0]
            vertex(chrom)                                                      # :46
            if typ in self.GENE:                                               # :49-52
                fid, parent = attrs.get("gene_id"), chrom
            elif typ in self.TRANSCRIPT:                                       # :53-76
                fid, parent = attrs.get("transcript_id"), attrs.get("gene_id")
                if fid == parent:
                    raise ValueError("Transcript ID is the same as Gene ID: " + str(fid))
                if parent not in self.children:
                    vertex(parent)
                    self.feature[parent] = [chrom, int(f[3]), int(f[4]), f[6][0], "gene"]
                    edge(chrom, parent)
                g = self.feature.get(parent)
                if g is not None:
                    g[1] = min(g[1], int(f[3]))
                    g[2] = max(g[2], int(f[4]))
            elif typ == "exon":                                                # :77-82
                parent = attrs.get("transcript_id")
                exon_counts[parent] = exon_counts.get(parent, 0) + 1
                fid = f"{parent}-e-{exon_counts[parent]}"
            else:
                continue
            self.feature[fid] = [f[0], int(f[3]), int(f[4]), f[6][0], typ]     # :87-90
            vertex(fid)
            if parent is not None:
                edge(parent, fid)                                              # :92-95

    def kids(self, v):
        return [c for c in self.children.get(v, []) if c != v]                 # :207-217

    def loci(self, fid):
        c, s, e, st, _ = self.feature[fid]                                     # :299-306
        return c, s, e, st

    def merged_exon_loci(self, fid: str, is_gene: bool):
        """GTF.getFasta (:223-239): the exon loci of the gene's transcripts (or of one transcript), de-duplicated,
        sorted by (chromosome, start) and merged when consecutive ones overlap on the same strand."""
        if fid not in self.children:
            return None
        targets = self.kids(fid)
        loci = []
        for t in targets:
            for ex in (self.kids(t) if is_gene else [t]):
                f = self.feature.get(ex)
                if f is not None:
                    l = (f[0], f[1], f[2], f[3])
                    if l not in loci:
                        loci.append(l)
        if not loci:
            return None
        loci.sort(key=lambda l: (l[0], l[1]))                                  # Loci.compareTo :413-419 (stable)
        merged = []
        for cur in loci:                                                       # :278-293
            if merged and merged[-1][0] == cur[0] and merged[-1][3] == cur[3] and merged[-1][1] <= cur[2] and cur[1] <= merged[-1][2]:
                last = merged[-1]
                merged[-1] = (last[0], min(last[1], cur[1]), max(last[2], cur[2]), last[3])
            else:
                merged.append(cur)
        return merged


# ---------------------------------------------------------------------------------------------------------
# KCF text (Data/KCFHeader.java:291-330, Data/Window.java:125-152,170-214, Data/Data.java:120-132)
# ---------------------------------------------------------------------------------------------------------
def kcf_header(reference: str, contigs, sample: str, window: int, step: int, kmer: int, n_windows: int,
               wi: float, wt: float, wr: float, cmd: str, date: str | None = None) -> str:
    date = date or datetime.date.today().strftime("%Y-%m-%d")
    out = [f"##format=KCF{KCF_VERSION}", f"##date={date}", "##source=kcftools", f"##reference={reference}"]
    out += [f"##contig=<ID={n},length={l}>" for n, l in contigs]
    out += ["##INFO=" + l for l in KCF_INFO_LINES]
    out += ["##FORMAT=" + l for l in KCF_FORMAT_LINES]
    params = [("window", str(window)), ("step", str(step)), ("kmer", str(kmer)), ("IBS", "false"),
              ("nwindow", str(n_windows)), ("wti", java_double_to_string(wi)), ("wtt", java_double_to_string(wt)),
              ("wtk", java_double_to_string(wr))]
    out += [f"##PARAM=<ID={k},value={v}>" for k, v in params]
    out.append("##CMD=" + cmd)
    out.append("#CHROM\tSTART\tEND\tID\tTOTAL_KMERS\tINFO\tFORMAT\t" + sample)
    return "\n".join(out) + "\n"


FLOAT_MIN_VALUE = 1.401298464324817e-45     # Float.MIN_VALUE: Window.calculateStats starts maxScore there (:185)


def kcf_row(seq: str, start: int, end: int, wid: str, res: dict, score: float, mean_count: float) -> str:
    """One window, one sample. calculateStats (:177-214): IS = score, XS = max(Float.MIN_VALUE, score) (so XS prints 0.00
    for scores <= 0), MS = score, IO = XO = obs, MO = (float) obs, IV = XV = var, MV = (float) var."""
    xs = score if score > FLOAT_MIN_VALUE else FLOAT_MIN_VALUE
    info = (f"EFFLEN={res['eff_length']};IS={java_fmt_2f(score)};XS={java_fmt_2f(xs)};MS={java_fmt_2f(score)};"
            f"IO={res['obs']};XO={res['obs']};MO={java_fmt_2f(float(res['obs']))};IV={res['var']};XV={res['var']};"
            f"MV={java_float_to_string(float(res['var']))}")
    data = ":".join(["N", str(res["var"]), str(res["obs"]), str(res["inner"]), str(res["left"]), str(res["right"]),
                     java_fmt_2f(mean_count), java_fmt_2f(score)])
    return "\t".join([seq, str(start), str(end), wid, str(res["total_kmers"]), info, "GT:VA:OB:ID:LD:RD:KD:SC", data])


# ---------------------------------------------------------------------------------------------------------
# cohort (Plugins/Cohort.java:71-119): S single-sample KCFs of the same windows -> one multi-sample KCF (row f-4)
# ---------------------------------------------------------------------------------------------------------
FLOAT_MAX_VALUE = 3.4028234663852886e38


def cohort_sample_from_text(field: str, total: int, eff: int, wi: float, wt: float, wr: float) -> dict:
    """Window.parseSampleData (Window.java:57-83) + Data ctor (Data.java:40-68): what `cohort` keeps of a sample column.
    The count sum is re-derived from the 2-decimal KD text: round(KD * obs) (Math.round = floor(x + 0.5))."""
    import math
    f = field.split(":")
    var, obs, inner, left, right = (int(x) for x in f[1:6])
    kmer_count = math.floor(float(f[6]) * obs + 0.5)                          # Window.java:70
    mean = kmer_count / obs if kmer_count > 0 else 0.0                         # Data.java:57-59
    if obs == 0 or total == 0 or eff == 0:                                      # Data.java:96-98
        sc = 0.0
    else:
        sc = ((wr * (obs / total)) + (wi * (1.0 - inner / eff)) + (wt * (1.0 - (left + right) / eff))) * 100.0
    return dict(ibs=f[0], var=var, obs=obs, inner=inner, left=left, right=right, mean=mean, score=sc)


def cohort_row(seq: str, start: int, end: int, wid: str, total: int, eff: int, samples: list[dict]) -> str:
    """Window.toString for several samples (Window.java:125-152, calculateStats :177-214): min / max / mean over the
    samples; observed-k-mer and variation means are accumulated in float; maxScore starts at Float.MIN_VALUE."""
    import numpy as np
    min_s, max_s, mean_s = FLOAT_MAX_VALUE, FLOAT_MIN_VALUE, 0.0
    mo = np.float32(0)
    mv = np.float32(0)
    for d in samples:
        min_s = min(min_s, d["score"]) if d["score"] < min_s else min_s
        max_s = d["score"] if d["score"] > max_s else max_s
        mean_s += d["score"]
        mo = np.float32(mo + np.float32(d["obs"]))
        mv = np.float32(mv + np.float32(d["var"]))
    n = len(samples)
    mo = np.float32(mo / np.float32(n))
    mv = np.float32(mv / np.float32(n))
    mean_s /= n
    info = (f"EFFLEN={eff};IS={java_fmt_2f(min_s)};XS={java_fmt_2f(max_s)};MS={java_fmt_2f(mean_s)};"
            f"IO={min(d['obs'] for d in samples)};XO={max(d['obs'] for d in samples)};MO={java_fmt_2f(float(mo))};"
            f"IV={min(d['var'] for d in samples)};XV={max(d['var'] for d in samples)};MV={java_float_to_string(float(mv))}")
    cols = [":".join([d["ibs"], str(d["var"]), str(d["obs"]), str(d["inner"]), str(d["left"]), str(d["right"]),
                      java_fmt_2f(d["mean"]), java_fmt_2f(d["score"])]) for d in samples]
    return "\t".join([seq, str(start), str(end), wid, str(total), info, "GT:VA:OB:ID:LD:RD:KD:SC"] + cols)
