"""Test helper: a Python restatement of the reference's report tail (src/RNASeQC.cpp:397-676) used to
check our understanding against the reference's golden outputs.  Not product code."""
import math

import numpy as np

MAD_FACTOR = 1.4826


def fmt(x):
    """std::ostream default formatting of a double (precision 6), as in the reference's metrics.tsv."""
    if isinstance(x, (int, np.integer)):
        return str(int(x))
    if math.isnan(x):
        return "nan"
    if math.isinf(x):
        return "inf" if x > 0 else "-inf"
    s = "%.6g" % x
    if "e" in s:                      # C++ prints e-05, Python prints e-05 as well; normalise exponent width
        m, e = s.split("e")
        sign = e[0]
        e = e[1:].lstrip("0").rjust(2, "0")
        s = "%se%s%s" % (m, sign, e)
    return s


def frac(c, a, b):
    return np.float64(c[a]) / np.float64(c[b]) if True else 0


def metrics_rates(c):
    """The rate block of metrics.tsv (src/RNASeQC.cpp:526-551) from the raw counters `c` (dict)."""
    with np.errstate(divide="ignore", invalid="ignore"):
        f = lambda a, b: float(np.float64(c[a]) / np.float64(c[b]))
        rows = [
            ("Mapping Rate", f("Mapped Reads", "Unique Mapping, Vendor QC Passed Reads")),
            ("Unique Rate of Mapped", f("Mapped Unique Reads", "Mapped Reads")),
            ("Duplicate Rate of Mapped", f("Mapped Duplicate Reads", "Mapped Reads")),
            ("Duplicate Rate of Mapped, excluding Globins", f("Non-Globin Duplicate Reads", "Non-Globin Reads")),
            ("Base Mismatch", f("Mismatched Bases", "Total Bases")),
            ("End 1 Mapping Rate", 2.0 * f("End 1 Mapped Reads", "Unique Mapping, Vendor QC Passed Reads")),
            ("End 2 Mapping Rate", 2.0 * f("End 2 Mapped Reads", "Unique Mapping, Vendor QC Passed Reads")),
            ("End 1 Mismatch Rate", f("End 1 Mismatches", "End 1 Bases")),
            ("End 2 Mismatch Rate", f("End 2 Mismatches", "End 2 Bases")),
            ("Expression Profiling Efficiency", f("Exonic Reads", "Unique Mapping, Vendor QC Passed Reads")),
            ("High Quality Rate", f("High Quality Reads", "Mapped Reads")),
            ("Exonic Rate", f("Exonic Reads", "Mapped Reads")),
            ("Intronic Rate", f("Intronic Reads", "Mapped Reads")),
            ("Intergenic Rate", f("Intergenic Reads", "Mapped Reads")),
            ("Intragenic Rate", f("Intragenic Reads", "Mapped Reads")),
            ("Ambiguous Alignment Rate", f("Ambiguous Reads", "Mapped Reads")),
            ("Discard Rate", float(np.float64(c["Mapped Reads"] - c["Reads used for Intron/Exon counts"]) / np.float64(c["Mapped Reads"]))),
            ("rRNA Rate", f("rRNA Reads", "Mapped Reads")),
            ("End 1 Sense Rate", float(np.float64(c["End 1 Sense"]) / np.float64(c["End 1 Sense"] + c["End 1 Antisense"]))),
            ("End 2 Sense Rate", float(np.float64(c["End 2 Sense"]) / np.float64(c["End 2 Sense"] + c["End 2 Antisense"]))),
            ("Avg. Splits per Read", f("Alignment Blocks", "Mapped Reads") - 1.0) if "Alignment Blocks" in c else None,
        ]
    return [r for r in rows if r is not None]


def fragment_stats(hist, median):
    """src/RNASeQC.cpp:570-606: hist = {size: count}; median = the quirky computeMedian."""
    sizes = sorted(hist)
    expanded = [s for s in sizes for _ in range(hist[s])]
    n = float(len(expanded))
    med = median(expanded)
    avg = 0.0
    for s in sizes:
        avg += float(s * hist[s]) / n
    dev = sorted(abs(float(s) - med) for s in expanded)
    mad = median(dev) * MAD_FACTOR
    sd = 0.0
    for s in sizes:
        for _ in range(hist[s]):
            sd += (float(s) - avg) ** 2 / n
    sd = sd ** 0.5
    return avg, med, sd, mad
