"""Known-answer tests against the golden OUTPUTS the reference ships under test_data/*.output/
(its inputs are not in the tree).  The figures checked here were copied out of those files into
tests/golden/reference_known_answers.json by tests/golden/make_known_answers.py."""
import json
import os

import numpy as np
import pytest

from tests import report_ref

HERE = os.path.dirname(os.path.abspath(__file__))
KA = json.load(open(os.path.join(HERE, "golden", "reference_known_answers.json")))


@pytest.mark.parametrize("name", ["chr1", "downsampled"])
def test_quirky_median_reproduces_golden_cv_median(oracle_lib, name):
    # "Median of Transcript Coverage CV": finite CVs of coverage.tsv, sorted, computeMedian (src/RNASeQC.cpp:610-628)
    d = KA[name]
    cvs = sorted(d["coverage_cv_finite"])
    golden = float(d["metrics"]["Median of Transcript Coverage CV"])
    # the file holds 6 significant digits, so allow the last printed digit to move
    assert abs(oracle_lib.median(cvs) - golden) <= 1.5e-5 * golden
    if name == "chr1":      # n = 1791 (odd): the textbook median is a different number
        assert abs(float(np.median(cvs)) - golden) > 1e-3 * golden
    # means / stds: valid zero-coverage genes print exactly like masked-out ones ("0 0 nan"), so their number k
    # is not recoverable from the file; one k must explain both medians
    means, stds = sorted(d["coverage_mean_nonzero"]), sorted(d["coverage_std_nonzero"])
    gm, gs = float(d["metrics"]["Median of Avg Transcript Coverage"]), float(d["metrics"]["Median of Transcript Coverage Std"])
    def qmed(k, vals):          # computeMedian of k zeros followed by the sorted values, without building the list
        n = k + len(vals)
        el = lambda i: 0.0 if i < k else vals[i - k]
        if n == 1:
            return el(0)
        mid = (n - 1) // 2
        return (el(mid) + el(mid + 1)) / 2.0 if n % 2 else el(mid)
    assert qmed(3, means) == oracle_lib.median([0.0] * 3 + means)
    hits = [k for k in range(0, d["n_zero_rows"] + 1, 1)
            if abs(qmed(k, means) - gm) <= 1.5e-5 * gm and abs(qmed(k, stds) - gs) <= 1.5e-5 * gs]
    assert hits, "no zero-row count explains the golden medians"


def test_fragment_statistics_reproduce_golden(oracle_lib):
    d = KA["downsampled"]
    hist = {int(k): int(v) for k, v in d["fragment_sizes"].items()}
    avg, med, sd, mad = report_ref.fragment_stats(hist, oracle_lib.median)
    m = d["metrics"]
    assert report_ref.fmt(avg) == m["Average Fragment Length"]
    assert report_ref.fmt(med) == m["Fragment Length Median"]
    assert report_ref.fmt(sd) == m["Fragment Length Std"]
    assert report_ref.fmt(mad) == m["Fragment Length MAD_Std"]


@pytest.mark.parametrize("name", ["chr1", "downsampled", "single_pair"])
def test_rate_block_reproduces_golden(name):
    m = KA[name]["metrics"]
    counters = {k: int(v) for k, v in m.items() if v.lstrip("-").isdigit()}
    if "Alignment Blocks" not in counters:
        counters.pop("Alignment Blocks", None)
    for key, val in report_ref.metrics_rates(counters):
        if key in m:
            assert report_ref.fmt(val) == m[key], (key, report_ref.fmt(val), m[key])


@pytest.mark.parametrize("name", ["chr1", "downsampled"])
def test_exon_gct_header_counts_nonzero_rows(name):
    # Q7: the header row count of exon_reads.gct is exonCounts.size(), i.e. exons with a committed fraction
    d = KA[name]
    assert d["exon_gct_header_rows"] == d["exon_gct_nonzero_rows"]
    assert d["exon_gct_rows"] > d["exon_gct_header_rows"]
    # every counted record adds 1 to a gene and fractions summing to 1 to its exons
    assert abs(d["exon_reads_sum"] - d["gene_reads_sum"]) < 1e-3 * d["gene_reads_sum"]
    assert d["gene_fragments_sum"] <= d["gene_reads_sum"]
