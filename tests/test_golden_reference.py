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


def test_single_pair_golden_reconstruction(oracle_lib):
    """Every raw counter the reference's single_pair golden metrics.tsv lists must come out of a
    reconstruction of that input (see tests/cases.py), plus the golden GCT values."""
    from rnaseqc_amd import abi
    from tests import cases
    ann, batch = cases.single_pair_case()
    r = oracle_lib.run_oracle(abi.default_params(), ann, [batch])
    m = KA["single_pair"]["metrics"]
    got = r.counter_dict()
    checked = 0
    for k, v in m.items():
        if k in got:
            assert got[k] == int(v), (k, got[k], v)
            checked += 1
    assert checked >= 30
    assert r.read_length == int(m["Read Length"])
    assert list(r.gene_reads) == [2] and list(r.gene_fragments) == [1]
    # exon_reads.gct: 13 rows, only ..._13 is 2.0 and the header says 1
    assert r.exon_reads[ann.exon_ids.index("ENSG00000227232.4_13")] == 2.0 and r.exon_reads.sum() == 2.0
    assert int(r.exon_hit.sum()) == 1
    # "Median of Avg Transcript Coverage 0", CV list empty -> 0, "Median Exon CV nan"
    assert list(r.gene_cov_valid) == [1] and r.gene_cov_mean[0] == 0 and np.isnan(r.gene_cov_cv[0])
    assert int(r.exon_cv_valid.sum()) == 0
    # Genes Detected 0 (needs 5 unique reads), bias genes 0
    assert int((r.gene_unique >= 5).sum()) == int(m["Genes Detected"])
    assert int(((r.bias_three + r.bias_five) > 0).sum()) == int(m["Genes used in 3' bias"])


def test_gc_moment_lines_reproduce_golden(oracle_lib, tmp_path):
    """chr1.cram golden (--fasta run): the four "Fragment GC Content" lines of metrics.tsv must come out of the golden
    gc_content.tsv histogram through OUR report tail, digit for digit, and gc_content.tsv itself must be re-emitted
    byte for byte (labels i/100 in default ostream format)."""
    import ctypes as C
    from rnaseqc_amd import abi, bamio
    from tests import test_fasta_gc as tg
    from tests.test_host_cli_pieces import _results_struct, load_annotation, read_table
    import subprocess
    root = os.path.dirname(HERE)
    subprocess.check_call(["make", "-s", "-C", os.path.join(root, "rnaseqc_amd", "csrc"), "../lib/librsqc_host.so"])
    host = C.CDLL(os.path.join(root, "rnaseqc_amd", "lib", "librsqc_host.so"))
    host.host_annotation_load.restype = C.c_void_p
    host.host_annotation_load.argtypes = [C.c_char_p, C.c_char_p, C.POINTER(C.c_char_p), C.c_int, C.POINTER(C.c_int)]
    host.host_write_reports.argtypes = [C.c_void_p, C.POINTER(abi.ResultsStruct), C.c_char_p, C.c_char_p, C.c_int, C.c_int, C.c_int,
                                        C.c_uint, C.POINTER(C.c_char_p), C.c_int, C.POINTER(C.c_int), C.c_int]
    host.host_annotation_free.argtypes = [C.c_void_p]
    ann, batch, ref = tg.gc_case()
    r = oracle_lib.run_oracle(abi.default_params(coverage_mask=0), ann, [batch], reference=ref)
    g = KA["chr1_cram"]
    r.gc_bins[:] = np.array(g["gc_bins"], dtype=np.uint64)
    gtf = str(tmp_path / "q.gtf"); bamio.write_gtf(gtf, ann)
    h, err = load_annotation(host, gtf, ["chr1", "chr2"])
    assert err == 0
    rs, keep = _results_struct(r)
    out = str(tmp_path / "out"); os.makedirs(out)
    visit = (C.c_int * 2)(0, 1)
    assert host.host_write_reports(h, C.byref(rs), out.encode(), b"g.bam", 0, 0, 1, 5, None, 0, visit, 2) == 0
    m = dict(read_table(os.path.join(out, "g.bam.metrics.tsv")))
    for k in ("Fragment GC Content Mean", "Fragment GC Content Std", "Fragment GC Content Skewness", "Fragment GC Content Kurtosis"):
        assert m[k] == g["metrics"][k], (k, m[k], g["metrics"][k])
    rows = read_table(os.path.join(out, "g.bam.gc_content.tsv"), 1)
    assert [x[0] for x in rows] == g["gc_bin_labels"] and [int(x[1]) for x in rows] == g["gc_bins"]
    assert list(m)[-4:] == g["metrics_keys"][-4:]
    host.host_annotation_free(h)


def test_legacy_golden_invariants(oracle_lib):
    """legacy.output golden (--legacy run of the downsampled case): properties of the rule set that do not depend on the
    missing input, checked on the golden AND on our restatement: no globin and no "Ambiguous" counters, every counted
    read in exactly one of exonic / intronic / intergenic, intragenic = exonic + intronic, "Split Reads" printed between
    "rRNA Reads" and "Total Bases"."""
    from rnaseqc_amd import abi, synth
    g = KA["legacy"]["metrics"]
    our = None
    ann = synth.make_annotation(seed=3, contigs=[("chrA", 3_000_000, 300), ("chrB", 1_500_000, 150)])
    batch = synth.make_reads(ann, 20000, seed=4, dup_frac=0.1, contig_lengths=np.array([3_000_000, 1_500_000]))
    r = oracle_lib.run_oracle(abi.default_params(legacy=1, mapq_threshold=4), ann, [batch])
    our = {k: str(v) for k, v in r.counter_dict().items()}
    for m in (g, our):
        i = {k: int(m[k]) for k in ("Exonic Reads", "Intronic Reads", "Intergenic Reads", "Intragenic Reads",
                                    "Reads used for Intron/Exon counts", "Non-Globin Reads", "Non-Globin Duplicate Reads",
                                    "Ambiguous Reads", "Split Reads")}
        assert i["Non-Globin Reads"] == 0 and i["Non-Globin Duplicate Reads"] == 0 and i["Ambiguous Reads"] == 0
        assert i["Exonic Reads"] + i["Intronic Reads"] + i["Intergenic Reads"] == i["Reads used for Intron/Exon counts"]
        assert i["Intragenic Reads"] == i["Exonic Reads"] + i["Intronic Reads"]
        assert 0 < i["Split Reads"] <= i["Exonic Reads"]
    keys = KA["legacy"]["metrics_keys"]
    assert keys.index("rRNA Reads") + 1 == keys.index("Split Reads") == keys.index("Total Bases") - 1
