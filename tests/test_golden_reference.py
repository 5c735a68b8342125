"""Known-answer tests against the golden OUTPUTS the reference ships under test_data/*.output/
(its inputs are not in the tree).  The figures checked here were copied out of those files into
tests/golden/reference_known_answers.json by tests/golden/make_known_answers.py."""
import json
import os

import numpy as np
import pytest

from tests import report_ref
from tests.test_host_cli_pieces import host  # noqa: F401  (fixture: the C++ host library)

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


def test_gct_writers_reproduce_chr1_golden(host, tmp_path):
    """The reference's chr1 golden GCT files, re-emitted BYTE FOR BYTE by our report writer from the values they hold:
    gene_reads / gene_fragments (cast to long, src/RNASeQC.cpp:441-442), exon_reads (std::fixed 6 decimals, header
    count = exons with a map entry while every exon gets a row, :510-521), the `#1.2` / `rows\\t1` / column header
    lines (:427-436), and "Genes Detected" = genes with uniqueGeneCounts >= 5 (:461; the golden run has no duplicate
    reads, so unique == reads).  The annotation is rebuilt from the ids in the tables (coordinates are irrelevant here)."""
    import ctypes as C
    import gzip
    import json
    from rnaseqc_amd import abi
    from tests.test_host_cli_pieces import load_annotation, _results_struct, read_table
    T = json.load(gzip.open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "chr1_gct_tables.json.gz"), "rt"))
    gids, gdesc = T["gene_reads"]["id"], T["gene_reads"]["desc"]
    eids = T["exon_reads"]["id"]
    exons_of = {}
    for e in eids:
        exons_of.setdefault(e.rsplit("_", 1)[0], []).append(e)
    gtf = str(tmp_path / "g.gtf")
    with open(gtf, "w") as f:
        pos = 1000
        for g, nm in zip(gids, gdesc):
            ex = exons_of.get(g, [])
            span = 300 * max(len(ex), 1)
            attr = 'gene_id "%s"; transcript_id "%s"; gene_name "%s"; transcript_type "protein_coding";' % (g, g, nm)
            f.write("chr1\tx\tgene\t%d\t%d\t.\t+\t.\t%s\n" % (pos, pos + span, attr))
            for k, e in enumerate(ex):
                f.write("chr1\tx\texon\t%d\t%d\t.\t+\t.\t%s exon_id \"%s\";\n" % (pos + 300 * k, pos + 300 * k + 99, attr, e))
            pos += span + 500
    h, err = load_annotation(host, gtf, ["chr1"])
    assert err == 0
    G, E = len(gids), len(eids)

    class R:            # the fields _results_struct reads
        pass
    r = R()
    r.gene_reads = np.array([int(v) for v in T["gene_reads"]["value"]], np.uint64)
    r.gene_unique = r.gene_reads.copy()
    r.gene_fragments = np.array([int(v) for v in T["gene_fragments"]["value"]], np.uint64)
    r.exon_reads = np.array([float(v) for v in T["exon_reads"]["value"]], np.float64)
    r.exon_hit = (r.exon_reads > 0).astype(np.uint8)
    m = KA["chr1"]["metrics"]
    r.counters = np.zeros(abi.N_COUNTERS, np.uint64)
    for k, name in enumerate(abi.COUNTER_NAMES):
        if name in m and m[name].isdigit():
            r.counters[k] = int(m[name])
    r.read_length = int(m["Read Length"])
    r.gene_cov_mean = np.zeros(G); r.gene_cov_std = np.zeros(G); r.gene_cov_cv = np.zeros(G); r.gene_cov_valid = np.zeros(G, np.uint8)
    r.gene_cov_valid[0] = 1                                        # (one gene with coverage: the median lines need a non-empty list)
    r.exon_cv = np.zeros(E); r.exon_cv_valid = np.zeros(E, np.uint8)
    r.bias_three = np.zeros(G, np.uint64); r.bias_five = np.zeros(G, np.uint64)
    r.fragment_size = np.zeros(0, np.int64); r.fragment_count = np.zeros(0, np.uint64); r.fragment_samples_remaining = 0
    r.have_reference = 0
    rs, keep = _results_struct(r)
    out = str(tmp_path / "out"); os.makedirs(out)
    visit = (C.c_int * 1)(0)
    assert host.host_write_reports(h, C.byref(rs), out.encode(), b"chr1.bam", 0, 0, 0, 5, None, 0, visit, 1) == 0
    for f in ("gene_reads", "gene_fragments", "exon_reads"):
        t = T[f]
        want = "\n".join(t["header"]) + "\n" + "".join("%s\t%s\t%s\n" % x for x in zip(t["id"], t["desc"], t["value"]))
        assert open(os.path.join(out, "chr1.bam.%s.gct" % f)).read() == want, f
    assert int(T["exon_reads"]["header"][1].split("\t")[0]) == int(r.exon_hit.sum()) < E
    ours = dict(read_table(os.path.join(out, "chr1.bam.metrics.tsv")))
    assert ours["Genes Detected"] == m["Genes Detected"] == str(int((r.gene_reads >= 5).sum()))
    # TPM needs the coding lengths of the missing GTF; what the golden table does pin: zero exactly where the count is zero, sum 1e6
    tpm = np.array([float(v) for v in T["gene_tpm"]["value"]])
    assert ((tpm > 0) == (r.gene_reads > 0)).all() and abs(tpm.sum() - 1e6) < 1e-3
    t_ours = read_table(os.path.join(out, "chr1.bam.gene_tpm.gct"), 3)
    assert abs(sum(float(x[2]) for x in t_ours) - 1e6) < 1e-2 and all((float(x[2]) > 0) == (c > 0) for x, c in zip(t_ours, r.gene_reads))
    host.host_annotation_free(h)
