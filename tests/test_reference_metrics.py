"""Pins the oracle's coverage / bias / statistics restatement against the REFERENCE's
own src/Metrics.cpp, compiled unmodified into oracle/_ref/libref_metrics.so."""
import numpy as np
import pytest

from rnaseqc_amd import abi
from rnaseqc_amd.model import Annotation, Batch

M = abi.CIG_M


@pytest.fixture(scope="module")
def ref(oracle_lib):
    if oracle_lib.ref_lib() is None:
        try:
            oracle_lib.build_ref()
        except Exception:
            pass
    if oracle_lib.ref_lib() is None:
        pytest.skip("oracle/_ref/libref_metrics.so not built (needs /root/reference)")
    return oracle_lib


def deep_coverage_case(seed, n_genes=18, mask=500):
    """Non-overlapping multi-exon genes with shaped deep coverage (0-600x) so that the
    bias gate (>=100 around the peak), the 5th-percentile trim (Q14) and both window
    medians are exercised, plus short genes that the mask swallows."""
    rng = np.random.default_rng(seed)
    rows, commits = [], []
    pos = 1000
    exon_row = 0
    gene_exon_off, exon_len, gene_strand = [0], [], []
    recs = []
    read_id = 0
    for g in range(n_genes):
        nex = int(rng.integers(1, 7))
        lens = rng.integers(30, 700, nex)
        if g % 7 == 0:
            lens = rng.integers(20, 60, nex)          # short gene: < 200 coding, no bias, mask eats it
        strand = "+-."[int(rng.integers(0, 3)) if g % 5 == 0 else int(rng.integers(0, 2))]
        gstart = pos
        ex = []
        for L in lens:
            ex.append((pos, pos + int(L) - 1))
            pos += int(L) + int(rng.integers(50, 400))
        gend = ex[-1][1]
        gid = "G%d" % g
        rows.append(dict(contig="c1", type="gene", start=gstart, end=gend, strand=strand, gene_id=gid))
        for k, (s, e) in enumerate(ex):
            rows.append(dict(contig="c1", type="exon", start=s, end=e, strand=strand, gene_id=gid, exon_id="%s_%d" % (gid, k)))
            exon_len.append(e - s + 1)
        gene_exon_off.append(gene_exon_off[-1] + nex)
        gene_strand.append({"+": 0, "-": 1, ".": 2}[strand])
        depth = [0, 3, 40, 150, 300, 450][int(rng.integers(0, 6))]
        coding = int(sum(lens))
        nreads = depth * coding // 50
        # 5'/3' skew: sample transcript position from a beta distribution
        a, b = [(1, 1), (2, 5), (5, 2), (0.7, 0.7)][int(rng.integers(0, 4))]
        tpos = (rng.beta(a, b, nreads) * coding).astype(np.int64)
        cum = np.concatenate([[0], np.cumsum(lens)])
        for t in tpos.tolist():
            k = int(np.searchsorted(cum, t, side="right") - 1)
            off = t - int(cum[k])
            L = int(min(50, lens[k] - off))
            if L <= 0:
                continue
            hq = rng.random() > 0.05
            recs.append(dict(qname="r%d" % read_id, tid=0, pos=ex[k][0] + off - 1, cigar=[(M, L)], flag=99,
                             mapq=255 if hq else 3, l_qseq=50))
            if hq:
                commits.append((exon_row + k, off, L, read_id))
            read_id += 1
        exon_row += nex
        pos += int(rng.integers(500, 3000))
    order = sorted(range(len(recs)), key=lambda i: recs[i]["pos"])
    recs = [recs[i] for i in order]
    ann = Annotation.from_rows(["c1"], rows)
    return ann, Batch.from_records(recs), gene_exon_off, exon_len, gene_strand, commits


@pytest.mark.parametrize("seed,mask,offset,window", [(11, 500, 0, 100), (12, 500, 0, 100), (13, 100, 0, 100),
                                                       (14, 500, 20, 50), (15, 0, 0, 100), (16, 500, 150, 100),
                                                       (17, 500, 60, 100)])
def test_coverage_bias_vs_reference_metrics_cpp(ref, seed, mask, offset, window):
    ann, batch, geo, elen, gstrand, commits = deep_coverage_case(seed)
    p = abi.default_params(coverage_mask=mask, bias_offset=offset, bias_window=window)
    c = np.array(commits, dtype=np.int64).reshape(-1, 4)
    try:
        got = ref.run_oracle(p, ann, [batch])
    except ref.OracleError as e:
        # e.g. --offset 150 with a trimmed transcript of 200..249 bases: the right window is
        # empty and computeMedian throws std::range_error (exit code 2 in the reference)
        assert e.code == abi.ERR_EMPTY_MEDIAN
        got = None
    # the reference commits in file order; sort commits by read position like the batch
    exon_start = ann.exon_row_start.astype(np.int64)   # rows are already in generation order here
    key = exon_start[c[:, 0]] + c[:, 1]
    c = c[np.argsort(key, kind="stable")]
    if got is None:
        with pytest.raises(ref.OracleError) as ei:
            ref.ref_coverage_run(geo, elen, gstrand, c[:, 0], c[:, 1], c[:, 2], c[:, 3], mask=mask,
                                 bias_offset=offset, bias_window=window)
        assert ei.value.code == abi.ERR_EMPTY_MEDIAN
        return
    want = ref.ref_coverage_run(geo, elen, gstrand, c[:, 0], c[:, 1], c[:, 2], c[:, 3], mask=mask,
                                bias_offset=offset, bias_window=window)
    np.testing.assert_array_equal(got.gene_cov_valid, want["gene_valid"])
    v = want["gene_valid"].astype(bool)
    # bit-exact: the oracle performs the same operations in the same order
    np.testing.assert_array_equal(got.gene_cov_mean[v], want["gene_mean"][v])
    np.testing.assert_array_equal(got.gene_cov_std[v], want["gene_std"][v])
    np.testing.assert_array_equal(got.gene_cov_cv[v], want["gene_cv"][v])
    # exon ids == exon rows in this construction
    np.testing.assert_array_equal(got.exon_cv_valid, want["exon_cv_valid"])
    ev = want["exon_cv_valid"].astype(bool)
    np.testing.assert_array_equal(got.exon_cv[ev], want["exon_cv"][ev])
    tot = (got.bias_three + got.bias_five).astype(np.float64)
    ratio = np.where(tot > 0, got.bias_three / np.where(tot > 0, tot, 1), -1.0)
    np.testing.assert_array_equal(ratio, want["bias_ratio"])
    assert int((tot > 0).sum()) == want["counted_genes"]
    if mask == 500 and offset == 0:
        assert want["counted_genes"] >= 3      # the bias path really ran


def test_median_and_statistics_vs_reference(ref):
    rng = np.random.default_rng(5)
    for n in [1, 2, 3, 4, 5, 6, 7, 10, 11, 100, 101]:
        d = np.sort(rng.random(n) * 10)
        assert ref.median(d) == ref.ref_median(d)
        assert ref.statistics(d.copy()) == ref.ref_statistics(d)
    with pytest.raises(ref.OracleError):
        ref.ref_median([])


def test_collector_vs_reference_class(ref):
    """Collector::{add, queryGene, collect} (src/Metrics.h:43-59, src/Metrics.cpp:48-81) -- the per-read staging of exon
    fractions behind exonCounts -- is driven with exactly the calls the oracle's restatement of exonAlignmentMetrics makes
    (recorded call by call) and must leave bit-identical sums, the same set of map entries (the `exon_reads.gct` header
    count, src/RNASeQC.cpp:513) and the same queryGene answers; incl. zero-length blocks, whose fraction 0 (or 0/0) the
    `coverage > 0` test drops (src/Metrics.cpp:51)."""
    import ctypes as C
    from rnaseqc_amd import synth
    contigs = [("chrA", 600_000, 80), ("chrB", 300_000, 30)]
    ann = synth.make_annotation(seed=51, contigs=contigs)
    base = synth.make_reads(ann, 12000, seed=52, dup_frac=0.1, contig_lengths=np.array([c[1] for c in contigs]))
    # hand-made additions on the first contig: zero-length aligned blocks inside exons
    rows = np.flatnonzero((ann.exon_row_contig == 0) & ((ann.exon_row_end - ann.exon_row_start) > 120))
    extra = []
    for k, row in enumerate(rows[:6]):
        s = int(ann.exon_row_start[row])
        extra.append(dict(tid=0, pos=s + 4, mpos=s + 4, flag=99, mapq=255, qname="zero%d" % k, nm=0, l_qseq=50,
                          cigar=[(M, 0)] if k % 2 == 0 else [(M, 50), (abi.CIG_I, 3), (M, 0), (M, 20)]))
    zb = Batch.from_records(extra)
    base.qname = base.qname_off = None
    zb.qname = zb.qname_off = None
    recs = Batch.concat([base, zb]).coordinate_sorted()
    assert recs.n == base.n + 6
    o = ref.Oracle(abi.default_params())
    o.set_annotation(ann)
    o._check(o._l.oracle_enable_collector_trace(o._h))
    o.submit(recs)
    want = o.finalize()

    class T(C.Structure):
        _fields_ = [("n", C.c_uint64), ("kind", C.c_void_p), ("read", C.c_void_p), ("gene", C.c_void_p), ("exon", C.c_void_p),
                    ("frac", C.c_void_p), ("query", C.c_void_p)]
    t = T()
    o._check(o._l.oracle_get_collector_trace(o._h, C.byref(t)))
    n = int(t.n)
    kind = abi._view(t.kind, n, np.uint8); read = abi._view(t.read, n, np.uint32); gene = abi._view(t.gene, n, np.uint32)
    exon = abi._view(t.exon, n, np.uint32); frac = abi._view(t.frac, n, np.float64); query = abi._view(t.query, n, np.uint8)
    o.close()
    assert (kind == 0).sum() > 5000 and (kind == 2).sum() > 3000
    adds = kind == 0
    assert (adds & ~(frac > 0)).sum() >= 3                      # the zero-length blocks: fraction 0 or NaN, dropped by add()
    E = ann.n_exons
    val = np.zeros(E); entry = np.zeros(E, np.uint8); q = np.zeros(n, np.uint8); mx = C.c_double()
    rl = ref.ref_lib()
    rl.ref_collector_replay.argtypes = [C.c_uint64] + [C.c_void_p] * 5 + [C.c_uint32] + [C.c_void_p] * 3 + [C.POINTER(C.c_double)]
    assert rl.ref_collector_replay(n, abi.ptr(kind), abi.ptr(read), abi.ptr(gene), abi.ptr(exon), abi.ptr(frac), E,
                                   abi.ptr(val), abi.ptr(entry), abi.ptr(q), C.byref(mx)) == 0
    np.testing.assert_array_equal(val, want.exon_reads)         # same additions in the same order: bit-identical
    np.testing.assert_array_equal(entry, want.exon_hit)
    np.testing.assert_array_equal(q[kind == 1], query[kind == 1])
    assert 0.99 < mx.value <= 2.0 + 1e-9                        # Collector::sum(): a read counted to two overlapping genes adds up twice
    # geneCounts only moves when queryGene says so (src/Expression.cpp:380-382)
    assert int(want.gene_reads.sum()) == int(query[kind == 1].sum())


# ---- src/BED.cpp (extractBED), compiled unmodified into the same library (oracle/ref_bed_harness.cpp) -------------------
def _bed_read(fn, path, cap=4096):
    import ctypes as C
    fn.restype = C.c_longlong
    fn.argtypes = [C.c_char_p, C.c_longlong, C.c_void_p, C.c_void_p, C.c_void_p, C.c_char_p, C.c_longlong, C.c_char_p, C.c_longlong]
    ci = np.zeros(cap, np.int32); st = np.zeros(cap, np.int64); en = np.zeros(cap, np.int64)
    names = C.create_string_buffer(1 << 16); err = C.create_string_buffer(512)
    n = fn(path.encode(), cap, ci.ctypes.data, st.ctypes.data, en.ctypes.data, names, len(names), err, len(err))
    threw = n <= -2
    k = -2 - n if threw else n
    return dict(status="open" if n == -1 else ("threw" if threw else "ok"), rows=[] if n == -1 else list(zip(ci[:k].tolist(), st[:k].tolist(), en[:k].tolist())),
                names=names.value.decode("latin-1").split("\n")[:-1] if n != -1 else [], err=err.value.decode("latin-1") if threw else "")


HOSTILE_BEDS = {
    "plain": "chr1\t100\t200\nchr1\t300\t400\nchr2\t5\t50\n",
    "no_trailing_newline": "chr1\t100\t200\nchr2 7 9",
    "comments_and_mixed_space": "#track name=x\nchr1   10 \t 20\textra\tfields\t+\n# not a comment: leading space\n#c\nchrM\t0\t1\n",
    "short_line_repeats_previous_token": "chr1\t100\nchr2\t7\t9\n",
    "one_token_line_throws": "chr1\t1\t2\nchrX\nchr1\t5\t6\n",
    "empty_line_throws": "chr1\t1\t2\n\nchr1\t5\t6\n",
    "whitespace_only_line_throws": "chr1\t1\t2\n   \t \nchr1\t5\t6\n",
    "trailing_junk_and_signs": "chr1\t100abc\t200.7\nchr1\t+5\t-5\nchr1\t-1\t0x10\n",
    "not_a_number": "chr1\tabc\t5\n",
    "beyond_64_bits": "chr1\t1\t99999999999999999999999\nchr1\t2\t3\n",
    "max_u64_wraps_to_zero": "chr1\t18446744073709551615\t18446744073709551614\n",
    "comment_tail_only": "chr1\t1\t2\n#end\n#really\n",
    "crlf": "chr1\t1\t2\r\nchr2\t3\t4\r\n",
    "vertical_tab_and_formfeed": "chr1\v1\f2\n",
    "name_reuse_and_order": "b\t1\t2\na\t3\t4\nb\t5\t6\nc\t7\t8\na\t9\t10\n",
    "hash_inside_line": "chr#1\t1\t2\n chr1\t#3\t4\n",
    "only_comments": "#a\n#b\n",
    "empty_file": "",
}


@pytest.mark.parametrize("name", sorted(HOSTILE_BEDS))
def test_bed_loader_vs_reference_extractBED(ref, name, tmp_path):
    """The product's BED loader (rnaseqc_amd/csrc/host/gtf.cpp, Annotation::load_bed) against the reference's own
    extractBED (src/BED.cpp:18-46, compiled unmodified; the loop of src/RNASeQC.cpp:185 around it) on well-formed and
    hostile files: same rows (+1/+1 coordinates), same contig names in the same first-sight order, and an exception with the
    same message at the same row."""
    import ctypes as C
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    subprocess.check_call(["make", "-s", "-C", os.path.join(root, "rnaseqc_amd", "csrc"), "../lib/librsqc_host.so"])
    host = C.CDLL(os.path.join(root, "rnaseqc_amd", "lib", "librsqc_host.so"))
    rl = ref.ref_lib()
    if not hasattr(rl, "ref_bed_read"):
        pytest.skip("oracle/_ref/libref_metrics.so predates the BED harness")
    path = str(tmp_path / (name + ".bed"))
    with open(path, "wb") as f:
        f.write(HOSTILE_BEDS[name].encode("latin-1"))
    want = _bed_read(rl.ref_bed_read, path)
    got = _bed_read(host.host_bed_read, path)
    assert got == want, (name, got, want)
    if name in ("one_token_line_throws", "empty_line_throws", "whitespace_only_line_throws", "not_a_number", "beyond_64_bits"):
        assert want["status"] == "threw" and want["err"].startswith("Encountered an unknown error while parsing the BED: ")
    if name == "short_line_repeats_previous_token":
        assert want["rows"][0] == (0, 101, 101)
    if name == "trailing_junk_and_signs":
        assert want["rows"] == [(0, 101, 201), (0, 6, -4), (0, 0, 1)]


def test_bed_missing_file_both_report_open_failure(ref, tmp_path):
    import ctypes as C
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    host = C.CDLL(os.path.join(root, "rnaseqc_amd", "lib", "librsqc_host.so"))
    rl = ref.ref_lib()
    if not hasattr(rl, "ref_bed_read"):
        pytest.skip("oracle/_ref/libref_metrics.so predates the BED harness")
    missing = str(tmp_path / "nope.bed")
    assert _bed_read(rl.ref_bed_read, missing)["status"] == "open" and _bed_read(host.host_bed_read, missing)["status"] == "open"
