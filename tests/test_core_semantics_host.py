"""The product's per-record core (rsqc_read.h + rsqc_index.h), compiled for the host by the
test harness, against the oracle.  This is the same source the HIP kernels compile; the
oracle walks a trimmed window, the core queries a static index."""
import numpy as np
import pytest

from rnaseqc_amd import abi, synth
from tests import cases, hostemu


def _compare(o, r):
    names = abi.COUNTER_NAMES
    for i, n in enumerate(names):
        assert int(o.counters[i]) == int(r.counters[i]), n
    np.testing.assert_array_equal(o.gene_reads, r.gene_reads)
    np.testing.assert_array_equal(o.gene_unique, r.gene_unique)
    np.testing.assert_array_equal(o.gene_fragments, r.gene_fragments)
    np.testing.assert_allclose(o.exon_reads, r.exon_reads, rtol=0, atol=1e-9)
    assert o.read_length == r.read_length


def test_quirk_case(oracle_lib):
    ann, batch = cases.quirk_case()
    p = abi.default_params()
    _compare(hostemu.run(p, ann, batch), oracle_lib.run_oracle(p, ann, [batch]))


@pytest.mark.parametrize("kw", [dict(), dict(stranded=abi.STRAND_REVERSE), dict(stranded=abi.STRAND_FORWARD, unpaired=1),
                                dict(unpaired=1, mapq_threshold=3, n_filter_tags=1, exclude_chimeric=1),
                                dict(base_mismatch=1, chimeric_distance=100)])
def test_synthetic_vs_oracle(oracle_lib, kw):
    ann = synth.make_annotation(seed=3, contigs=[("chrA", 3_000_000, 300), ("chrB", 1_500_000, 150), ("chrC", 400_000, 0)])
    batch = synth.make_reads(ann, 20000, seed=4, dup_frac=0.1, chimeric_tag_frac=0.01, filter_tag_frac=0.02,
                             contig_lengths=np.array([3_000_000, 1_500_000, 400_000]))
    p = abi.default_params(**kw)
    o = hostemu.run(p, ann, batch)
    r = oracle_lib.run_oracle(p, ann, [batch])
    _compare(o, r)
    assert r.gene_reads.sum() > 1000


def test_many_overlapping_genes_take_the_slow_path(oracle_lib):
    # 12 genes stacked on the same exon: more than FAST_SET genes per block
    rows = []
    for g in range(12):
        rows.append(dict(contig="c", type="gene", start=100, end=2000, strand="+-"[g % 2], gene_id="G%d" % g))
        rows.append(dict(contig="c", type="exon", start=100 + g, end=1500 + g, strand="+-"[g % 2], gene_id="G%d" % g,
                         exon_id="E%d" % g))
    from rnaseqc_amd.model import Annotation, Batch
    ann = Annotation.from_rows(["c"], rows)
    recs = [dict(qname="a%d" % i, tid=0, pos=200 + i, cigar=[(abi.CIG_M, 50), (abi.CIG_N, 100), (abi.CIG_M, 50)], flag=99)
            for i in range(40)]
    b = Batch.from_records(recs)
    p = abi.default_params()
    o = hostemu.run(p, ann, b)
    assert o.n_overflow == 40
    _compare(o, oracle_lib.run_oracle(p, ann, [b]))
    assert list(o.gene_reads) == [40] * 12


def test_hostile_records_under_sanitizers():
    """The per-record core on records whose fields are anywhere inside the format's ranges (positions next to 2^31,
    operations of 2^28 - 1 bases, thousands of N operations, any flag / mate / NM / l_seq): every access stays inside the
    annotation's tables and the coverage array.  tests/hostemu/core_fuzz.py with hostemu.cpp built under the address and
    undefined-behaviour sanitizers (32-bit wrap-around is left alone: it is what the device does with such input)."""
    import os
    import subprocess
    import sys
    asan = subprocess.check_output(["g++", "-print-file-name=libasan.so"], text=True).strip()
    if not os.path.isabs(asan):
        pytest.skip("no libasan in this toolchain")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-m", "tests.hostemu.core_fuzz", "3", "7"], cwd=root, capture_output=True, text=True, timeout=900,
                       env=dict(os.environ, LD_PRELOAD=asan, ASAN_OPTIONS="detect_leaks=0"))
    assert r.returncode == 0 and "core_fuzz: 9 runs" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]


@pytest.mark.parametrize("seed", range(6))
def test_interval_index_equals_row_index(oracle_lib, seed):
    """The two feature stages of the core -- elementary intervals + rank bit vector (exon_metrics_ei, what the per-record
    kernel runs) and the start-sorted row table (exon_metrics_fast, still the general code's index) -- on hostile
    annotations: every output incl. the per-base coverage difference array is identical, and equals the oracle's."""
    from tests.test_legacy_rules import hostile_case, stacked_case
    for ann, batch in (hostile_case(100 + seed), stacked_case(200 + seed)):
        for kw in (dict(), dict(stranded=abi.STRAND_FORWARD), dict(stranded=abi.STRAND_REVERSE, unpaired=1)):
            p = abi.default_params(mapq_threshold=4, **kw)
            a0 = hostemu.run(p, ann, batch, mode=0, want_cov=True)
            a1 = hostemu.run(p, ann, batch, mode=1, want_cov=True)
            _compare(a1, a0)
            np.testing.assert_array_equal(a1.cov, a0.cov)
            _compare(a1, oracle_lib.run_oracle(p, ann, [batch]))


def _outside_rows():
    return [dict(contig="c", type="gene", start=100, end=900, strand="+", gene_id="G0"),
            dict(contig="c", type="exon", start=100, end=300, strand="+", gene_id="G0", exon_id="E0"),
            dict(contig="c", type="exon", start=800, end=1000, strand="+", gene_id="G0", exon_id="E1"),     # 100 bases beyond the gene row
            dict(contig="c", type="gene", start=2000, end=3000, strand="-", gene_id="G1"),
            dict(contig="c", type="exon", start=1900, end=2400, strand="-", gene_id="G1", exon_id="E2"),    # starts before its gene row
            dict(contig="c", type="exon", start=2600, end=3000, strand="-", gene_id="G1", exon_id="E3")]


def _pair(name, p1, p2, n=50):
    M = abi.CIG_M
    return [dict(qname=name, tid=0, pos=p1, cigar=[(M, n)], flag=99, mapq=255, nm=0, mpos=p2, mtid=0, isize=p2 + n - p1),
            dict(qname=name, tid=0, pos=p2, cigar=[(M, n)], flag=147, mapq=255, nm=0, mpos=p1, mtid=0, isize=-(p2 + n - p1))]


def test_exon_outside_its_gene_row_is_accepted(oracle_lib):
    """An exon that sticks out of its gene's row (rounds 3-4 refused such an annotation; the reference runs it).  As long as no
    record that is counted to the gene STARTS behind the gene row's end -- i.e. the reference has not retired the gene yet
    (src/Expression.cpp:84-93) -- the static index gives the reference's streamed result in full: counts, fragments, coverage.
    Records inside the part of E1 that lies beyond the row (they start at or before the row's end), and records in the part of E2
    in front of its gene row."""
    from rnaseqc_amd.model import Annotation, Batch
    ann = Annotation.from_rows(["c"], _outside_rows())
    recs = _pair("a", 150, 820) + _pair("b", 810, 880, n=100) + _pair("c", 1905, 2650) + _pair("d", 1950, 2300) + _pair("e", 120, 200)
    recs.sort(key=lambda r: r["pos"])
    b = Batch.from_records(recs)
    for kw in (dict(), dict(stranded=abi.STRAND_FORWARD)):
        p = abi.default_params(coverage_mask=0, **kw)
        want = oracle_lib.run_oracle(p, ann, [b])
        a0 = hostemu.run(p, ann, b, mode=0, want_cov=True)
        a1 = hostemu.run(p, ann, b, mode=1, want_cov=True)
        _compare(a1, a0); _compare(a1, want)
        k = hostemu.run_k1(p, ann, b, grid=1, want_cov=True)
        _compare(k, want); np.testing.assert_array_equal(k.cov, a1.cov)
    assert int(want.gene_reads.sum()) > 0


def test_records_behind_a_retired_gene_row(oracle_lib):
    """... and where the two differ, stated: a record that STARTS behind the end of G0's row (901 > 900) inside E1.  The reference has
    retired G0 by then: it still counts the record to the gene and the exon (maps keyed by id), counts its name into a fresh
    fragment set and ignores its coverage with a warning (src/Metrics.cpp:108-112).  The static index counts the same reads and
    exon fractions and every classification counter; its fragment count is the number of distinct names over the whole file."""
    from rnaseqc_amd.model import Annotation, Batch
    ann = Annotation.from_rows(["c"], _outside_rows())
    recs = _pair("a", 810, 905, n=60) + _pair("late", 910, 930, n=40)
    recs.sort(key=lambda r: r["pos"])
    b = Batch.from_records(recs)
    p = abi.default_params(coverage_mask=0)
    want = oracle_lib.run_oracle(p, ann, [b])
    got = hostemu.run(p, ann, b, mode=1)
    for i, n in enumerate(abi.COUNTER_NAMES):
        assert int(got.counters[i]) == int(want.counters[i]), n
    np.testing.assert_array_equal(got.gene_reads, want.gene_reads)
    np.testing.assert_array_equal(got.gene_unique, want.gene_unique)
    np.testing.assert_allclose(got.exon_reads, want.exon_reads, rtol=0, atol=1e-9)
    assert int(got.gene_reads[0]) == 4 and int(got.gene_fragments[0]) == 2          # two names
    assert int(want.gene_fragments[0]) == 3                                         # the reference: "a" once before the gene retired, once after
