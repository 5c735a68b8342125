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


def test_exon_outside_its_gene_row_is_refused():
    """An exon that sticks out of its gene's row: the reference's counts then depend on when the gene leaves its window
    (src/Metrics.cpp:106-112, "Gene encountered after computing coverage"); the static index cannot follow that, so the
    annotation is refused by the index builder (rsqc_set_annotation / the CLI: exit 11 with the message) instead of being
    counted differently."""
    from rnaseqc_amd.model import Annotation, Batch
    rows = [dict(contig="c", type="gene", start=100, end=900, strand="+", gene_id="G0"),
            dict(contig="c", type="exon", start=100, end=300, strand="+", gene_id="G0", exon_id="E0"),
            dict(contig="c", type="exon", start=800, end=1000, strand="+", gene_id="G0", exon_id="E1")]      # 100 bases beyond the gene row
    ann = Annotation.from_rows(["c"], rows)
    b = Batch.from_records([dict(qname="a", tid=0, pos=150, cigar=[(abi.CIG_M, 50)], flag=99)])
    with pytest.raises(RuntimeError) as e:
        hostemu.run(abi.default_params(), ann, b)
    assert "rc=-1" in str(e.value)
