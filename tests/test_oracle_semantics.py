"""The CPU oracle against expectations derived by hand from the reference source."""
import numpy as np
import pytest

from rnaseqc_amd import abi
from tests import cases


def test_quirk_case_counters(oracle_lib):
    ann, batch = cases.quirk_case()
    r = oracle_lib.run_oracle(abi.default_params(), ann, [batch])
    got = r.counter_dict()
    for k, v in cases.QUIRK_COUNTERS.items():
        assert got[k] == v, (k, got[k], v)
    assert list(r.gene_reads) == cases.QUIRK_GENE_READS
    assert list(r.gene_unique) == cases.QUIRK_GENE_UNIQUE
    assert list(r.gene_fragments) == cases.QUIRK_GENE_FRAGMENTS
    np.testing.assert_allclose(r.exon_reads, cases.QUIRK_EXON_READS, atol=1e-12)
    assert list(r.exon_hit) == [1, 1, 1, 1, 0, 1, 1, 1]
    assert r.read_length == cases.QUIRK_READ_LENGTH


def test_quirk_case_coverage(oracle_lib):
    ann, batch = cases.quirk_case()
    r = oracle_lib.run_oracle(abi.default_params(), ann, [batch])
    # GA: coding 1503; mask leaves GA_2[299..300] + GA_3[0..500]; ones = d1 100 + k1 88 + k1 8
    assert list(r.gene_cov_valid) == [1, 1, 0, 0, 1]
    m = 196 / 503
    assert r.gene_cov_mean[0] == pytest.approx(m, rel=1e-12)
    assert r.gene_cov_std[0] == pytest.approx(np.sqrt(m * (1 - m)), rel=1e-12)
    assert r.gene_cov_cv[0] == pytest.approx(np.sqrt(m * (1 - m)) / m, rel=1e-12)
    # GB: only GB_1 covered, which the 500-base mask removes -> mean 0, std 0, cv nan
    assert r.gene_cov_mean[1] == 0 and r.gene_cov_std[1] == 0 and np.isnan(r.gene_cov_cv[1])
    # GC: 2001 bases, mask leaves [500,1501): three adjacent reads cover [500,800)
    m = 300 / 1001
    assert r.gene_cov_mean[4] == pytest.approx(m, rel=1e-12)
    assert r.gene_cov_cv[4] == pytest.approx(np.sqrt(m * (1 - m)) / m, rel=1e-12)
    # exon CVs: only GA_3 and GC_1 have a finite CV
    assert list(r.exon_cv_valid) == [0, 0, 1, 0, 0, 0, 0, 1]
    m3 = 196 / 501
    assert r.exon_cv[2] == pytest.approx(np.sqrt(m3 * (1 - m3)) / m3, rel=1e-12)
    assert not r.bias_three.any() and not r.bias_five.any()


def test_quirk_case_batch_split_invariance(oracle_lib):
    ann, batch = cases.quirk_case()
    whole = oracle_lib.run_oracle(abi.default_params(), ann, [batch])
    parts = [batch.slice(0, 5), batch.slice(5, 6), batch.slice(6, 19), batch.slice(19, batch.n)]
    split = oracle_lib.run_oracle(abi.default_params(), ann, parts)
    assert (whole.counters == split.counters).all()
    assert (whole.gene_fragments == split.gene_fragments).all()
    np.testing.assert_array_equal(whole.exon_reads, split.exon_reads)
    np.testing.assert_array_equal(whole.gene_cov_mean, split.gene_cov_mean)


def test_quirky_median(oracle_lib):
    # src/Metrics.h:147-160: odd n -> mean of [mid],[mid+1]; even n -> [mid]
    assert oracle_lib.median([5.0]) == 5.0
    assert oracle_lib.median([1.0, 2.0]) == 1.0
    assert oracle_lib.median([1.0, 2.0, 4.0]) == 3.0
    assert oracle_lib.median([1.0, 2.0, 4.0, 8.0]) == 2.0
    assert oracle_lib.median([1.0, 2.0, 4.0, 8.0, 16.0]) == 6.0
    with pytest.raises(oracle_lib.OracleError):
        oracle_lib.median([])


def test_exclude_chimeric_and_tags(oracle_lib):
    ann, _ = cases.quirk_case()
    recs = cases.quirk_records()
    recs[5]["tags"] = [True]           # i1 carries the --tag
    recs[7]["ch"] = True               # n1 carries the chimeric tag (READ1)
    from rnaseqc_amd.model import Batch
    b = Batch.from_records(recs)
    r = oracle_lib.run_oracle(abi.default_params(n_filter_tags=1, exclude_chimeric=1), ann, [b])
    c = r.counter_dict()
    assert c["Filtered by tag: 0"] == 1
    assert c["Chimeric Fragments_tag"] == 1
    assert c["Chimeric Fragments_auto"] == 2
    # i1 filtered by tag, n1 (ch tag) and c1 (mate elsewhere) excluded as chimeric
    assert c["Reads used for Intron/Exon counts"] == 18 - 3
    assert c["Intronic Reads"] == 0
    assert c["Mapped Reads"] == 18     # the exclusions happen after the mapped counters
    assert c["Total Bases"] == 1810 - 200   # n1 and c1 leave before :317; i1 does not


def test_stranded(oracle_lib):
    ann, batch = cases.quirk_case()
    # --stranded RF: READ1 forward -> feature strand Reverse (Expression.cpp:119-125)
    r = oracle_lib.run_oracle(abi.default_params(stranded=abi.STRAND_REVERSE), ann, [batch])
    c = r.counter_dict()
    # p1/1 is forward READ1 -> looks only at '-' features: GA is '+', so it is intergenic now;
    # p1/2 is reverse READ2 -> target = !(rev) .. = '-' as well.  h1 (forward READ1) sees GH ('-').
    assert list(r.gene_reads) == [0, 1, 0, 1, 0]
    assert c["End 1 Sense"] == 0 and c["End 1 Antisense"] == 3   # t1 and k1 (last block touches GB_1) on GB, h1 on GH
    r2 = oracle_lib.run_oracle(abi.default_params(stranded=abi.STRAND_FORWARD), ann, [batch])
    assert list(r2.gene_reads) == [6, 0, 1, 0, 3]


def _collision_case():
    """Two read pairs whose NAMES differ but share their 64-bit rsqc_qname_hash (tests/golden/qname_hash_collision.json, found by
    tools/qname_collision.c), both counted to the one gene of a one-contig annotation; a third, ordinary pair beside them."""
    import json
    import os
    from rnaseqc_amd.model import Annotation, Batch
    fx = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "qname_hash_collision.json")))
    rows = [dict(contig="c", type="gene", start=100, end=5000, strand="+", gene_id="G0"),
            dict(contig="c", type="exon", start=100, end=5000, strand="+", gene_id="G0", exon_id="E0")]
    ann = Annotation.from_rows(["c"], rows)
    M = abi.CIG_M
    recs = []
    for k, name in enumerate((fx["a"], fx["b"], "ordinary")):
        p1, p2 = 200 + 300 * k, 400 + 300 * k
        recs.append(dict(qname=name, tid=0, pos=p1, cigar=[(M, 100)], flag=99, mapq=255, nm=0, mpos=p2, mtid=0))
        recs.append(dict(qname=name, tid=0, pos=p2, cigar=[(M, 100)], flag=147, mapq=255, nm=0, mpos=p1, mtid=0))
    recs.sort(key=lambda r: r["pos"])
    return fx, ann, Batch.from_records(recs)


def test_crafted_qname_hash_collision_fixture(oracle_lib):
    """The fixture is what it says (two different names, one 64-bit hash, different second hashes), and it shows what the name
    identity of the batch format means: the oracle counts THREE fragments from the names (the reference's std::set<std::string>,
    src/Expression.cpp:383-387), THREE from the 96-bit identity (qhash, qhash2) the library's ingest paths provide, and TWO from
    the 64-bit hash alone (a caller that leaves rsqc_batch.qhash2 NULL).  DESIGN.md 5."""
    fx, ann, batch = _collision_case()
    assert fx["a"] != fx["b"]
    assert abi.qname_hash(fx["a"].encode()) == abi.qname_hash(fx["b"].encode()) == int(fx["hash"], 16)
    assert abi.qname_hash2(fx["a"].encode()) != abi.qname_hash2(fx["b"].encode())
    p = abi.default_params()
    exact = oracle_lib.run_oracle(p, ann, [batch])
    assert int(exact.gene_reads[0]) == 6 and int(exact.gene_fragments[0]) == 3
    import copy
    hashed = copy.copy(batch)
    hashed.qname = None; hashed.qname_off = None
    by_hashes = oracle_lib.run_oracle(p, ann, [hashed])
    assert int(by_hashes.gene_reads[0]) == 6 and int(by_hashes.gene_fragments[0]) == 3
    narrow = copy.copy(hashed)
    narrow.qhash2 = None
    by_hash64 = oracle_lib.run_oracle(p, ann, [narrow])
    assert int(by_hash64.gene_reads[0]) == 6 and int(by_hash64.gene_fragments[0]) == 2


def _collision_pairing_case():
    """The colliding names of tests/golden/qname_hash_collision.json with their mates INTERLEAVED (a1 b1 a2 b2) inside one BED interval /
    one exon, plus an ordinary pair: the two QNAME-keyed maps beside the fragment tracker -- fragmentSizeMetrics
    (src/Expression.cpp:511-531) and the GC branch (:461-474) -- pair a1 with a2 and b1 with b2 when they compare names, and
    a1 with a2 only (b1 finds a's entry and leaves it, b2 then opens an entry that nobody closes) when `a` and `b` are one key."""
    import json
    import os
    from rnaseqc_amd.model import Annotation, Batch, Bed, Reference
    fx = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "qname_hash_collision.json")))
    rows = [dict(contig="c", type="gene", start=100, end=5000, strand="+", gene_id="G0"),
            dict(contig="c", type="exon", start=100, end=5000, strand="+", gene_id="G0", exon_id="E0")]
    ann = Annotation.from_rows(["c"], rows)
    M = abi.CIG_M
    def pair(name, p1, p2):
        size = p2 + 100 - p1
        return [dict(qname=name, tid=0, pos=p1, cigar=[(M, 100)], flag=99, mapq=255, nm=0, mpos=p2, mtid=0, isize=size),
                dict(qname=name, tid=0, pos=p2, cigar=[(M, 100)], flag=147, mapq=255, nm=0, mpos=p1, mtid=0, isize=-size)]
    recs = pair(fx["a"], 200, 420) + pair(fx["b"], 300, 560) + pair("ordinary", 1000, 1300)
    recs.sort(key=lambda r: r["pos"])
    bed = Bed.from_intervals([0], [99], [5000])
    seq = np.full(6000, ord("A"), np.uint8); seq[:350] = ord("G")       # G/C only in front: the fragments' GC shares tell the pairings apart
    ref = Reference(contig=[0], sequence=[seq])
    return ann, Batch.from_records(recs), bed, ref


def test_collision_in_the_pairing_maps(oracle_lib):
    """Fragment sizes and fragment GC content of the interleaved collision case: by names and by the 96-bit identity three
    fragments (sizes 320, 360, 400), by the 64-bit hash alone two.  The device keys every QNAME-keyed path on the 96 bits since
    round 5 (rsqc_k5.h); tests/test_gpu_parity.py runs the same case through it."""
    import copy
    ann, batch, bed, ref = _collision_pairing_case()
    p = abi.default_params(coverage_mask=0)
    by_name = oracle_lib.run_oracle(p, ann, [batch], bed=bed, reference=ref)
    assert sorted(int(x) for x in by_name.fragment_size) == [320, 360, 400] and int(by_name.fragment_count.sum()) == 3
    assert int(by_name.gc_bins[46]) == 1 and int(by_name.gc_bins[13]) == 1 and int(by_name.gc_bins[0]) == 1      # a: [200, 520), b: [300, 660), ordinary
    hashed = copy.copy(batch); hashed.qname = None; hashed.qname_off = None
    by96 = oracle_lib.run_oracle(p, ann, [hashed], bed=bed, reference=ref)
    np.testing.assert_array_equal(by96.fragment_size, by_name.fragment_size); np.testing.assert_array_equal(by96.fragment_count, by_name.fragment_count)
    np.testing.assert_array_equal(by96.gc_bins, by_name.gc_bins)
    narrow = copy.copy(hashed); narrow.qhash2 = None
    by64 = oracle_lib.run_oracle(p, ann, [narrow], bed=bed, reference=ref)
    assert int(by64.fragment_count.sum()) == 2
    assert int(by64.gc_bins[75]) == 1 and int(by64.gc_bins[0]) == 2          # b1 closes a1's entry: [200, 400); b2 closes a2's: [420, 660)
