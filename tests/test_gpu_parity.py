"""-m gpu: the HIP hot path, called through the C ABI, against the oracle."""
import numpy as np
import pytest

from rnaseqc_amd import abi, engine, synth
from rnaseqc_amd.model import Annotation, Batch
from tests import cases
from tests.compare import assert_results_match

pytestmark = pytest.mark.gpu

SMALL_CONTIGS = [("chrA", 3_000_000, 300), ("chrB", 1_500_000, 150), ("chrC", 400_000, 0)]
SMALL_LENGTHS = np.array([3_000_000, 1_500_000, 400_000])


def small_inputs(n_pairs=20000, seed=4, **kw):
    ann = synth.make_annotation(seed=3, contigs=SMALL_CONTIGS)
    batch = synth.make_reads(ann, n_pairs, seed=seed, contig_lengths=SMALL_LENGTHS, **kw)
    return ann, batch


def test_quirk_case_hand_derived(oracle_lib):
    ann, batch = cases.quirk_case()
    p = abi.default_params()
    got = engine.run_engine(p, ann, [batch])
    c = got.counter_dict()
    for k, v in cases.QUIRK_COUNTERS.items():
        assert c[k] == v, (k, c[k], v)
    assert list(got.gene_reads) == cases.QUIRK_GENE_READS
    assert list(got.gene_unique) == cases.QUIRK_GENE_UNIQUE
    assert list(got.gene_fragments) == cases.QUIRK_GENE_FRAGMENTS
    np.testing.assert_allclose(got.exon_reads, cases.QUIRK_EXON_READS, atol=1e-12)
    assert got.read_length == cases.QUIRK_READ_LENGTH
    assert_results_match(got, oracle_lib.run_oracle(p, ann, [batch]))


@pytest.mark.parametrize("kw", [dict(), dict(stranded=abi.STRAND_REVERSE), dict(stranded=abi.STRAND_FORWARD, unpaired=1),
                                dict(unpaired=1, mapq_threshold=3, n_filter_tags=1, exclude_chimeric=1),
                                dict(base_mismatch=1, chimeric_distance=100), dict(coverage_mask=100),
                                dict(coverage_mask=0, bias_window=50, bias_offset=10)])
def test_synthetic_vs_oracle(oracle_lib, kw):
    ann, batch = small_inputs(dup_frac=0.1, chimeric_tag_frac=0.01, filter_tag_frac=0.02)
    p = abi.default_params(**kw)
    assert_results_match(engine.run_engine(p, ann, [batch]), oracle_lib.run_oracle(p, ann, [batch]))


@pytest.mark.parametrize("kw", [dict(), dict(bias_window=300, bias_offset=25, bias_gene_length=700),
                                dict(bias_window=700, bias_offset=5, bias_gene_length=1500, coverage_mask=50)])
def test_deep_coverage_bias_path(oracle_lib, kw):
    # few genes, most reads on them: depth in the hundreds so the bias gate (>=100) opens; windows beyond 128 bases use
    # the wide-window instantiation of the coverage kernel
    ann = synth.make_annotation(seed=8, contigs=[("chrA", 400_000, 40)])
    batch = synth.make_reads(ann, 150000, seed=9, frac=(0.97, 0.01, 0.01, 0.01), expr_sigma=1.0,
                             contig_lengths=np.array([400_000]))
    p = abi.default_params(**kw)
    want = oracle_lib.run_oracle(p, ann, [batch])
    assert int(((want.bias_three + want.bias_five) > 0).sum()) >= (5 if not kw else 1)
    assert_results_match(engine.run_engine(p, ann, [batch]), want)


def test_gene_with_more_exons_than_lanes(oracle_lib):
    """150 exons in one gene: the coverage kernel hands a wave its exon rows 64 at a time (tests/cases.py)."""
    ann, batch = cases.many_exon_case()
    p = abi.default_params(unpaired=1, coverage_mask=0)
    want = oracle_lib.run_oracle(p, ann, [batch])
    assert want.exon_cv_valid.sum() >= 140
    assert_results_match(engine.run_engine(p, ann, [batch]), want)


@pytest.mark.parametrize("n_exons", [5, 10])
def test_depth_beyond_16_bits_on_a_long_gene(oracle_lib, n_exons):
    """The longest genes keep 16-bit depths in LDS; a base covered >= 65 536 times must send the gene through the
    in-memory path with identical results (two long genes here: one deep, one shallow).  5 exons x 4 kb = 20 kb of coding
    length: the 64 KB instantiation of the 1024-thread class; 10 exons = 40 kb: the 146 KB one."""
    rows = []
    for gid, base in (("deep", 10_000), ("shallow", 200_000)):
        rows.append(dict(contig="c", type="gene", start=base, end=base + 10_000 * n_exons + 10_000, strand="+", gene_id=gid, gene_name=gid))
        for k in range(n_exons):
            rows.append(dict(contig="c", type="exon", start=base + k * 10_000, end=base + k * 10_000 + 3_999, strand="+",
                             gene_id=gid, exon_id="%s_e%d" % (gid, k)))
    from rnaseqc_amd.model import Annotation, Batch
    ann = Annotation.from_rows(["c"], rows)
    n_deep = 70_000
    n = n_deep + 3_000
    pos = np.concatenate([np.full(n_deep, 10_000 + 10_100, np.int32) - 1,          # all on the same 100 bases of exon 1
                          (200_000 + np.sort(np.random.default_rng(5).integers(0, 3_900, 3_000))).astype(np.int32) - 1])
    qh = abi.qname_hash_bytes(np.frombuffer(b"".join(b"%015d" % i for i in range(n)), np.uint8).reshape(n, 15))
    batch = Batch(pos=pos, mpos=pos.copy(), isize=np.zeros(n, np.int32), qhash=qh, cigar_off=np.arange(n, dtype=np.uint32),
                  flag=np.zeros(n, np.uint16), l_qseq=np.full(n, 100, np.uint16), mapq=np.full(n, 255, np.uint8),
                  nm=np.zeros(n, np.uint8), tagbits=np.full(n, abi.TB_HAS_NM | abi.TB_MTID_SAME, np.uint8),
                  n_cigar=np.ones(n, np.uint8), cigar=np.full(n, (100 << 4) | abi.CIG_M, np.uint32),
                  seg_tid=np.array([0], np.int32), seg_start=np.array([0, n], np.uint64),
                  wide_index=np.zeros(0, np.uint64), wide_nm=np.zeros(0, np.int32), wide_l_qseq=np.zeros(0, np.int32),
                  wide_n_cigar=np.zeros(0, np.uint32))
    p = abi.default_params(unpaired=1)
    want = oracle_lib.run_oracle(p, ann, [batch])
    got = engine.run_engine(p, ann, [batch])
    assert_results_match(got, want)
    assert got.gene_reads[0] == n_deep and got.gene_cov_valid[0] == 1 and got.gene_cov_mean[0] > 100


def test_batch_split_invariance(oracle_lib):
    ann, batch = small_inputs(n_pairs=8000)
    p = abi.default_params()
    cuts = [0, 1, 257, 4096, 9000, batch.n]
    parts = [batch.slice(a, b) for a, b in zip(cuts[:-1], cuts[1:])]
    whole = engine.run_engine(p, ann, [batch])
    split = engine.run_engine(p, ann, parts)
    assert_results_match(split, whole)
    assert_results_match(split, oracle_lib.run_oracle(p, ann, parts))


def test_overlapping_genes_slow_path(oracle_lib):
    rows = []
    for g in range(12):
        rows.append(dict(contig="c", type="gene", start=100, end=2000, strand="+-"[g % 2], gene_id="G%d" % g))
        rows.append(dict(contig="c", type="exon", start=100 + g, end=1500 + g, strand="+-"[g % 2], gene_id="G%d" % g,
                         exon_id="E%d" % g))
    ann = Annotation.from_rows(["c"], rows)
    recs = [dict(qname="a%d" % (i // 2), tid=0, pos=200 + i, cigar=[(abi.CIG_M, 50), (abi.CIG_N, 100), (abi.CIG_M, 50)],
                 flag=99 if i % 2 == 0 else 147) for i in range(300)]
    b = Batch.from_records(recs)
    p = abi.default_params()
    got = engine.run_engine(p, ann, [b])
    assert list(got.gene_reads) == [300] * 12
    assert list(got.gene_fragments) == [150] * 12
    assert_results_match(got, oracle_lib.run_oracle(p, ann, [b]))


def test_resident_batches_and_reset(oracle_lib):
    ann, batch = small_inputs(n_pairs=6000)
    p = abi.default_params()
    want = oracle_lib.run_oracle(p, ann, [batch])
    e = engine.Engine(p)
    e.set_annotation(ann)
    h = e.upload(batch)
    for _ in range(3):
        e.reset()
        e.submit_resident(h)
        assert_results_match(e.finalize(), want)
    t = e.timing()
    assert t["classify_launches"] == 3 and t["classify_records"] == 3 * batch.n
    assert t["classify_bytes"] == 3 * batch.algorithmic_bytes and t["classify_ms"] > 0
    e.close()


def test_edge_inputs(oracle_lib):
    ann, batch = small_inputs(n_pairs=500)
    p = abi.default_params()
    # empty batch, one-record batch, records with wide (escaped) fields
    e = engine.Engine(p)
    e.set_annotation(ann)
    e.submit(batch.slice(0, 0))
    r = e.finalize()
    assert int(r.counters.sum()) == 0 and r.read_length == 0
    e.close()
    one = batch.slice(10, 11)
    assert_results_match(engine.run_engine(p, ann, [one]), oracle_lib.run_oracle(p, ann, [one]))
    recs = [dict(qname="w1", tid=0, pos=5000, cigar=[(abi.CIG_M, 70000)], flag=99, nm=300, l_qseq=70000),
            dict(qname="w2", tid=0, pos=6000, cigar=[(abi.CIG_M, 1), (abi.CIG_I, 1)] * 150, flag=147, nm=3),
            dict(qname="w3", tid=0, pos=7000, cigar=[(abi.CIG_S, 50)], flag=99, nm=0),          # no reference base
            dict(qname="w4", tid=5, pos=7000, cigar=[(abi.CIG_M, 50)], flag=99, nm=0),          # unrecognised RefID
            dict(qname="w5", tid=-1, pos=-1, cigar=[], flag=77, mapq=0, nm=None, l_qseq=50)]
    wide = Batch.from_records(recs)
    assert len(wide.wide_index) == 2
    assert_results_match(engine.run_engine(p, ann, [wide]), oracle_lib.run_oracle(p, ann, [wide]))


def test_bad_cigar_is_an_error(oracle_lib):
    ann, _ = small_inputs(n_pairs=10)
    b = Batch.from_records([dict(qname="b", tid=0, pos=100, cigar=[(9, 50)], flag=99)])
    p = abi.default_params()
    with pytest.raises(oracle_lib.OracleError) as eo:
        oracle_lib.run_oracle(p, ann, [b])
    assert eo.value.code == abi.ERR_BAD_CIGAR
    with pytest.raises(engine.EngineError) as eg:
        engine.run_engine(p, ann, [b])
    assert eg.value.code == abi.ERR_BAD_CIGAR


def test_contig_ownership_shards(oracle_lib):
    # multi-GPU by contig: each shard sees only its contigs' records and owns only their genes
    ann, batch = small_inputs(n_pairs=8000)
    p = abi.default_params()
    whole = oracle_lib.run_oracle(p, ann, [batch])
    tid = batch.tid_per_record()
    a_end = int(np.searchsorted(tid, 1))
    shard0, shard1 = batch.slice(0, a_end), batch.slice(a_end, batch.n)
    r0 = engine.run_engine(p, ann, [shard0], owned=[1, 0, 0])
    r1 = engine.run_engine(p, ann, [shard1], owned=[0, 1, 1])
    np.testing.assert_array_equal(r0.gene_reads + r1.gene_reads, whole.gene_reads)
    np.testing.assert_array_equal(r0.gene_fragments + r1.gene_fragments, whole.gene_fragments)
    np.testing.assert_array_equal(r0.counters + r1.counters, whole.counters)
    assert not (r0.gene_cov_valid & r1.gene_cov_valid).any()
    np.testing.assert_array_equal(r0.gene_cov_valid | r1.gene_cov_valid, whole.gene_cov_valid)
    np.testing.assert_allclose(r0.gene_cov_mean + r1.gene_cov_mean, whole.gene_cov_mean, rtol=1e-9, atol=1e-12)
    np.testing.assert_array_equal(r0.bias_three + r1.bias_three, whole.bias_three)


def test_group_reduce_rccl_and_peer_paths(oracle_lib):
    """rsqc_reduce_group, the exchange step of `rnaseqc --gpus N`: (i) a group of ONE context takes the RCCL path (librccl bound at
    run time, ncclCommInitAll, three ncclReduce in a group call on the context's stream) and leaves the results what they were;
    (ii) two contexts sharded by contig on the box's one device cannot share a communicator and take the peer-copy path: the sum
    on the first context equals the unsharded run.  (Two devices, RCCL path: the same code with n = 2.)"""
    ann, batch = small_inputs(n_pairs=8000)
    p = abi.default_params()
    whole = oracle_lib.run_oracle(p, ann, [batch])
    e = engine.Engine(p)
    try:
        e.set_annotation(ann)
        e.submit(batch); e.wait()
        e.finalize_device()
        assert engine.Engine.reduce_group([e]) is True          # RCCL, one rank
        assert_results_match(e.refresh_results(), whole)
    finally:
        e.close()
    tid = batch.tid_per_record()
    a_end = int(np.searchsorted(tid, 1))
    e0, e1 = engine.Engine(p), engine.Engine(p)
    try:
        e0.set_annotation(ann, np.array([1, 0, 0], np.uint8)); e1.set_annotation(ann, np.array([0, 1, 1], np.uint8))
        e0.submit(batch.slice(0, a_end)); e1.submit(batch.slice(a_end, batch.n)); e0.wait(); e1.wait()
        e0.finalize_device(); e1.finalize_device()
        assert engine.Engine.reduce_group([e0, e1]) is False     # same device twice: peer copies
        got = e0.refresh_results()
        np.testing.assert_array_equal(got.gene_reads, whole.gene_reads)
        np.testing.assert_array_equal(got.gene_fragments, whole.gene_fragments)
        np.testing.assert_array_equal(got.counters, whole.counters)
        np.testing.assert_allclose(got.exon_reads, whole.exon_reads, rtol=1e-9, atol=1e-6)
        np.testing.assert_array_equal(got.gene_cov_valid, whole.gene_cov_valid)
    finally:
        e0.close(); e1.close()


@pytest.mark.gpu
def test_group_handle_reused_across_steps(oracle_lib):
    """rsqc_group_create / rsqc_group_reduce / rsqc_group_destroy: the communicators are made once, BEFORE the contexts hold an
    annotation, and serve several end-of-file exchanges (reset between them); the one-rank group takes the RCCL path, a group of
    two contexts on one device is still a valid group (peer copies) and says why."""
    ann, batch = small_inputs(n_pairs=6000)
    p = abi.default_params()
    whole = oracle_lib.run_oracle(p, ann, [batch])
    e = engine.Engine(p)
    g = engine.Engine.Group([e])                                  # before set_annotation: the group only needs the devices
    try:
        info = g.info()
        assert info["uses_rccl"] and info["init_ms"] >= 0.0
        e.set_annotation(ann)
        for _ in range(2):
            e.reset(); e.submit(batch); e.wait(); e.finalize_device()
            assert g.reduce() is True
            assert_results_match(e.refresh_results(), whole)
        assert g.info()["last_reduce_ms"] > 0.0
    finally:
        g.close(); e.close()
    e0, e1 = engine.Engine(p), engine.Engine(p)
    g = engine.Engine.Group([e0, e1])
    try:
        info = g.info()
        assert not info["uses_rccl"] and "share a device" in info["note"]
    finally:
        g.close(); e0.close(); e1.close()


@pytest.mark.gpu
def test_group_reduce_on_two_devices(oracle_lib):
    """The N >= 2 RCCL branch on REAL devices: two contexts on two GPUs sharded by contig, communicators from ncclCommInitAll,
    grouped ncclReduce onto the first.  Runs wherever the box has two GPUs (hipGetDeviceCount() >= 2); the 1-GPU boxes skip it."""
    ann, batch = small_inputs(n_pairs=8000)
    p = abi.default_params()
    whole = oracle_lib.run_oracle(p, ann, [batch])
    tid = batch.tid_per_record()
    a_end = int(np.searchsorted(tid, 1))
    p0, p1 = abi.default_params(), abi.default_params()
    p0.device, p1.device = 0, 1
    e0 = engine.Engine(p0)
    try:
        e1 = engine.Engine(p1)                                    # rsqc_create on device 1: fails on a one-GPU box
    except engine.EngineError:
        e0.close()
        pytest.skip("needs two GPUs (hipGetDeviceCount() < 2 here)")
    g = engine.Engine.Group([e0, e1])
    try:
        assert g.info()["uses_rccl"], g.info()
        e0.set_annotation(ann, np.array([1, 0, 0], np.uint8)); e1.set_annotation(ann, np.array([0, 1, 1], np.uint8))
        e0.submit(batch.slice(0, a_end)); e1.submit(batch.slice(a_end, batch.n)); e0.wait(); e1.wait()
        e0.finalize_device(); e1.finalize_device()
        assert g.reduce() is True
        got = e0.refresh_results()
        np.testing.assert_array_equal(got.gene_reads, whole.gene_reads)
        np.testing.assert_array_equal(got.gene_fragments, whole.gene_fragments)
        np.testing.assert_array_equal(got.counters, whole.counters)
        np.testing.assert_allclose(got.exon_reads, whole.exon_reads, rtol=1e-9, atol=1e-6)
    finally:
        g.close(); e0.close(); e1.close()


def test_crafted_qname_hash_collision_counts_exactly(oracle_lib, tmp_path):
    """tests/golden/qname_hash_collision.json through the device: two different names with ONE 64-bit rsqc_qname_hash in one gene.
    With the second hash of the batch format (rsqc_batch.qhash2, filled by Batch.from_records as by the library's own BAM
    decode) the end-of-file stage compares 96 bits and counts what the reference's std::set<std::string> counts
    (src/Expression.cpp:383-387): THREE fragments.  A caller that leaves qhash2 NULL gets the 64-bit identity (two), which is
    what the oracle says for those inputs.  The same file through the device-side BAM decode: three.  DESIGN.md 5."""
    from tests.test_oracle_semantics import _collision_case
    from tests.test_gpu_decode import decode_file
    from rnaseqc_amd import bamio
    import copy
    _fx, ann, batch = _collision_case()
    assert batch.qhash2 is not None
    p = abi.default_params()
    exact = oracle_lib.run_oracle(p, ann, [batch])                       # by the names
    got = engine.run_engine(p, ann, [batch])
    assert_results_match(got, exact)
    assert int(got.gene_fragments[0]) == 3 and int(got.gene_reads[0]) == 6
    narrow = copy.copy(batch); narrow.qname = None; narrow.qname_off = None; narrow.qhash2 = None
    got64 = engine.run_engine(p, ann, [narrow])
    assert_results_match(got64, oracle_lib.run_oracle(p, ann, [narrow]))
    assert int(got64.gene_fragments[0]) == 2
    path = str(tmp_path / "collide.bam")
    bamio.write_bam(path, [("c", 10_000)], batch)
    e = engine.Engine(p)
    try:
        e.set_annotation(ann)
        decode_file(e, path, 1, collect=False)
        e.wait()
        assert_results_match(e.finalize(), exact)
    finally:
        e.close()


def test_chr1_scale_million_reads(oracle_lib):
    ann = synth.make_annotation(seed=1)                       # chr1-like: 5 234 genes
    batch = synth.make_reads(ann, 500_000, seed=2)
    p = abi.default_params()
    got = engine.run_engine(p, ann, [batch])
    want = oracle_lib.run_oracle(p, ann, [batch])
    assert_results_match(got, want)
    # size-independent invariants of the path: every counted record adds 1 to a
    # gene and its fractions add 1 to that gene's exons when no record is counted to two genes
    assert abs(got.exon_reads.sum() - got.gene_reads.sum()) < 1e-3 * max(1, got.gene_reads.sum())
    assert got.counter("Exonic Reads") + got.counter("Intronic Reads") + got.counter("Intergenic Reads") + \
        got.counter("Ambiguous Reads") == got.counter("Reads used for Intron/Exon counts")
    assert (got.gene_fragments <= got.gene_reads).all() and (got.gene_unique <= got.gene_reads).all()


def test_finalize_device_then_refresh_matches_finalize(oracle_lib):
    """The distributed flow: end-of-file stage without read-back, then one read-back after the reduction."""
    ann = synth.make_annotation(seed=11, contigs=[("c1", 2_000_000, 80), ("c2", 1_000_000, 40)])
    batch = synth.make_reads(ann, 40_000, seed=12)
    p = abi.default_params()
    want = oracle_lib.run_oracle(p, ann, [batch])
    e = engine.Engine(p)
    try:
        e.set_annotation(ann)
        e.submit(batch); e.wait()
        e.finalize_device()
        got = e.refresh_results()
        assert_results_match(got, want)
        again = e.finalize()                 # already finalized: plain read-back
        assert_results_match(again, want)
    finally:
        e.close()


def test_genome_scale_annotation(oracle_lib):
    """GENCODE-shaped annotation (25 contigs, 56 202 genes) in two batches: contig segments, the
    per-contig bin tables and boundary tiles at full annotation size (BASELINE.json configs[2] shape)."""
    ann = synth.make_annotation(seed=3, contigs=synth.human_contigs())
    batch = synth.make_reads(ann, 1_000_000, seed=4)
    half = batch.n // 2 + 17                                   # not a multiple of the tile size
    parts = [batch.slice(0, half), batch.slice(half, batch.n)]
    p = abi.default_params()
    got = engine.run_engine(p, ann, parts)
    want = oracle_lib.run_oracle(p, ann, parts)
    assert_results_match(got, want)
    assert got.counter("Total Alignments") == batch.n


@pytest.mark.parametrize("samples", [1000000, 100, 7])
def test_fragment_sizes_with_bed(oracle_lib, samples):
    # --bed: pairs whose mates both sit inside one BED interval are sampled in file order (K5)
    ann = synth.make_annotation(seed=8, contigs=[("chrA", 600_000, 60), ("chrB", 300_000, 30)])
    bed = synth.make_bed(ann, min_len=250)
    assert len(bed.contig) >= 10
    batch = synth.make_reads(ann, 60000, seed=10, frac=(0.9, 0.04, 0.03, 0.03), expr_sigma=1.0,
                             contig_lengths=np.array([600_000, 300_000]))
    p = abi.default_params(fragment_samples=samples)
    want = oracle_lib.run_oracle(p, ann, [batch], bed=bed)
    assert want.fragment_count.sum() == min(samples, want.fragment_count.sum()) and want.fragment_count.sum() > 0
    if samples == 1000000:
        assert want.fragment_count.sum() > 200
    got = engine.run_engine(p, ann, [batch], bed=bed)
    assert_results_match(got, want)
    assert got.fragment_samples_remaining == want.fragment_samples_remaining
    # split into batches: same result
    parts = [batch.slice(0, 50000), batch.slice(50000, batch.n)]
    assert_results_match(engine.run_engine(p, ann, parts, bed=bed), want)


def test_bed_zero_length_first_block_at_an_interval_end(oracle_lib):
    """the wave-level BED cursor of classify_ei_kernel<true> ("is any interval near this tile's candidates?") must not skip a
    record whose ZERO-LENGTH aligned block sits one position behind the end of an interval -- fragmentSizeMetrics' block test
    (src/Expression.cpp:490-507) accepts it: start <= bs and end >= be - 1 with be = bs.  Records of one zero-length block only (clips around
    it), placed exactly at, one before and one behind the ends of BED intervals, beside ordinary pairs; device = oracle in full."""
    rows = [dict(contig="c", type="gene", start=1000, end=90000, strand="+", gene_id="G"),
            dict(contig="c", type="exon", start=1000, end=90000, strand="+", gene_id="G", exon_id="E")]
    ann = Annotation.from_rows(["c"], rows)
    from rnaseqc_amd.model import Bed
    starts = [2000 + 3000 * k for k in range(20)]
    bed = Bed.from_intervals([0] * 20, starts, [x + 1500 for x in starts])
    M, S = abi.CIG_M, abi.CIG_S
    recs = []
    for k, x in enumerate(starts):
        end = x + 1500
        for d in (-2, -1, 0, 1, 2):                           # the zero-length block around the interval's end (1-based block start = pos + 1)
            q = "z%d_%d" % (k, d)                             # first mate an ordinary read inside the interval, second mate the zero-length record
            recs.append(dict(qname=q, tid=0, pos=x + 1300 + d, cigar=[(M, 100)], flag=99, mapq=255, nm=0, mpos=end + d, mtid=0, isize=300 + k))
            recs.append(dict(qname=q, tid=0, pos=end + d, cigar=[(S, 60), (M, 0), (S, 40)], flag=147, mapq=255, nm=0, mpos=x + 1300 + d, mtid=0, isize=-(300 + k)))
        q = "p%d" % k                                         # an ordinary pair inside the interval
        recs.append(dict(qname=q, tid=0, pos=x + 100, cigar=[(M, 100)], flag=99, mapq=255, nm=0, mpos=x + 400, mtid=0, isize=400))
        recs.append(dict(qname=q, tid=0, pos=x + 400, cigar=[(M, 100)], flag=147, mapq=255, nm=0, mpos=x + 100, mtid=0, isize=-400))
    recs.sort(key=lambda r: r["pos"])
    b = Batch.from_records(recs)
    p = abi.default_params()
    want = oracle_lib.run_oracle(p, ann, [b], bed=bed)
    assert want.fragment_count.sum() > 20                      # (the ordinary pairs give 20: some zero-length mates are sampled too)
    got = engine.run_engine(p, ann, [b], bed=bed)
    assert_results_match(got, want)
    assert got.fragment_samples_remaining == want.fragment_samples_remaining


def test_fragment_sizes_duplicate_qnames(oracle_lib):
    # hand-made QNAME groups: a third record after a completed pair starts a new pending entry; a
    # failing second mate leaves the entry in place (src/Expression.cpp:528)
    rows = [dict(contig="c", type="gene", start=1000, end=9000, strand="+", gene_id="G"),
            dict(contig="c", type="exon", start=1000, end=9000, strand="+", gene_id="G", exon_id="E")]
    ann = Annotation.from_rows(["c"], rows)
    from rnaseqc_amd.model import Bed
    bed = Bed.from_intervals([0, 0], [1500, 5000], [4000, 8000])
    M = abi.CIG_M
    def rec(q, pos, flag, mpos, isize):
        return dict(qname=q, tid=0, pos=pos, cigar=[(M, 100)], flag=flag, mpos=mpos, isize=isize)
    recs = [rec("a", 2000, 99, 2200, 300), rec("b", 2050, 99, 2250, 300), rec("c", 2100, 163, 2300, 310),
            rec("a", 2200, 147, 2000, -300),          # sample 300
            rec("b", 2250, 83 | 0x20, 2050, -300),    # mate reverse flag set -> no sample, entry stays
            rec("b", 2260, 147, 2050, -310),          # same name again: compared with the FIRST b -> sample 310
            rec("c", 2300, 83, 2100, -310),           # sample 310
            rec("a", 2400, 147, 2000, -500),          # a was erased: becomes a new pending entry
            rec("a", 2500, 147, 2000, -600),          # ends after the pending a -> sample 600
            rec("d", 5100, 99, 2600, 2600), rec("d", 2600, 147, 5100, -2600)]
    recs.sort(key=lambda r: r["pos"])
    b = Batch.from_records(recs)
    p = abi.default_params()
    want = oracle_lib.run_oracle(p, ann, [b], bed=bed)
    assert dict(zip(want.fragment_size.tolist(), want.fragment_count.tolist())) == {300: 1, 310: 2, 600: 1}
    got = engine.run_engine(p, ann, [b], bed=bed)
    assert_results_match(got, want)


def test_single_pair_golden_reconstruction_gpu(oracle_lib):
    import json, os
    ka = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "reference_known_answers.json")))
    ann, batch = cases.single_pair_case()
    p = abi.default_params()
    got = engine.run_engine(p, ann, [batch])
    c = got.counter_dict()
    for k, v in ka["single_pair"]["metrics"].items():
        if k in c:
            assert c[k] == int(v), (k, c[k], v)
    assert list(got.gene_reads) == [2] and list(got.gene_fragments) == [1] and got.read_length == 76
    assert_results_match(got, oracle_lib.run_oracle(p, ann, [batch]))


# ---- --legacy counting rules (src/Expression.cpp:129-304, src/RNASeQC.cpp:258-287) -------------------------------
@pytest.mark.parametrize("kw", [dict(), dict(stranded=abi.STRAND_REVERSE), dict(unpaired=1, n_filter_tags=1, exclude_chimeric=1),
                                dict(coverage_mask=100, stranded=abi.STRAND_FORWARD)])
def test_legacy_rules_vs_oracle(oracle_lib, kw):
    ann, batch = small_inputs(dup_frac=0.1, chimeric_tag_frac=0.01, filter_tag_frac=0.02)
    p = abi.default_params(legacy=1, mapq_threshold=4, **kw)
    want = oracle_lib.run_oracle(p, ann, [batch])
    assert int(want.counters[abi.COUNTER_INDEX["Split Reads"]]) > 100 and int(want.gene_reads.sum()) > 1000
    assert_results_match(engine.run_engine(p, ann, [batch]), want)


def test_legacy_hand_derived_and_hostile_cases(oracle_lib):
    from tests import test_legacy_rules as tl
    ann = cases.quirk_annotation()
    got = engine.run_engine(tl._legacy_params(), ann, [Batch.from_records(tl._records())])
    tl._check(got, ann)
    for seed in range(4):
        ann, batch = tl.hostile_case(seed)
        for kw in (dict(), dict(stranded=abi.STRAND_FORWARD)):
            p = tl._legacy_params(coverage_mask=20, **kw)
            assert_results_match(engine.run_engine(p, ann, [batch]), oracle_lib.run_oracle(p, ann, [batch]))
    for seed in range(2):                         # stacked genes: more genes per read than the kernel returns in registers
        ann, batch = tl.stacked_case(seed)
        for p in (tl._legacy_params(coverage_mask=20), abi.default_params(mapq_threshold=4, coverage_mask=20)):
            assert_results_match(engine.run_engine(p, ann, [batch]), oracle_lib.run_oracle(p, ann, [batch]))


def test_legacy_deep_coverage_and_batches(oracle_lib):
    # the same records in one batch and in three: coverage / bias of the end-of-file stage under the legacy commits
    ann = synth.make_annotation(seed=8, contigs=[("chrA", 400_000, 40)])
    batch = synth.make_reads(ann, 60000, seed=9, frac=(0.97, 0.01, 0.01, 0.01), expr_sigma=1.0, contig_lengths=np.array([400_000]))
    p = abi.default_params(legacy=1, mapq_threshold=4)
    want = oracle_lib.run_oracle(p, ann, [batch])
    assert_results_match(engine.run_engine(p, ann, [batch]), want)
    n = batch.n
    parts = [batch.slice(0, n // 3), batch.slice(n // 3, 2 * n // 3), batch.slice(2 * n // 3, n)]
    assert_results_match(engine.run_engine(p, ann, parts), want)


# ---- --fasta GC statistics (src/Expression.cpp:459-477, src/Metrics.cpp:299-303) -----------------------------------
def test_fasta_hand_derived_case(oracle_lib):
    from tests import test_fasta_gc as tg
    ann, batch, ref = tg.gc_case()
    p = abi.default_params(coverage_mask=0)
    got = engine.run_engine(p, ann, [batch], reference=ref)
    tg.check_gc_case(got, ann)
    assert_results_match(got, oracle_lib.run_oracle(p, ann, [batch], reference=ref))


@pytest.mark.parametrize("kw", [dict(), dict(stranded=abi.STRAND_REVERSE), dict(coverage_mask=50, unpaired=1)])
def test_fasta_gc_vs_oracle(oracle_lib, kw):
    ann, batch = small_inputs(n_pairs=40000, dup_frac=0.05)
    ref = synth.make_reference(SMALL_LENGTHS[:2], seed=11)            # chrC is not in the FASTA index
    p = abi.default_params(**kw)
    want = oracle_lib.run_oracle(p, ann, [batch], reference=ref)
    assert int(want.gc_bins.sum()) > 500 and int((want.gc_bins > 0).sum()) > 20
    got = engine.run_engine(p, ann, [batch], reference=ref)
    assert_results_match(got, want)
    n = batch.n                                                      # mates of a fragment in different batches
    assert_results_match(engine.run_engine(p, ann, [batch.slice(0, n // 2), batch.slice(n // 2, n)], reference=ref), want)


def test_crafted_collision_in_the_pairing_stages(oracle_lib):
    """tests/test_oracle_semantics.py::_collision_pairing_case through the device with --bed and --fasta: the fragment-size sampler
    and the fragment GC pairing (rsqc_k5.h) key a name on (qhash, qhash2) like the fragment de-duplication -- three fragments with
    the reference's sizes and GC bins, not the two of the 64-bit hash alone.  Also through rsqc_submit in two
    batches (the candidates of the first are retired into the arenas) and for a caller without qhash2."""
    import copy
    from tests.test_oracle_semantics import _collision_pairing_case
    ann, batch, bed, ref = _collision_pairing_case()
    p = abi.default_params(coverage_mask=0)
    exact = oracle_lib.run_oracle(p, ann, [batch], bed=bed, reference=ref)            # by the names
    got = engine.run_engine(p, ann, [batch], bed=bed, reference=ref)
    assert_results_match(got, exact)
    assert sorted(int(x) for x in got.fragment_size) == [320, 360, 400]
    np.testing.assert_array_equal(got.gc_bins, exact.gc_bins)
    parts = [batch.slice(0, 3), batch.slice(3, batch.n)]
    split = engine.run_engine(p, ann, parts, bed=bed, reference=ref)
    assert_results_match(split, exact)
    np.testing.assert_array_equal(split.gc_bins, exact.gc_bins)
    narrow = copy.copy(batch); narrow.qname = None; narrow.qname_off = None; narrow.qhash2 = None
    got64 = engine.run_engine(p, ann, [narrow], bed=bed, reference=ref)
    want64 = oracle_lib.run_oracle(p, ann, [narrow], bed=bed, reference=ref)
    assert_results_match(got64, want64)
    np.testing.assert_array_equal(got64.gc_bins, want64.gc_bins)
    assert int(got64.fragment_count.sum()) == 2


def test_name_identity_is_one_per_pass(oracle_lib):
    """A pass whose batches disagree about rsqc_batch.qhash2 is refused (RSQC_ERR_ARG): the mates of a fragment would carry two
    different identities."""
    import copy
    from tests.test_oracle_semantics import _collision_case
    _fx, ann, batch = _collision_case()
    with_h2, without = batch.slice(0, 3), batch.slice(3, batch.n)
    without = copy.copy(without); without.qhash2 = None
    e = engine.Engine(abi.default_params())
    try:
        e.set_annotation(ann)
        e.submit(with_h2)
        with pytest.raises(Exception):
            e.submit(without)
    finally:
        e.close()


def test_exon_outside_its_gene_row_on_the_device(oracle_lib):
    """tests/test_core_semantics_host.py::test_exon_outside_its_gene_row_is_accepted on the device: rsqc_set_annotation accepts the
    annotation with a warning in rsqc_last_error (the reference's "Gene encountered after computing coverage") and the result is the
    reference's streamed one for records that reach the gene before its row retires."""
    from rnaseqc_amd.model import Annotation, Batch
    from tests.test_core_semantics_host import _outside_rows, _pair
    ann = Annotation.from_rows(["c"], _outside_rows())
    recs = _pair("a", 150, 820) + _pair("b", 810, 880, n=100) + _pair("c", 1905, 2650) + _pair("d", 1950, 2300) + _pair("e", 120, 200)
    recs.sort(key=lambda r: r["pos"])
    b = Batch.from_records(recs)
    p = abi.default_params(coverage_mask=0)
    e = engine.Engine(p)
    try:
        import warnings
        with warnings.catch_warnings(record=True) as seen:
            warnings.simplefilter("always")
            w = e.set_annotation(ann)
        assert w and "outside the row of its gene" in w and "outside the row of its gene" in e.last_error()
        assert any("outside the row of its gene" in str(x.message) for x in seen)          # Engine.set_annotation logs it itself
        e.submit(b)
        got = e.finalize()
        assert got.exons_outside_gene_row >= 1                                               # ... and the results carry the flag
        assert_results_match(got, oracle_lib.run_oracle(p, ann, [b]))
    finally:
        e.close()


def test_one_batch_of_file_ranges_equals_one_batch_per_range(oracle_lib):
    """rsqc_batch.seg_file_index: the non-adjacent contigs of a shard as ONE batch give the results AND the shard summary (per-range
    Read-Length functions, kept fragment samples) of the same ranges submitted one batch each; mixed read lengths, --bed with a
    cut-off, a range that is a contig's unmapped tail."""
    from rnaseqc_amd.model import Batch
    from tests.test_distributed_gloo import _inputs, _shards, CONTIGS
    ann, batch, bed = _inputs()
    rank_of = np.array([0, 1, 0, 1])
    owned = np.array([1, 0, 1, 0], np.uint8)
    runs = _shards(batch, rank_of, 0, 2, split=False)
    assert len(runs) == 2
    p = abi.default_params(fragment_samples=150)
    out = []
    for parts in (runs, [Batch.concat_ranges(runs)]):
        e = engine.Engine(p)
        try:
            e.set_annotation(ann, owned); e.set_bed(bed)
            for b in parts:
                e.submit(b)
            r = e.finalize(); si = e.shard_summary()
            out.append((r, si))
        finally:
            e.close()
    (ra, sa), (rb, sb) = out
    assert_results_match(rb, ra)
    assert rb.read_length == ra.read_length
    for f in ("batch_file_index", "batch_records", "rl_offset", "rl_span", "rl_state"):
        np.testing.assert_array_equal(getattr(sb, f), getattr(sa, f), err_msg=f)
    oa = np.argsort(sa.sample_file_index); ob = np.argsort(sb.sample_file_index)
    np.testing.assert_array_equal(sb.sample_file_index[ob], sa.sample_file_index[oa])
    np.testing.assert_array_equal(sb.sample_size[ob], sa.sample_size[oa])
    assert len(set(int(x) for x in sa.rl_state)) > 1


def test_constant_read_names_with_bed_and_fasta(oracle_lib):
    """3 000 HQ paired records that all carry ONE read name (stripped / constant QNAMEs) inside one BED interval and one exon: the
    pairing stages' bucket for that name is beyond the LDS sort and is sorted in memory (rsqc_k5.h: pair_bucket_big_sort) -- round 4
    failed such a run with RSQC_ERR_CAPACITY.  Fragment sizes, their cut-off and the GC histogram equal the oracle's
    walk of its QNAME-keyed maps."""
    from rnaseqc_amd.model import Bed, Reference
    rows = [dict(contig="c", type="gene", start=1000, end=90000, strand="+", gene_id="G"),
            dict(contig="c", type="exon", start=1000, end=90000, strand="+", gene_id="G", exon_id="E")]
    ann = Annotation.from_rows(["c"], rows)
    bed = Bed.from_intervals([0], [1500], [88000])
    rng = np.random.default_rng(9)
    ref = Reference(contig=[0], sequence=[np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, 100000)]])
    M = abi.CIG_M
    recs = []
    for k in range(1500):
        p1 = 2000 + 50 * k; p2 = p1 + int(rng.integers(120, 600))
        size = p2 + 100 - p1
        recs.append(dict(qname="x", tid=0, pos=p1, cigar=[(M, 100)], flag=99, mapq=255, nm=0, mpos=p2, mtid=0, isize=size))
        recs.append(dict(qname="x", tid=0, pos=p2, cigar=[(M, 100)], flag=147, mapq=255, nm=0, mpos=p1, mtid=0, isize=-size))
    recs.sort(key=lambda r: r["pos"])
    b = Batch.from_records(recs)
    for samples in (1000000, 37):
        p = abi.default_params(fragment_samples=samples, coverage_mask=0)
        want = oracle_lib.run_oracle(p, ann, [b], bed=bed, reference=ref)
        assert int(want.fragment_count.sum()) == min(samples, int(want.fragment_count.sum())) and int(want.fragment_count.sum()) > 30
        got = engine.run_engine(p, ann, [b], bed=bed, reference=ref)
        assert_results_match(got, want)


@pytest.mark.parametrize("n_names,ok", [(33, True), (34, False)])
def test_names_sharing_one_64_bit_hash_boundary(oracle_lib, n_names, ok):
    """The set-aside list of the fragment count (rsqc_k4.h: names whose 64-bit hash is already in a partition's set under ANOTHER
    second hash) holds 32 entries per partition: 33 names with one 64-bit hash in one gene (the set's owner + 32) are counted exactly,
    the 34th is a clean RSQC_ERR_CAPACITY, never a miscount (include/rnaseqc_amd.h documents the limit).  Hash-only
    batch: the hashes are written directly -- one crafted 64-bit collision costs 5e9 hash evaluations (tools/qname_collision.c)."""
    rows = [dict(contig="c", type="gene", start=100, end=9000, strand="+", gene_id="G0"),
            dict(contig="c", type="exon", start=100, end=9000, strand="+", gene_id="G0", exon_id="E0")]
    ann = Annotation.from_rows(["c"], rows)
    recs = [dict(qname="n%d" % k, tid=0, pos=200 + 10 * k, cigar=[(abi.CIG_M, 50)], flag=0, mapq=255, nm=0) for k in range(n_names)]
    b = Batch.from_records(recs)
    b.qname = None; b.qname_off = None
    b.qhash = np.full(b.n, 0x1234567890ABCDEF, np.uint64)
    b.qhash2 = (np.arange(b.n, dtype=np.uint32) * np.uint32(2654435761) + np.uint32(17)).astype(np.uint32)
    p = abi.default_params(unpaired=1)
    if ok:
        got = engine.run_engine(p, ann, [b])
        assert int(got.gene_fragments[0]) == n_names == int(got.gene_reads[0])
        assert_results_match(got, oracle_lib.run_oracle(p, ann, [b]))
    else:
        with pytest.raises(engine.EngineError) as e:
            engine.run_engine(p, ann, [b])
        assert e.value.code == abi.ERR_CAPACITY


@pytest.mark.parametrize("seed", range(2))
def test_spliced_reads_on_shared_junctions(oracle_lib, seed):
    """The two-block calls of one junction's reads (classify_ei_kernel's wave-uniform two-block path, rsqc_k1.h k1e_uniform2) beside calls
    that splice elsewhere (general path): two genes sharing an exon, an exon under a second exon of its gene, a globin, an rRNA gene, soft
    clips / insertions (mixed aligned lengths), zero-length blocks -- device = oracle in full, unstranded (uniform path on) and stranded (off).
    The reads have MIXED lengths in fewer than 64 tiles: the case in which read_length_kernel's tile gate must re-open tiles when a
    spliced record lowers the state (src/RNASeQC.cpp:275-278; the gate was only ever narrowed before round 6's third session)."""
    from tests.test_k1_wave_emulation import junction_case
    ann, batch = junction_case(900 + seed, n_per_junction=2500)
    for kw in (dict(), dict(unpaired=1, mapq_threshold=3), dict(stranded=abi.STRAND_FORWARD)):
        p = abi.default_params(**kw)
        want = oracle_lib.run_oracle(p, ann, [batch])
        assert want.gene_reads.sum() > 3000
        assert_results_match(engine.run_engine(p, ann, [batch]), want)
        parts = [batch.slice(0, 4001), batch.slice(4001, batch.n)]
        assert_results_match(engine.run_engine(p, ann, parts), want)
