"""The per-record KERNELS (rnaseqc_amd/csrc/rsqc_k1.h: classify_ei_kernel; rsqc_k1s.h: classify_slow_kernel on its overflow list;
rsqc_kr.h: read_length_kernel; with rsqc_wave.h), unmodified, on a 64-lane SIMT emulation for the host (tests/hostemu/wavemu.h: one fiber per lane; ballot / shuffle / mbcnt /
LDS and memory atomics / __syncthreads on a cooperative scheduler) against the oracle: per-wave LDS queues sorted by block
count, the feature stage on full and on drained tiles, workgroup tables, pair chunks, the long-CIGAR queue, the overflow
list, Read-Length inputs per tile.  The GPU tests run the same source on the device."""
import numpy as np
import pytest

from rnaseqc_amd import abi, synth
from rnaseqc_amd.model import Annotation, Batch
from tests import cases, hostemu
from tests.test_legacy_rules import hostile_case, stacked_case


def _compare(o, r, cov=None):
    for i, n in enumerate(abi.COUNTER_NAMES):
        assert int(o.counters[i]) == int(r.counters[i]), n
    np.testing.assert_array_equal(o.gene_reads, r.gene_reads)
    np.testing.assert_array_equal(o.gene_unique, r.gene_unique)
    np.testing.assert_array_equal(o.gene_fragments, r.gene_fragments)
    np.testing.assert_allclose(o.exon_reads, r.exon_reads, rtol=0, atol=1e-9)
    assert o.read_length == r.read_length
    if cov is not None:
        np.testing.assert_array_equal(o.cov, cov)


def test_quirk_case(oracle_lib):
    ann, batch = cases.quirk_case()
    p = abi.default_params()
    for grid in (1, 3):
        _compare(hostemu.run_k1(p, ann, batch, grid=grid), oracle_lib.run_oracle(p, ann, [batch]))


@pytest.mark.parametrize("kw", [dict(), dict(stranded=abi.STRAND_REVERSE), dict(stranded=abi.STRAND_FORWARD, unpaired=1),
                                dict(unpaired=1, mapq_threshold=3, n_filter_tags=1, exclude_chimeric=1),
                                dict(base_mismatch=1, chimeric_distance=100)])
def test_synthetic_vs_oracle(oracle_lib, kw):
    """Three contigs (one without features), ~12 k records: several tiles per wave, queues that fill and drain, contig
    boundaries inside tiles, the records of the last contig."""
    ann = synth.make_annotation(seed=3, contigs=[("chrA", 3_000_000, 300), ("chrB", 1_500_000, 150), ("chrC", 400_000, 0)])
    batch = synth.make_reads(ann, 6000, seed=4, dup_frac=0.1, chimeric_tag_frac=0.01, filter_tag_frac=0.02,
                             contig_lengths=np.array([3_000_000, 1_500_000, 400_000]))
    p = abi.default_params(**kw)
    r = oracle_lib.run_oracle(p, ann, [batch])
    ref = hostemu.run(p, ann, batch, mode=1, want_cov=True)
    for grid in (1, 2, 5):
        o = hostemu.run_k1(p, ann, batch, grid=grid, want_cov=True)
        _compare(o, r, ref.cov)
        assert o.n_pairs > 1000
        assert o.n_coarse > 200                       # records in empty stretches: answered by the coarse table, no rank word
    assert r.gene_reads.sum() > 1000


@pytest.mark.parametrize("seed", range(4))
def test_hostile_annotations(oracle_lib, seed):
    """Overlapping / nested genes and exons (intervals covered by more than two exons -> overflow list), 1-4 blocks per record,
    zero-length operations, long CIGARs, records at position 0."""
    for ann, batch in (hostile_case(400 + seed), stacked_case(500 + seed)):
        for kw in (dict(), dict(stranded=abi.STRAND_FORWARD), dict(stranded=abi.STRAND_REVERSE, unpaired=1)):
            p = abi.default_params(mapq_threshold=4, **kw)
            ref = hostemu.run(p, ann, batch, mode=1, want_cov=True)
            want = oracle_lib.run_oracle(p, ann, [batch])
            o = hostemu.run_k1(p, ann, batch, grid=2, want_cov=True)                      # overflow list -> classify_slow_kernel
            _compare(o, want, ref.cov)
            _compare(hostemu.run_k1(p, ann, batch, grid=2, want_cov=True, slow_kernel=False), want, ref.cov)   # -> per-record code on the host
            assert o.n_overflow > 0


def test_default_build_without_the_coarse_table(oracle_lib):
    """The product's default configuration (no -DK1E_COARSE: every look-up through the rank words) on the same input."""
    ann = synth.make_annotation(seed=3, contigs=[("chrA", 3_000_000, 300), ("chrB", 1_500_000, 150), ("chrC", 400_000, 0)])
    batch = synth.make_reads(ann, 6000, seed=4, dup_frac=0.1, contig_lengths=np.array([3_000_000, 1_500_000, 400_000]))
    p = abi.default_params()
    r = oracle_lib.run_oracle(p, ann, [batch])
    o = hostemu.run_k1(p, ann, batch, grid=3, coarse=False)
    _compare(o, r)
    assert o.n_coarse == 0


def test_small_and_ragged_batches(oracle_lib):
    """Fewer records than lanes, one record, sizes around the tile and queue boundaries, more workgroups than tiles."""
    ann = synth.make_annotation(seed=5, contigs=[("chrA", 400_000, 60), ("chrB", 200_000, 30)])
    full = synth.make_reads(ann, 500, seed=6, contig_lengths=np.array([400_000, 200_000]))
    p = abi.default_params()
    for n in (1, 2, 63, 64, 65, 127, 128, 129, 191, 257, 640, full.n):
        b = full.slice(0, min(n, full.n))
        r = oracle_lib.run_oracle(p, ann, [b])
        for grid in (1, 4):
            _compare(hostemu.run_k1(p, ann, b, grid=grid), r)


def test_wide_records_and_long_cigars(oracle_lib):
    """Escape values (l_qseq >= 65535, NM >= 255, >= 255 CIGAR operations) come from the wide table in both kernels; CIGARs of
    5-8 operations are walked from registers, longer ones from memory; 3- and 4-block records; more than 4 blocks -> general code."""
    rows = [dict(contig="c", type="gene", start=100, end=90000, strand="+", gene_id="G0"),
            dict(contig="c", type="exon", start=100, end=90000, strand="+", gene_id="G0", exon_id="E0"),
            dict(contig="c", type="gene", start=95000, end=99000, strand="-", gene_id="G1"),
            dict(contig="c", type="exon", start=95000, end=96000, strand="-", gene_id="G1", exon_id="E1"),
            dict(contig="c", type="exon", start=97000, end=99000, strand="-", gene_id="G1", exon_id="E2")]
    ann = Annotation.from_rows(["c"], rows)
    M, I, D, N, S = abi.CIG_M, abi.CIG_I, abi.CIG_D, abi.CIG_N, abi.CIG_S
    recs = []
    rng = np.random.default_rng(11)
    for i in range(300):
        kind = i % 10
        pos = 150 + 10 * i
        if kind == 0:
            cig = [(M, 40)] + [(I, 1), (M, 1)] * 140                  # 281 operations: wide n_cigar, 141 blocks -> general code
        elif kind == 1:
            cig = [(S, 3), (M, 30), (N, 20), (M, 30), (N, 20), (M, 30), (S, 2)]   # 7 operations, 3 blocks
        elif kind == 2:
            cig = [(M, 20), (N, 5), (M, 20), (D, 2), (M, 20), (N, 7), (M, 20)]    # 4 blocks
        elif kind == 3:
            cig = [(M, 10), (I, 2), (M, 10), (D, 1), (M, 10), (I, 1), (M, 10), (N, 30), (M, 10), (S, 4)]   # 10 operations, 5 blocks
        elif kind == 4:
            cig = [(M, 70000)]                                         # l_qseq beyond 16 bits
        elif kind == 5:
            cig = [(S, 5), (M, 95), (S, 5), (abi.CIG_H, 3), (abi.CIG_P, 0)]       # 5 operations, 1 block
        else:
            cig = [(M, 50), (N, 100), (M, 50)] if rng.random() < 0.5 else [(M, 100)]
        recs.append(dict(qname="q%d" % (i // 2), tid=0, pos=pos, cigar=cig, flag=99 if i % 2 == 0 else 147, mapq=255,
                         nm=300 if kind == 6 else 1, mpos=pos + 50, mtid=0))
    b = Batch.from_records(recs)
    assert len(b.wide_index) > 0
    p = abi.default_params()
    r = oracle_lib.run_oracle(p, ann, [b])
    for grid in (1, 2):
        o = hostemu.run_k1(p, ann, b, grid=grid)
        _compare(o, r)
        assert o.n_overflow >= 60


def test_every_record_deferred_fills_the_long_kernels_chunks(oracle_lib):
    """Every record has four blocks inside the exons of two overlapping genes: classify_ei_kernel defers them all, classify_long_kernel's
    own pair chunks (sized like a K1 workgroup's, but there are fewer of them than K1 workgroups here) fill up, and the rest of the pairs
    take the per-lane path into the K1 workgroups' chunks.  Also: the three-block ring's surplus (tiles of 64 three-block records)."""
    rows = [dict(contig="c", type="gene", start=100, end=200000, strand="+", gene_id="G0"),
            dict(contig="c", type="exon", start=100, end=200000, strand="+", gene_id="G0", exon_id="E0"),
            dict(contig="c", type="gene", start=150, end=190000, strand="-", gene_id="G1"),
            dict(contig="c", type="exon", start=150, end=190000, strand="-", gene_id="G1", exon_id="E1")]
    ann = Annotation.from_rows(["c"], rows)
    M, N = abi.CIG_M, abi.CIG_N
    for four, n in ((True, 4000), (False, 1500)):
        recs = []
        for i in range(n):
            # (second input: a wave's ring holds 63 three-block records when a tile of 64 more arrives -- 21 fit, the rest is surplus)
            three = [(M, 30), (N, 5), (M, 30), (N, 7), (M, 30)] if (i >= 192 or (i < 189 and i % 3 == 0)) else [(M, 90)]
            cig = [(M, 20), (N, 5), (M, 20), (N, 5), (M, 20), (N, 7), (M, 20)] if four else three
            recs.append(dict(qname="q%d" % (i // 2), tid=0, pos=300 + 20 * i, cigar=cig, flag=99 if i % 2 == 0 else 147, mapq=255, nm=0,
                             mpos=300 + 20 * (i ^ 1), mtid=0))
        b = Batch.from_records(recs)
        p = abi.default_params()
        r = oracle_lib.run_oracle(p, ann, [b])
        assert int(r.gene_reads.sum()) == 2 * n                       # every record is counted to both genes: two pairs each
        o = hostemu.run_k1(p, ann, b, grid=3 if four else 1)
        _compare(o, r)
        assert (o.n_deferred == n) if four else (0 < o.n_deferred < n)


def test_bed_instance_candidates(oracle_lib):
    """The --bed instance classify_ei_kernel<true> under emulation (its wave-level cursor into the BED rows and the scalar
    interval look-up had none): the fragment-size candidates it leaves must be exactly the records that pass the per-record rule
    (gate cascade + bed_interval_of, no shortcut) -- a synthetic BED over three contigs, and zero-length first blocks around interval ends."""
    from rnaseqc_amd.model import Bed
    ann = synth.make_annotation(seed=8, contigs=[("chrA", 600_000, 60), ("chrB", 300_000, 30), ("chrC", 100_000, 0)])
    bed = synth.make_bed(ann, min_len=250)
    batch = synth.make_reads(ann, 5000, seed=10, frac=(0.9, 0.04, 0.03, 0.03), expr_sigma=1.0, contig_lengths=np.array([600_000, 300_000, 100_000]))
    p = abi.default_params()
    r = oracle_lib.run_oracle(p, ann, [batch])
    for grid in (1, 3):
        o = hostemu.run_k1(p, ann, batch, grid=grid, bed=bed)
        _compare(o, r)
        assert o.n_candidates > 300
    rows = [dict(contig="c", type="gene", start=1000, end=90000, strand="+", gene_id="G"),
            dict(contig="c", type="exon", start=1000, end=90000, strand="+", gene_id="G", exon_id="E")]
    ann2 = Annotation.from_rows(["c"], rows)
    starts = [2000 + 3000 * k for k in range(20)]
    bed2 = Bed.from_intervals([0] * 20, starts, [x + 1500 for x in starts])
    M, S = abi.CIG_M, abi.CIG_S
    recs = []
    for k, x in enumerate(starts):
        end = x + 1500
        for dd in (-2, -1, 0, 1, 2):
            q = "z%d_%d" % (k, dd)
            recs.append(dict(qname=q, tid=0, pos=x + 1300 + dd, cigar=[(M, 100)], flag=99, mapq=255, nm=0, mpos=end + dd, mtid=0, isize=300 + k))
            recs.append(dict(qname=q, tid=0, pos=end + dd, cigar=[(S, 60), (M, 0), (S, 40)], flag=147, mapq=255, nm=0, mpos=x + 1300 + dd, mtid=0, isize=-(300 + k)))
    recs.sort(key=lambda t: t["pos"])
    b2 = Batch.from_records(recs)
    o = hostemu.run_k1(p, ann2, b2, grid=2, bed=bed2)
    _compare(o, oracle_lib.run_oracle(p, ann2, [b2]))
    assert o.n_candidates >= 20 * 3 * 2 - 20                       # (first mates at three of the five offsets, and the zero-length mates beside them)


def test_many_small_contigs_in_one_tile(oracle_lib):
    """Several contigs inside one 64-record tile: the records beyond the tile's first contig take the general code and find their
    contig themselves; for one with a long CIGAR the general code also counts "Alignment Blocks" and checks the operations
    (a 1-GPU / by-contig-sharded difference of 9 blocks in 60 k records found the missing count on the GPU)."""
    rows = []
    names = ["c%d" % k for k in range(12)]
    for k, nm in enumerate(names):
        rows.append(dict(contig=nm, type="gene", start=100, end=5000, strand="+-"[k % 2], gene_id="G%d" % k))
        rows.append(dict(contig=nm, type="exon", start=200, end=1200, strand="+-"[k % 2], gene_id="G%d" % k, exon_id="E%d" % k))
    ann = Annotation.from_rows(names, rows)
    recs = []
    for k in range(12):
        for j in range(5 + 7 * (k % 3)):
            cig = [(abi.CIG_M, 100)] if j % 3 else [(abi.CIG_M, 30), (abi.CIG_N, 50), (abi.CIG_M, 30), (abi.CIG_N, 50), (abi.CIG_M, 30)]
            recs.append(dict(qname="q%d_%d" % (k, j // 2), tid=k, pos=150 + 40 * j, cigar=cig, flag=99 if j % 2 == 0 else 147,
                             mapq=255, nm=0, mpos=300, mtid=k))
    b = Batch.from_records(recs)
    p = abi.default_params()
    r = oracle_lib.run_oracle(p, ann, [b])
    for grid in (1, 2):
        _compare(hostemu.run_k1(p, ann, b, grid=grid), r)
    assert r.gene_reads.sum() > 50


def test_names_sharing_a_64_bit_hash_stay_distinct(oracle_lib):
    """rsqc_batch.qhash2: one name in sixty takes the 64-bit hash of its predecessor in the sorted list of hashes (the two
    then collide wherever they meet in a gene); the second hashes stay as they were.  The oracle keyed by (qhash, qhash2) and
    the kernels (per-read emission of both words, frag_local / frag_count comparing both) must agree gene by gene -- and
    differ from the 64-bit-only answer.  (A partition absorbs 32 such entries; beyond that the stage reports
    RSQC_ERR_CAPACITY -- sixty-fold this rate, unreachable with real names -- instead of miscounting.)"""
    import copy
    ann = synth.make_annotation(seed=3, contigs=[("chrA", 200_000, 12)])
    batch = synth.make_reads(ann, 3000, seed=9, contig_lengths=np.array([200_000]))
    assert batch.qhash2 is not None
    batch.qname = None; batch.qname_off = None
    keys = np.unique(np.asarray(batch.qhash, np.uint64))
    remap = {int(k): int(keys[i - 1]) for i, k in enumerate(keys) if i % 60 == 5}
    batch.qhash = np.array([remap.get(int(h), int(h)) for h in batch.qhash], np.uint64)
    p = abi.default_params()
    r = oracle_lib.run_oracle(p, ann, [batch])
    o = hostemu.run_k1(p, ann, batch, grid=2)
    _compare(o, r)
    narrow = copy.copy(batch); narrow.qhash2 = None
    r64 = oracle_lib.run_oracle(p, ann, [narrow])
    _compare(hostemu.run_k1(p, ann, narrow, grid=2), r64)
    assert int(r.gene_fragments.sum()) > int(r64.gene_fragments.sum()) + 10


def dense_case(seed, n_genes=7, n_reads=14000, span=2400):
    """Short one-block reads packed onto a few overlapping genes (both strands, a globin, an rRNA gene, exons that overlap exons of
    their own and of other genes, an interval under three exons): most 64-record feature-stage calls of one-block records then
    have every block inside ONE elementary interval -- the wave-uniform path of classify_ei_kernel (rsqc_k1.h, k1e_uniform1) --
    and the calls that straddle a breakpoint take the general path next to them."""
    from rnaseqc_amd.abi import CIG_M as M, CIG_S as S, CIG_N as N, CIG_I as I
    rng = np.random.default_rng(seed)
    rows = []
    for g in range(n_genes):
        gs = int(rng.integers(1, span - 500)); ge = gs + int(rng.integers(120, 480))
        strand = "+-."[int(rng.integers(0, 3))]
        ttype = "rRNA" if g % 5 == 1 else "protein_coding"
        name = "HBB" if g % 6 == 2 else "N%d" % g
        rows.append(dict(contig="c", type="gene", start=gs, end=ge, strand=strand, gene_id="G%d" % g, gene_name=name, transcript_type=ttype))
        for e in range(int(rng.integers(1, 4))):
            es = int(rng.integers(gs, ge - 60)); ee = min(ge, es + int(rng.integers(60, 300)))
            rows.append(dict(contig="c", type="exon", start=es, end=ee, strand=strand, gene_id="G%d" % g, exon_id="G%d_e%d" % (g, e),
                             gene_name=name, transcript_type=ttype))
    ann = Annotation.from_rows(["c"], rows)
    recs = []
    starts = np.sort(rng.integers(0, span, n_reads))
    for pos in starts:
        u = rng.random()
        if u < 0.85:
            cig = [(M, int(rng.integers(0 if rng.random() < 0.02 else 1, 14)))]
            if rng.random() < 0.2:
                cig = [(S, 3)] + cig
            if rng.random() < 0.1:
                cig = cig + [(I, 2)]
        elif u < 0.95:
            cig = [(M, int(rng.integers(1, 30))), (N, int(rng.choice([1, 40, 200]))), (M, int(rng.integers(1, 30)))]
        else:
            cig = [(M, 10), (N, 30), (M, 10), (N, 30), (M, 10)]
        flag = (0x1 if rng.random() < 0.9 else 0) | (0x2 if rng.random() < 0.9 else 0) | (0x10 if rng.random() < 0.5 else 0) | \
               (0x40 if rng.random() < 0.5 else 0x80) | (0x400 if rng.random() < 0.1 else 0)
        recs.append(dict(qname="q%d" % int(rng.integers(0, n_reads // 2)), tid=0, pos=int(pos), cigar=cig, flag=flag,
                         mapq=int(rng.choice([0, 3, 60, 255, 255, 255])), nm=int(rng.integers(0, 9)) if rng.random() < 0.9 else None,
                         mpos=int(pos) + int(rng.integers(0, 300)), mtid=0))
    return ann, Batch.from_records(recs)


def test_cached_interval_does_not_survive_a_contig_change(oracle_lib):
    """The uniform path keeps the last interval it looked up per wave (K1eTables::ucache).  Positions start again on every contig: the
    same coordinates that lie inside an exon on the first contig are intergenic on the second and inside another gene's exon on the
    third -- a cached [lo, hi) carried across the boundary would count the later contigs' reads to the first contig's gene."""
    from rnaseqc_amd.abi import CIG_M as M
    rows = [dict(contig="a", type="gene", start=1000, end=3000, strand="+", gene_id="GA", gene_name="NA", transcript_type="protein_coding"),
            dict(contig="a", type="exon", start=1000, end=3000, strand="+", gene_id="GA", exon_id="GA_e0", gene_name="NA", transcript_type="protein_coding"),
            dict(contig="b", type="gene", start=9000, end=9500, strand="+", gene_id="GB", gene_name="NB", transcript_type="protein_coding"),
            dict(contig="b", type="exon", start=9000, end=9500, strand="+", gene_id="GB", exon_id="GB_e0", gene_name="NB", transcript_type="protein_coding"),
            dict(contig="c", type="gene", start=900, end=3100, strand="-", gene_id="GC", gene_name="NC", transcript_type="protein_coding"),
            dict(contig="c", type="exon", start=900, end=3100, strand="-", gene_id="GC", exon_id="GC_e0", gene_name="NC", transcript_type="protein_coding")]
    ann = Annotation.from_rows(["a", "b", "c"], rows)
    rng = np.random.default_rng(5)
    recs = []
    for tid in range(3):
        for pos in np.sort(rng.integers(1200, 2600, 700)):
            recs.append(dict(qname="q%d_%d" % (tid, len(recs) // 2), tid=tid, pos=int(pos), cigar=[(M, 40)], flag=0x1 | 0x2 | (0x40 if len(recs) % 2 else 0x80),
                             mapq=255, nm=0, mpos=int(pos) + 50, mtid=tid))
    batch = Batch.from_records(recs)
    p = abi.default_params()
    want = oracle_lib.run_oracle(p, ann, [batch])
    assert list(want.gene_reads) == [700, 0, 700]
    for grid in (1, 2):
        o = hostemu.run_k1(p, ann, batch, grid=grid, want_cov=False, coarse=False)
        _compare(o, want)
        assert o.n_ucache_hits >= 8, (o.n_ucache_hits, o.n_uniform)


@pytest.mark.parametrize("seed", range(3))
def test_dense_one_block_tiles_take_the_uniform_path(oracle_lib, seed):
    ann, batch = dense_case(700 + seed)
    for kw in (dict(), dict(unpaired=1, mapq_threshold=3), dict(stranded=abi.STRAND_FORWARD)):
        p = abi.default_params(**kw)
        want = oracle_lib.run_oracle(p, ann, [batch])
        ref = hostemu.run(p, ann, batch, mode=1, want_cov=True)
        for grid in (1, 3):
            o = hostemu.run_k1(p, ann, batch, grid=grid, want_cov=True, coarse=False)
            _compare(o, want, ref.cov)
            if "stranded" not in kw:
                assert o.n_uniform >= 40, o.n_uniform        # (of ~190 one-block calls)
                assert 10 <= o.n_ucache_hits < o.n_uniform, (o.n_ucache_hits, o.n_uniform)    # ... many in the interval of the call before
            else:
                assert o.n_uniform == 0 and o.n_ucache_hits == 0     # --stranded: the containing exons are a per-lane property
    assert want.gene_reads.sum() > 300 and want.counters[abi.COUNTER_NAMES.index("rRNA Reads")] > 0


def junction_case(seed, n_per_junction=420):
    """Spliced reads packed onto the junctions of a few genes: two genes that share an exon stretch (two-gene sets, the intersection over
    the blocks drops one of them), a globin, an rRNA gene on the other strand, an exon under a second exon of its own gene.  Most reads
    end their first block ON the last base of an exon and start the second on the first base of the next (the rule for a spliced
    alignment); some splice inside the exons, some carry soft clips / insertions (another aligned length), a zero-length block, low
    MAPQ, the duplicate flag.  The 64-record calls of the two-block queue are then mostly ONE junction's reads: k1e_uniform2."""
    from rnaseqc_amd.abi import CIG_M as M, CIG_S as S, CIG_N as N, CIG_I as I
    rng = np.random.default_rng(seed)
    rows = []
    def gene(gid, name, strand, ttype, exons, extra=()):
        gs, ge = min(e[0] for e in exons), max(e[1] for e in exons)
        rows.append(dict(contig="c", type="gene", start=gs, end=ge, strand=strand, gene_id=gid, gene_name=name, transcript_type=ttype))
        for k, (es, ee) in enumerate(list(exons) + list(extra)):
            rows.append(dict(contig="c", type="exon", start=es, end=ee, strand=strand, gene_id=gid, exon_id="%s_e%d" % (gid, k), gene_name=name, transcript_type=ttype))
    g1 = [(1000, 1120), (1400, 1490), (1800, 1950)]
    gene("G1", "N1", "+", "protein_coding", g1)
    gene("G2", "N2", "+", "protein_coding", [(1400, 1490), (2300, 2380)])            # shares G1's middle exon, then leaves it
    g3 = [(3000, 3100), (3300, 3420), (3700, 3800)]
    gene("G3", "HBB", "-", "protein_coding", g3, extra=[(3320, 3400)])               # an exon under a second exon of its own gene
    g4 = [(5000, 5090), (5200, 5300)]
    gene("G4", "N4", "-", "rRNA", g4)
    ann = Annotation.from_rows(["c"], rows)
    junctions = [(g1[0][1], g1[1][0]), (g1[1][1], g1[2][0]), (1490, 2300), (g3[0][1], g3[1][0]), (g3[1][1], g3[2][0]), (g4[0][1], g4[1][0])]
    recs = []
    for jn, (e_end, s_next) in enumerate(junctions):                               # 1-based closed exon end / next exon start
        noise = 0.003 if jn % 2 == 0 else 0.15                                     # every other junction: many reads spliced somewhere else
        for _ in range(n_per_junction):
            l0, l1 = int(rng.integers(1, 30)), int(rng.integers(1, 30))
            shift = 0 if rng.random() >= noise else int(rng.integers(-6, 7))        # ... into the intron or inside the exon
            pos = e_end - l0 + shift                                               # 0-based start: the block's last base is e_end + shift
            gap = (s_next - 1) - (pos + l0)
            if rng.random() < noise:
                gap += int(rng.integers(-4, 5))
            if gap < 1:
                gap = 1
            cig = [(M, l0), (N, gap), (M, 0 if rng.random() < 0.01 else l1)]
            v = rng.random()
            if v < 0.12:
                cig = [(S, 4)] + cig
            elif v < 0.2:
                cig = cig + [(I, 3)]
            flag = (0x1 if rng.random() < 0.9 else 0) | (0x2 if rng.random() < 0.9 else 0) | (0x10 if rng.random() < 0.5 else 0) | \
                   (0x40 if rng.random() < 0.5 else 0x80) | (0x400 if rng.random() < 0.1 else 0)
            recs.append(dict(qname="q%d" % int(rng.integers(0, n_per_junction * 3)), tid=0, pos=int(pos), cigar=cig, flag=flag,
                             mapq=int(rng.choice([0, 3, 60, 255, 255, 255, 255])), nm=int(rng.integers(0, 7)) if rng.random() < 0.9 else None,
                             mpos=int(pos) + int(rng.integers(0, 300)), mtid=0))
    for pos in rng.integers(900, 5400, 600):                                       # one-block reads between them
        recs.append(dict(qname="s%d" % len(recs), tid=0, pos=int(pos), cigar=[(M, int(rng.integers(5, 40)))], flag=0x1 | 0x2 | 0x40, mapq=255, nm=0,
                         mpos=int(pos) + 100, mtid=0))
    recs.sort(key=lambda r: r["pos"])
    return ann, Batch.from_records(recs)


@pytest.mark.parametrize("seed", range(3))
def test_spliced_tiles_take_the_two_block_uniform_path(oracle_lib, seed):
    ann, batch = junction_case(900 + seed)
    for kw in (dict(), dict(unpaired=1, mapq_threshold=3), dict(stranded=abi.STRAND_REVERSE)):
        p = abi.default_params(**kw)
        want = oracle_lib.run_oracle(p, ann, [batch])
        ref = hostemu.run(p, ann, batch, mode=1, want_cov=True)
        for grid in (1, 3):
            o = hostemu.run_k1(p, ann, batch, grid=grid, want_cov=True, coarse=False)     # the product's configuration: general path only
            _compare(o, want, ref.cov)
            assert o.n_uniform2 == 0
            o = hostemu.run_k1(p, ann, batch, grid=grid, want_cov=True, coarse=True)      # the superset build: -DK1E_UNIFORM2=1
            _compare(o, want, ref.cov)
            if "stranded" not in kw:
                assert o.n_uniform2 >= 8, o.n_uniform2           # (of ~40 two-block calls; the calls with a read spliced elsewhere take the general path)
            else:
                assert o.n_uniform2 == 0                         # --stranded: the containing exons are a per-lane property
    names = list(ann.gene_ids[:ann.n_genes_listed]) if hasattr(ann, "gene_ids") else []
    assert want.gene_reads.sum() > 300 and want.counters[abi.COUNTER_NAMES.index("rRNA Reads")] > 0
    assert int((want.gene_reads > 0).sum()) >= 4, names
