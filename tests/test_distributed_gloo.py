"""World size 2 over gloo: the contig-sharded merge (rnaseqc_amd/distributed.py) reproduces the single-process result,
including the two order-dependent outputs -- the fragment-size histogram with its first-N cut-off and Read Length on
input with MIXED read lengths.  CPU test: the per-shard compute is the oracle (no GPU in this container).  GPU test
(-m gpu): the per-shard compute is the HIP path, two processes sharing the box's GPU; bench.py runs the same merge
over RCCL."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CONTIGS = [("cA", 2_000_000, 150), ("cB", 1_500_000, 120), ("cC", 900_000, 60), ("cD", 700_000, 50)]
FRAGMENT_SAMPLES = 300          # far below what the shards produce: the cut-off decides


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _inputs():
    from rnaseqc_amd import synth
    ann = synth.make_annotation(seed=21, contigs=CONTIGS)
    batch = synth.make_reads(ann, 15000, seed=22, dup_frac=0.05, contig_lengths=np.array([c[1] for c in CONTIGS]))
    # mixed read lengths: Read Length becomes a genuine state machine over the file (src/RNASeQC.cpp:275-278)
    rng = np.random.default_rng(5)
    batch.l_qseq = rng.choice(np.array([36, 50, 76, 101, 150], np.uint16), size=batch.n).astype(np.uint16)
    bed = synth.make_bed(ann, min_len=300)
    return ann, batch, bed


def _shards(batch, rank_of, rank, world, split=True):
    """The records of this rank's contigs as batches that are contiguous file ranges (one per run of records)."""
    tid = batch.tid_per_record()
    mine = np.isin(tid, np.flatnonzero(rank_of == rank))
    if rank == world - 1:
        mine |= tid < 0                       # the unmapped tail goes to the last rank
    idx = np.flatnonzero(mine)
    runs = np.split(idx, np.flatnonzero(np.diff(idx) != 1) + 1) if len(idx) else []
    parts = []
    for r in runs:                            # split long runs further: several batches per contig
        lo, hi = int(r[0]), int(r[-1]) + 1
        mid = lo + (hi - lo) // 3
        parts += [batch.slice(lo, mid), batch.slice(mid, hi)] if (split and mid > lo) else [batch.slice(lo, hi)]
    return parts


def _check(merged, whole):
    np.testing.assert_array_equal(merged.gene_reads, whole.gene_reads)
    np.testing.assert_array_equal(merged.gene_unique, whole.gene_unique)
    np.testing.assert_array_equal(merged.gene_fragments, whole.gene_fragments)
    np.testing.assert_array_equal(merged.counters, whole.counters)
    np.testing.assert_allclose(merged.exon_reads, whole.exon_reads, rtol=1e-9, atol=1e-9)
    np.testing.assert_array_equal(merged.gene_cov_valid, whole.gene_cov_valid)
    np.testing.assert_allclose(merged.gene_cov_mean, whole.gene_cov_mean, rtol=1e-9, atol=1e-9)
    np.testing.assert_allclose(merged.gene_cov_std, whole.gene_cov_std, rtol=1e-9, atol=1e-9)
    np.testing.assert_array_equal(np.isnan(merged.gene_cov_cv), np.isnan(whole.gene_cov_cv))
    np.testing.assert_array_equal(merged.exon_cv_valid, whole.exon_cv_valid)
    np.testing.assert_allclose(merged.exon_cv, whole.exon_cv, rtol=1e-8, atol=1e-9)
    np.testing.assert_array_equal(merged.bias_three, whole.bias_three)
    assert merged.read_length == whole.read_length
    np.testing.assert_array_equal(merged.fragment_size, whole.fragment_size)
    np.testing.assert_array_equal(merged.fragment_count, whole.fragment_count)
    assert int(np.asarray(whole.fragment_count).sum()) == FRAGMENT_SAMPLES
    assert merged.fragment_samples_remaining == whole.fragment_samples_remaining == 0


def _worker(rank, world, port, out_dir, use_gpu):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist
    from rnaseqc_amd import abi, synth, distributed
    from oracle import binding
    dist.init_process_group("gloo", rank=rank, world_size=world)
    ann, batch, bed = _inputs()
    tid = batch.tid_per_record()
    per_contig = np.array([(tid == k).sum() for k in range(len(CONTIGS))])
    rank_of = distributed.assign_contigs(per_contig, world)
    assert len(set(rank_of.tolist())) == world
    # LPT does not keep contigs in file order on a rank: that is the point of composing per batch
    parts = _shards(batch, rank_of, rank, world)
    owned = distributed.owned_mask(rank_of, rank)
    p = abi.default_params(fragment_samples=FRAGMENT_SAMPLES)
    if use_gpu:
        from rnaseqc_amd import engine
        e = engine.Engine(p)
        e.set_annotation(ann, owned); e.set_bed(bed)
        if use_gpu == 2:                          # the rank's file ranges as ONE batch (rsqc_batch.seg_file_index): one kernel launch
            from rnaseqc_amd.model import Batch
            parts = [Batch.concat_ranges(_shards(batch, rank_of, rank, world, split=False))]
            assert len(parts[0].seg_file_index) >= 2
        for b in parts:
            e.submit(b)
        local = e.finalize()
        shard = e.shard_summary()
        e.close()
    else:
        o = binding.Oracle(abi.default_params(fragment_samples=0xFFFFFFFF))      # the cut-off is applied after the merge
        o.set_annotation(ann, owned); o.set_bed(bed)
        o.enable_trace()
        for b in parts:
            o.submit(b)
        local = o.finalize()
        shard = o.shard_info()
        o.close()
    merged = distributed.merge_results(local, dist, shard=shard, fragment_samples=FRAGMENT_SAMPLES)
    if rank == 0:
        whole = binding.run_oracle(p, ann, [batch], bed=bed)
        assert len({int(x) for si in merged.shard_infos for x in si.rl_state}) > 1      # the walks really differ
        _check(merged, whole)
        open(os.path.join(out_dir, "ok"), "w").write("ok")
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_contig_sharding_gloo(tmp_path, oracle_lib):
    import torch.multiprocessing as mp
    mp.spawn(_worker, args=(2, _free_port(), str(tmp_path), False), nprocs=2, join=True)
    assert (tmp_path / "ok").exists()


@pytest.mark.gpu
def test_two_rank_contig_sharding_hip(tmp_path, oracle_lib):
    import torch.multiprocessing as mp
    mp.spawn(_worker, args=(2, _free_port(), str(tmp_path), True), nprocs=2, join=True)
    assert (tmp_path / "ok").exists()


@pytest.mark.gpu
def test_two_rank_contig_sharding_hip_one_batch_of_ranges(tmp_path, oracle_lib):
    """The same with every rank's contigs submitted as ONE batch of non-adjacent file ranges: per-range
    Read-Length functions and file indices, one launch per rank."""
    import torch.multiprocessing as mp
    mp.spawn(_worker, args=(2, _free_port(), str(tmp_path), 2), nprocs=2, join=True)
    assert (tmp_path / "ok").exists()


def test_fasta_histogram_is_additive(oracle_lib):
    """--fasta: both mates of a GC fragment lie in one exon, hence on one contig: per-shard histograms add up."""
    from rnaseqc_amd import abi, synth, distributed
    ann, batch, _ = _inputs()
    ref = synth.make_reference([c[1] for c in CONTIGS], seed=23)
    tid = batch.tid_per_record()
    rank_of = distributed.assign_contigs(np.array([(tid == k).sum() for k in range(len(CONTIGS))]), 2)
    p = abi.default_params()
    locals_ = [oracle_lib.run_oracle(p, ann, _shards(batch, rank_of, r, 2), owned=distributed.owned_mask(rank_of, r), reference=ref)
               for r in range(2)]
    whole = oracle_lib.run_oracle(p, ann, [batch], reference=ref)
    np.testing.assert_array_equal(locals_[0].gc_bins + locals_[1].gc_bins, whole.gc_bins)
    assert int(whole.gc_bins.sum()) > 100


def test_read_length_composition_matches_the_state_machine():
    from rnaseqc_amd import distributed
    rng = np.random.default_rng(3)
    for trial in range(30):
        n = int(rng.integers(1, 400))
        lq = rng.choice([36, 50, 76, 101, 150], size=n)
        span = np.where(rng.random(n) < 0.15, lq + rng.integers(50, 5000, n), lq)
        r = 0
        for s_, l_ in zip(span, lq):
            if s_ > r:
                r = int(l_)
        cuts = sorted({0, n, *rng.integers(0, n + 1, 4).tolist()})
        infos, base = [], 0
        for lo, hi in zip(cuts[:-1], cuts[1:]):
            k, v = distributed.read_length_transfer(span[lo:hi], lq[lo:hi])
            infos.append(distributed.ShardInfo(np.array([lo], np.uint64), np.array([hi - lo], np.uint64), np.array([0, len(k)], np.uint32),
                                               k, v, np.zeros(0, np.uint64), np.zeros(0, np.uint32)))
        assert distributed.compose_read_length(infos[::-1]) == r          # order of the shards does not matter: file index does


def test_lpt_assignment():
    from rnaseqc_amd import distributed
    r = distributed.assign_contigs([100, 10, 90, 20, 80], 2)
    loads = [sum(x for x, k in zip([100, 10, 90, 20, 80], r) if k == j) for j in range(2)]
    assert abs(loads[0] - loads[1]) <= 40 and set(r) == {0, 1}   # LPT: 130 vs 170


def _merge_worker(rank, port, out):
    """One packed gather for the usual step, a second exactly sized one when a rank's payload exceeds the fixed width (a --bed run's
    kept samples): both forms must hand every rank every shard's arrays unchanged."""
    import torch
    import torch.distributed as dist
    sys.path.insert(0, ROOT)
    from rnaseqc_amd import distributed
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=2)
    try:
        for big in (5, 20000):                                          # below / above the first gather's 8192 words
            n = big if rank == 1 else 5
            sh = distributed.ShardInfo(np.array([rank * 100], np.uint64), np.array([50], np.uint64), np.array([0, 1], np.uint32), np.array([151], np.uint32),
                                       np.array([150], np.int32), np.arange(n, dtype=np.uint64) * 2 + rank, (np.arange(n) % 500 + 100).astype(np.int64))
            rl, sizes, counts, rem, infos = distributed.merge_order_dependent(sh, dist, torch.device("cpu"), 1000000)
            assert rl == 150 and len(infos) == 2
            assert len(infos[0].sample_size) == 5 and len(infos[1].sample_size) == big
            assert (infos[1].sample_file_index == np.arange(big) * 2 + 1).all() and (infos[0].batch_file_index == [0]).all() and (infos[1].batch_file_index == [100]).all()
            assert int(np.asarray(counts).sum()) == 5 + big and rem == 1000000 - 5 - big
        if rank == 0:
            open(os.path.join(out, "ok"), "w").write("ok")
    finally:
        dist.destroy_process_group()


def test_order_dependent_merge_one_or_two_gathers(tmp_path):
    import torch.multiprocessing as mp
    mp.spawn(_merge_worker, args=(_free_port(), str(tmp_path)), nprocs=2, join=True)
    assert (tmp_path / "ok").exists()
