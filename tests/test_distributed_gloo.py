"""CPU, world_size 2 over gloo: the contig-sharded merge (rnaseqc_amd/distributed.py) reproduces
the single-process result.  The per-shard compute is done by the oracle here (no GPU in this
container); on the GPU box bench.py runs the same merge over RCCL."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist
    from rnaseqc_amd import abi, synth, distributed
    from oracle import binding
    dist.init_process_group("gloo", rank=rank, world_size=world)
    contigs = [("cA", 2_000_000, 150), ("cB", 1_500_000, 120), ("cC", 900_000, 60)]
    ann = synth.make_annotation(seed=21, contigs=contigs)
    batch = synth.make_reads(ann, 15000, seed=22, dup_frac=0.05, contig_lengths=np.array([c[1] for c in contigs]))
    tid = batch.tid_per_record()
    per_contig = np.array([(tid == k).sum() for k in range(3)])
    rank_of = distributed.assign_contigs(per_contig, world)
    mine = np.isin(tid, np.flatnonzero(rank_of == rank))
    if rank == 0:
        mine |= tid < 0                       # the unmapped tail goes to rank 0
    idx = np.flatnonzero(mine)
    # records of a rank are contiguous runs per contig: build the shard from those runs
    runs = np.split(idx, np.flatnonzero(np.diff(idx) != 1) + 1) if len(idx) else []
    parts = [batch.slice(int(r[0]), int(r[-1]) + 1) for r in runs]
    p = abi.default_params()
    ref = synth.make_reference([c[1] for c in contigs], seed=23)
    local = binding.run_oracle(p, ann, parts, owned=distributed.owned_mask(rank_of, rank), reference=ref)
    merged = distributed.merge_results(local, dist)
    if rank == 0:
        whole = binding.run_oracle(p, ann, [batch], reference=ref)
        np.testing.assert_array_equal(merged.gc_bins, whole.gc_bins)
        assert int(whole.gc_bins.sum()) > 300 and merged.gc_out_of_range == whole.gc_out_of_range
        np.testing.assert_array_equal(merged.gene_reads, whole.gene_reads)
        np.testing.assert_array_equal(merged.gene_unique, whole.gene_unique)
        np.testing.assert_array_equal(merged.gene_fragments, whole.gene_fragments)
        np.testing.assert_array_equal(merged.counters, whole.counters)
        np.testing.assert_allclose(merged.exon_reads, whole.exon_reads, rtol=1e-12, atol=1e-9)
        np.testing.assert_array_equal(merged.gene_cov_valid, whole.gene_cov_valid)
        np.testing.assert_array_equal(merged.gene_cov_mean, whole.gene_cov_mean)
        np.testing.assert_array_equal(merged.gene_cov_std, whole.gene_cov_std)
        np.testing.assert_array_equal(np.isnan(merged.gene_cov_cv), np.isnan(whole.gene_cov_cv))
        np.testing.assert_array_equal(merged.exon_cv_valid, whole.exon_cv_valid)
        np.testing.assert_array_equal(merged.exon_cv, whole.exon_cv)
        np.testing.assert_array_equal(merged.bias_three, whole.bias_three)
        assert merged.read_length == whole.read_length
        open(os.path.join(out_dir, "ok"), "w").write("ok")
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_contig_sharding_gloo(tmp_path, oracle_lib):
    import torch.multiprocessing as mp
    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    assert (tmp_path / "ok").exists()


def test_lpt_assignment():
    from rnaseqc_amd import distributed
    r = distributed.assign_contigs([100, 10, 90, 20, 80], 2)
    loads = [sum(x for x, k in zip([100, 10, 90, 20, 80], r) if k == j) for j in range(2)]
    assert abs(loads[0] - loads[1]) <= 40 and set(r) == {0, 1}   # LPT: 130 vs 170
