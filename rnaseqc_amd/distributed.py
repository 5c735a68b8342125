"""Multi-GPU host logic: contig -> rank assignment and the end-of-file merge.

The path shards by contig: genes never span contigs, both mates of a
fragment counted to a gene are on the gene's contig, coverage and bias are per gene.  So a
rank that owns a set of contigs produces final per-gene values for them and partial sums for
everything that is additive; the only exchange is one sum-reduction of the count vectors and
scalar counters (RCCL all_reduce on the GPU box, gloo in the CPU tests) plus an ownership-
masked merge of the per-gene statistics.
"""
from __future__ import annotations

import numpy as np


def assign_contigs(records_per_contig, world: int):
    """Longest-processing-time bin packing of contigs onto ranks; returns rank per contig."""
    order = np.argsort(-np.asarray(records_per_contig, dtype=np.int64), kind="stable")
    load = np.zeros(world, dtype=np.int64)
    rank_of = np.zeros(len(records_per_contig), dtype=np.int32)
    for c in order:
        r = int(np.argmin(load))
        rank_of[c] = r
        load[r] += int(records_per_contig[c])
    return rank_of


def owned_mask(rank_of, rank: int) -> np.ndarray:
    return (np.asarray(rank_of) == rank).astype(np.uint8)


class MergedResults:
    pass


def merge_results(local, dist=None, device=None):
    """All-reduce the additive parts of a rank's abi.Results and merge the owner-only parts.

    `dist` is torch.distributed (initialised) or None for a single rank.  Additive: gene reads /
    unique / fragments, exon fractions, scalar counters, fragment-size histogram.  Owner-only
    (zero on non-owners, so a sum is a merge): per-gene coverage mean/std/CV/valid, exon CV,
    bias accumulators.  --fasta: the fragment GC histogram is additive.  Read Length: exact when every shard's eligible records share one l_qseq
    (see DESIGN.md); shards are combined in contig order.
    """
    import torch
    m = MergedResults()

    def allsum(a, dtype):
        t = torch.as_tensor(np.ascontiguousarray(a).astype(dtype), device=device)
        if dist is not None:
            dist.all_reduce(t)
        return t.cpu().numpy()

    m.gene_reads = allsum(local.gene_reads, np.int64).astype(np.uint64)
    m.gene_unique = allsum(local.gene_unique, np.int64).astype(np.uint64)
    m.gene_fragments = allsum(local.gene_fragments, np.int64).astype(np.uint64)
    m.exon_reads = allsum(local.exon_reads, np.float64)
    m.exon_hit = (allsum(local.exon_hit, np.int64) > 0).astype(np.uint8)
    m.counters = allsum(local.counters, np.int64).astype(np.uint64)
    valid = local.gene_cov_valid.astype(bool)
    m.gene_cov_valid = (allsum(local.gene_cov_valid, np.int64) > 0).astype(np.uint8)
    m.gene_cov_mean = allsum(np.where(valid, local.gene_cov_mean, 0.0), np.float64)
    m.gene_cov_std = allsum(np.where(valid, local.gene_cov_std, 0.0), np.float64)
    # CV may be NaN (mean 0) on its owner: carry the NaN as a flag, sum the finite part
    cv_nan = valid & np.isnan(local.gene_cov_cv)
    cv_fin = np.where(valid & ~np.isnan(local.gene_cov_cv), local.gene_cov_cv, 0.0)
    m.gene_cov_cv = allsum(cv_fin, np.float64)
    m.gene_cov_cv[allsum(cv_nan.astype(np.int64), np.int64) > 0] = np.nan
    ev = local.exon_cv_valid.astype(bool)
    m.exon_cv_valid = (allsum(local.exon_cv_valid, np.int64) > 0).astype(np.uint8)
    m.exon_cv = allsum(np.where(ev, local.exon_cv, 0.0), np.float64)
    m.bias_three = allsum(local.bias_three, np.int64).astype(np.uint64)
    m.bias_five = allsum(local.bias_five, np.int64).astype(np.uint64)
    # Read Length: (rank, value) -- the last shard (highest rank with mapped records) decides when all
    # shards are uniform; ranks hold contigs in increasing order in the benchmarks and the CLI
    world = dist.get_world_size() if dist is not None else 1
    rank = dist.get_rank() if dist is not None else 0
    rl = np.zeros(world, dtype=np.int64)
    rl[rank] = local.read_length
    rl = allsum(rl, np.int64)
    nz = np.flatnonzero(rl)
    m.read_length = int(rl[nz[-1]]) if len(nz) else 0
    m.read_length_per_rank = rl
    # --fasta: both mates of a GC fragment lie in one exon, hence on one contig -> the histogram is additive; exon GC
    # values depend on the annotation and the reference only (identical on every rank)
    m.have_reference = int(getattr(local, "have_reference", 0))
    if m.have_reference:
        m.gc_bins = allsum(local.gc_bins, np.int64).astype(np.uint64)
        m.gc_out_of_range = int(allsum(np.array([local.gc_out_of_range]), np.int64)[0])
        m.exon_gc = np.array(local.exon_gc)
    return m
