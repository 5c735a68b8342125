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


from dataclasses import dataclass


@dataclass
class ShardInfo:
    """The order-dependent outputs of one shard (include/rnaseqc_amd.h, rsqc_shard_info)."""
    batch_file_index: np.ndarray       # [B] file index of the first record of every submitted batch
    batch_records: np.ndarray          # [B]
    rl_offset: np.ndarray              # [B + 1] into rl_span / rl_state
    rl_span: np.ndarray                # Read-Length transfer function of the batch: ascending span keys ...
    rl_state: np.ndarray               # ... and the state the batch leaves when entered with a state below the key
    sample_file_index: np.ndarray      # fragment-size samples kept by the shard (its first N by file index, any order)
    sample_size: np.ndarray


def read_length_transfer(span, lq):
    """The Read-Length transfer function (src/RNASeQC.cpp:275-278) of a run of eligible records, as the table
    rsqc_shard_info describes: keys = the prefix maxima of span, value k = the final state of the walk that starts by
    firing prefix maximum k.  Literal restatement for the tests (the device kernel runs all the walks at once)."""
    span = np.asarray(span, dtype=np.int64); lq = np.asarray(lq, dtype=np.int64)
    keys, vals = [], []
    cur = 0
    starts = []
    for i in range(len(span)):
        if span[i] > cur:
            cur = int(span[i]); starts.append(i)
    for i in starts:
        r = int(lq[i])
        for j in range(i + 1, len(span)):
            if span[j] > r:
                r = int(lq[j])
        keys.append(int(span[i])); vals.append(r)
    return np.array(keys, np.uint32), np.array(vals, np.int32)


def compose_read_length(infos):
    """Read Length of the whole file from the shards' per-batch transfer functions: batches of all shards in
    ascending file index, from state 0."""
    items = []
    for si in infos:
        for b in range(len(si.batch_file_index)):
            lo, hi = int(si.rl_offset[b]), int(si.rl_offset[b + 1])
            items.append((int(si.batch_file_index[b]), si.rl_span[lo:hi], si.rl_state[lo:hi]))
    items.sort(key=lambda t: t[0])
    r = 0
    for _, keys, vals in items:
        k = int(np.searchsorted(keys, r, side="right"))          # first key > r (keys ascend)
        if k < len(keys):
            r = int(vals[k])
    return r


def merge_fragment_samples(infos, max_samples: int):
    """First `max_samples` samples of the union in file order -> (sizes ascending, counts), remaining."""
    f = np.concatenate([np.asarray(si.sample_file_index, np.uint64) for si in infos]) if infos else np.zeros(0, np.uint64)
    z = np.concatenate([np.asarray(si.sample_size, np.uint32) for si in infos]) if infos else np.zeros(0, np.uint32)
    keep = min(len(f), int(max_samples))
    if keep < len(f):
        idx = np.argpartition(f, keep - 1)[:keep] if keep else np.zeros(0, np.int64)
        z = z[idx]
    sizes, counts = np.unique(z.astype(np.int64), return_counts=True)
    return sizes.astype(np.int64), counts.astype(np.uint64), int(max_samples) - keep


def merge_order_dependent(shard: ShardInfo, dist=None, device=None, fragment_samples: int = 1000000):
    """Gathers every rank's ShardInfo (a handful of small all_gathers) and returns
    (read_length, fragment sizes, fragment counts, samples remaining, [ShardInfo per rank])."""
    import torch
    world = dist.get_world_size() if dist is not None else 1
    rank = dist.get_rank() if dist is not None else 0

    # Seven 1-D arrays of different lengths per rank travel as ONE packed int64 buffer [7 lengths | payload] of a fixed width, so that the
    # usual step costs ONE collective and one host copy (round 6; round 5: a gather of the lengths, then a gather of the payload padded to
    # the longest rank: two latency-bound collectives with a device-to-host copy behind each, most of `collective_host_merge_ms`).  A
    # rank whose payload does not fit the fixed width (a --bed run's kept samples) makes every rank take a second, exactly sized gather.
    # uint32 / uint64 travel as int64: gloo and RCCL both carry it, values are far below 2^63.
    fields = [np.ascontiguousarray(x).astype(np.int64).ravel() for x in
              (shard.batch_file_index, shard.batch_records, shard.rl_offset, shard.rl_span, shard.rl_state, shard.sample_file_index, shard.sample_size)]
    if dist is None:
        cols = [[f] for f in fields]
    else:
        W0 = 8192                                                                     # int64 words of payload in the first gather (64 KB per rank)
        lens = np.array([len(f) for f in fields], np.int64)
        cat = np.concatenate(fields) if int(lens.sum()) else np.zeros(0, np.int64)
        mine = np.zeros(7 + W0, np.int64)
        mine[:7] = lens
        if len(cat) <= W0:
            mine[7:7 + len(cat)] = cat
        buf = torch.from_numpy(mine).to(device)
        out = [torch.empty_like(buf) for _ in range(world)]
        dist.all_gather(out, buf)
        flats = [o.cpu().numpy() for o in out]
        all_lens = np.stack([f[:7] for f in flats])                                   # [world, 7]
        payload = [f[7:] for f in flats]
        width = int(all_lens.sum(axis=1).max())
        if width > W0:                                                                # (every rank sees the same lengths: the same decision everywhere)
            mine2 = np.zeros(width, np.int64)
            mine2[:len(cat)] = cat
            buf2 = torch.from_numpy(mine2).to(device)
            out2 = [torch.empty_like(buf2) for _ in range(world)]
            dist.all_gather(out2, buf2)
            payload = [o.cpu().numpy() for o in out2]
        cols = [[] for _ in fields]
        for k in range(world):
            flat = payload[k]
            at = 0
            for j in range(len(fields)):
                n = int(all_lens[k, j]); cols[j].append(flat[at:at + n].copy()); at += n
    infos = [ShardInfo(*(c[k] for c in cols)) for k in range(world)]
    sizes, counts, remaining = merge_fragment_samples(infos, fragment_samples)
    return compose_read_length(infos), sizes, counts, remaining, infos


class MergedResults:
    pass


def merge_results(local, dist=None, device=None, shard: ShardInfo | None = None, fragment_samples: int = 1000000):
    """All-reduce the additive parts of a rank's abi.Results and merge the owner-only parts.

    `shard` (this rank's ShardInfo: Engine.shard_summary(), or built from the oracle's trace in the CPU tests) carries
    the two order-dependent outputs: with it, Read Length is composed exactly from the per-batch transfer functions of
    all ranks in file order, and the fragment-size histogram is rebuilt from the ranks' samples with the
    --fragment-samples cut-off applied in file order (src/Expression.cpp:482-540).  Without it those two fields are
    left unset (a consumer that needs them fails loudly).

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
    if shard is not None:
        m.read_length, m.fragment_size, m.fragment_count, m.fragment_samples_remaining, m.shard_infos = \
            merge_order_dependent(shard, dist, device, fragment_samples)
    # --fasta: both mates of a GC fragment lie in one exon, hence on one contig -> the histogram is additive; exon GC
    # values depend on the annotation and the reference only (identical on every rank)
    m.have_reference = int(getattr(local, "have_reference", 0))
    if m.have_reference:
        m.gc_bins = allsum(local.gc_bins, np.int64).astype(np.uint64)
        m.gc_out_of_range = int(allsum(np.array([local.gc_out_of_range]), np.int64)[0])
        m.exon_gc = np.array(local.exon_gc)
    return m
