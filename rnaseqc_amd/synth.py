"""Seeded synthetic inputs shaped like the BASELINE.json workloads.

No real GENCODE GTF or BAM exists offline, so both are generated:
  * make_annotation: GENCODE-like collapsed gene models (exons of a gene disjoint
    and sorted, ~10 % opposite-strand overlaps, ~1 % rRNA, the 12 globin names),
    56 202 genes / 323 826 exons at full scale (reference model: 323 418), 5 234 / 29 190 for the chr1 subset.
  * make_reads: coordinate-sorted paired 2x150 records: ~78 % from transcripts
    (spliced CIGARs), ~11 % intronic, ~7 % intergenic, ~4 % straddling exon edges,
    plus S/I/D CIGARs, low-MAPQ, NM>6, secondary/supplementary/unmapped records.
Everything is vectorised numpy; a 10 M-record batch takes tens of seconds.
"""
from __future__ import annotations

import numpy as np

from . import abi
from .model import Annotation, Batch, Bed, GLOBINS

# GRCh38-like contig lengths (chr1-22, X, Y, M) and a gene share roughly proportional
# to GENCODE gene density
HUMAN_CONTIGS = [
    ("chr1", 248956422, 5234), ("chr2", 242193529, 3900), ("chr3", 198295559, 2950),
    ("chr4", 190214555, 2500), ("chr5", 181538259, 2800), ("chr6", 170805979, 2850),
    ("chr7", 159345973, 2800), ("chr8", 145138636, 2300), ("chr9", 138394717, 2200),
    ("chr10", 133797422, 2150), ("chr11", 135086622, 3150), ("chr12", 133275309, 2850),
    ("chr13", 114364328, 1250), ("chr14", 107043718, 2150), ("chr15", 101991189, 2050),
    ("chr16", 90338345, 2350), ("chr17", 83257441, 2900), ("chr18", 80373285, 1100),
    ("chr19", 58617616, 2900), ("chr20", 64444167, 1350), ("chr21", 46709983, 800),
    ("chr22", 50818468, 1300), ("chrX", 156040895, 2300), ("chrY", 57227415, 500),
    ("chrM", 16569, 37),
]
# sums to 56 202 genes after the fix-up in human_contigs()
NEX_SMALL_FRAC = 0.592     # tuned so that human_contigs() yields ~323 k exons (the reference's GENCODE v19 collapsed model: 323 418)


def human_contigs(total_genes: int = 56202):
    c = [list(x) for x in HUMAN_CONTIGS]
    diff = total_genes - sum(x[2] for x in c)
    c[1][2] += diff
    return [tuple(x) for x in c]


def make_annotation(seed: int = 1, contigs=None, bam_contigs=None, shuffle_rows: bool = False) -> Annotation:
    """contigs: [(name, length, n_genes)].  Default: the chr1-like subset."""
    rng = np.random.default_rng(seed)
    if contigs is None:
        contigs = [HUMAN_CONTIGS[0]]
    if bam_contigs is None:
        bam_contigs = [c[0] for c in contigs]
    contig_id = {n: i for i, n in enumerate(bam_contigs)}
    names = list(bam_contigs)
    g_contig, g_start, g_end, g_flags = [], [], [], []
    e_contig, e_start, e_end, e_gene = [], [], [], []
    gene_base = 0
    for cname, clen, ng in contigs:
        if cname not in contig_id:
            contig_id[cname] = len(names)
            names.append(cname)
        cid = contig_id[cname]
        if ng == 0:
            continue
        # exons per gene: the collapsed GENCODE shape recoverable from the reference's golden exon_reads.gct
        # (mean 5.75, median 2, p90 16, p99 39, max 359) = a mixture of 1-3-exon genes (lncRNA,
        # pseudogenes) and protein-coding-like genes, plus a handful of very long ones
        nex = np.where(rng.random(ng) < NEX_SMALL_FRAC, 1 + rng.poisson(0.5, ng),
                       np.clip(np.rint(np.exp(rng.normal(np.log(9.0), 0.72, ng))), 1, 359)).astype(np.int64)
        giant = rng.random(ng) < 0.0004
        nex[giant] = rng.integers(150, 360, int(giant.sum()))
        budget = max(clen - 1000, 1000)
        E = int(nex.sum())
        elen = np.clip(np.rint(np.exp(rng.normal(np.log(150.0), 0.8, E))), 20, 12000).astype(np.int64)
        ilen = np.clip(np.rint(np.exp(rng.normal(np.log(1500.0), 1.2, E))), 50, 150000).astype(np.int64)
        first = np.cumsum(nex) - nex                     # index of each gene's first exon
        ilen[first] = 0                                  # no intron before the first exon
        if elen.sum() + ilen.sum() > 0.6 * budget:
            if elen.sum() < 0.45 * budget:               # gene-dense contig (chr19-like): shorter introns, same exons
                ilen = np.maximum((ilen * ((0.6 * budget - elen.sum()) / ilen.sum())).astype(np.int64), 30)
                ilen[first] = 0
            else:                                        # too small for spliced models (chrM: 37 genes in 16.5 kb):
                nex = np.ones(ng, np.int64); E = ng      # single-exon genes sized to the contig, like the real one
                elen = np.maximum((0.85 * budget / ng * (0.4 + 1.2 * rng.random(ng))).astype(np.int64), 30)
                ilen = np.zeros(ng, np.int64)
                first = np.arange(ng)
        gidx = np.repeat(np.arange(ng), nex)
        step = elen + ilen
        cum = np.cumsum(step)
        rel_end = cum - (cum[first] - step[first])[gidx]  # exon end offset within gene (exclusive)
        rel_start = rel_end - elen
        span = rel_end[first + nex - 1]
        if span.sum() > 0.9 * budget:
            raise ValueError("contig %s is too small for %d genes" % (cname, ng))
        free = budget - span.sum()
        overlap = rng.random(ng) < 0.10
        overlap[0] = False
        gap = rng.exponential(1.0, ng)
        gap[overlap] = 0
        gap = np.floor(gap / max(gap.sum(), 1e-9) * max(free, 0) * 0.98).astype(np.int64)
        gstart = np.empty(ng, np.int64)
        pos = 500
        strand = rng.integers(0, 2, ng)
        prev_start, prev_span = 500, 0
        # sequential placement (ng <= ~6k per contig, cheap)
        back = rng.random(ng)
        for i in range(ng):
            if overlap[i]:
                gstart[i] = prev_start + int(back[i] * 0.8 * prev_span)
                strand[i] = 1 - strand[i - 1]
                pos = max(pos, gstart[i] + span[i] + 1)
            else:
                gstart[i] = pos + gap[i]
                pos = gstart[i] + span[i] + 1
            prev_start, prev_span = gstart[i], span[i]
        flags = strand.astype(np.uint8)
        dot = rng.random(ng) < 0.005
        flags[dot] = abi.STRAND_UNKNOWN
        ribo = rng.random(ng) < 0.01
        flags[ribo] |= abi.FF_RIBOSOMAL
        g_contig.append(np.full(ng, cid, np.int32))
        g_start.append(gstart + 1)
        g_end.append(gstart + span)
        g_flags.append(flags)
        e_contig.append(np.full(E, cid, np.int32))
        e_start.append(gstart[gidx] + rel_start + 1)
        e_end.append(gstart[gidx] + rel_end)
        e_gene.append(gidx + gene_base)
        gene_base += ng
    g_contig = np.concatenate(g_contig); g_start = np.concatenate(g_start); g_end = np.concatenate(g_end)
    g_flags = np.concatenate(g_flags)
    e_contig = np.concatenate(e_contig); e_start = np.concatenate(e_start); e_end = np.concatenate(e_end)
    e_gene = np.concatenate(e_gene)
    G, E = len(g_start), len(e_start)
    # GTF order = generation order (gene by gene); ids = that order
    gene_ids = ["SYNG%08d.1" % i for i in range(G)]
    gene_names = ["SG%d" % i for i in range(G)]
    glob = sorted(GLOBINS)
    pick = rng.choice(G, size=min(len(glob), G), replace=False)
    for nm, gi in zip(glob, pick):
        gene_names[int(gi)] = nm
    first_of_gene = np.zeros(G + 1, np.int64)
    np.add.at(first_of_gene, e_gene + 1, 1)
    first_of_gene = np.cumsum(first_of_gene)
    exon_no = np.arange(E) - first_of_gene[e_gene] + 1
    exon_ids = ["%s_%d" % (gene_ids[g], k) for g, k in zip(e_gene.tolist(), exon_no.tolist())]
    exon_gene_names = [gene_names[g] for g in e_gene.tolist()]
    e_flags = g_flags[e_gene]
    # sorted rows: by (contig, start), ties in GTF order (stable)
    go = np.lexsort((np.arange(G), g_start, g_contig))
    eo = np.lexsort((np.arange(E), e_start, e_contig))
    # exonsForGene CSR over sorted exon rows
    row_gene = e_gene[eo]
    order_by_gene = np.argsort(row_gene, kind="stable")
    off = np.zeros(G + 1, np.uint32)
    np.add.at(off, row_gene + 1, 1)
    off = np.cumsum(off).astype(np.uint32)
    coding = np.zeros(G, np.int64)
    np.add.at(coding, e_gene, e_end - e_start + 1)
    globin = np.array([1 if n in GLOBINS else 0 for n in gene_names], np.uint8)
    return Annotation(
        contig_names=names, n_ref=len(bam_contigs), gene_ids=gene_ids, gene_names=gene_names,
        exon_ids=exon_ids, exon_gene_names=exon_gene_names, n_genes=G,
        gene_row_contig=g_contig[go].astype(np.int32), gene_row_start=g_start[go].astype(np.int32),
        gene_row_end=g_end[go].astype(np.int32), gene_row_flags=g_flags[go].astype(np.uint8),
        gene_row_id=go.astype(np.uint32),
        exon_row_contig=e_contig[eo].astype(np.int32), exon_row_start=e_start[eo].astype(np.int32),
        exon_row_end=e_end[eo].astype(np.int32), exon_row_flags=e_flags[eo].astype(np.uint8),
        exon_row_id=eo.astype(np.uint32), exon_row_gene=row_gene.astype(np.uint32),
        gene_is_globin=globin, gene_exon_off=off, gene_exon_row=order_by_gene.astype(np.uint32),
        coding_length=coding,
    )


def _ragged_take(values, starts, counts):
    """Concatenate values[starts[i] : starts[i]+counts[i]] for all i."""
    total = int(counts.sum())
    if total == 0:
        return values[:0]
    out_off = np.cumsum(counts) - counts
    idx = np.arange(total) - np.repeat(out_off, counts) + np.repeat(starts, counts)
    return values[idx]


def make_reads(ann: Annotation, n_pairs: int, seed: int = 2, read_len: int = 150,
               frac=(0.78, 0.11, 0.07, 0.04), dup_frac: float = 0.0, keep_qnames: bool = False,
               expr_sigma: float = 2.0, low_mapq_frac: float = 0.05, secondary_frac: float = 0.01,
               supplementary_frac: float = 0.005, unmapped_frac: float = 0.01, indel_frac: float = 0.02,
               chimeric_tag_frac: float = 0.0, filter_tag_frac: float = 0.0, expr_genes: int | None = None,
               contig_lengths=None, only_contig: int | None = None, fid_base: int = 0) -> Batch:
    """fid_base: first fragment number (QNAMEs are "SYN:%012d" of the fragment number), so that batches made
    contig by contig (make_reads_sharded) never share a name."""
    rng = np.random.default_rng(seed)
    rl = read_len
    G = ann.n_genes_listed
    # transcriptome coordinates over exon rows in exonsForGene order
    ge_row = ann.gene_exon_row.astype(np.int64)
    ex_s = ann.exon_row_start.astype(np.int64)[ge_row]      # 1-based
    ex_e = ann.exon_row_end.astype(np.int64)[ge_row]
    ex_len = ex_e - ex_s + 1
    T_end = np.cumsum(ex_len)
    T_start = T_end - ex_len
    goff = ann.gene_exon_off.astype(np.int64)
    gene_T0 = np.where(goff[:-1] < len(T_start), T_start[np.minimum(goff[:-1], len(T_start) - 1)], 0)
    coding = ann.coding_length[:G].astype(np.int64)
    row_of_gene = np.empty(G, np.int64)
    row_of_gene[ann.gene_row_id.astype(np.int64)] = np.arange(G)
    gene_contig = ann.gene_row_contig.astype(np.int64)[row_of_gene]
    gene_s = ann.gene_row_start.astype(np.int64)[row_of_gene]
    gene_e = ann.gene_row_end.astype(np.int64)[row_of_gene]
    if contig_lengths is None:
        contig_lengths = np.zeros(ann.n_ref, np.int64)
        np.maximum.at(contig_lengths, gene_contig, gene_e + 5000)
    n_tx, n_intr, n_inter, n_edge = [int(round(f * n_pairs)) for f in frac]
    n_tx = n_pairs - n_intr - n_inter - n_edge
    if only_contig is not None:
        # a contig without a usable gene (chrM-like: every transcript shorter than a read, no gene long enough for an
        # intronic pair) sends those pairs to the intergenic class instead
        on = gene_contig == only_contig
        if not (on & (coding >= rl)).any():
            n_inter += n_tx; n_tx = 0
        if not (on & ((gene_e - gene_s) > 3 * rl + 400)).any():
            n_inter += n_intr; n_intr = 0
        if not (ann.exon_row_contig == only_contig).any():
            n_inter += n_edge; n_edge = 0

    # ---- fragment-level: genomic block lists for both mates -----------------------
    # every mate is described by (contig, list of (gstart1, len)); we build flat op arrays
    def transcript_mates(n):
        expr = np.exp(rng.normal(0.0, expr_sigma, G))
        ok = coding >= rl
        if only_contig is not None:
            ok &= gene_contig == only_contig
        if expr_genes is not None:
            keep = np.zeros(G, bool)
            keep[rng.choice(G, size=min(expr_genes, G), replace=False)] = True
            ok &= keep
        w = expr * coding * ok
        w = w / w.sum()
        g = rng.choice(G, size=n, p=w)
        L = np.clip(np.rint(rng.normal(250, 60, n)), rl, None).astype(np.int64)
        L = np.minimum(L, coding[g])
        t0 = (rng.random(n) * (coding[g] - L + 1)).astype(np.int64)
        TA = gene_T0[g] + t0
        TB = gene_T0[g] + t0 + L - rl
        return g, TA, TB

    def to_blocks(T0, length):
        """transcriptome interval [T0, T0+length) -> per-read (pos0, ops ragged)."""
        n = len(T0)
        r0 = np.searchsorted(T_end, T0, side="right")
        Tstop = T0 + length
        ops_list, lens_list = [], []
        k = 0
        row = r0.copy()
        active = np.ones(n, bool)
        pos1 = ex_s[r0] + (T0 - T_start[r0])              # 1-based genomic start
        nops = np.zeros(n, np.int64)
        cols = []
        while active.any():
            rr = np.minimum(row, len(T_end) - 1)
            seg_s = np.maximum(T0, T_start[rr])
            seg_e = np.minimum(Tstop, T_end[rr])
            mlen = np.where(active, seg_e - seg_s, 0)
            more = active & (T_end[rr] < Tstop)
            nxt = np.minimum(rr + 1, len(T_end) - 1)
            gap = np.where(more, ex_s[nxt] - ex_e[rr] - 1, 0)
            cols.append((active.copy(), mlen, more.copy(), gap))
            row = row + 1
            active = more
            k += 1
            if k > 400:
                raise RuntimeError("too many blocks")
        K = len(cols)
        opm = np.zeros((n, 2 * K), np.uint32)
        valid = np.zeros((n, 2 * K), bool)
        for j, (act, mlen, more, gap) in enumerate(cols):
            opm[:, 2 * j] = (mlen.astype(np.uint32) << 4) | abi.CIG_M
            valid[:, 2 * j] = act
            # adjacent exons (gap 0) merge: emit N only when gap > 0, else a 0-length N is dropped
            opm[:, 2 * j + 1] = (gap.astype(np.uint32) << 4) | abi.CIG_N
            valid[:, 2 * j + 1] = more & (gap > 0)
        ncig = valid.sum(axis=1)
        flat = opm[valid]
        return pos1 - 1, ncig.astype(np.int64), flat

    def genomic_mates(contig, start1, n):
        """contiguous genomic fragments: mates are plain 150M reads."""
        L = np.clip(np.rint(rng.normal(250, 60, n)), rl, None).astype(np.int64)
        posA = start1 - 1
        posB = start1 - 1 + L - rl
        ncig = np.ones(n, np.int64)
        flat = np.full(n, (rl << 4) | abi.CIG_M, np.uint32)
        return posA, posB, ncig, flat, L

    parts = []   # per category: dict(contig, posA, ncigA, flatA, posB, ncigB, flatB)
    if n_tx:
        g, TA, TB = transcript_mates(n_tx)
        pA, nA, fA = to_blocks(TA, rl)
        pB, nB, fB = to_blocks(TB, rl)
        parts.append((gene_contig[g], pA, nA, fA, pB, nB, fB))
    if n_intr:
        big = (gene_e - gene_s) > 3 * rl + 400
        if only_contig is not None:
            big &= gene_contig == only_contig
        big = np.flatnonzero(big)
        g = big[rng.integers(0, len(big), n_intr)]
        s1 = gene_s[g] + (rng.random(n_intr) * (gene_e[g] - gene_s[g] - 2 * rl - 300)).astype(np.int64)
        pA, pB, nc, fl, _ = genomic_mates(gene_contig[g], s1, n_intr)
        parts.append((gene_contig[g], pA, nc, fl, pB, nc.copy(), fl.copy()))
    if n_inter:
        cw = np.asarray(contig_lengths, dtype=np.float64).copy()
        if only_contig is not None:
            keep = np.zeros(len(cw), bool); keep[only_contig] = True
            cw[~keep] = 0
        cw = cw / cw.sum()
        c = rng.choice(len(contig_lengths), size=n_inter, p=cw)
        s1 = 1 + (rng.random(n_inter) * np.maximum(contig_lengths[c] - 1000, 1)).astype(np.int64)
        pA, pB, nc, fl, _ = genomic_mates(c, s1, n_inter)
        parts.append((c, pA, nc, fl, pB, nc.copy(), fl.copy()))
    if n_edge:
        ex_c_all = ann.exon_row_contig.astype(np.int64)[ge_row]
        cand = np.arange(len(ex_s)) if only_contig is None else np.flatnonzero(ex_c_all == only_contig)
        r = cand[rng.integers(0, len(cand), n_edge)]
        s1 = np.maximum(ex_s[r] - rng.integers(1, rl, n_edge), 1)
        c = ann.exon_row_contig.astype(np.int64)[ge_row][r]
        pA, pB, nc, fl, _ = genomic_mates(c, s1, n_edge)
        parts.append((c, pA, nc, fl, pB, nc.copy(), fl.copy()))

    contig = np.concatenate([p[0] for p in parts])
    posA = np.concatenate([p[1] for p in parts]); ncA = np.concatenate([p[2] for p in parts])
    flA = np.concatenate([p[3] for p in parts])
    posB = np.concatenate([p[4] for p in parts]); ncB = np.concatenate([p[5] for p in parts])
    flB = np.concatenate([p[6] for p in parts])
    nf = len(contig)

    def ref_len(nc, fl):
        owner = np.repeat(np.arange(len(nc)), nc)
        out = np.zeros(len(nc), np.int64)
        np.add.at(out, owner, (fl >> 4).astype(np.int64))
        return out
    endA = posA + ref_len(ncA, flA)
    endB = posB + ref_len(ncB, flB)
    # A is the leftmost mate (forward), B the rightmost (reverse)
    a_is_r1 = rng.random(nf) < 0.5
    flagA = np.where(a_is_r1, 99, 163).astype(np.uint16)
    flagB = np.where(a_is_r1, 147, 83).astype(np.uint16)
    isz = (np.maximum(endA, endB) - np.minimum(posA, posB)).astype(np.int32)
    frag_id = np.arange(nf, dtype=np.int64) + int(fid_base)

    # ---- record-level arrays (2 per fragment) --------------------------------------
    tid = np.concatenate([contig, contig]).astype(np.int32)
    pos = np.concatenate([posA, posB]).astype(np.int64)
    mpos = np.concatenate([posB, posA]).astype(np.int64)
    isize = np.concatenate([isz, -isz]).astype(np.int32)
    flag = np.concatenate([flagA, flagB]).astype(np.uint16)
    ncig = np.concatenate([ncA, ncB])
    flat = np.concatenate([flA, flB])
    fid = np.concatenate([frag_id, frag_id])
    n = 2 * nf
    coff = np.cumsum(ncig) - ncig
    lq = np.full(n, rl, np.int64)

    # S / I / D variants on single-block reads
    single = np.flatnonzero(ncig == 1)
    nmod = int(indel_frac * n)
    if nmod and len(single):
        pickm = rng.choice(single, size=min(nmod, len(single)), replace=False)
        kind = rng.integers(0, 3, len(pickm))
        extra_ops = []     # (record, [ops])
        new_nc = ncig.copy()
        repl = {}
        for rec_i, kd in zip(pickm.tolist(), kind.tolist()):
            if kd == 0:      # 10S140M, pos += 10
                repl[rec_i] = [(10 << 4) | abi.CIG_S, ((rl - 10) << 4) | abi.CIG_M]
                pos[rec_i] += 10
            elif kd == 1:    # 70M2I78M
                repl[rec_i] = [(70 << 4) | abi.CIG_M, (2 << 4) | abi.CIG_I, ((rl - 72) << 4) | abi.CIG_M]
            else:            # 70M3D80M
                repl[rec_i] = [(70 << 4) | abi.CIG_M, (3 << 4) | abi.CIG_D, ((rl - 70) << 4) | abi.CIG_M]
            new_nc[rec_i] = len(repl[rec_i])
        new_off = np.cumsum(new_nc) - new_nc
        new_flat = np.zeros(int(new_nc.sum()), np.uint32)
        keep = np.ones(n, bool)
        keep[list(repl.keys())] = False
        # copy untouched
        src_idx = _ragged_take(np.arange(len(flat)), coff[keep], ncig[keep])
        dst_idx = _ragged_take(np.arange(len(new_flat)), new_off[keep], new_nc[keep])
        new_flat[dst_idx] = flat[src_idx]
        for rec_i, ops in repl.items():
            new_flat[new_off[rec_i]:new_off[rec_i] + len(ops)] = ops
        flat, ncig, coff = new_flat, new_nc, new_off
        # mates of soft-clipped reads see the shifted position
        mate = (np.arange(n) + nf) % n
        mpos = pos[mate]

    mapq = np.where(rng.random(n) < low_mapq_frac, 3, 255).astype(np.uint8)
    nm = rng.integers(0, 3, n).astype(np.int64)
    nm[rng.random(n) < 0.01] = 8
    tagbits = np.full(n, abi.TB_HAS_NM | abi.TB_MTID_SAME, np.uint8)
    if chimeric_tag_frac:
        tagbits[rng.random(n) < chimeric_tag_frac] |= abi.TB_HAS_CH
    if filter_tag_frac:
        tagbits[rng.random(n) < filter_tag_frac] |= abi.TB_FILTER0
    if dup_frac:
        d = rng.random(nf) < dup_frac
        flag[np.concatenate([d, d])] |= abi.FDUP

    # secondary / supplementary copies
    def copies(fraction, bit):
        k = int(fraction * n)
        if not k:
            return None
        src = rng.integers(0, n, k)
        return src, bit
    extra = [c for c in (copies(secondary_frac, abi.FSECONDARY), copies(supplementary_frac, abi.FSUPP)) if c]
    if extra:
        src = np.concatenate([e[0] for e in extra])
        bits = np.concatenate([np.full(len(e[0]), e[1], np.uint16) for e in extra])
        tid = np.concatenate([tid, tid[src]]); pos = np.concatenate([pos, pos[src]])
        mpos = np.concatenate([mpos, mpos[src]]); isize = np.concatenate([isize, isize[src]])
        flag = np.concatenate([flag, flag[src] | bits]); mapq = np.concatenate([mapq, mapq[src]])
        nm = np.concatenate([nm, nm[src]]); tagbits = np.concatenate([tagbits, tagbits[src]])
        lq = np.concatenate([lq, lq[src]]); fid = np.concatenate([fid, fid[src]])
        add_flat = _ragged_take(flat, coff[src], ncig[src])
        flat = np.concatenate([flat, add_flat]); ncig = np.concatenate([ncig, ncig[src]])
        coff = np.cumsum(ncig) - ncig
    # unmapped tail (pairs)
    nu = int(unmapped_frac * n) // 2 * 2
    if nu:
        uf = int(fid_base) + nf + np.arange(nu // 2, dtype=np.int64)
        tid = np.concatenate([tid, np.full(nu, -1, np.int32)])
        pos = np.concatenate([pos, np.full(nu, -1, np.int64)]); mpos = np.concatenate([mpos, np.full(nu, -1, np.int64)])
        isize = np.concatenate([isize, np.zeros(nu, np.int32)])
        flag = np.concatenate([flag, np.tile(np.array([77, 141], np.uint16), nu // 2)])
        mapq = np.concatenate([mapq, np.zeros(nu, np.uint8)]); nm = np.concatenate([nm, np.zeros(nu, np.int64)])
        tagbits = np.concatenate([tagbits, np.full(nu, abi.TB_MTID_SAME, np.uint8)])
        lq = np.concatenate([lq, np.full(nu, rl, np.int64)]); fid = np.concatenate([fid, np.repeat(uf, 2)])
        ncig = np.concatenate([ncig, np.zeros(nu, np.int64)])
        coff = np.cumsum(ncig) - ncig
    N = len(pos)

    # ---- coordinate sort (unmapped last), stable -------------------------------------
    tkey = np.where(tid < 0, np.iinfo(np.int32).max, tid).astype(np.int64)
    order = np.lexsort((pos, tkey))
    tid, pos, mpos, isize, flag, mapq, nm, tagbits, lq, fid = (
        a[order] for a in (tid, pos, mpos, isize, flag, mapq, nm, tagbits, lq, fid))
    new_nc = ncig[order]
    flat = _ragged_take(flat, coff[order], new_nc)
    ncig = new_nc
    coff = np.cumsum(ncig) - ncig

    # qnames: fixed width "SYN:%012d"
    digits = np.zeros((N, 16), np.uint8)
    digits[:, :4] = np.frombuffer(b"SYN:", np.uint8)
    v = fid.copy()
    for j in range(15, 3, -1):
        digits[:, j] = 48 + (v % 10)
        v //= 10
    qhash = abi.qname_hash_bytes(digits)

    change = np.flatnonzero(np.diff(tid)) + 1
    starts = np.concatenate([[0], change]) if N else np.zeros(0, np.int64)
    seg_tid = tid[starts] if N else np.zeros(0, np.int32)
    seg_start = np.concatenate([starts, [N]]).astype(np.uint64)

    b = Batch(pos=pos.astype(np.int32), mpos=mpos.astype(np.int32), isize=isize.astype(np.int32), qhash=qhash,
              cigar_off=coff.astype(np.uint32), flag=flag.astype(np.uint16), l_qseq=lq.astype(np.uint16),
              mapq=mapq.astype(np.uint8), nm=nm.astype(np.uint8), tagbits=tagbits.astype(np.uint8),
              n_cigar=ncig.astype(np.uint8), cigar=flat.astype(np.uint32), seg_tid=seg_tid.astype(np.int32),
              seg_start=seg_start, qhash2=abi.qname_hash2_bytes(digits))
    if keep_qnames:
        b.qname = digits.reshape(-1).copy()
        b.qname_off = (np.arange(N + 1, dtype=np.uint32) * 16).astype(np.uint32)
    return b


def contig_pair_shares(ann: Annotation, n_pairs: int) -> np.ndarray:
    """Read pairs per BAM contig for make_reads_sharded: proportional to the contig's gene count (largest
    remainder), so the split is known before anything is generated -- the LPT assignment of contigs to GPUs
    (distributed.assign_contigs) works from these numbers."""
    G = ann.n_genes_listed
    per = np.bincount(ann.gene_row_contig.astype(np.int64), minlength=ann.n_ref)[:ann.n_ref].astype(np.float64)
    if per.sum() == 0:
        per[:] = 1.0
    exact = per / per.sum() * n_pairs
    share = np.floor(exact).astype(np.int64)
    rest = int(n_pairs - share.sum())
    if rest:
        share[np.argsort(-(exact - share), kind="stable")[:rest]] += 1
    return share


_SHARD_CTX = {}


def _shard_job(c):
    a = _SHARD_CTX
    return c, make_reads(a["ann"], int(a["share"][c]), seed=a["seed"] * 100003 + c, only_contig=c, unmapped_frac=0.0,
                         fid_base=int(a["fid_base"][c]), contig_lengths=a["contig_lengths"], **a["kw"])


def make_reads_sharded(ann: Annotation, n_pairs: int, seed: int = 2, contigs=None, workers: int = 0,
                       unmapped_frac: float = 0.01, with_unmapped: bool = True, contig_lengths=None, as_parts: bool = False, **kw):
    """The BASELINE-scale input (100 M records): every contig is generated on its own (own seed, own fragment
    numbers) and the pieces are concatenated in contig order, which IS coordinate order; the unmapped tail comes
    last.  `contigs` restricts the output to a subset (a GPU's shard): the union over a partition of the contigs
    equals the full input record for record.  workers > 1 forks that many generator processes (call this
    before the HIP runtime is initialised).  Returns (batch, records_per_contig[n_ref]); with as_parts, the list of
    per-contig batches instead of their concatenation."""
    share = contig_pair_shares(ann, n_pairs)
    fid_base = np.concatenate([[0], np.cumsum(share)])[:-1]
    if contig_lengths is None:
        contig_lengths = np.zeros(ann.n_ref, np.int64)
        np.maximum.at(contig_lengths, ann.gene_row_contig.astype(np.int64), ann.gene_row_end.astype(np.int64) + 5000)
    want = [c for c in (range(ann.n_ref) if contigs is None else contigs) if share[c] > 0]
    _SHARD_CTX.clear()
    _SHARD_CTX.update(ann=ann, share=share, seed=seed, fid_base=fid_base, kw=kw, contig_lengths=np.asarray(contig_lengths, np.int64))
    order = sorted(want, key=lambda c: -int(share[c]))              # longest first
    pieces = {}
    if workers > 1 and len(order) > 1:
        import multiprocessing as mp
        with mp.get_context("fork").Pool(min(workers, len(order))) as pool:
            for c, b in pool.imap_unordered(_shard_job, order, chunksize=1):
                pieces[c] = b
    else:
        for c in order:
            pieces[c] = _shard_job(c)[1]
    _SHARD_CTX.clear()
    per_contig = np.zeros(ann.n_ref, np.int64)
    for c, b in pieces.items():
        per_contig[c] = b.n
    parts = [pieces[c] for c in sorted(pieces)]
    if with_unmapped and unmapped_frac:
        nu = int(unmapped_frac * 2 * n_pairs) // 2 * 2
        if nu:
            rl = int(kw.get("read_len", 150))
            fid = int(share.sum()) + np.repeat(np.arange(nu // 2, dtype=np.int64), 2)
            digits = np.zeros((nu, 16), np.uint8)
            digits[:, :4] = np.frombuffer(b"SYN:", np.uint8)
            v = fid.copy()
            for j in range(15, 3, -1):
                digits[:, j] = 48 + (v % 10)
                v //= 10
            parts.append(Batch(pos=np.full(nu, -1, np.int32), mpos=np.full(nu, -1, np.int32), isize=np.zeros(nu, np.int32),
                               qhash=abi.qname_hash_bytes(digits), cigar_off=np.zeros(nu, np.uint32),
                               flag=np.tile(np.array([77, 141], np.uint16), nu // 2), l_qseq=np.full(nu, rl, np.uint16),
                               mapq=np.zeros(nu, np.uint8), nm=np.zeros(nu, np.uint8),
                               tagbits=np.full(nu, abi.TB_MTID_SAME, np.uint8), n_cigar=np.zeros(nu, np.uint8),
                               cigar=np.zeros(0, np.uint32), seg_tid=np.array([-1], np.int32),
                               seg_start=np.array([0, nu], np.uint64), qhash2=abi.qname_hash2_bytes(digits)))
    if as_parts:
        # one batch per contig (+ the unmapped tail), each a contiguous range of the file.  file_index_base is a VIRTUAL
        # file index, contig << 32: monotone in file order, which is all the boundary asks for (gaps are allowed), and
        # computable by a rank that generated only its own contigs
        keys = sorted(pieces)
        for c, b in zip(keys + [ann.n_ref], parts):
            b.file_index_base = int(c) << 32
        return parts, per_contig
    return Batch.concat(parts), per_contig


def make_bed(ann: Annotation, min_len: int = 1000) -> Bed:
    """Exons >= min_len that overlap no other exon row (mirrors the intent of
    python/rnaseqc/insert_size_intervals.py:75-92), 0-based half-open, sorted."""
    s = ann.exon_row_start.astype(np.int64)
    e = ann.exon_row_end.astype(np.int64)
    c = ann.exon_row_contig.astype(np.int64)
    n = len(s)
    if n == 0:
        return Bed.from_intervals([], [], [])
    key_prev_end = np.maximum.accumulate(np.where(np.arange(n) >= 0, e + c * (1 << 40), 0))
    prev_end = np.concatenate([[-1], key_prev_end[:-1]])
    next_start = np.concatenate([s[1:] + c[1:] * (1 << 40), [1 << 62]])
    me_s, me_e = s + c * (1 << 40), e + c * (1 << 40)
    ok = (prev_end < me_s) & (next_start > me_e) & ((e - s + 1) >= min_len)
    idx = np.flatnonzero(ok)
    return Bed.from_intervals(c[idx], s[idx] - 1, e[idx])


def make_reference(contig_lengths, seed: int = 0, contigs=None, gc_wave: int = 50_000, uniform: bool = False):
    """Random base strings for the given contigs (ids default to 0..n-1): GC fraction drifting between 0.3 and 0.7
    along the contig, a few lower-case stretches and N runs (only G/g/C/c count, src/Fasta.cpp:67-74)."""
    from .model import Reference
    rng = np.random.default_rng(seed)
    seqs = []
    for L in contig_lengths:
        L = int(L)
        if uniform:                                                 # large contigs (bench): uniform ACGT, no drift
            seqs.append(np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, L, dtype=np.uint8)])
            continue
        x = np.arange(L, dtype=np.float64)
        pgc = 0.5 + 0.2 * np.sin(x / gc_wave * 2 * np.pi + rng.random() * 6.28)
        is_gc = rng.random(L) < pgc
        pick = rng.random(L) < 0.5
        b = np.where(is_gc, np.where(pick, ord("G"), ord("C")), np.where(pick, ord("A"), ord("T"))).astype(np.uint8)
        for _ in range(max(1, L // 200_000)):                       # soft-masked stretches and N runs
            a = int(rng.integers(0, max(1, L - 2000))); n = int(rng.integers(50, 2000))
            b[a:a + n] |= 0x20
            a = int(rng.integers(0, max(1, L - 500))); n = int(rng.integers(10, 500))
            b[a:a + n] = ord("N")
        seqs.append(b)
    ids = list(range(len(seqs))) if contigs is None else list(contigs)
    return Reference(contig=ids, sequence=seqs)
