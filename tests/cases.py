"""Hand-built inputs shared by the CPU (oracle) and GPU (HIP) tests.

`quirk_case()` is a 22-record, 2-contig, 5-gene input whose expected outputs
were derived BY HAND from the reference source (see the comments) -- they are
not produced by any of our code.
"""
import numpy as np

from rnaseqc_amd import abi
from rnaseqc_amd.model import Annotation, Batch

M, I, D, N, S = abi.CIG_M, abi.CIG_I, abi.CIG_D, abi.CIG_N, abi.CIG_S


def quirk_annotation():
    rows = [
        dict(contig="chr1", type="gene", start=1000, end=5000, strand="+", gene_id="GA", gene_name="AAA", transcript_type="protein_coding"),
        dict(contig="chr1", type="exon", start=1000, end=1200, strand="+", gene_id="GA", exon_id="GA_1", gene_name="AAA", transcript_type="protein_coding"),
        dict(contig="chr1", type="exon", start=2000, end=2300, strand="+", gene_id="GA", exon_id="GA_2", gene_name="AAA", transcript_type="protein_coding"),
        dict(contig="chr1", type="exon", start=4000, end=5000, strand="+", gene_id="GA", exon_id="GA_3", gene_name="AAA", transcript_type="protein_coding"),
        dict(contig="chr1", type="gene", start=4500, end=8000, strand="-", gene_id="GB", gene_name="BBB", transcript_type="protein_coding"),
        dict(contig="chr1", type="exon", start=4500, end=4800, strand="-", gene_id="GB", exon_id="GB_1", gene_name="BBB", transcript_type="protein_coding"),
        dict(contig="chr1", type="exon", start=7000, end=8000, strand="-", gene_id="GB", exon_id="GB_2", gene_name="BBB", transcript_type="protein_coding"),
        dict(contig="chr1", type="gene", start=10000, end=10500, strand="+", gene_id="GR", gene_name="RRR", transcript_type="rRNA"),
        dict(contig="chr1", type="exon", start=10000, end=10500, strand="+", gene_id="GR", exon_id="GR_1", gene_name="RRR", transcript_type="rRNA"),
        dict(contig="chr1", type="gene", start=12000, end=12600, strand="-", gene_id="GH", gene_name="HBB", transcript_type="protein_coding"),
        dict(contig="chr1", type="exon", start=12000, end=12600, strand="-", gene_id="GH", exon_id="GH_1", gene_name="HBB", transcript_type="protein_coding"),
        dict(contig="chr2", type="gene", start=100, end=2100, strand="+", gene_id="GC", gene_name="CCC", transcript_type="protein_coding"),
        dict(contig="chr2", type="exon", start=100, end=2100, strand="+", gene_id="GC", exon_id="GC_1", gene_name="CCC", transcript_type="protein_coding"),
    ]
    return Annotation.from_rows(["chr1", "chr2"], rows)


def quirk_records():
    def rec(q, tid, pos, cigar, flag, mapq=255, nm=0, mpos=None, mtid=None, isize=0, **kw):
        return dict(qname=q, tid=tid, pos=pos, cigar=cigar, flag=flag, mapq=mapq, nm=nm,
                    mpos=pos if mpos is None else mpos, mtid=tid if mtid is None else mtid, isize=isize, **kw)
    R = [
        rec("p1", 0, 1049, [(M, 100)], 99, mpos=1079, isize=130),
        rec("p1", 0, 1079, [(M, 100)], 147, mpos=1049, isize=-130),
        rec("s1", 0, 1150, [(M, 50), (N, 800), (M, 50)], 99),
        rec("s2", 0, 1150, [(M, 50), (N, 300), (M, 50)], 99),
        rec("a1", 0, 1899, [(M, 100)], 99),
        rec("i1", 0, 2999, [(M, 100)], 99),
        rec("l1", 0, 4099, [(M, 100)], 99, mapq=3),
        rec("n1", 0, 4199, [(M, 100)], 99, nm=8),
        rec("d1", 0, 4299, [(M, 100)], 99 | 0x400),
        rec("x1", 0, 4300, [(M, 100)], 99 | 0x100),
        rec("x2", 0, 4301, [(M, 100)], 99 | 0x800),
        rec("x3", 0, 4302, [(M, 100)], 99 | 0x200),
        rec("k1", 0, 4399, [(S, 10), (M, 40), (I, 2), (M, 48), (D, 5), (M, 10)], 99),
        rec("t1", 0, 4549, [(M, 100)], 99),
        rec("g1", 0, 8999, [(M, 100)], 99),
        rec("r1", 0, 10099, [(M, 100)], 99),
        rec("h1", 0, 12099, [(M, 100)], 99),
        rec("c1", 0, 12999, [(M, 100)], 97, mtid=1, mpos=500),
        rec("u1", 1, 599, [(M, 100)], 99),
        rec("u2", 1, 699, [(M, 100)], 99),
        rec("u3", 1, 799, [(M, 100)], 99),
        rec("um1", -1, -1, [], 77, mapq=0, nm=None, l_qseq=100),
    ]
    return R


def quirk_case():
    return quirk_annotation(), Batch.from_records(quirk_records())


# ---- expected values, derived by hand from the reference source ----------------
QUIRK_COUNTERS = {
    "Total Alignments": 22,                       # src/RNASeQC.cpp:245,397
    "Alternative Alignments": 1,                  # x1                                   :254
    "Supplementary Alignments": 1,                # x2                                   :255
    "Failed Vendor QC": 1,                        # x3                                   :256
    "Low Mapping Quality": 2,                     # l1 (3 < 255) and the unmapped um1 (0) :257
    "Chimeric Fragments_auto": 2,                 # x2 (SUP, no ch tag :258-260), c1 (mate on chr2 :287-289)
    "Chimeric Fragments_tag": 0,
    "Unique Mapping, Vendor QC Passed Reads": 19,  # 22 - x1 - x2 - x3                    :263-265
    "Unpaired Reads": 0,
    "Mapped Reads": 18,                           # - um1                                :268-270
    "Mapped Duplicate Reads": 1,                  # d1
    "Mapped Unique Reads": 17,
    "Total Mapped Pairs": 17,                     # every mapped READ1 (all but p1/2)     :284-286
    "End 1 Mapped Reads": 17, "End 2 Mapped Reads": 1,
    "End 1 Mismatches": 8, "End 2 Mismatches": 0, "Mismatched Bases": 8,   # n1 has NM 8  :294-316
    "End 1 Bases": 1710, "End 2 Bases": 100, "Total Bases": 1810,          # k1 has l_qseq 110
    "Duplicate Pairs": 1, "Unique Fragments": 16,
    "High Quality Reads": 15,                     # 18 - l1 (mapq) - n1 (NM>6) - c1 (not proper) :330
    "Low Quality Reads": 3,
    "Reads used for Intron/Exon counts": 18,
    "Alignment Blocks": 22,                       # s1 2 + s2 2 + k1 3 + 15 x 1           :360
    "Non-Globin Reads": 17,                       # all 18 but h1 (HBB)          Expression.cpp:395-404
    "Non-Globin Duplicate Reads": 1,
    "Exonic Reads": 13, "HQ Exonic Reads": 11,    # p1 p1 s1 l1 n1 d1 k1 t1 r1 h1 u1 u2 u3
    "Ambiguous Reads": 2, "HQ Ambiguous Reads": 2,  # s2 (block in intron), a1 (touches exon start: Q1)
    "Intronic Reads": 1, "HQ Intronic Reads": 1,  # i1
    "Intergenic Reads": 2, "HQ Intergenic Reads": 1,  # g1, c1(not HQ)
    "Intragenic Reads": 14, "HQ Intragenic Reads": 12,
    "rRNA Reads": 1,
    "End 1 Sense": 12, "End 1 Antisense": 1,      # h1: forward read on a '-' gene
    "End 2 Sense": 0, "End 2 Antisense": 1,       # p1/2: reverse read on a '+' gene
}
# geneList order: GA GB GR GH GC
QUIRK_GENE_READS = [6, 1, 1, 1, 3]       # GA: p1 p1 s1 d1 k1 t1; t1 also counts for GB (Q2)
QUIRK_GENE_UNIQUE = [5, 1, 1, 1, 3]      # d1 is a duplicate
QUIRK_GENE_FRAGMENTS = [5, 1, 1, 1, 3]   # p1's two mates are one fragment
# exonList order: GA_1 GA_2 GA_3 GB_1 GB_2 GR_1 GH_1 GC_1
QUIRK_EXON_READS = [2.5, 0.5, 3.0, 1.0, 0.0, 1.0, 1.0, 3.0]
QUIRK_READ_LENGTH = 110                  # k1: span 103 > 100 -> l_qseq 110 (Q3)


def single_pair_case():
    """A reconstruction of the reference's test_data/single_pair: its inputs are not in the tree, but the
    golden outputs (single_pair.output/) fix what they must look like: one '-' strand gene with 13 exons
    (WASH7P), one proper pair of 76-base reads, both inside exon _13, READ1 reverse (End 1 Sense = 1),
    READ2 forward (End 2 Antisense = 1), within 500 bases of a transcript end (coverage mean 0)."""
    exons = [(14363, 14829), (14970, 15038), (15796, 15947), (16607, 16765), (16858, 17055), (17233, 17368),
             (17606, 17742), (17915, 18061), (18268, 18366), (24738, 24891), (29534, 29806), (30000, 30200), (30300, 30500)]
    rows = [dict(contig="1", type="gene", start=14363, end=30500, strand="-", gene_id="ENSG00000227232.4", gene_name="WASH7P",
                 transcript_type="pseudogene")]
    for k, (s, e) in enumerate(exons):
        rows.append(dict(contig="1", type="exon", start=s, end=e, strand="-", gene_id="ENSG00000227232.4",
                         exon_id="ENSG00000227232.4_%d" % (13 - k), gene_name="WASH7P", transcript_type="pseudogene"))
    ann = Annotation.from_rows(["1"], rows)
    recs = [dict(qname="pair", tid=0, pos=14400, cigar=[(M, 76)], flag=163, mpos=14500, isize=176, nm=0),
            dict(qname="pair", tid=0, pos=14500, cigar=[(M, 76)], flag=83, mpos=14400, isize=-176, nm=0)]
    return ann, Batch.from_records(recs)


def many_exon_case():
    """One gene of 150 thirty-base exons (more exons than a wave has lanes: the end-of-file coverage kernel gathers a wave's exon
    rows 64 at a time) + one ordinary gene; 6 000 one-block records of 24 bases."""
    from rnaseqc_amd.model import Annotation, Batch
    from rnaseqc_amd import abi
    rows = [dict(contig="c", type="gene", start=1_000, end=1_000 + 150 * 100, strand="+", gene_id="many", gene_name="many")]
    for k in range(150):
        rows.append(dict(contig="c", type="exon", start=1_000 + k * 100, end=1_000 + k * 100 + 29, strand="+", gene_id="many", exon_id="m_e%d" % k))
    rows.append(dict(contig="c", type="gene", start=40_000, end=42_000, strand="-", gene_id="plain", gene_name="plain"))
    rows.append(dict(contig="c", type="exon", start=40_000, end=42_000, strand="-", gene_id="plain", exon_id="p_e0"))
    ann = Annotation.from_rows(["c"], rows)
    rng = np.random.default_rng(21)
    n = 6_000
    ex = rng.integers(0, 150, n - 1_000); off = rng.integers(0, 6, n - 1_000)
    pos = np.sort(np.concatenate([1_000 + ex * 100 + off, 40_000 + rng.integers(0, 1_900, 1_000)])).astype(np.int32) - 1
    qh = abi.qname_hash_bytes(np.frombuffer(b"".join(b"%015d" % i for i in range(n)), np.uint8).reshape(n, 15))
    batch = Batch(pos=pos, mpos=pos.copy(), isize=np.zeros(n, np.int32), qhash=qh, cigar_off=np.arange(n, dtype=np.uint32),
                  flag=np.zeros(n, np.uint16), l_qseq=np.full(n, 24, np.uint16), mapq=np.full(n, 255, np.uint8),
                  nm=np.zeros(n, np.uint8), tagbits=np.full(n, abi.TB_HAS_NM | abi.TB_MTID_SAME, np.uint8),
                  n_cigar=np.ones(n, np.uint8), cigar=np.full(n, (24 << 4) | abi.CIG_M, np.uint32),
                  seg_tid=np.array([0], np.int32), seg_start=np.array([0, n], np.uint64),
                  wide_index=np.zeros(0, np.uint64), wide_nm=np.zeros(0, np.int32), wide_l_qseq=np.zeros(0, np.int32),
                  wide_n_cigar=np.zeros(0, np.uint32))
    return ann, batch
