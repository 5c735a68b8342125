"""Host-side containers for the boundary: Annotation, Bed, Batch.

They hold numpy arrays laid out exactly as include/rnaseqc_amd.h wants them and
hand out the ctypes structs.  `Annotation.from_rows` is the Python mirror of
the reference's GTF bookkeeping (id assignment, ordering, exonsForGene), used
by the tests and the synthetic generators; the C++ loader
(rnaseqc_amd/csrc/gtf.cpp) must produce the same arrays from GTF text.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Iterable, Sequence

import numpy as np

from . import abi

# blacklistedGlobins, reference src/Expression.cpp:24
GLOBINS = {"HBA1", "HBA2", "HBB", "HBD", "HBG1", "HBG2", "HBE1", "HBM", "HBQ1", "HBZ", "HBBP1", "HBZP1"}

_STRAND = {"+": abi.STRAND_FORWARD, "-": abi.STRAND_REVERSE}


@dataclass
class Annotation:
    contig_names: list            # id -> name; [0, n_ref) = BAM header order
    n_ref: int
    gene_ids: list                # listed gene_id strings, geneList order
    gene_names: list              # geneNames[gene_id]
    exon_ids: list                # exonList order
    exon_gene_names: list         # geneNames[exon_id] (Description column of exon_reads.gct)
    n_genes: int                  # listed + phantom
    gene_row_contig: np.ndarray
    gene_row_start: np.ndarray
    gene_row_end: np.ndarray
    gene_row_flags: np.ndarray
    gene_row_id: np.ndarray
    exon_row_contig: np.ndarray
    exon_row_start: np.ndarray
    exon_row_end: np.ndarray
    exon_row_flags: np.ndarray
    exon_row_id: np.ndarray
    exon_row_gene: np.ndarray
    gene_is_globin: np.ndarray
    gene_exon_off: np.ndarray
    gene_exon_row: np.ndarray
    coding_length: np.ndarray = field(default=None)   # geneCodingLengths by gene id
    gene_row_order: np.ndarray = field(default=None)  # optional GTF positions of the rows (legacy rules only)
    exon_row_order: np.ndarray = field(default=None)

    @property
    def n_genes_listed(self) -> int:
        return len(self.gene_ids)

    @property
    def n_exons(self) -> int:
        return len(self.exon_row_id)

    @property
    def n_contigs(self) -> int:
        return len(self.contig_names)

    # ------------------------------------------------------------------ build
    @staticmethod
    def from_rows(bam_contigs: Sequence[str], rows: Iterable[dict]) -> "Annotation":
        """rows: GTF order; each {contig,type('gene'|'exon'),start,end,strand,gene_id,
        exon_id(optional),gene_name(optional),transcript_type(optional)}.
        Mirrors src/GTF.cpp:30-131 + src/RNASeQC.cpp:127-156 (incl. the
        transcript_type state leak Q16 and exon-id inference Q11)."""
        contig_id = {n: i for i, n in enumerate(bam_contigs)}
        contig_names = list(bam_contigs)
        gene_index, gene_ids, gene_names = {}, [], []
        exon_ids, exon_gene_names, exon_seen = [], [], set()
        exon_counter = {}
        g_rows, e_rows = [], []
        pending_gene_refs = []
        last_ttype = ""
        for order, r in enumerate(rows):
            cname = r["contig"]
            if cname not in contig_id:
                contig_id[cname] = len(contig_names)
                contig_names.append(cname)
            cid = contig_id[cname]
            if "transcript_type" in r and r["transcript_type"] is not None:
                last_ttype = r["transcript_type"]
            flags = _STRAND.get(r.get("strand", "."), abi.STRAND_UNKNOWN)
            if "rRNA" in last_ttype:
                flags |= abi.FF_RIBOSOMAL
            start, end = int(r["start"]), int(r["end"])
            if end < start:
                raise ValueError("feature with end < start is not supported")
            if r["type"] == "gene":
                gid = r["gene_id"]
                if gid in gene_index:
                    raise ValueError("Detected non-unique Gene ID: " + gid)
                gene_index[gid] = len(gene_ids)
                gene_ids.append(gid)
                gene_names.append(r.get("gene_name") or gid)
                g_rows.append((cid, start, order, end, flags, gid))
            elif r["type"] == "exon":
                gid = r["gene_id"]
                eid = r.get("exon_id")
                if eid is None:
                    exon_counter[gid] = exon_counter.get(gid, 0) + 1
                    eid = "%s_%d" % (gid, exon_counter[gid])
                if eid in exon_seen:
                    raise ValueError("Detected non-unique Exon ID: " + eid)
                exon_seen.add(eid)
                exon_ids.append(eid)
                exon_gene_names.append(r.get("gene_name") or gid)
                e_rows.append((cid, start, order, end, flags, len(exon_ids) - 1, gid))
        # phantom genes: gene_ids that only exon rows name
        listed = len(gene_ids)
        phantom = {}
        for row in e_rows:
            gid = row[6]
            if gid not in gene_index and gid not in phantom:
                phantom[gid] = listed + len(phantom)
        n_genes = listed + len(phantom)

        def gidx(gid):
            return gene_index[gid] if gid in gene_index else phantom[gid]

        g_rows.sort(key=lambda t: (t[0], t[1], t[2]))
        e_rows.sort(key=lambda t: (t[0], t[1], t[2]))
        gr = np.array([(t[0], t[1], t[3], t[4], gidx(t[5])) for t in g_rows], dtype=np.int64).reshape(-1, 5)
        er = np.array([(t[0], t[1], t[3], t[4], t[5], gidx(t[6])) for t in e_rows], dtype=np.int64).reshape(-1, 6)
        coding = np.zeros(n_genes, dtype=np.int64)
        per_gene = [[] for _ in range(n_genes)]
        for row_i in range(er.shape[0]):
            g = int(er[row_i, 5])
            per_gene[g].append(row_i)
            coding[g] += er[row_i, 2] - er[row_i, 1] + 1
        off = np.zeros(n_genes + 1, dtype=np.uint32)
        for g in range(n_genes):
            off[g + 1] = off[g] + len(per_gene[g])
        ge_row = np.array([x for lst in per_gene for x in lst], dtype=np.uint32)
        globin = np.zeros(n_genes, dtype=np.uint8)
        for i, nm in enumerate(gene_names):
            globin[i] = 1 if nm in GLOBINS else 0
        return Annotation(
            contig_names=contig_names, n_ref=len(bam_contigs), gene_ids=gene_ids, gene_names=gene_names,
            exon_ids=exon_ids, exon_gene_names=exon_gene_names, n_genes=n_genes,
            gene_row_contig=gr[:, 0].astype(np.int32), gene_row_start=gr[:, 1].astype(np.int32),
            gene_row_end=gr[:, 2].astype(np.int32), gene_row_flags=gr[:, 3].astype(np.uint8),
            gene_row_id=gr[:, 4].astype(np.uint32),
            exon_row_contig=er[:, 0].astype(np.int32), exon_row_start=er[:, 1].astype(np.int32),
            exon_row_end=er[:, 2].astype(np.int32), exon_row_flags=er[:, 3].astype(np.uint8),
            exon_row_id=er[:, 4].astype(np.uint32), exon_row_gene=er[:, 5].astype(np.uint32),
            gene_is_globin=globin, gene_exon_off=off, gene_exon_row=ge_row, coding_length=coding,
            gene_row_order=np.array([t[2] for t in g_rows], dtype=np.uint32),
            exon_row_order=np.array([t[2] for t in e_rows], dtype=np.uint32),
        )

    # ------------------------------------------------------------------ ABI
    def to_struct(self) -> abi.AnnotationStruct:
        s = abi.AnnotationStruct()
        s.n_ref, s.n_contigs = self.n_ref, self.n_contigs
        s.n_genes, s.n_genes_listed, s.n_exons = self.n_genes, self.n_genes_listed, self.n_exons
        for f in ("gene_row_contig", "gene_row_start", "gene_row_end", "gene_row_flags", "gene_row_id",
                  "exon_row_contig", "exon_row_start", "exon_row_end", "exon_row_flags", "exon_row_id",
                  "exon_row_gene", "gene_is_globin", "gene_exon_off", "gene_exon_row"):
            a = np.ascontiguousarray(getattr(self, f))
            setattr(self, f, a)
            setattr(s, f, abi.ptr(a))
        for f in ("gene_row_order", "exon_row_order"):
            if getattr(self, f) is not None:
                a = np.ascontiguousarray(getattr(self, f), dtype=np.uint32)
                setattr(self, f, a)
                setattr(s, f, abi.ptr(a))
        return s

    def to_gtf_lines(self):
        """GTF text with the same content (for CLI tests).  Rows are emitted in
        gene order, each gene row followed by its exons."""
        lines = []
        row_of_gene = {int(g): i for i, g in enumerate(self.gene_row_id)}
        for g, gid in enumerate(self.gene_ids):
            i = row_of_gene[g]
            strand = {0: "+", 1: "-", 2: "."}[int(self.gene_row_flags[i]) & 3]
            ttype = "rRNA" if int(self.gene_row_flags[i]) & abi.FF_RIBOSOMAL else "protein_coding"
            cname = self.contig_names[int(self.gene_row_contig[i])]
            attr = 'gene_id "%s"; transcript_id "%s"; gene_name "%s"; transcript_type "%s";' % (
                gid, gid, self.gene_names[g], ttype)
            lines.append("\t".join([cname, "synth", "gene", str(int(self.gene_row_start[i])),
                                    str(int(self.gene_row_end[i])), ".", strand, ".", attr]))
            for k in range(int(self.gene_exon_off[g]), int(self.gene_exon_off[g + 1])):
                r = int(self.gene_exon_row[k])
                eid = self.exon_ids[int(self.exon_row_id[r])]
                lines.append("\t".join([cname, "synth", "exon", str(int(self.exon_row_start[r])),
                                        str(int(self.exon_row_end[r])), ".", strand, ".",
                                        attr + ' exon_id "%s";' % eid]))
        return lines


@dataclass
class Bed:
    contig: np.ndarray
    start: np.ndarray     # stored +1 (src/BED.cpp:31)
    end: np.ndarray       # stored +1 (src/BED.cpp:33)

    @staticmethod
    def from_intervals(contig, bed_start, bed_end) -> "Bed":
        return Bed(np.asarray(contig, dtype=np.int32), np.asarray(bed_start, dtype=np.int32) + 1,
                   np.asarray(bed_end, dtype=np.int32) + 1)

    def to_struct(self) -> abi.BedStruct:
        s = abi.BedStruct()
        s.n_intervals = len(self.contig)
        for f in ("contig", "start", "end"):
            a = np.ascontiguousarray(getattr(self, f), dtype=np.int32)
            setattr(self, f, a)
            setattr(s, f, abi.ptr(a))
        return s


@dataclass
class Reference:
    """--fasta: base strings of the contigs the FASTA index names (rsqc_reference)."""
    contig: list                 # boundary contig ids
    sequence: list               # one uint8 array (ASCII) per contig

    def to_struct(self) -> abi.ReferenceStruct:
        import ctypes as C
        s = abi.ReferenceStruct()
        s.n = len(self.contig)
        self._contig = np.ascontiguousarray(self.contig, dtype=np.int32)
        self._length = np.array([len(x) for x in self.sequence], dtype=np.uint64)
        self.sequence = [np.ascontiguousarray(x, dtype=np.uint8) for x in self.sequence]
        self._ptrs = (C.c_void_p * max(s.n, 1))(*[x.ctypes.data for x in self.sequence])
        s.contig, s.length = abi.ptr(self._contig), abi.ptr(self._length)
        s.sequence = C.cast(self._ptrs, C.c_void_p)
        return s


@dataclass
class Batch:
    """Structure-of-arrays batch of alignment records in file order."""
    pos: np.ndarray
    mpos: np.ndarray
    isize: np.ndarray
    qhash: np.ndarray
    cigar_off: np.ndarray
    flag: np.ndarray
    l_qseq: np.ndarray
    mapq: np.ndarray
    nm: np.ndarray
    tagbits: np.ndarray
    n_cigar: np.ndarray
    cigar: np.ndarray
    seg_tid: np.ndarray
    seg_start: np.ndarray
    wide_index: np.ndarray = field(default_factory=lambda: np.zeros(0, np.uint64))
    wide_nm: np.ndarray = field(default_factory=lambda: np.zeros(0, np.int32))
    wide_l_qseq: np.ndarray = field(default_factory=lambda: np.zeros(0, np.int32))
    wide_n_cigar: np.ndarray = field(default_factory=lambda: np.zeros(0, np.uint32))
    qname_off: np.ndarray | None = None
    qname: np.ndarray | None = None
    file_index_base: int = 0
    qhash2: np.ndarray | None = None      # rsqc_batch.qhash2 (second name hash, uint32 per record); None = the 64-bit identity
    seg_file_index: np.ndarray | None = None   # rsqc_batch.seg_file_index: the batch is several file ranges, one per segment (uint64 per segment)

    _DT = dict(pos=np.int32, mpos=np.int32, isize=np.int32, qhash=np.uint64, cigar_off=np.uint32,
               flag=np.uint16, l_qseq=np.uint16, mapq=np.uint8, nm=np.uint8, tagbits=np.uint8,
               n_cigar=np.uint8, cigar=np.uint32, seg_tid=np.int32, seg_start=np.uint64,
               wide_index=np.uint64, wide_nm=np.int32, wide_l_qseq=np.int32, wide_n_cigar=np.uint32)

    @property
    def n(self) -> int:
        return len(self.pos)

    @property
    def algorithmic_bytes(self) -> int:
        """32 B per record + 4 B per CIGAR op."""
        return 32 * self.n + 4 * len(self.cigar)

    def tid_per_record(self) -> np.ndarray:
        t = np.empty(self.n, dtype=np.int32)
        for s in range(len(self.seg_tid)):
            t[int(self.seg_start[s]):int(self.seg_start[s + 1])] = self.seg_tid[s]
        return t

    def to_struct(self) -> abi.BatchStruct:
        s = abi.BatchStruct()
        for f, dt in self._DT.items():
            a = np.ascontiguousarray(getattr(self, f), dtype=dt)
            setattr(self, f, a)
        s.n = self.n
        s.file_index_base = self.file_index_base
        # pack the two 16-byte half-record arrays of the boundary
        core = np.empty(self.n, dtype=abi.REC_CORE)
        aux = np.empty(self.n, dtype=abi.REC_AUX)
        for f in ("pos", "mpos", "isize", "cigar_off"):
            core[f] = getattr(self, f)
        for f in ("qhash", "flag", "l_qseq", "mapq", "nm", "tagbits", "n_cigar"):
            aux[f] = getattr(self, f)
        self._core, self._aux = core, aux
        s.core, s.aux = abi.ptr(core), abi.ptr(aux)
        for f in ("cigar", "seg_tid", "seg_start", "wide_index", "wide_nm", "wide_l_qseq", "wide_n_cigar"):
            setattr(s, f, abi.ptr(getattr(self, f)))
        s.n_cigar_total = len(self.cigar)
        s.n_seg = len(self.seg_tid)
        s.n_wide = len(self.wide_index)
        if self.qname is not None:
            self.qname_off = np.ascontiguousarray(self.qname_off, dtype=np.uint32)
            self.qname = np.ascontiguousarray(self.qname, dtype=np.uint8)
            s.qname_off, s.qname = abi.ptr(self.qname_off), abi.ptr(self.qname)
        if self.qhash2 is not None:
            self.qhash2 = np.ascontiguousarray(self.qhash2, dtype=np.uint32)
            s.qhash2 = abi.ptr(self.qhash2)
        if self.seg_file_index is not None:
            self.seg_file_index = np.ascontiguousarray(self.seg_file_index, dtype=np.uint64)
            assert len(self.seg_file_index) == len(self.seg_tid)
            s.seg_file_index = abi.ptr(self.seg_file_index)
        return s

    # ---------------------------------------------------------------- builders
    @staticmethod
    def from_records(records: Sequence[dict], hash_names: bool = True) -> "Batch":
        """records (file order): dict(tid,pos,mpos,mtid,isize,flag,mapq,l_qseq,cigar=[(op,len)..],
        nm=None|int, ch=bool, tags=[bool..], qname=str).  Small-case builder for tests."""
        n = len(records)
        b = dict(pos=np.zeros(n, np.int32), mpos=np.zeros(n, np.int32), isize=np.zeros(n, np.int32),
                 qhash=np.zeros(n, np.uint64), cigar_off=np.zeros(n, np.uint32), flag=np.zeros(n, np.uint16),
                 l_qseq=np.zeros(n, np.uint16), mapq=np.zeros(n, np.uint8), nm=np.zeros(n, np.uint8),
                 tagbits=np.zeros(n, np.uint8), n_cigar=np.zeros(n, np.uint8))
        cig, wide = [], []
        seg_tid, seg_start = [], []
        names, noff = bytearray(), [0]
        qh2 = np.zeros(n, np.uint32)
        for i, r in enumerate(records):
            tid = int(r.get("tid", 0))
            if not seg_tid or seg_tid[-1] != tid:
                seg_tid.append(tid)
                seg_start.append(i)
            b["pos"][i] = r.get("pos", 0)
            b["mpos"][i] = r.get("mpos", 0)
            b["isize"][i] = r.get("isize", 0)
            b["flag"][i] = r.get("flag", 0)
            b["mapq"][i] = r.get("mapq", 255)
            qn = r.get("qname", "q%d" % i).encode()
            names += qn
            noff.append(len(names))
            b["qhash"][i] = abi.qname_hash(qn)
            qh2[i] = abi.qname_hash2(qn)
            tb = 0
            nmv = r.get("nm", 0)
            if nmv is not None:
                tb |= abi.TB_HAS_NM
            else:
                nmv = 0
            if r.get("ch", False):
                tb |= abi.TB_HAS_CH
            if r.get("mtid", tid) == tid:
                tb |= abi.TB_MTID_SAME
            for k, present in enumerate(r.get("tags", [])):
                if present:
                    tb |= abi.TB_FILTER0 << k
            b["tagbits"][i] = tb
            c = r.get("cigar", [])
            b["cigar_off"][i] = len(cig)
            for op, ln in c:
                cig.append((int(ln) << 4) | int(op))
            lq = int(r.get("l_qseq", sum(ln for op, ln in c if op in (abi.CIG_M, abi.CIG_I, abi.CIG_S, abi.CIG_EQ, abi.CIG_X))))
            need_wide = lq >= abi.LQSEQ_ESCAPE or nmv >= abi.NM_ESCAPE or nmv < 0 or len(c) >= abi.NCIGAR_ESCAPE
            b["l_qseq"][i] = abi.LQSEQ_ESCAPE if lq >= abi.LQSEQ_ESCAPE else lq
            b["nm"][i] = abi.NM_ESCAPE if (nmv >= abi.NM_ESCAPE or nmv < 0) else nmv
            b["n_cigar"][i] = abi.NCIGAR_ESCAPE if len(c) >= abi.NCIGAR_ESCAPE else len(c)
            if need_wide:
                wide.append((i, nmv, lq, len(c)))
        seg_start.append(n)
        return Batch(cigar=np.array(cig, dtype=np.uint32), seg_tid=np.array(seg_tid, np.int32),
                     seg_start=np.array(seg_start, np.uint64),
                     wide_index=np.array([w[0] for w in wide], np.uint64),
                     wide_nm=np.array([w[1] for w in wide], np.int32),
                     wide_l_qseq=np.array([w[2] for w in wide], np.int32),
                     wide_n_cigar=np.array([w[3] for w in wide], np.uint32),
                     qname_off=np.array(noff, np.uint32), qname=np.frombuffer(bytes(names), dtype=np.uint8).copy(),
                     qhash2=qh2, **b)

    @staticmethod
    def concat(parts: Sequence["Batch"]) -> "Batch":
        """The batches one after the other as ONE batch (cigar pool, segment table and wide table re-based).
        Adjacent segments of the same contig are merged."""
        parts = [p for p in parts if p.n]
        if not parts:
            return Batch.from_records([])
        if len(parts) == 1:
            return parts[0]
        n_at = np.concatenate([[0], np.cumsum([p.n for p in parts])]).astype(np.int64)
        c_at = np.concatenate([[0], np.cumsum([len(p.cigar) for p in parts])]).astype(np.int64)
        if c_at[-1] >= (1 << 32):
            raise ValueError("CIGAR pool of the concatenation exceeds 32-bit offsets")
        cat = lambda f: np.concatenate([getattr(p, f) for p in parts])
        seg_tid, seg_start = [], []
        for k, p in enumerate(parts):
            for s in range(len(p.seg_tid)):
                if int(p.seg_start[s + 1]) == int(p.seg_start[s]):
                    continue
                if seg_tid and seg_tid[-1] == int(p.seg_tid[s]):
                    continue
                seg_tid.append(int(p.seg_tid[s])); seg_start.append(int(p.seg_start[s]) + int(n_at[k]))
        seg_start.append(int(n_at[-1]))
        kw = {}
        if all(p.qname is not None for p in parts):
            q_at = np.concatenate([[0], np.cumsum([len(p.qname) for p in parts])]).astype(np.int64)
            kw = dict(qname=cat("qname"),
                      qname_off=np.concatenate([(p.qname_off[:-1].astype(np.int64) + q_at[k]) for k, p in enumerate(parts)] +
                                               [q_at[-1:]]).astype(np.uint32))
        return Batch(pos=cat("pos"), mpos=cat("mpos"), isize=cat("isize"), qhash=cat("qhash"),
                     cigar_off=np.concatenate([(p.cigar_off.astype(np.int64) + c_at[k]) for k, p in enumerate(parts)]).astype(np.uint32),
                     flag=cat("flag"), l_qseq=cat("l_qseq"), mapq=cat("mapq"), nm=cat("nm"), tagbits=cat("tagbits"),
                     n_cigar=cat("n_cigar"), cigar=cat("cigar"), seg_tid=np.asarray(seg_tid, np.int32),
                     seg_start=np.asarray(seg_start, np.uint64),
                     wide_index=np.concatenate([(p.wide_index.astype(np.int64) + n_at[k]) for k, p in enumerate(parts)]).astype(np.uint64),
                     wide_nm=cat("wide_nm"), wide_l_qseq=cat("wide_l_qseq"), wide_n_cigar=cat("wide_n_cigar"),
                     file_index_base=parts[0].file_index_base,
                     qhash2=(np.concatenate([p.qhash2 for p in parts]) if all(p.qhash2 is not None for p in parts) else None), **kw)

    @staticmethod
    def concat_ranges(parts: Sequence["Batch"]) -> "Batch":
        """Batches that are NON-ADJACENT ranges of one file (the contigs a GPU owns in a contig-sharded run), in file order, as ONE
        batch: every segment keeps the file index of its first record (rsqc_batch.seg_file_index), so that one kernel launch
        serves all of them and the order-dependent outputs are still kept per range.  A part's own file_index_base (+ the offset
        of the segment inside the part) is the segment's index."""
        every = list(parts)
        parts = [p for p in every if p.n]
        if not parts:                                   # a rank that owns no records: an empty batch of ranges
            import copy
            if not every:
                raise ValueError("concat_ranges: no parts")
            one = copy.copy(every[0])
            one.seg_file_index = np.zeros(0, np.uint64)
            return one
        # (a single part goes through concat as well: it drops the part's empty segments, which the index list below skips too)
        one = Batch.concat(parts)
        idx = []
        for p in parts:
            for s in range(len(p.seg_tid)):
                if int(p.seg_start[s + 1]) > int(p.seg_start[s]):
                    idx.append(int(p.file_index_base) + int(p.seg_start[s]))
        if len(idx) != len(one.seg_tid):
            raise ValueError("concat_ranges: adjacent parts share a contig (segments were merged)")
        import copy
        one = copy.copy(one)
        one.seg_file_index = np.asarray(idx, np.uint64)
        one.file_index_base = idx[0] if idx else 0
        return one

    def take(self, idx) -> "Batch":
        """The records idx[0], idx[1], ... as a new batch (any order, e.g. a coordinate sort of a concatenation)."""
        idx = np.asarray(idx, dtype=np.int64)
        n = self.n
        cend = np.empty(n + 1, dtype=np.int64); cend[:n] = self.cigar_off; cend[n] = len(self.cigar)
        # CIGAR length of a record = distance to the next record's offset IN STORAGE ORDER
        order = np.argsort(self.cigar_off, kind="stable")
        clen = np.zeros(n, np.int64)
        nxt = np.append(self.cigar_off[order][1:].astype(np.int64), len(self.cigar))
        clen[order] = nxt - self.cigar_off[order].astype(np.int64)
        counts = clen[idx]
        total = int(counts.sum())
        out_off = np.cumsum(counts) - counts
        flat = self.cigar[(np.arange(total) - np.repeat(out_off, counts) + np.repeat(self.cigar_off[idx].astype(np.int64), counts))] if total else self.cigar[:0]
        tid = self.tid_per_record()[idx]
        m = len(idx)
        if m:
            change = np.flatnonzero(np.diff(tid)) + 1
            starts = np.concatenate([[0], change])
            seg_tid, seg_start = tid[starts], np.concatenate([starts, [m]])
        else:
            seg_tid, seg_start = np.zeros(0, np.int32), np.zeros(1, np.uint64)
        pos_of = {int(w): k for k, w in enumerate(self.wide_index)}
        wsel = [(j, pos_of[int(i)]) for j, i in enumerate(idx) if int(i) in pos_of] if len(pos_of) else []
        kw = {}
        if self.qname is not None:
            qlen = (self.qname_off[1:].astype(np.int64) - self.qname_off[:-1].astype(np.int64))[idx]
            qo = np.concatenate([[0], np.cumsum(qlen)])
            qflat = self.qname[(np.arange(int(qlen.sum())) - np.repeat(qo[:-1], qlen) + np.repeat(self.qname_off[:-1].astype(np.int64)[idx], qlen))]
            kw = dict(qname=qflat, qname_off=qo.astype(np.uint32))
        return Batch(pos=self.pos[idx], mpos=self.mpos[idx], isize=self.isize[idx], qhash=self.qhash[idx],
                     cigar_off=out_off.astype(np.uint32), flag=self.flag[idx], l_qseq=self.l_qseq[idx], mapq=self.mapq[idx],
                     nm=self.nm[idx], tagbits=self.tagbits[idx], n_cigar=self.n_cigar[idx], cigar=flat,
                     seg_tid=np.asarray(seg_tid, np.int32), seg_start=np.asarray(seg_start, np.uint64),
                     wide_index=np.array([w[0] for w in wsel], np.uint64), wide_nm=self.wide_nm[[w[1] for w in wsel]] if wsel else np.zeros(0, np.int32),
                     wide_l_qseq=self.wide_l_qseq[[w[1] for w in wsel]] if wsel else np.zeros(0, np.int32),
                     wide_n_cigar=self.wide_n_cigar[[w[1] for w in wsel]] if wsel else np.zeros(0, np.uint32),
                     qhash2=(self.qhash2[idx] if self.qhash2 is not None else None), **kw)

    def coordinate_sorted(self) -> "Batch":
        """Stable sort by (contig, position), unplaced records last: what a coordinate-sorted BAM holds."""
        tid = self.tid_per_record().astype(np.int64)
        key = np.where(tid < 0, np.int64(1) << 40, tid)
        return self.take(np.lexsort((self.pos.astype(np.int64), key)))

    def slice(self, lo: int, hi: int) -> "Batch":
        """Records [lo, hi) as an independent batch (cigar pool re-based)."""
        lo, hi = int(lo), int(hi)
        n = self.n
        cend = np.empty(n + 1, dtype=np.int64)
        cend[:n] = self.cigar_off
        cend[n] = len(self.cigar)
        c0, c1 = int(cend[lo]), int(cend[hi]) if hi <= n else len(self.cigar)
        tid = self.tid_per_record()[lo:hi]
        seg_tid, seg_start = [], []
        if hi > lo:
            change = np.flatnonzero(np.diff(tid)) + 1
            starts = np.concatenate([[0], change])
            seg_tid = tid[starts]
            seg_start = np.concatenate([starts, [hi - lo]])
        else:
            seg_start = [0]
        wm = (self.wide_index >= lo) & (self.wide_index < hi)
        kw = {}
        if self.qname is not None:
            q0, q1 = int(self.qname_off[lo]), int(self.qname_off[hi])
            kw = dict(qname_off=(self.qname_off[lo:hi + 1] - q0).astype(np.uint32), qname=self.qname[q0:q1].copy())
        return Batch(pos=self.pos[lo:hi].copy(), mpos=self.mpos[lo:hi].copy(), isize=self.isize[lo:hi].copy(),
                     qhash=self.qhash[lo:hi].copy(), cigar_off=(self.cigar_off[lo:hi] - c0).astype(np.uint32),
                     flag=self.flag[lo:hi].copy(), l_qseq=self.l_qseq[lo:hi].copy(), mapq=self.mapq[lo:hi].copy(),
                     nm=self.nm[lo:hi].copy(), tagbits=self.tagbits[lo:hi].copy(), n_cigar=self.n_cigar[lo:hi].copy(),
                     cigar=self.cigar[c0:c1].copy(), seg_tid=np.asarray(seg_tid, np.int32),
                     seg_start=np.asarray(seg_start, np.uint64),
                     wide_index=(self.wide_index[wm] - lo).astype(np.uint64), wide_nm=self.wide_nm[wm].copy(),
                     wide_l_qseq=self.wide_l_qseq[wm].copy(), wide_n_cigar=self.wide_n_cigar[wm].copy(),
                     file_index_base=self.file_index_base + lo,
                     qhash2=(self.qhash2[lo:hi].copy() if self.qhash2 is not None else None), **kw)
