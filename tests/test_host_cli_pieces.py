"""CPU: the CLI's host-side C++ (GTF/BED ingest, BAM decode, report writers, library-complexity search)
through rnaseqc_amd/lib/librsqc_host.so."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from rnaseqc_amd import abi, bamio, synth
from rnaseqc_amd.model import Annotation, Batch
from tests import cases, report_ref

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SO = os.path.join(ROOT, "rnaseqc_amd", "lib", "librsqc_host.so")


@pytest.fixture(scope="module")
def host():
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "rnaseqc_amd", "csrc"), "../lib/librsqc_host.so"])
    lib = C.CDLL(SO)
    lib.host_annotation_load.restype = C.c_void_p
    lib.host_annotation_load.argtypes = [C.c_char_p, C.c_char_p, C.POINTER(C.c_char_p), C.c_int, C.POINTER(C.c_int)]
    lib.host_annotation_load_ex.restype = C.c_void_p
    lib.host_annotation_load_ex.argtypes = [C.c_char_p, C.c_char_p, C.POINTER(C.c_char_p), C.c_int, C.c_int, C.POINTER(C.c_int)]
    lib.host_annotation_struct.restype = C.POINTER(abi.AnnotationStruct); lib.host_annotation_struct.argtypes = [C.c_void_p]
    lib.host_annotation_bed.restype = C.POINTER(abi.BedStruct); lib.host_annotation_bed.argtypes = [C.c_void_p]
    for f in ("host_annotation_gene_name", "host_annotation_gene_id", "host_annotation_exon_id"):
        getattr(lib, f).restype = C.c_char_p; getattr(lib, f).argtypes = [C.c_void_p, C.c_int]
    lib.host_annotation_coding_length.restype = C.c_longlong; lib.host_annotation_coding_length.argtypes = [C.c_void_p, C.c_int]
    lib.host_annotation_free.argtypes = [C.c_void_p]
    lib.host_write_reports.argtypes = [C.c_void_p, C.POINTER(abi.ResultsStruct), C.c_char_p, C.c_char_p, C.c_int, C.c_int, C.c_int,
                                       C.c_uint, C.POINTER(C.c_char_p), C.c_int, C.POINTER(C.c_int), C.c_int]
    lib.host_library_complexity.restype = C.c_uint; lib.host_library_complexity.argtypes = [C.c_double, C.c_double, C.c_double]
    lib.host_bam_read_all.restype = C.c_void_p; lib.host_bam_read_all.argtypes = [C.c_char_p, C.c_char_p, C.POINTER(C.c_char_p), C.c_int]
    lib.host_bam_batch.restype = C.POINTER(abi.BatchStruct); lib.host_bam_batch.argtypes = [C.c_void_p]
    lib.host_bam_n_contigs.argtypes = [C.c_void_p]
    lib.host_bam_contig.restype = C.c_char_p; lib.host_bam_contig.argtypes = [C.c_void_p, C.c_int]
    lib.host_bam_free.argtypes = [C.c_void_p]
    return lib


def _arr(p, n, dt):
    return abi._view(p, n, dt)


def load_annotation(host, gtf, contigs, bed=None):
    names = (C.c_char_p * len(contigs))(*[c.encode() for c in contigs])
    err = C.c_int()
    h = host.host_annotation_load(gtf.encode(), (bed or "").encode(), names, len(contigs), C.byref(err))
    return h, err.value


def test_gtf_ingest_matches_python_mirror(host, tmp_path):
    ann = synth.make_annotation(seed=31, contigs=[("chrA", 2_000_000, 120), ("chrB", 900_000, 60), ("chrC", 500_000, 20)])
    gtf = str(tmp_path / "a.gtf")
    bamio.write_gtf(gtf, ann)
    # BAM header order differs from GTF order and has an extra contig; chrC is GTF-only
    bam_contigs = ["chrB", "chrZ", "chrA"]
    h, err = load_annotation(host, gtf, bam_contigs)
    assert err == 0
    s = host.host_annotation_struct(h).contents
    assert (s.n_ref, s.n_contigs, s.n_genes_listed, s.n_exons) == (3, 4, ann.n_genes_listed, ann.n_exons)
    # python mirror on the same rows with the same BAM contig order
    rows = []
    row_of_gene = {int(g): i for i, g in enumerate(ann.gene_row_id)}
    for g, gid in enumerate(ann.gene_ids):
        i = row_of_gene[g]
        strand = {0: "+", 1: "-", 2: "."}[int(ann.gene_row_flags[i]) & 3]
        tt = "rRNA" if int(ann.gene_row_flags[i]) & abi.FF_RIBOSOMAL else "protein_coding"
        cname = ann.contig_names[int(ann.gene_row_contig[i])]
        rows.append(dict(contig=cname, type="gene", start=int(ann.gene_row_start[i]), end=int(ann.gene_row_end[i]), strand=strand,
                         gene_id=gid, gene_name=ann.gene_names[g], transcript_type=tt))
        for k in range(int(ann.gene_exon_off[g]), int(ann.gene_exon_off[g + 1])):
            r = int(ann.gene_exon_row[k])
            rows.append(dict(contig=cname, type="exon", start=int(ann.exon_row_start[r]), end=int(ann.exon_row_end[r]), strand=strand,
                             gene_id=gid, exon_id=ann.exon_ids[int(ann.exon_row_id[r])], gene_name=ann.gene_names[g], transcript_type=tt))
    py = Annotation.from_rows(bam_contigs, rows)
    L, E = py.n_genes_listed, py.n_exons
    for f, n, dt in [("gene_row_contig", L, np.int32), ("gene_row_start", L, np.int32), ("gene_row_end", L, np.int32),
                     ("gene_row_flags", L, np.uint8), ("gene_row_id", L, np.uint32), ("exon_row_contig", E, np.int32),
                     ("exon_row_start", E, np.int32), ("exon_row_end", E, np.int32), ("exon_row_flags", E, np.uint8),
                     ("exon_row_id", E, np.uint32), ("exon_row_gene", E, np.uint32), ("gene_is_globin", py.n_genes, np.uint8),
                     ("gene_exon_off", py.n_genes + 1, np.uint32), ("gene_exon_row", E, np.uint32)]:
        np.testing.assert_array_equal(_arr(getattr(s, f), n, dt), getattr(py, f), err_msg=f)
    assert host.host_annotation_gene_name(h, 5).decode() == ann.gene_names[5]
    assert host.host_annotation_coding_length(h, 7) == int(ann.coding_length[7])
    host.host_annotation_free(h)


def test_gtf_ingest_legacy_leaves_out_one_base_features(host, tmp_path):
    # src/RNASeQC.cpp:129-135: under --legacy a feature with end == start is not added to the feature lists (and a
    # 1-base exon takes 1 off its gene's coding length), but operator>> has already entered it in geneList / exonList
    lines = ['c1\tx\tgene\t100\t900\t.\t+\t.\tgene_id "A"; gene_name "A";',
             'c1\tx\texon\t100\t300\t.\t+\t.\tgene_id "A"; exon_id "A1";',
             'c1\tx\texon\t500\t500\t.\t+\t.\tgene_id "A"; exon_id "A2";',          # 1-base exon
             'c1\tx\texon\t700\t900\t.\t+\t.\tgene_id "A"; exon_id "A3";',
             'c1\tx\tgene\t950\t950\t.\t-\t.\tgene_id "B"; gene_name "B";',          # 1-base gene
             'c1\tx\texon\t950\t950\t.\t-\t.\tgene_id "B"; exon_id "B1";']
    gtf = str(tmp_path / "l.gtf")
    open(gtf, "w").write("\n".join(lines) + "\n")
    names = (C.c_char_p * 1)(b"c1")
    err = C.c_int()
    h = host.host_annotation_load_ex(gtf.encode(), b"", names, 1, 1, C.byref(err))
    assert err.value == 0
    s = host.host_annotation_struct(h).contents
    assert (s.n_ref, s.n_contigs, s.n_genes_listed, s.n_genes, s.n_exons) == (1, 2, 2, 3, 4)      # + the parking contig and gene
    assert list(_arr(s.gene_row_contig, 2, np.int32)) == [0, 1] and list(_arr(s.gene_row_id, 2, np.uint32)) == [0, 1]
    assert list(_arr(s.exon_row_contig, 4, np.int32)) == [0, 0, 1, 1]
    assert list(_arr(s.exon_row_id, 4, np.uint32)) == [0, 2, 1, 3]                                 # exonList order kept as ids
    assert list(_arr(s.exon_row_gene, 4, np.uint32)) == [0, 0, 2, 2]
    assert list(_arr(s.gene_exon_off, 4, np.uint32)) == [0, 2, 2, 4]
    assert list(_arr(s.gene_row_order, 2, np.uint32)) == [0, 4] and list(_arr(s.exon_row_order, 4, np.uint32)) == [1, 3, 2, 5]
    assert host.host_annotation_coding_length(h, 0) == 201 + 1 + 201 - 1 and host.host_annotation_coding_length(h, 1) == 0
    host.host_annotation_free(h)
    # without --legacy the same file keeps every row
    h = host.host_annotation_load_ex(gtf.encode(), b"", names, 1, 0, C.byref(err))
    s = host.host_annotation_struct(h).contents
    assert (s.n_contigs, s.n_genes, s.n_exons) == (1, 2, 4) and host.host_annotation_coding_length(h, 0) == 403
    host.host_annotation_free(h)


def test_gtf_ingest_genome_scale(host, tmp_path):
    """332 k lines (56 202 genes, 25 contigs): the interned-id table grows several times; the flattened annotation must
    equal the generator's, which was flattened by the Python mirror of the same rules."""
    ann = synth.make_annotation(seed=3, contigs=synth.human_contigs())
    gtf = str(tmp_path / "g.gtf")
    bamio.write_gtf(gtf, ann)
    h, err = load_annotation(host, gtf, list(ann.contig_names))
    assert err == 0
    s = host.host_annotation_struct(h).contents
    L, E = ann.n_genes_listed, ann.n_exons
    assert (s.n_genes_listed, s.n_exons, s.n_genes) == (L, E, ann.n_genes)
    for f, n, dt in [("gene_row_contig", L, np.int32), ("gene_row_start", L, np.int32), ("gene_row_end", L, np.int32),
                     ("gene_row_flags", L, np.uint8), ("gene_row_id", L, np.uint32), ("exon_row_contig", E, np.int32),
                     ("exon_row_start", E, np.int32), ("exon_row_end", E, np.int32), ("exon_row_flags", E, np.uint8),
                     ("exon_row_id", E, np.uint32), ("exon_row_gene", E, np.uint32), ("gene_is_globin", ann.n_genes, np.uint8),
                     ("gene_exon_off", ann.n_genes + 1, np.uint32), ("gene_exon_row", E, np.uint32)]:
        np.testing.assert_array_equal(_arr(getattr(s, f), n, dt), np.asarray(getattr(ann, f)), err_msg=f)
    assert host.host_annotation_gene_name(h, L - 1).decode() == ann.gene_names[L - 1]
    assert host.host_annotation_coding_length(h, L // 2) == int(ann.coding_length[L // 2])
    host.host_annotation_free(h)


def test_gtf_quirks(host, tmp_path):
    # Q11 exon-id inference, Q16 transcript_type leak, duplicate ids and a blank line are fatal (exit 11)
    g = tmp_path / "q.gtf"
    g.write_text("#c\n"
                 'c1\tx\tgene\t100\t900\t.\t+\t.\tgene_id "G1"; gene_name "HBB"; transcript_type "rRNA";\n'
                 'c1\tx\texon\t100\t300\t.\t+\t.\tgene_id "G1";\n'
                 'c1\tx\ttranscript\t100\t900\t.\t+\t.\tgene_id "G1"; transcript_id "T1";\n'
                 'c1\tx\tgene\t2000\t2500\t.\t-\t.\tgene_id "G2";\n'
                 'c1\tx\texon\t2000\t2500\t.\t-\t.\tgene_id "G2"; exon_id "E2";\n')
    h, err = load_annotation(host, str(g), ["c1"])
    assert err == 0
    s = host.host_annotation_struct(h).contents
    assert host.host_annotation_exon_id(h, 0).decode() == "G1_1"
    # G2 and its exon carry no transcript_type: they inherit "rRNA" from the lines before
    assert list(_arr(s.gene_row_flags, 2, np.uint8)) == [abi.STRAND_FORWARD | abi.FF_RIBOSOMAL, abi.STRAND_REVERSE | abi.FF_RIBOSOMAL]
    assert list(_arr(s.gene_is_globin, 2, np.uint8)) == [1, 0]
    assert host.host_annotation_gene_name(h, 1).decode() == "G2"
    host.host_annotation_free(h)
    for bad in ['c1\tx\tgene\t1\t5\t.\t+\t.\tgene_id "A";\nc1\tx\tgene\t7\t9\t.\t+\t.\tgene_id "A";\n',
                'c1\tx\tgene\t1\t5\t.\t+\t.\tgene_id "A";\n\nc1\tx\texon\t1\t5\t.\t+\t.\tgene_id "A";\n',
                'c1\tx\texon\t1\t5\t.\t+\t.\ttranscript_id "A";\n']:
        g.write_text(bad)
        assert load_annotation(host, str(g), ["c1"]) == (None, 11)
    assert load_annotation(host, str(tmp_path / "missing.gtf"), ["c1"]) == (None, 10)


def test_bam_decode_round_trip(host, tmp_path):
    ann = synth.make_annotation(seed=33, contigs=[("chrA", 800_000, 50), ("chrB", 400_000, 25)])
    batch = synth.make_reads(ann, 3000, seed=34, keep_qnames=True, chimeric_tag_frac=0.02, filter_tag_frac=0.03,
                             contig_lengths=np.array([800_000, 400_000]))
    # a record with wide fields and an unplaced mate on another contig
    path = str(tmp_path / "t.bam")
    bamio.write_bam(path, [("chrA", 800_000), ("chrB", 400_000)], batch)
    tags = (C.c_char_p * 1)(b"XF")
    h = host.host_bam_read_all(path.encode(), b"ch", tags, 1)
    assert h
    b = host.host_bam_batch(h).contents
    assert b.n == batch.n and b.n_cigar_total == len(batch.cigar)
    core = _arr(b.core, b.n, abi.REC_CORE); aux = _arr(b.aux, b.n, abi.REC_AUX)
    for f in ("pos", "mpos", "isize", "cigar_off"):
        np.testing.assert_array_equal(core[f], getattr(batch, f), err_msg=f)
    for f in ("qhash", "flag", "l_qseq", "mapq", "nm", "tagbits", "n_cigar"):
        np.testing.assert_array_equal(aux[f], getattr(batch, f), err_msg=f)
    np.testing.assert_array_equal(_arr(b.cigar, b.n_cigar_total, np.uint32), batch.cigar)
    np.testing.assert_array_equal(_arr(b.seg_tid, b.n_seg, np.int32), batch.seg_tid)
    np.testing.assert_array_equal(_arr(b.seg_start, b.n_seg + 1, np.uint64), batch.seg_start)
    assert [host.host_bam_contig(h, i).decode() for i in range(host.host_bam_n_contigs(h))] == ["chrA", "chrB"]
    host.host_bam_free(h)
    assert not host.host_bam_read_all(str(tmp_path / "nope.bam").encode(), b"ch", tags, 0)


@pytest.mark.parametrize("threads,batch_records,group_bytes", [(1, 1 << 20, 0), (8, 1 << 20, 0), (5, 7777, 0), (16, 100, 0),
                                                               (8, 1 << 20, 200_000), (3, 5000, 70_000)])
def test_bam_decode_parallel_matches_input(host, tmp_path, threads, batch_records, group_bytes, monkeypatch):
    """Parallel BGZF inflate + parallel record parsing: many blocks, records spanning block borders,
    read_batch calls that end in the middle of an inflated group."""
    ann = synth.make_annotation(seed=35, contigs=[("chrA", 3_000_000, 120), ("chrB", 1_000_000, 40), ("chrC", 500_000, 10)])
    batch = synth.make_reads(ann, 30_000, seed=36, keep_qnames=True, chimeric_tag_frac=0.02, filter_tag_frac=0.03,
                             contig_lengths=np.array([3_000_000, 1_000_000, 500_000]))
    path = str(tmp_path / "p.bam")
    bamio.write_bam(path, [("chrA", 3_000_000), ("chrB", 1_000_000), ("chrC", 500_000)], batch)
    if group_bytes:                       # many small inflated groups: every hand-over between producer and parser
        monkeypatch.setenv("RSQC_HOST_GROUP_BYTES", str(group_bytes))
    tags = (C.c_char_p * 1)(b"XF")
    host.host_bam_read_all_ex.restype = C.c_void_p
    h = host.host_bam_read_all_ex(path.encode(), b"ch", tags, 1, threads, C.c_ulonglong(batch_records))
    assert h
    b = host.host_bam_batch(C.c_void_p(h)).contents
    assert b.n == batch.n and b.n_cigar_total == len(batch.cigar)
    core = _arr(b.core, b.n, abi.REC_CORE); aux = _arr(b.aux, b.n, abi.REC_AUX)
    for f in ("pos", "mpos", "isize", "cigar_off"):
        np.testing.assert_array_equal(core[f], getattr(batch, f), err_msg=f)
    for f in ("qhash", "flag", "l_qseq", "mapq", "nm", "tagbits", "n_cigar"):
        np.testing.assert_array_equal(aux[f], getattr(batch, f), err_msg=f)
    np.testing.assert_array_equal(_arr(b.cigar, b.n_cigar_total, np.uint32), batch.cigar)
    np.testing.assert_array_equal(_arr(b.seg_tid, b.n_seg, np.int32), batch.seg_tid)
    np.testing.assert_array_equal(_arr(b.seg_start, b.n_seg + 1, np.uint64), batch.seg_start)
    np.testing.assert_array_equal(_arr(b.wide_index, b.n_wide, np.uint64), batch.wide_index)
    host.host_bam_free(C.c_void_p(h))


def test_bam_decode_many_default_groups(host, tmp_path):
    """1.2 M records (~300 MB inflated): several 64 MB groups handed from the producer thread to the parser at the
    default settings, batches of 2^20 records ending inside a group."""
    ann = synth.make_annotation(seed=37, contigs=[("chrA", 30_000_000, 600), ("chrB", 10_000_000, 200)])
    batch = synth.make_reads(ann, 600_000, seed=38, keep_qnames=True, contig_lengths=np.array([30_000_000, 10_000_000]))
    path = str(tmp_path / "big.bam")
    bamio.write_bam_fast(path, [("chrA", 30_000_000), ("chrB", 10_000_000)], batch, threads=4)
    host.host_bam_read_all_ex.restype = C.c_void_p
    h = host.host_bam_read_all_ex(path.encode(), b"ch", None, 0, 4, C.c_ulonglong(1 << 20))
    assert h
    b = host.host_bam_batch(C.c_void_p(h)).contents
    assert b.n == batch.n and b.n_cigar_total == len(batch.cigar)
    core = _arr(b.core, b.n, abi.REC_CORE); aux = _arr(b.aux, b.n, abi.REC_AUX)
    for f in ("pos", "mpos", "isize", "cigar_off"):
        np.testing.assert_array_equal(core[f], getattr(batch, f), err_msg=f)
    for f in ("qhash", "flag", "l_qseq", "mapq", "nm", "tagbits", "n_cigar"):
        np.testing.assert_array_equal(aux[f], getattr(batch, f), err_msg=f)
    np.testing.assert_array_equal(_arr(b.cigar, b.n_cigar_total, np.uint32), batch.cigar)
    np.testing.assert_array_equal(_arr(b.seg_start, b.n_seg + 1, np.uint64), batch.seg_start)
    host.host_bam_free(C.c_void_p(h))


def test_bam_decode_long_record_spans_framing_chunks(host, tmp_path):
    """A 4.5 MB record (long read, larger than the reader's 4 MB head room) between short ones: the parallel framer's chunks inside it have no record
    start to guess, and its SEQ/QUAL bytes (0x11 / 0xff runs) must not be taken for records."""
    recs = []
    for i in range(3000):
        recs.append(dict(tid=0, pos=100 + i, mpos=100 + i, isize=0, flag=0, cigar=[(abi.CIG_M, 100)], qname="s%d" % i))
    recs.append(dict(tid=0, pos=5000, mpos=5000, isize=0, flag=0, cigar=[(abi.CIG_M, 3_000_000)], qname="long"))
    for i in range(3000):
        recs.append(dict(tid=1 if i > 1500 else 0, pos=6000 + i, mpos=6000 + i, isize=0, flag=0, cigar=[(abi.CIG_M, 90), (abi.CIG_S, 10)], qname="t%d" % i))
    batch = Batch.from_records(recs)
    path = str(tmp_path / "l.bam")
    bamio.write_bam(path, [("chrA", 3_000_000), ("chrB", 1_000_000)], batch)
    host.host_bam_read_all_ex.restype = C.c_void_p
    for threads, per, group in ((1, 1 << 20, 0), (8, 1 << 20, 0), (4, 1000, 0), (8, 1 << 20, 1_000_000), (2, 300, 400_000)):
        if group:                         # the record then spans several groups and outgrows the head room
            os.environ["RSQC_HOST_GROUP_BYTES"] = str(group)
        else:
            os.environ.pop("RSQC_HOST_GROUP_BYTES", None)
        h = host.host_bam_read_all_ex(path.encode(), b"ch", None, 0, threads, C.c_ulonglong(per))
        assert h
        b = host.host_bam_batch(C.c_void_p(h)).contents
        assert b.n == batch.n
        core = _arr(b.core, b.n, abi.REC_CORE); aux = _arr(b.aux, b.n, abi.REC_AUX)
        np.testing.assert_array_equal(core["pos"], batch.pos)
        np.testing.assert_array_equal(aux["qhash"], batch.qhash)
        np.testing.assert_array_equal(aux["l_qseq"], batch.l_qseq)
        np.testing.assert_array_equal(_arr(b.cigar, b.n_cigar_total, np.uint32), batch.cigar)
        np.testing.assert_array_equal(_arr(b.wide_index, b.n_wide, np.uint64), batch.wide_index)
        np.testing.assert_array_equal(_arr(b.wide_l_qseq, b.n_wide, np.int32), batch.wide_l_qseq)
        np.testing.assert_array_equal(_arr(b.seg_tid, b.n_seg, np.int32), batch.seg_tid)
        np.testing.assert_array_equal(_arr(b.seg_start, b.n_seg + 1, np.uint64), batch.seg_start)
        host.host_bam_free(C.c_void_p(h))
    os.environ.pop("RSQC_HOST_GROUP_BYTES", None)


def test_library_complexity_matches_the_literal_loop(host, oracle_lib):
    rng = np.random.default_rng(7)
    for _ in range(40):
        unique = float(rng.integers(10, 40000))
        dup = float(rng.integers(1, int(unique)))
        limit = float(unique + rng.integers(1, 300000))
        assert host.host_library_complexity(dup, unique, limit) == oracle_lib.library_complexity(dup, unique, limit), (dup, unique, limit)
    assert host.host_library_complexity(0.0, 1000.0, 1e9) == 0
    # full-size run of the bracketed search (the literal loop needs ~6 s here)
    assert host.host_library_complexity(3000.0, 30000.0, 1e9) > 30000


def _results_struct(r):
    """abi.Results (numpy) -> ResultsStruct for the C++ report writer."""
    keep = []
    rs = abi.ResultsStruct()
    rs.n_genes_listed, rs.n_exons = len(r.gene_reads), len(r.exon_reads)
    for f, dt in [("gene_reads", np.uint64), ("gene_unique", np.uint64), ("gene_fragments", np.uint64), ("exon_reads", np.float64),
                  ("exon_hit", np.uint8), ("gene_cov_mean", np.float64), ("gene_cov_std", np.float64), ("gene_cov_cv", np.float64),
                  ("gene_cov_valid", np.uint8), ("exon_cv", np.float64), ("exon_cv_valid", np.uint8), ("bias_three", np.uint64),
                  ("bias_five", np.uint64), ("fragment_size", np.int64), ("fragment_count", np.uint64)]:
        a = np.ascontiguousarray(getattr(r, f), dtype=dt)
        keep.append(a)
        setattr(rs, f, abi.ptr(a))
    for i, v in enumerate(r.counters):
        rs.counters[i] = int(v)
    rs.read_length = r.read_length
    rs.n_fragment_sizes = len(r.fragment_size)
    rs.fragment_samples_remaining = r.fragment_samples_remaining
    if getattr(r, "have_reference", 0):
        rs.have_reference = 1
        rs.gc_out_of_range = int(r.gc_out_of_range)
        for f, dt in [("gc_bins", np.uint64), ("exon_gc", np.float64)]:
            a = np.ascontiguousarray(getattr(r, f), dtype=dt)
            keep.append(a)
            setattr(rs, f, abi.ptr(a))
    return rs, keep


def read_table(path, skip=0):
    rows = [l.rstrip("\n").split("\t") for l in open(path)]
    return rows[skip:]


def test_report_writer_on_the_quirk_case(host, oracle_lib, tmp_path):
    ann, batch = cases.quirk_case()
    gtf = str(tmp_path / "q.gtf")
    bamio.write_gtf(gtf, ann)
    h, err = load_annotation(host, gtf, ["chr1", "chr2"])
    assert err == 0
    p = abi.default_params()
    r = oracle_lib.run_oracle(p, ann, [batch])
    rs, keep = _results_struct(r)
    out = str(tmp_path / "out")
    os.makedirs(out)
    visit = (C.c_int * 2)(0, 1)
    assert host.host_write_reports(h, C.byref(rs), out.encode(), b"quirk.bam", 0, 0, 1, 5, None, 0, visit, 2) == 0
    m = dict(read_table(os.path.join(out, "quirk.bam.metrics.tsv")))
    c = r.counter_dict()
    # the rate block must equal the Python restatement that reproduces the reference's golden files
    for key, val in report_ref.metrics_rates(c):
        assert m[key] == report_ref.fmt(val), key
    # the counter block against the REFERENCE's own operator<<(ofstream&, Metrics&) when it is built
    if oracle_lib.ref_lib() is not None:
        ref_path = str(tmp_path / "ref_counters.tsv")
        oracle_lib.ref_metrics_print({k: v for k, v in c.items() if not k.startswith("Filtered by tag")}, ref_path)
        ref_rows = read_table(ref_path)
        ours = read_table(os.path.join(out, "quirk.bam.metrics.tsv"))
        start = [i for i, row in enumerate(ours) if row[0] == "Total Alignments"][0]
        assert ours[start:start + len(ref_rows)] == ref_rows
    assert m["Read Length"] == "110" and m["Genes Detected"] == "1" and m["Estimated Library Complexity"] != ""
    assert m["Median of Avg Transcript Coverage"] == report_ref.fmt(oracle_lib.median(sorted(r.gene_cov_mean[r.gene_cov_valid.astype(bool)])))
    # GCTs
    g = read_table(os.path.join(out, "quirk.bam.gene_reads.gct"))
    assert g[0] == ["#1.2"] and g[1] == ["5", "1"] and g[2] == ["Name", "Description", "Counts"]
    assert [row[2] for row in g[3:]] == ["6", "1", "1", "1", "3"] and g[3][:2] == ["GA", "AAA"]
    e = read_table(os.path.join(out, "quirk.bam.exon_reads.gct"))
    assert e[1] == ["7", "1"] and [row[2] for row in e[3:]] == ["2.500000", "0.500000", "3.000000", "1.000000", "0.000000", "1.000000", "1.000000", "3.000000"]
    t = read_table(os.path.join(out, "quirk.bam.gene_tpm.gct"))
    tp = np.array([6 / 1503, 1 / 1302, 1 / 501, 1 / 601, 3 / 2001]) * 1000
    np.testing.assert_allclose([float(row[2]) for row in t[3:]], tp / (tp.sum() / 1e6), rtol=1e-6)
    f = read_table(os.path.join(out, "quirk.bam.gene_fragments.gct"))
    assert [row[2] for row in f[3:]] == ["5", "1", "1", "1", "3"]
    cov = read_table(os.path.join(out, "quirk.bam.coverage.tsv"))
    assert cov[0] == ["gene_id", "coverage_mean", "coverage_std", "coverage_CV"]
    assert [row[0] for row in cov[1:]] == ["GA", "GB", "GR", "GH", "GC"]          # exit order: chr1 by start, then chr2
    assert cov[3][1:] == ["0", "0", "nan"] and cov[1][1] == report_ref.fmt(196 / 503)
    cv = read_table(os.path.join(out, "quirk.bam.exon_cv.tsv"))
    assert cv[0] == ["Exon ID", "Exon CV"] and [row[0] for row in cv[1:]] == ["GA_3", "GC_1"]
    host.host_annotation_free(h)


def test_report_writer_reproduces_single_pair_golden(host, oracle_lib, tmp_path):
    """Every line of the reference's golden single_pair metrics.tsv that the current code still prints must
    be reproduced verbatim from the reconstructed input (tests/cases.py)."""
    import json
    ka = json.load(open(os.path.join(ROOT, "tests", "golden", "reference_known_answers.json")))["single_pair"]["metrics"]
    ann, batch = cases.single_pair_case()
    gtf = str(tmp_path / "s.gtf")
    bamio.write_gtf(gtf, ann)
    h, err = load_annotation(host, gtf, ["1"])
    r = oracle_lib.run_oracle(abi.default_params(), ann, [batch])
    rs, keep = _results_struct(r)
    out = str(tmp_path / "out"); os.makedirs(out)
    visit = (C.c_int * 1)(0)
    assert host.host_write_reports(h, C.byref(rs), out.encode(), b"single_pair.bam", 0, 0, 0, 5, None, 0, visit, 1) == 0
    ours = dict(read_table(os.path.join(out, "single_pair.bam.metrics.tsv")))
    stale = {"Duplicate Reads"}                      # printed by the version that made the golden, not by src/
    same = 0
    for k, v in ka.items():
        if k in stale:
            continue
        assert k in ours, k
        got = ours[k].replace("-nan", "nan")
        assert got == v, (k, got, v)
        same += 1
    assert same >= 75
    g = read_table(os.path.join(out, "single_pair.bam.gene_tpm.gct"))
    assert g[3] == ["ENSG00000227232.4", "WASH7P", "1000000.000000"]
    e = read_table(os.path.join(out, "single_pair.bam.exon_reads.gct"))
    assert e[1] == ["1", "1"] and len(e) == 16
    host.host_annotation_free(h)


def test_report_writer_gc_outputs(host, oracle_lib, tmp_path):
    """--fasta outputs of the report tail (src/RNASeQC.cpp:628-674): gc_content.tsv, the GC column of exon_cv.tsv and the
    four moment lines, the latter against the reference's own getAdvancedStatistics when oracle/_ref is built."""
    from tests import test_fasta_gc as tg
    ann, batch, ref = tg.gc_case()
    r = oracle_lib.run_oracle(abi.default_params(coverage_mask=0), ann, [batch], reference=ref)
    r.gc_bins[40] += 7; r.gc_bins[41] += 2; r.gc_bins[77] += 5              # a less degenerate histogram
    gtf = str(tmp_path / "q.gtf")
    bamio.write_gtf(gtf, ann)
    h, err = load_annotation(host, gtf, ["chr1", "chr2"])
    assert err == 0
    rs, keep = _results_struct(r)
    out = str(tmp_path / "out"); os.makedirs(out)
    visit = (C.c_int * 2)(0, 1)
    assert host.host_write_reports(h, C.byref(rs), out.encode(), b"g.bam", 0, 0, 1, 5, None, 0, visit, 2) == 0
    gc = read_table(os.path.join(out, "g.bam.gc_content.tsv"))
    assert gc[0] == ["Content Bin", "Count"] and len(gc) == 101
    assert gc[1] == ["0", str(int(r.gc_bins[0]))] and gc[34] == ["0.33", str(int(r.gc_bins[33]))] and gc[100][0] == "0.99"
    cv = read_table(os.path.join(out, "g.bam.exon_cv.tsv"))
    assert cv[0] == ["Exon ID", "Exon CV", "GC Content"]
    rows = {row[0]: row for row in cv[1:]}
    assert rows["GA_1"][2] == "-1" and rows["GC_1"][2] == report_ref.fmt(tg.ref_gc(100, 2001))
    m = dict(read_table(os.path.join(out, "g.bam.metrics.tsv")))
    values = np.repeat(np.arange(100), r.gc_bins.astype(np.int64))
    if oracle_lib.ref_lib() is not None and hasattr(oracle_lib.ref_lib(), "ref_advanced_statistics"):
        avg, skew, sd, kurt = oracle_lib.ref_advanced_statistics(values)
    else:                                                                  # moments by definition (looser)
        avg, sd = values.mean(), values.std()
        skew = ((values - avg) ** 3).mean() / sd ** 3; kurt = ((values - avg) ** 4).mean() / sd ** 4 - 3
    for key, val in (("Fragment GC Content Mean", avg / 100.0), ("Fragment GC Content Std", sd / 100.0),
                     ("Fragment GC Content Skewness", skew), ("Fragment GC Content Kurtosis", kurt)):
        assert abs(float(m[key]) - val) <= 1e-5 * max(1.0, abs(val)), key
    assert list(m)[-4:] == ["Fragment GC Content Mean", "Fragment GC Content Std", "Fragment GC Content Skewness", "Fragment GC Content Kurtosis"]
    host.host_annotation_free(h)


def test_bam_decode_inflate_backends_agree(host, tmp_path):
    """BGZF inflate goes through libdeflate when the shared library is present and through zlib otherwise
    (RSQC_HOST_ZLIB=1 forces zlib): same records either way, and a damaged block is an error in both."""
    import sys
    ann = synth.make_annotation(seed=37, contigs=[("chrA", 2_000_000, 100)])
    batch = synth.make_reads(ann, 20_000, seed=38, keep_qnames=True, contig_lengths=np.array([2_000_000]))
    path = str(tmp_path / "b.bam")
    bamio.write_bam(path, [("chrA", 2_000_000)], batch)
    bad = str(tmp_path / "bad.bam")
    raw = bytearray(open(path, "rb").read())
    raw[len(raw) // 2] ^= 0x5A; raw[len(raw) // 2 + 1] ^= 0xA5          # inside some block's deflate stream
    open(bad, "wb").write(bytes(raw))
    code = (
        "import ctypes as C, sys, numpy as np\n"
        "sys.path.insert(0, %r)\n"
        "from rnaseqc_amd import abi\n"
        "lib = C.CDLL(%r)\n"
        "lib.host_bam_read_all_ex.restype = C.c_void_p\n"
        "lib.host_bam_read_all_ex.argtypes = [C.c_char_p, C.c_char_p, C.POINTER(C.c_char_p), C.c_int, C.c_int, C.c_ulonglong]\n"
        "lib.host_bam_batch.restype = C.POINTER(abi.BatchStruct); lib.host_bam_batch.argtypes = [C.c_void_p]\n"
        "h = lib.host_bam_read_all_ex(sys.argv[1].encode(), b'ch', None, 0, 4, 1 << 20)\n"
        "if not h: print('ERR'); sys.exit(0)\n"
        "b = lib.host_bam_batch(h).contents\n"
        "core = abi._view(b.core, b.n, abi.REC_CORE); aux = abi._view(b.aux, b.n, abi.REC_AUX)\n"
        "print(b.n, int(core['pos'].astype(np.int64).sum()), int(aux['qhash'].sum() & np.uint64(0xFFFFFFFF)), b.n_cigar_total)\n"
    ) % (ROOT, SO)
    outs = {}
    for name, env in (("libdeflate", {}), ("zlib", {"RSQC_HOST_ZLIB": "1"})):
        for f in (path, bad):
            p = subprocess.run([sys.executable, "-c", code, f], stdout=subprocess.PIPE, env=dict(os.environ, **env))
            outs[(name, f)] = p.stdout.decode().strip()
    assert outs[("libdeflate", path)] == outs[("zlib", path)] and outs[("zlib", path)].split()[0] == str(batch.n)
    assert int(outs[("zlib", path)].split()[1]) == int(batch.pos.astype(np.int64).sum())
    assert outs[("libdeflate", bad)] == "ERR" and outs[("zlib", bad)] == "ERR"


def test_bam_decode_walks_every_aux_type(host, tmp_path):
    """Records carry random extra aux fields of every BAM type (A c C s S i I f d Z H and B arrays of each element type)
    around NM / the chimeric tag / the filter tag, including look-alike tag names: the decoded tag bits and NM must not
    change (SeqLib GetIntTag / GetZTag / GetTag semantics at src/RNASeQC.cpp:295,320-327,780-800)."""
    import struct
    rng = np.random.default_rng(5)
    ann = synth.make_annotation(seed=39, contigs=[("chrA", 1_000_000, 60)])
    batch = synth.make_reads(ann, 4000, seed=40, keep_qnames=True, chimeric_tag_frac=0.05, filter_tag_frac=0.05,
                             contig_lengths=np.array([1_000_000]))

    def field():
        name = bytes(rng.choice(list(b"ABXYZabmn"), 2).tolist())
        if name in (b"NM", b"ch", b"XF"):
            name = b"Zz"
        t = rng.choice(list("AcCsSiIfdZHB"))
        if t == "A": v = bytes([int(rng.integers(33, 126))])
        elif t in "cC": v = bytes([int(rng.integers(0, 256))])
        elif t in "sS": v = struct.pack("<H", int(rng.integers(0, 65536)))
        elif t in "iI": v = struct.pack("<I", int(rng.integers(0, 2 ** 32)))
        elif t == "f": v = struct.pack("<f", float(rng.random()))
        elif t == "d": v = struct.pack("<d", float(rng.random()))               # 8 bytes: htslib skips it, so do we
        elif t == "Z": v = bytes(rng.integers(33, 126, int(rng.integers(0, 40))).tolist()) + b"\x00"
        elif t == "H": v = b"".join(b"%02X" % int(x) for x in rng.integers(0, 256, int(rng.integers(0, 8)))) + b"\x00"
        else:
            st = rng.choice(list("cCsSiIf")); cnt = int(rng.integers(0, 20))
            es = {"c": 1, "C": 1, "s": 2, "S": 2}.get(st, 4)
            v = st.encode() + struct.pack("<I", cnt) + bytes(rng.integers(0, 256, es * cnt).tolist())
        return name + t.encode() + v

    def extra(i):
        return b"".join(field() for _ in range(int(rng.integers(0, 4)))), b"".join(field() for _ in range(int(rng.integers(0, 4))))
    path = str(tmp_path / "x.bam")
    bamio.write_bam(path, [("chrA", 1_000_000)], batch, extra_aux=extra)
    tags = (C.c_char_p * 1)(b"XF")
    host.host_bam_read_all_ex.restype = C.c_void_p
    h = host.host_bam_read_all_ex(path.encode(), b"ch", tags, 1, 4, C.c_ulonglong(1 << 20))
    assert h
    b = host.host_bam_batch(C.c_void_p(h)).contents
    assert b.n == batch.n
    aux = _arr(b.aux, b.n, abi.REC_AUX); core = _arr(b.core, b.n, abi.REC_CORE)
    for f in ("qhash", "flag", "l_qseq", "mapq", "nm", "tagbits", "n_cigar"):
        np.testing.assert_array_equal(aux[f], getattr(batch, f), err_msg=f)
    np.testing.assert_array_equal(core["pos"], batch.pos)
    np.testing.assert_array_equal(_arr(b.cigar, b.n_cigar_total, np.uint32), batch.cigar)
    host.host_bam_free(C.c_void_p(h))


def test_bam_decode_long_cigar_from_cg_tag(host, tmp_path):
    """A CIGAR of more than 65535 operations is stored in the CG:B,I tag behind the placeholder <l_seq>S<ref_len>N
    (SAM spec 4.2.2); htslib restores it on read (bam_tag2cigar), so the reference's extractBlocks sees the real
    operations -- and so must the boundary's batch."""
    import struct, zlib
    n_ops = 70001                                                   # 35001 x 1M interleaved with 35000 x 1D
    ops = [(1 << 4) | (0 if k % 2 == 0 else 2) for k in range(n_ops)]
    l_seq = (n_ops + 1) // 2
    ref_len = n_ops
    name = b"long1\x00"
    cg = b"CGBI" + struct.pack("<I", n_ops) + struct.pack("<%dI" % n_ops, *ops)
    tags = b"NMC" + bytes([3]) + cg + b"XYd" + struct.pack("<d", 1.5) + b"chZx\x00"      # a 'd' field before the chimeric tag
    placeholder = struct.pack("<II", (l_seq << 4) | 4, (ref_len << 4) | 3)
    rec = struct.pack("<iiBBHHHiiii", 0, 100, len(name), 255, 4680, 2, 0, l_seq, -1, -1, 0) + name + placeholder + \
        b"\x11" * ((l_seq + 1) // 2) + b"\xff" * l_seq + tags
    plain = struct.pack("<iiBBHHHiiii", 0, 200, len(name), 255, 4680, 1, 0, 50, -1, -1, 0) + name + struct.pack("<I", (50 << 4)) + \
        b"\x11" * 25 + b"\xff" * 50
    text = b"@HD\tVN:1.6\tSO:coordinate\n@SQ\tSN:chrA\tLN:1000000\n"
    raw = b"BAM\x01" + struct.pack("<I", len(text)) + text + struct.pack("<I", 1) + struct.pack("<I", 5) + b"chrA\x00" + struct.pack("<I", 1000000)
    raw += struct.pack("<I", len(rec)) + rec + struct.pack("<I", len(plain)) + plain
    path = str(tmp_path / "cg.bam")
    with open(path, "wb") as f:
        for o in range(0, len(raw), 60000):
            f.write(bamio._bgzf_block(raw[o:o + 60000]))
        f.write(bamio._EOF)
    h = host.host_bam_read_all(path.encode(), b"ch", None, 0)
    assert h
    b = host.host_bam_batch(h).contents
    assert b.n == 2 and b.n_wide == 1
    aux = _arr(b.aux, 2, abi.REC_AUX); core = _arr(b.core, 2, abi.REC_CORE)
    assert aux["n_cigar"][0] == abi.NCIGAR_ESCAPE and _arr(b.wide_n_cigar, 1, np.uint32)[0] == n_ops
    cig = _arr(b.cigar, b.n_cigar_total, np.uint32)
    assert b.n_cigar_total == n_ops + 1 and list(cig[:n_ops]) == ops and cig[n_ops] == (50 << 4)
    assert core["cigar_off"][1] == n_ops
    assert aux["tagbits"][0] & abi.TB_HAS_NM and aux["nm"][0] == 3 and aux["tagbits"][0] & abi.TB_HAS_CH   # found behind CG and the 'd' field
    host.host_bam_free(h)


def test_bam_decode_rejects_oversized_isize(host, tmp_path):
    """A BGZF trailer that claims more than 64 KiB is refused before any buffer is sized from it."""
    import struct
    ann, batch = cases.quirk_case()
    path = str(tmp_path / "q.bam")
    bamio.write_bam(path, [("chr1", 100000), ("chr2", 100000)], batch)
    d = bytearray(open(path, "rb").read())
    bsize = struct.unpack_from("<H", d, 16)[0] + 1
    struct.pack_into("<I", d, bsize - 4, 0x7FFFFFF0)                 # ISIZE of the first block
    bad = str(tmp_path / "bad.bam")
    open(bad, "wb").write(bytes(d))
    assert not host.host_bam_read_all(bad.encode(), b"ch", None, 0)
    h = host.host_bam_read_all(path.encode(), b"ch", None, 0)
    assert h
    host.host_bam_free(h)


def test_bam_decode_reports_unsorted_input_and_unknown_refids(host, tmp_path):
    """What the reference prints on stderr from its loop: positions going backwards inside a contig
    (src/RNASeQC.cpp:354-355) and records whose RefID the header does not define (:333-337)."""
    ann = synth.make_annotation(seed=39, contigs=[("chrA", 1_000_000, 60), ("chrB", 500_000, 20)])
    batch = synth.make_reads(ann, 3000, seed=41, keep_qnames=True, contig_lengths=np.array([1_000_000, 500_000]))
    contigs = [("chrA", 1_000_000), ("chrB", 500_000)]
    host.host_bam_unsorted.argtypes = [C.c_void_p]; host.host_bam_bad_refid_count.argtypes = [C.c_void_p]
    host.host_bam_bad_refid.argtypes = [C.c_void_p, C.c_int]; host.host_bam_bad_refid.restype = C.c_char_p
    host.host_bam_read_all_ex.restype = C.c_void_p

    def decode(b, contigs_):
        path = str(tmp_path / "t.bam")
        bamio.write_bam(path, contigs_, b)
        out = []
        for threads, per in ((1, 1 << 20), (3, 700)):
            h = host.host_bam_read_all_ex(path.encode(), b"ch", None, 0, threads, C.c_ulonglong(per))
            assert h
            out.append((host.host_bam_unsorted(C.c_void_p(h)), [host.host_bam_bad_refid(C.c_void_p(h), i).decode()
                                                                  for i in range(host.host_bam_bad_refid_count(C.c_void_p(h)))]))
            host.host_bam_free(C.c_void_p(h))
        assert out[0] == out[1]
        return out[0]

    assert decode(batch, contigs) == (0, [])
    # two mapped primary records of one contig swapped
    tid = batch.tid_per_record()
    ok = np.flatnonzero((tid == 0) & ((batch.flag & 0x904) == 0))
    i, j = int(ok[10]), int(ok[400])
    assert batch.pos[i] < batch.pos[j]
    sw = batch.slice(0, batch.n)
    sw.pos[i], sw.pos[j] = batch.pos[j], batch.pos[i]
    assert decode(sw, contigs)[0] == 1
    # a header that names only the first contig: the records of the second one carry an undefined RefID
    u, names = decode(batch, contigs[:1])
    assert u == 0 and len(names) > 0 and all(n.startswith("SYN:") for n in names)


def test_bam_by_contig_reading_through_the_index(host, tmp_path):
    """BamReader::load_index + seek: the records of one reference sequence, located by the .bai's per-reference
    virtual offsets (the counterpart of an index-driven region reader; what a contig-sharded run reads)."""
    contigs = [("cA", 2_000_000, 150), ("cB", 1_500_000, 120), ("cC", 900_000, 0), ("cD", 700_000, 50)]
    ann = synth.make_annotation(seed=21, contigs=contigs)
    batch = synth.make_reads(ann, 20000, seed=22, contig_lengths=np.array([c[1] for c in contigs]))
    path = str(tmp_path / "i.bam")
    voff = bamio.write_bam_fast(path, [(c[0], c[1]) for c in contigs], batch, threads=3, bai=True)
    assert len(voff) == len(batch.seg_tid) + 1 and (np.diff(voff.astype(np.int64)) > 0).all()
    host.host_bam_read_contig.restype = C.c_void_p
    tid = batch.tid_per_record()
    for c in range(len(contigs)):
        for threads in (1, 3):
            h = host.host_bam_read_contig(path.encode(), b"ch", None, 0, threads, c)
            assert h
            b = host.host_bam_batch(C.c_void_p(h)).contents
            m = tid == c
            assert b.n == int(m.sum())
            core = _arr(b.core, b.n, abi.REC_CORE); aux = _arr(b.aux, b.n, abi.REC_AUX)
            np.testing.assert_array_equal(core["pos"], batch.pos[m]); np.testing.assert_array_equal(aux["flag"], batch.flag[m])
            assert b.n_seg == 1 and _arr(b.seg_tid, 1, np.int32)[0] == c
            host.host_bam_free(C.c_void_p(h))


def test_host_reader_on_damaged_records_under_sanitizers(tmp_path):
    """The host BAM reader (parallel framing from guessed record starts + chain verification, host/bam.cpp) on files whose
    records are damaged behind valid BGZF blocks: an exception or a record count, never an access outside its buffers or a
    loop that does not end; the same for damaged containers (BGZF headers, trailers, cut files), and for the device decode's
    feeder (host/bgzf_feed.cpp) on every file.  tests/hostemu/reader_fuzz.cpp under the address / undefined-behaviour sanitizers."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "reader_fuzz")
    libdir = os.path.join(root, "rnaseqc_amd", "lib")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=all",
                           os.path.join(root, "tests", "hostemu", "reader_fuzz.cpp"), os.path.join(root, "rnaseqc_amd", "csrc", "host", "bam.cpp"),
                           os.path.join(root, "rnaseqc_amd", "csrc", "host", "bgzf_feed.cpp"),
                           "-o", exe, "-L" + libdir, "-lrnaseqc_amd", "-lz", "-ldl", "-lpthread", "-Wl,-rpath," + libdir])
    r = subprocess.run([exe, "500", "11", str(tmp_path / "f.bam")], capture_output=True, text=True, timeout=900,
                       env=dict(os.environ, ASAN_OPTIONS="detect_leaks=0"))
    assert r.returncode == 0 and "500 cases" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]


def test_annotation_ingest_on_damaged_text_under_sanitizers(tmp_path):
    """GTF / BED ingest (host/gtf.cpp) on damaged lines (cut, fields missing or empty, coordinates that are no numbers or beyond
    64 bits, unbalanced quotes, CRLF, garbage): the reference's error classes (exit codes 10 / 11) or a loaded annotation,
    nothing else.  tests/hostemu/gtf_fuzz.cpp under the address / undefined-behaviour sanitizers."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "gtf_fuzz")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=all",
                           os.path.join(root, "tests", "hostemu", "gtf_fuzz.cpp"), os.path.join(root, "rnaseqc_amd", "csrc", "host", "gtf.cpp"), "-o", exe])
    r = subprocess.run([exe, "1500", "5", str(tmp_path)], capture_output=True, text=True, timeout=900, env=dict(os.environ, ASAN_OPTIONS="detect_leaks=0"))
    assert r.returncode == 0 and "1500 cases" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]
