"""Minimal BAM / GTF / BED writers for the synthetic inputs (tests and the CLI benchmark).
Only what the CLI's decoder reads is written faithfully: header, core fields, CIGAR, NM / chimeric /
filter tags.  SEQ is all 'A', QUAL 0xff."""
from __future__ import annotations

import struct
import zlib

import numpy as np

from . import abi


def _bgzf_block(data: bytes, level: int = 1) -> bytes:
    comp = zlib.compressobj(level, zlib.DEFLATED, -15)
    c = comp.compress(data) + comp.flush()
    bsize = len(c) + 25
    return (b"\x1f\x8b\x08\x04\x00\x00\x00\x00\x00\xff\x06\x00BC\x02\x00" + struct.pack("<H", bsize) + c +
            struct.pack("<II", zlib.crc32(data) & 0xFFFFFFFF, len(data)))


_EOF = bytes.fromhex("1f8b08040000000000ff0600424302001b0003000000000000000000")


def write_bam(path, contigs, batch, qnames=None, filter_tag="XF", ch_tag="ch", level=1, extra_aux=None):
    """contigs: [(name, length)]; batch: model.Batch (file order).  qnames: list of bytes or None (uses batch.qname).
    extra_aux: optional callable i -> (bytes before, bytes after) of further aux fields around the standard ones."""
    n = batch.n
    tid = batch.tid_per_record()
    hdr_text = ("@HD\tVN:1.6\tSO:coordinate\n" + "".join("@SQ\tSN:%s\tLN:%d\n" % (nm, ln) for nm, ln in contigs)).encode()
    out = bytearray()
    out += b"BAM\x01" + struct.pack("<I", len(hdr_text)) + hdr_text + struct.pack("<I", len(contigs))
    for nm, ln in contigs:
        out += struct.pack("<I", len(nm) + 1) + nm.encode() + b"\x00" + struct.pack("<I", ln)
    wide = {int(i): k for k, i in enumerate(batch.wide_index)}
    cend = np.append(batch.cigar_off, len(batch.cigar)).astype(np.int64)
    chunks = [bytes(out)]
    cur = bytearray()
    for i in range(n):
        if qnames is not None:
            name = qnames[i]
        else:
            name = bytes(batch.qname[int(batch.qname_off[i]):int(batch.qname_off[i + 1])])
        lq, nm, nc = int(batch.l_qseq[i]), int(batch.nm[i]), int(batch.n_cigar[i])
        if i in wide:
            k = wide[i]
            lq, nm, nc = int(batch.wide_l_qseq[k]), int(batch.wide_nm[k]), int(batch.wide_n_cigar[k])
        cig = batch.cigar[int(batch.cigar_off[i]):int(batch.cigar_off[i]) + nc]
        tb = int(batch.tagbits[i])
        t = int(tid[i])
        mtid = t if (tb & abi.TB_MTID_SAME) else (t + 1 if t + 1 < len(contigs) else (0 if t != 0 else -1))
        tags = b""
        if tb & abi.TB_HAS_NM:
            tags += b"NM" + (b"C" + struct.pack("<B", nm) if 0 <= nm < 256 else b"i" + struct.pack("<i", nm))
        if tb & abi.TB_HAS_CH:
            tags += ch_tag.encode() + b"Z1\x00"
        if tb & abi.TB_FILTER0:
            tags += filter_tag.encode() + b"i" + struct.pack("<i", 1)
        if extra_aux is not None:
            before, after = extra_aux(i)
            tags = before + tags + after
        rec = struct.pack("<iiBBHHHiiii", t, int(batch.pos[i]), len(name) + 1, int(batch.mapq[i]), 4680, nc,
                          int(batch.flag[i]), lq, mtid, int(batch.mpos[i]), int(batch.isize[i]))
        rec += name + b"\x00" + cig.astype("<u4").tobytes() + b"\x11" * ((lq + 1) // 2) + b"\xff" * lq + tags
        cur += struct.pack("<I", len(rec)) + rec
        if len(cur) > 60000:
            chunks.append(bytes(cur[:60000])); cur = cur[60000:]
    if cur:
        chunks.append(bytes(cur))
    with open(path, "wb") as f:
        # the header may exceed one block
        for c in chunks:
            for o in range(0, len(c), 60000):
                f.write(_bgzf_block(c[o:o + 60000], level))
        f.write(_EOF)


def write_gtf(path, ann):
    with open(path, "w") as f:
        f.write("##synthetic collapsed annotation\n")
        for line in ann.to_gtf_lines():
            f.write(line + "\n")


def write_bed(path, ann, bed):
    with open(path, "w") as f:
        for c, s, e in zip(bed.contig.tolist(), bed.start.tolist(), bed.end.tolist()):
            f.write("%s\t%d\t%d\n" % (ann.contig_names[c], s - 1, e - 1))


def write_bam_fast(path, contigs, batch, ch_tag="ch", filter_tag="XF", threads=16, seq_mode=0, bai=False, struct=None):
    """Same file as write_bam through the C++ host library's writer (rnaseqc_amd/lib/librsqc_host.so,
    host_bam_write_ex): used for multi-million-record CLI benchmarks.  Records without names get 16 hex digits of
    their qhash as QNAME (mates keep sharing a name).  seq_mode 0: SEQ all 'A', QUAL 0xff;
    1: random bases and binned Phred-like qualities (the compressibility of a real file).  bai: also write
    <path>.bai with the per-contig virtual offsets.  Returns the BGZF virtual offsets of the batch's segments
    (+ the end of the records)."""
    import ctypes as C
    import os
    lib = C.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "lib", "librsqc_host.so"))
    names = (C.c_char_p * len(contigs))(*[c[0].encode() for c in contigs])
    lens = (C.c_uint * len(contigs))(*[int(c[1]) for c in contigs])
    st = struct if struct is not None else batch.to_struct()
    voff = np.zeros(int(st.n_seg) + 1, np.uint64)
    lib.host_bam_write_ex.argtypes = [C.c_char_p, C.POINTER(C.c_char_p), C.POINTER(C.c_uint), C.c_int, C.c_void_p, C.c_char_p,
                                      C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_void_p]
    rc = lib.host_bam_write_ex(str(path).encode(), names, lens, len(contigs), C.byref(st), ch_tag.encode(), filter_tag.encode(),
                               threads, seq_mode, 1 if bai else 0, voff.ctypes.data)
    if rc:
        raise OSError("host_bam_write_ex failed: %d" % rc)
    return voff


def write_fasta(path, names, reference, line_bases=60, index_path=None):
    """FASTA + .fai for the contigs of a model.Reference (test input for --fasta)."""
    idx = []
    with open(path, "wb") as f:
        for name, seq in zip(names, reference.sequence):
            f.write(b">" + name.encode() + b"\n")
            off = f.tell()
            b = bytes(bytearray(seq))
            for i in range(0, len(b), line_bases):
                f.write(b[i:i + line_bases] + b"\n")
            idx.append((name, len(b), off, line_bases, line_bases + 1))
    with open(index_path or path + ".fai", "w") as f:
        for rec in idx:
            f.write("%s\t%d\t%d\t%d\t%d\n" % rec)


# ---- the host side of the device decode (rsqc_decode_*): file chunks and BGZF block tables from host/bgzf_feed.cpp ----
import ctypes as C  # noqa: E402
import os  # noqa: E402

def _feed_lib():
    l = C.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "lib", "librsqc_host.so"))
    l.host_feed_open.restype = C.c_void_p; l.host_feed_open.argtypes = [C.c_char_p]
    l.host_feed_first_voffset.restype = C.c_ulonglong; l.host_feed_first_voffset.argtypes = [C.c_void_p]
    l.host_feed_start.argtypes = [C.c_void_p, C.c_ulonglong, C.c_ulonglong, C.c_ulonglong, C.c_ulonglong, C.c_int]
    l.host_feed_next.argtypes = [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_ulonglong), C.POINTER(C.c_void_p), C.POINTER(C.c_uint32),
                                 C.POINTER(C.c_uint32), C.POINTER(C.c_ulonglong), C.POINTER(C.c_int)]
    l.host_feed_error.restype = C.c_char_p; l.host_feed_error.argtypes = [C.c_void_p]
    l.host_feed_free.argtypes = [C.c_void_p]
    return l


BGZF_BLOCK = np.dtype([("in_offset", "<u8"), ("in_bytes", "<u4"), ("out_bytes", "<u4"), ("crc32", "<u4"), ("flags", "<u4")])


def feed_chunks(path, voff_beg=None, voff_end=0, chunk_bytes=1 << 17, max_out=1 << 40, threads=2, cpu_share=None, reserve=None):
    """[(compressed bytes, block table, skip, limit, last)] of a range, through BgzfFeeder.
    reserve = (chunk_bytes, max_out): the buffers are set up before the range starts, as the command line does."""
    l = _feed_lib()
    h = l.host_feed_open(str(path).encode())
    assert h
    if voff_beg is None:
        voff_beg = l.host_feed_first_voffset(h)
        assert voff_beg != 2 ** 64 - 1, l.host_feed_error(h)
    if cpu_share:                                  # (threads, initial share, largest share): blocks the CPU inflates arrive flagged BGZF_INFLATED
        l.host_feed_cpu_share.argtypes = [C.c_void_p, C.c_int, C.c_double, C.c_double]
        l.host_feed_cpu_share(h, *cpu_share)
    if reserve:
        l.host_feed_reserve.argtypes = [C.c_void_p, C.c_ulonglong, C.c_ulonglong]
        assert l.host_feed_reserve(h, int(reserve[0]), int(reserve[1])) == 0, l.host_feed_error(h)
    assert l.host_feed_start(h, voff_beg, voff_end, chunk_bytes, max_out, threads) == 0
    out = []
    while True:
        data, nbytes, blocks, nb, skip, limit, last = C.c_void_p(), C.c_ulonglong(), C.c_void_p(), C.c_uint32(), C.c_uint32(), C.c_ulonglong(), C.c_int()
        rc = l.host_feed_next(h, C.byref(data), C.byref(nbytes), C.byref(blocks), C.byref(nb), C.byref(skip), C.byref(limit), C.byref(last))
        if rc < 0:
            err = l.host_feed_error(h).decode(); l.host_feed_free(h)
            raise RuntimeError(err)
        if rc == 0:
            break
        comp = C.string_at(data.value, nbytes.value)
        tab = np.frombuffer(C.string_at(blocks.value, nb.value * BGZF_BLOCK.itemsize), BGZF_BLOCK).copy() if nb.value else np.zeros(0, BGZF_BLOCK)
        out.append((comp, tab, skip.value, limit.value, bool(last.value)))
    l.host_feed_free(h)
    return out
