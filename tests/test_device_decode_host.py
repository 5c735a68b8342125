"""CPU tests of the device-decode cores, run as a wave of one lane: the DEFLATE decoder against zlib,
the CRC arithmetic of the 64-lane build, and BAM framing / parsing in windows against the records that were written.
The kernels that wrap these bodies are covered by the -m gpu tests (tests/test_gpu_decode.py)."""
import os
import random
import zlib

import numpy as np
import pytest

from rnaseqc_amd import abi, bamio, synth
from rnaseqc_amd.model import Batch
from tests.hostemu import decode as emu


def _payloads():
    rng = random.Random(5)
    yield b""
    yield b"a"
    yield b"A" * 65280
    yield bytes(rng.getrandbits(8) for _ in range(65280))
    yield bytes(rng.choice(b"ACGT") for _ in range(65280))
    yield (b"hello world, " * 6000)[:65280]
    yield np.random.default_rng(4).normal(30, 5, 60000).astype(np.int8).tobytes()
    for _ in range(12):
        n = rng.randrange(1, 65281)
        kind = rng.randrange(4)
        if kind == 0:
            yield bytes(rng.getrandbits(8) for _ in range(n))
        elif kind == 1:
            yield bytes(rng.choice(b"ACGTN") for _ in range(n))
        elif kind == 2:
            pieces = [bytes(rng.getrandbits(8) for _ in range(rng.randrange(1, 300))) for _ in range(20)]
            yield b"".join(rng.choice(pieces) for _ in range(n // 100 + 1))[:n]
        else:
            yield bytes([rng.randrange(2)]) * n


@pytest.mark.parametrize("variant", ["", "no_par_commit", "no_inwalk", "all", "vwalk1", "vwalk2", "vwalk3", "vwalk3_no_inwalk"])
def test_inflate_matches_zlib_on_every_block_type(variant):
    """Stored, fixed and dynamic blocks, several blocks per stream, small windows, long codes (Huffman-only on random bytes),
    runs (distance 1) and distances up to 32 KiB.  variant: "" is the product's configuration; the others switch off the one-pass
    commit of a round (INF_PAR_COMMIT_CFG) and / or the in-walk decode of long codes (INF_INWALK_CFG) of rsqc_inflate.h, or set the
    number of symbols the walk takes per step (INF_VWALK_CFG: 1 = the product, 2, 4, 8)."""
    n = 0
    for d in _payloads():
        for level in (0, 1, 6, 9):
            for strat in (zlib.Z_DEFAULT_STRATEGY, zlib.Z_FIXED, zlib.Z_HUFFMAN_ONLY, zlib.Z_RLE):
                for wbits in (-15, -9):
                    co = zlib.compressobj(level, zlib.DEFLATED, wbits, 9, strat)
                    half = len(d) // 2
                    comp = co.compress(d[:half]) + co.flush(zlib.Z_FULL_FLUSH) + co.compress(d[half:]) + co.flush()
                    rc, out = emu.inflate(comp, len(d), zlib.crc32(d), variant)
                    assert rc == 0 and out == d, (len(d), level, strat, wbits, rc)
                    n += 1
    assert n > 500


def test_inflate_reports_corruption():
    rng = random.Random(9)
    d = bytes(rng.choice(b"ACGT") for _ in range(30000))
    comp = zlib.compress(d, 6)[2:-4]
    for _ in range(200):                                  # any flipped bit: a decode error or, at the latest, the CRC
        c = bytearray(comp); c[rng.randrange(len(c))] ^= 1 << rng.randrange(8)
        rc, _o = emu.inflate(bytes(c), len(d), zlib.crc32(d))
        assert rc != 0
    assert emu.inflate(comp, len(d) - 1, zlib.crc32(d))[0] != 0          # ISIZE too small / too large / payload cut short
    assert emu.inflate(comp, len(d) + 1, zlib.crc32(d))[0] != 0
    assert emu.inflate(comp[:len(comp) // 2], len(d), zlib.crc32(d))[0] != 0
    assert emu.inflate(comp, len(d), zlib.crc32(d) ^ 1)[0] == 8           # INF_ERR_CRC


def test_crc_of_the_64_lane_build():
    """Pieces per lane, right alignment, the six-level combine and the advance of the running register -- with the lanes
    as an array, for chunk lengths around every boundary."""
    l = emu.lib()
    rng = random.Random(1)
    for n in [0, 1, 2, 63, 64, 65, 127, 1000, 5000, 16383, 16384, 16384 + 257]:
        a = bytes(rng.getrandbits(8) for _ in range(777)); b = bytes(rng.getrandbits(8) for _ in range(n))
        got = l.emu_crc_wave64(b, n, zlib.crc32(a) ^ 0xFFFFFFFF) ^ 0xFFFFFFFF
        assert got == zlib.crc32(a + b), n


def _check(dec, batch):
    assert dec.status == 0 and dec.n == batch.n
    for f in ("pos", "mpos", "isize", "cigar_off"):
        np.testing.assert_array_equal(dec.core[f], getattr(batch, f), err_msg=f)
    for f in ("qhash", "flag", "l_qseq", "mapq", "nm", "tagbits", "n_cigar"):
        np.testing.assert_array_equal(dec.aux[f], getattr(batch, f), err_msg=f)
    if batch.qhash2 is not None:
        np.testing.assert_array_equal(dec.qhash2, batch.qhash2, err_msg="qhash2")
    np.testing.assert_array_equal(dec.cigar, batch.cigar)
    np.testing.assert_array_equal(dec.seg_tid, batch.seg_tid)
    np.testing.assert_array_equal(dec.seg_start, batch.seg_start)
    np.testing.assert_array_equal(np.array([w[0] for w in dec.wide], np.uint64), batch.wide_index)


@pytest.mark.parametrize("window,threads,perturb", [(1 << 30, 7, 0), (100_000, 3, 0), (20_000, 1, 0), (333_333, 1024, 0), (150_000, 5, 1),
                                                    (1 << 30, 4, 2), (250_000, 4, 2), (1 << 30, 4, 3), (1 << 30, 2, 4), (90_000, 2, 4), (1 << 30, 3, 5)])
def test_window_decode_matches_the_written_records(tmp_path, window, threads, perturb):
    """BGZF blocks through the emulated inflate, then frame / chain / offsets / parse / lists in windows: records that
    span windows, windows smaller than a segment's worth of records, and guesses made wrong on purpose that the chain step has
    to repair (moved, random, a run of consecutive ones, none at all, walks flagged as garbage): the listed repair must leave
    the segments the plain sequential walk leaves (checked inside the emulation for every window)."""
    contigs = [("chrA", 3_000_000), ("chrB", 1_000_000), ("chrC", 500_000)]
    ann = synth.make_annotation(seed=35, contigs=[("chrA", 3_000_000, 120), ("chrB", 1_000_000, 40), ("chrC", 500_000, 10)])
    batch = synth.make_reads(ann, 12_000, seed=36, keep_qnames=True, chimeric_tag_frac=0.02, filter_tag_frac=0.03,
                             contig_lengths=np.array([3_000_000, 1_000_000, 500_000]))
    path = str(tmp_path / "p.bam")
    bamio.write_bam(path, contigs, batch)
    stream, first, n_ref = emu.inflate_bam(path)
    assert n_ref == 3
    dec = emu.decode_stream(stream, first, n_ref, window, "ch", ("XF",), threads=threads, perturb=perturb)
    _check(dec, batch)
    assert not dec.unsorted and not dec.bad_names


def test_guesses_near_the_window_end_are_checked(tmp_path):
    """A candidate record start whose record reaches past the window's end cannot be checked against the record behind it.  Taken
    at once (round 3), the two bytes in front of a true start -- read as a block_size of megabytes -- were the guess of every
    segment of a window's last megabytes on RefID 0: ~900 wrong guesses on this file, each repaired by one thread on the device.
    Now such a candidate is a last resort: no guess of this file is wrong, in either SEQ flavour."""
    ann = synth.make_annotation(seed=3, contigs=[("chrA", 3_000_000, 300), ("chrB", 1_500_000, 150)])
    batch = synth.make_reads(ann, 60_000, seed=4, contig_lengths=np.array([3_000_000, 1_500_000]))
    l = emu.lib()
    l.emu_decode_unconfirmed.restype = C.c_ulonglong
    for mode in (0, 1):
        path = str(tmp_path / ("g%d.bam" % mode))
        bamio.write_bam_fast(path, [("chrA", 3_000_000), ("chrB", 1_500_000)], batch, threads=3, seq_mode=mode)
        stream, first, n_ref = emu.inflate_bam(path, use_emu=False)
        l.emu_decode_unconfirmed(1)
        dec = emu.decode_stream(stream, first, n_ref, 8_000_000, threads=5)
        assert dec.n == batch.n and dec.windows >= 4
        np.testing.assert_array_equal(dec.core["pos"], batch.pos)
        assert l.emu_decode_unconfirmed(1) == 0


def test_window_decode_long_record_and_wide_fields(tmp_path):
    """A 4.5 MB record between short ones: segments inside it have no record start to guess, its SEQ / QUAL bytes must not
    be taken for records, and it spans several windows."""
    recs = []
    for i in range(3000):
        recs.append(dict(tid=0, pos=100 + i, mpos=100 + i, isize=0, flag=0, cigar=[(abi.CIG_M, 100)], qname="s%d" % i))
    recs.append(dict(tid=0, pos=5000, mpos=5000, isize=0, flag=0, cigar=[(abi.CIG_M, 3_000_000)], qname="long"))
    for i in range(3000):
        recs.append(dict(tid=1 if i > 1500 else 0, pos=6000 + i, mpos=6000 + i, isize=0, flag=0, cigar=[(abi.CIG_M, 90), (abi.CIG_S, 10)], qname="t%d" % i))
    batch = Batch.from_records(recs)
    path = str(tmp_path / "l.bam")
    bamio.write_bam(path, [("chrA", 3_000_000), ("chrB", 1_000_000)], batch)
    stream, first, n_ref = emu.inflate_bam(path, use_emu=False)
    for window in (1 << 30, 1_000_000, 70_000):
        dec = emu.decode_stream(stream, first, n_ref, window, threads=11)
        assert dec.n == batch.n
        np.testing.assert_array_equal(dec.core["pos"], batch.pos)
        np.testing.assert_array_equal(dec.aux["qhash"], batch.qhash)
        np.testing.assert_array_equal(dec.aux["l_qseq"], batch.l_qseq)
        np.testing.assert_array_equal(dec.cigar, batch.cigar)
        np.testing.assert_array_equal(np.array([w[0] for w in dec.wide], np.uint64), batch.wide_index)
        np.testing.assert_array_equal(np.array([w[2] for w in dec.wide], np.int32), batch.wide_l_qseq)
        np.testing.assert_array_equal(dec.seg_tid, batch.seg_tid)
        np.testing.assert_array_equal(dec.seg_start, batch.seg_start)


def test_window_decode_diagnostics(tmp_path):
    """The unsorted-input test and the unrecognised-RefID names (src/RNASeQC.cpp:333-337,354-355), across window borders."""
    recs = [dict(tid=0, pos=100 + 10 * i, mpos=0, isize=0, flag=0, cigar=[(abi.CIG_M, 50)], qname="r%d" % i) for i in range(2000)]
    recs[1500]["pos"] = 5                                                      # goes backwards
    recs[700]["tid"] = 7; recs[700]["qname"] = "alien"                          # RefID outside the header
    recs[701]["flag"] = 0x100; recs[701]["pos"] = 1                             # secondary: not judged
    batch = Batch.from_records(recs)
    path = str(tmp_path / "d.bam")
    bamio.write_bam(path, [("chrA", 3_000_000), ("chrB", 1_000_000)], batch)
    stream, first, n_ref = emu.inflate_bam(path, use_emu=False)
    for window in (1 << 30, 9_000):
        dec = emu.decode_stream(stream, first, n_ref, window)
        assert dec.n == 2000 and dec.unsorted and dec.bad_names == ["alien"]
    recs[1500]["pos"] = 100 + 15000
    batch = Batch.from_records(recs)
    bamio.write_bam(path, [("chrA", 3_000_000), ("chrB", 1_000_000)], batch)
    stream, first, n_ref = emu.inflate_bam(path, use_emu=False)
    # the only backwards step is exactly at a window border: judged against the carried record
    dec = emu.decode_stream(stream, first, n_ref, 9_000)
    assert not dec.unsorted
    recs[1600]["pos"] = 7
    bamio.write_bam(path, [("chrA", 3_000_000), ("chrB", 1_000_000)], Batch.from_records(recs))
    stream, first, n_ref = emu.inflate_bam(path, use_emu=False)
    for window in (400, 1 << 30):
        assert emu.decode_stream(stream, first, n_ref, window).unsorted


# ---- the host side of the device decode: file chunks and BGZF block tables (host/bgzf_feed.cpp) --------------------
import ctypes as C
from tests.hostemu.decode import feed_chunks, _feed_lib


def _inflate_chunks(chunks):
    """The record bytes of a range: every block inflated (zlib; CRC and ISIZE checked), skip / limit applied."""
    parts = []
    for comp, tab, skip, limit, _last in chunks:
        raw = []
        for b in tab:
            if b["flags"] & abi.BGZF_INFLATED:                     # the feeder's CPU share: the bytes are already inflated (and checked)
                d = comp[int(b["in_offset"]):int(b["in_offset"]) + int(b["in_bytes"])]
            else:
                d = zlib.decompress(comp[int(b["in_offset"]):int(b["in_offset"]) + int(b["in_bytes"])], -15)
            assert len(d) == b["out_bytes"] and zlib.crc32(d) == b["crc32"]
            raw.append(d)
        raw = b"".join(raw)
        parts.append(raw[skip:limit] if limit else raw[skip:])
    return b"".join(parts)


@pytest.mark.parametrize("chunk_bytes,max_out", [(1 << 17, 1 << 40), (200_000, 300_000), (48 << 20, 768 << 20)])
def test_feeder_whole_file_and_contig_ranges(tmp_path, chunk_bytes, max_out):
    contigs = [("chrA", 3_000_000), ("chrB", 1_000_000), ("chrC", 500_000)]
    ann = synth.make_annotation(seed=35, contigs=[("chrA", 3_000_000, 120), ("chrB", 1_000_000, 40), ("chrC", 500_000, 10)])
    batch = synth.make_reads(ann, 40_000, seed=36, keep_qnames=True, contig_lengths=np.array([3_000_000, 1_000_000, 500_000]))
    path = str(tmp_path / "f.bam")
    voff = bamio.write_bam_fast(path, contigs, batch, threads=3, seq_mode=1, bai=True)
    # whole file: header skipped through the first chunk's skip, every block exactly once
    chunks = feed_chunks(path, chunk_bytes=chunk_bytes, max_out=max_out)
    assert chunks[-1][4] and all(sum(int(x) for x in c[1]["out_bytes"]) <= max(max_out, 65536) for c in chunks)
    stream = _inflate_chunks(chunks)
    dec = emu.decode_stream(stream, 0, 3, 1 << 30)
    _check(dec, batch)
    # one contig at a time through the index's virtual offsets: exactly its records
    for s in range(len(batch.seg_tid)):
        lo, hi = int(batch.seg_start[s]), int(batch.seg_start[s + 1])
        part = _inflate_chunks(feed_chunks(path, int(voff[s]), int(voff[s + 1]), chunk_bytes=chunk_bytes, max_out=max_out))
        d = emu.decode_stream(part, 0, 3, 70_000)
        assert d.n == hi - lo and list(d.seg_tid) == [int(batch.seg_tid[s])]
        np.testing.assert_array_equal(d.core["pos"], batch.pos[lo:hi])
        np.testing.assert_array_equal(d.aux["qhash"], batch.qhash[lo:hi])


def test_feeder_sizes_its_calls_by_the_files_compression(tmp_path):
    """A call holds max_out inflated bytes whatever the file's compression: the feeder reads what that takes at the ratio of the chunk
    before (not chunk_bytes of a file compressed 12 x, most of which the call could not hold and the next one would read again), and
    buffers set up in advance for the file's sampled ratio are not outgrown -- a call gets smaller instead.  The stream is the same."""
    contigs = [("chrA", 3_000_000), ("chrB", 1_000_000)]
    ann = synth.make_annotation(seed=35, contigs=[("chrA", 3_000_000, 120), ("chrB", 1_000_000, 40)])
    batch = synth.make_reads(ann, 60_000, seed=36, keep_qnames=True, contig_lengths=np.array([3_000_000, 1_000_000]))
    for mode in (0, 1):
        path = str(tmp_path / ("r%d.bam" % mode))
        bamio.write_bam_fast(path, contigs, batch, threads=3, seq_mode=mode)
        size = os.path.getsize(path)
        plain = _inflate_chunks(feed_chunks(path, chunk_bytes=1 << 20))
        max_out = 2 << 20
        for reserve in (None, (size, max_out), (1 << 18, max_out), (size, max_out, (2, 0.2, 0.5))):
            share = reserve[2] if reserve and len(reserve) > 2 else None
            chunks = feed_chunks(path, chunk_bytes=size, max_out=max_out, reserve=reserve[:2] if reserve else None, cpu_share=share)
            assert _inflate_chunks(chunks) == plain
            outs = [int(c[1]["out_bytes"].sum()) for c in chunks]
            assert all(o <= max_out for o in outs)
            if reserve is None or reserve[0] == size:
                # steady state: calls filled to within two blocks of max_out, from file bytes that are a fraction of chunk_bytes
                assert len(chunks) >= 4 and all(o > max_out - 2 * 65536 for o in outs[1:-1])
                file_bytes = [int((c[1]["in_offset"][-1] + c[1]["in_bytes"][-1]) if not (c[1]["flags"] & abi.BGZF_INFLATED).any() else 0) for c in chunks]
                assert all(fb < size // 2 for fb in file_bytes)


def test_feeder_cpu_share(tmp_path):
    """The feeder inflates the tail of every chunk on CPU threads (RSQC_BGZF_INFLATED blocks behind the file bytes): one run at
    the end of the table, bytes one after the other, and the stream is the same."""
    contigs = [("chrA", 3_000_000), ("chrB", 1_000_000)]
    ann = synth.make_annotation(seed=35, contigs=[("chrA", 3_000_000, 120), ("chrB", 1_000_000, 40)])
    batch = synth.make_reads(ann, 60_000, seed=36, keep_qnames=True, contig_lengths=np.array([3_000_000, 1_000_000]))
    path = str(tmp_path / "c.bam")
    bamio.write_bam_fast(path, contigs, batch, threads=3, seq_mode=1)
    plain = _inflate_chunks(feed_chunks(path, chunk_bytes=1 << 20))
    seen = 0
    for share in ((3, 0.4, 0.5), (2, 0.05, 0.9)):
        chunks = feed_chunks(path, chunk_bytes=1 << 20, cpu_share=share)
        for comp, tab, _s, _l, _last in chunks:
            fl = (tab["flags"] & abi.BGZF_INFLATED) != 0
            seen += int(fl.sum())
            if fl.any():
                k = int(np.argmax(fl))
                assert fl[k:].all() and not fl[:k].any()
                assert (tab["in_bytes"][k:] == tab["out_bytes"][k:]).all()
                assert (np.diff(tab["in_offset"][k:].astype(np.int64)) == tab["out_bytes"][k:-1]).all()
                assert int(tab["in_offset"][-1]) + int(tab["out_bytes"][-1]) == len(comp)
        assert _inflate_chunks(chunks) == plain
    assert seen > 10


def test_feeder_reports_truncated_and_foreign_files(tmp_path):
    contigs = [("chrA", 3_000_000)]
    ann = synth.make_annotation(seed=3, contigs=[("chrA", 3_000_000, 50)])
    batch = synth.make_reads(ann, 5000, seed=4, keep_qnames=True, contig_lengths=np.array([3_000_000]))
    path = str(tmp_path / "t.bam")
    bamio.write_bam_fast(path, contigs, batch, threads=2)
    data = open(path, "rb").read()
    cut = str(tmp_path / "cut.bam")
    open(cut, "wb").write(data[:len(data) - 40])                      # ends inside the EOF marker block
    with pytest.raises(RuntimeError, match="truncated BGZF block"):
        feed_chunks(cut)
    junk = str(tmp_path / "junk.bam")
    open(junk, "wb").write(b"not a bam file at all, just text" * 10)
    l = _feed_lib(); h = l.host_feed_open(junk.encode())
    assert l.host_feed_first_voffset(h) == 2 ** 64 - 1 and b"BGZF" in l.host_feed_error(h)
    l.host_feed_free(h)


def test_periodic_copy_reciprocal_is_exact():
    """inflate_copy's j mod dist for dist < 64: (j * ceil(65536 / dist)) >> 16 == j // dist over the whole range it is used on."""
    for d in range(1, 64):
        recip = (65536 + d - 1) // d
        assert all(((j * recip) >> 16) == j // d for j in range(1040)), d


def _sanitized(tmp_path, name, libs=(), defs=()):
    """tests/hostemu/<name>.cpp built with the address and undefined-behaviour sanitizers (a finding aborts the run)."""
    import subprocess
    exe = str(tmp_path / name)
    src = os.path.join(os.path.dirname(os.path.abspath(__file__)), "hostemu", name + ".cpp")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=all", *defs, src, "-o", exe, *libs])
    return exe


@pytest.mark.parametrize("variant", ["", "all"])
def test_inflate_on_damaged_input_under_sanitizers(tmp_path, variant):
    """What the wave's DEFLATE decoder does with flipped bits, overwritten bytes, cut payloads, garbage and wrong ISIZE values:
    it stays inside its input, its ISIZE bytes of output and its tables, it ends, and it never hands on wrong bytes (on the
    GPU the first two are a dead device).  Verdicts are zlib's, see tests/hostemu/inflate_fuzz.cpp."""
    import subprocess
    exe = _sanitized(tmp_path, "inflate_fuzz", ["-lz"], emu.VARIANTS[variant])
    for seed in (1, 2):
        r = subprocess.run([exe, "4000", str(seed)], capture_output=True, text=True, timeout=600, env=dict(os.environ, ASAN_OPTIONS="detect_leaks=0"))
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
        assert "4000 cases" in r.stdout


def test_window_decode_on_damaged_records_under_sanitizers(tmp_path):
    """Framing, chain repair, offsets, parsing and lists on windows of damaged record bytes (a BGZF CRC only vouches for what
    the writer compressed): reads inside the window, writes inside buffers sized like the library sizes them, every loop
    ends, listed repair == sequential walk.  See tests/hostemu/decode_fuzz.cpp."""
    import subprocess
    exe = _sanitized(tmp_path, "decode_fuzz")
    for seed in (1, 2):
        r = subprocess.run([exe, "1500", str(seed)], capture_output=True, text=True, timeout=900, env=dict(os.environ, ASAN_OPTIONS="detect_leaks=0"))
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
        assert "1500 cases" in r.stdout


@pytest.mark.parametrize("level", [0, 6, 9])
def test_bam_of_other_compression_levels_through_the_emulation(tmp_path, level):
    """The files of tests/test_zz_gpu_decode_levels.py (stored blocks, zlib's default and best levels) through the emulated
    inflate and window decode."""
    contigs = [("chrA", 3_000_000), ("chrB", 1_000_000), ("chrC", 500_000)]
    ann = synth.make_annotation(seed=35, contigs=[("chrA", 3_000_000, 120), ("chrB", 1_000_000, 40), ("chrC", 500_000, 10)])
    batch = synth.make_reads(ann, 30_000, seed=37 + level, keep_qnames=True, chimeric_tag_frac=0.02, filter_tag_frac=0.03,
                             contig_lengths=np.array([3_000_000, 1_000_000, 500_000]))
    path = str(tmp_path / "l.bam")
    bamio.write_bam(path, contigs, batch, level=level)
    stream, first, n_ref = emu.inflate_bam(path)
    dec = emu.decode_stream(stream, first, n_ref, 1 << 30, "ch", ("XF",), threads=5)
    _check(dec, batch)
