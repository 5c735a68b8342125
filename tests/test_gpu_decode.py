"""-m gpu: the device-side BAM decode (rsqc_decode_*) through the C ABI -- BGZF inflate, record framing
and record parsing on the GPU -- against the records that were written (column for column, read back from the device),
against the host-decoded run of the same file (every result), and through the command line in both decode modes."""
import os
import subprocess

import numpy as np
import pytest

from rnaseqc_amd import abi, bamio, engine, synth
from rnaseqc_amd.model import Batch
from tests.compare import assert_results_match
from tests.hostemu.decode import feed_chunks
from tests.test_cli import cli, read_table, _compare_tables  # noqa: F401

pytestmark = pytest.mark.gpu

def decode_file(e, path, n_ref, ch_tag="ch", filter_tags=(), chunk_bytes=48 << 20, max_out=768 << 20, collect=True, voff=None, base=0, cpu_share=None):
    """Feeds a BAM file (or the range voff = (beg, end)) through rsqc_decode_*.  With collect, reads every decoded
    batch back from the device and returns it concatenated."""
    chunks = feed_chunks(path, *(voff or (None, 0)), chunk_bytes=chunk_bytes, max_out=max_out, cpu_share=cpu_share)
    e.decode_begin(n_ref, ch_tag, filter_tags, base)
    parts, runs_all, total = [], [], 0
    for comp, tab, skip, limit, _last in chunks:
        n, runs = e.decode_submit(comp, tab, skip, limit)
        runs_all += runs; total += n
        if collect and n:
            e.wait()
            s = e.last_decoded()
            assert s.n == n and s.n_seg == len(runs)
            rd = e.read_device
            parts.append(dict(core=rd(s.core, n, abi.REC_CORE), aux=rd(s.aux, n, abi.REC_AUX), qhash2=rd(s.qhash2, n, np.uint32), cigar=rd(s.cigar, s.n_cigar_total, np.uint32),
                              seg_tid=rd(s.seg_tid, s.n_seg, np.int32), seg_start=rd(s.seg_start, s.n_seg + 1, np.uint64),
                              wide_index=rd(s.wide_index, s.n_wide, np.uint64), wide_nm=rd(s.wide_nm, s.n_wide, np.int32),
                              wide_lq=rd(s.wide_l_qseq, s.n_wide, np.int32), wide_nc=rd(s.wide_n_cigar, s.n_wide, np.uint32),
                              base=s.file_index_base))
    info = e.decode_end()
    return parts, runs_all, total, info, len(chunks)


def check_columns(parts, batch, first_base=0):
    at, ops = 0, 0
    for p in parts:
        n = len(p["core"])
        assert p["base"] == at + first_base
        assert list(p["seg_start"][:1]) == [0] and int(p["seg_start"][-1]) == n
        np.testing.assert_array_equal(p["core"]["cigar_off"], batch.cigar_off[at:at + n] - ops)
        for f in ("pos", "mpos", "isize"):
            np.testing.assert_array_equal(p["core"][f], getattr(batch, f)[at:at + n], err_msg=f)
        for f in ("qhash", "flag", "l_qseq", "mapq", "nm", "tagbits", "n_cigar"):
            np.testing.assert_array_equal(p["aux"][f], getattr(batch, f)[at:at + n], err_msg=f)
        if batch.qhash2 is not None:
            np.testing.assert_array_equal(p["qhash2"], batch.qhash2[at:at + n], err_msg="qhash2")
        np.testing.assert_array_equal(p["cigar"], batch.cigar[ops:ops + len(p["cigar"])])
        at += n; ops += len(p["cigar"])
    assert at == batch.n and ops == len(batch.cigar)
    # contig runs and the wide table, in whole-file numbering
    tids, starts, wide = [], [], []
    at = 0
    for p in parts:
        for k, t in enumerate(p["seg_tid"]):
            if not tids or tids[-1] != int(t) or k > 0:
                tids.append(int(t)); starts.append(at + int(p["seg_start"][k]))
        wide += [(at + int(i), int(a), int(b), int(c)) for i, a, b, c in zip(p["wide_index"], p["wide_nm"], p["wide_lq"], p["wide_nc"])]
        at += len(p["core"])
    assert tids == [int(t) for t in batch.seg_tid] and starts == [int(x) for x in batch.seg_start[:-1]]
    np.testing.assert_array_equal(np.array([w[0] for w in wide], np.uint64), batch.wide_index)
    np.testing.assert_array_equal(np.array([w[1] for w in wide], np.int32), batch.wide_nm)
    np.testing.assert_array_equal(np.array([w[2] for w in wide], np.int32), batch.wide_l_qseq)
    np.testing.assert_array_equal(np.array([w[3] for w in wide], np.uint32), batch.wide_n_cigar)


@pytest.mark.parametrize("chunk_bytes,max_out", [(48 << 20, 768 << 20), (1 << 17, 1 << 40), (1 << 18, 200_000)])
def test_decode_columns_and_results(tmp_path, chunk_bytes, max_out):
    """Python-zlib BGZF blocks (dynamic codes, records across block and call borders), tags of every kind: the decoded
    columns equal the written records and the run's results equal the host-fed run's."""
    contigs = [("chrA", 3_000_000), ("chrB", 1_000_000), ("chrC", 500_000)]
    ann = synth.make_annotation(seed=35, contigs=[("chrA", 3_000_000, 120), ("chrB", 1_000_000, 40), ("chrC", 500_000, 10)])
    batch = synth.make_reads(ann, 30_000, seed=36, keep_qnames=True, chimeric_tag_frac=0.02, filter_tag_frac=0.03,
                             contig_lengths=np.array([3_000_000, 1_000_000, 500_000]))
    path = str(tmp_path / "p.bam")
    bamio.write_bam(path, contigs, batch)
    p = abi.default_params(); p.n_filter_tags = 1
    e = engine.Engine(p)
    e.set_annotation(ann)
    parts, runs, total, info, n_calls = decode_file(e, path, 3, "ch", ("XF",), chunk_bytes, max_out)
    assert total == batch.n and info[0] == batch.n and not info[1] and info[2] == 0
    if chunk_bytes < (1 << 20):
        assert n_calls > 3
    check_columns(parts, batch)
    got = e.finalize()
    e.close()
    want = engine.run_engine(p, ann, [batch])
    assert_results_match(got, want)


def test_decode_with_cpu_share(tmp_path):
    """The feeder's CPU threads inflate the tail of every chunk (RSQC_BGZF_INFLATED blocks): same columns, same results."""
    contigs = [("chrA", 3_000_000), ("chrB", 1_000_000), ("chrC", 500_000)]
    ann = synth.make_annotation(seed=35, contigs=[("chrA", 3_000_000, 120), ("chrB", 1_000_000, 40), ("chrC", 500_000, 10)])
    batch = synth.make_reads(ann, 40_000, seed=36, keep_qnames=True, chimeric_tag_frac=0.02, filter_tag_frac=0.03,
                             contig_lengths=np.array([3_000_000, 1_000_000, 500_000]))
    path = str(tmp_path / "p.bam")
    bamio.write_bam(path, contigs, batch)
    p = abi.default_params(); p.n_filter_tags = 1
    want = engine.run_engine(p, ann, [batch])
    for chunk, share in ((1 << 19, (3, 0.4, 0.5)), (48 << 20, (2, 0.3, 0.9))):
        e = engine.Engine(p)
        e.set_annotation(ann)
        parts, runs, total, info, n_calls = decode_file(e, path, 3, "ch", ("XF",), chunk, cpu_share=share)
        assert total == batch.n
        check_columns(parts, batch)
        got = e.finalize()
        e.close()
        assert_results_match(got, want)


def test_decode_pipelined_stream(tmp_path):
    """rsqc_decode_params.pipelined: a call returns once its kernels are enqueued and is completed by the next one (the next
    chunk of the file is copied beside its kernels); the counts arrive one call late, the last with rsqc_decode_end."""
    contigs = [("chrA", 3_000_000), ("chrB", 1_000_000), ("chrC", 500_000)]
    ann = synth.make_annotation(seed=35, contigs=[("chrA", 3_000_000, 120), ("chrB", 1_000_000, 40), ("chrC", 500_000, 10)])
    batch = synth.make_reads(ann, 60_000, seed=36, contig_lengths=np.array([3_000_000, 1_000_000, 500_000]))
    path = str(tmp_path / "p.bam")
    bamio.write_bam_fast(path, contigs, batch, threads=3, seq_mode=1)
    p = abi.default_params()
    want = engine.run_engine(p, ann, [batch])
    for chunk, reserve in ((1 << 17, 0), (1 << 18, 64 << 20), (48 << 20, 0)):
        e = engine.Engine(p)
        e.set_annotation(ann)
        e.decode_begin(3, pipelined=True, reserve=reserve)
        counts, runs = [], []
        chunks = feed_chunks(path, chunk_bytes=chunk)
        for comp, tab, skip, limit, _last in chunks:
            n, r = e.decode_submit(comp, tab, skip, limit)
            counts.append(n); runs += r
        info = e.decode_end()
        counts.append(e.decode_last[0]); runs += e.decode_last[1]
        assert counts[0] == 0 and sum(counts) == batch.n == info[0]
        assert [t for k, t in enumerate(runs) if k == 0 or runs[k - 1] != t] == [int(t) for t in batch.seg_tid]
        got = e.finalize()
        e.close()
        assert_results_match(got, want)


def test_decode_pipelined_record_larger_than_head_room(tmp_path):
    """A pipelined stream whose window origin moves while a call is in preparation: a 7.5 MB record (5 M bases) straddles two
    calls, the call that completes its predecessor finds a carried part larger than the 4 MB head room, the window is rebuilt
    around a larger head -- and the block table of the call being prepared, laid out for the old origin, has to follow
    (it did not: the inflate overwrote the carried bytes and the window was shifted)."""
    recs = []
    for i in range(2000):
        recs.append(dict(tid=0, pos=100 + i, mpos=100 + i, isize=0, flag=0, cigar=[(abi.CIG_M, 100)], qname="s%d" % i))
    recs.append(dict(tid=0, pos=5000, mpos=5000, isize=0, flag=0, cigar=[(abi.CIG_M, 5_000_000)], qname="long"))
    for i in range(2000):
        recs.append(dict(tid=1 if i > 1000 else 0, pos=6000 + i, mpos=6000 + i, isize=0, flag=0, cigar=[(abi.CIG_M, 90), (abi.CIG_S, 10)], qname="t%d" % i))
    batch = Batch.from_records(recs)
    path = str(tmp_path / "pl.bam")
    bamio.write_bam(path, [("chrA", 8_000_000), ("chrB", 1_000_000)], batch)
    ann = synth.make_annotation(seed=3, contigs=[("chrA", 8_000_000, 50), ("chrB", 1_000_000, 20)])
    p = abi.default_params()
    want = engine.run_engine(p, ann, [batch])
    for chunk in (1 << 17, 1 << 20):
        e = engine.Engine(p)
        e.set_annotation(ann)
        e.decode_begin(2, pipelined=True)
        total = 0
        for comp, tab, skip, limit, _last in feed_chunks(path, chunk_bytes=chunk):
            total += e.decode_submit(comp, tab, skip, limit)[0]
        info = e.decode_end()
        total += e.decode_last[0]
        assert total == batch.n == info[0]
        got = e.finalize()
        e.close()
        assert_results_match(got, want)


def test_decode_fast_writer_blocks_long_record_and_ranges(tmp_path):
    """libdeflate-written blocks (the CLI benchmark's files), a 3 MB record that outgrows the head room kept for records that
    straddle two calls, and one contig at a time through the index's virtual offsets."""
    recs = []
    for i in range(3000):
        recs.append(dict(tid=0, pos=100 + i, mpos=100 + i, isize=0, flag=0, cigar=[(abi.CIG_M, 100)], qname="s%d" % i))
    recs.append(dict(tid=0, pos=5000, mpos=5000, isize=0, flag=0, cigar=[(abi.CIG_M, 3_000_000)], qname="long"))
    for i in range(3000):
        recs.append(dict(tid=1 if i > 1500 else 0, pos=6000 + i, mpos=6000 + i, isize=0, flag=0, cigar=[(abi.CIG_M, 90), (abi.CIG_S, 10)], qname="t%d" % i))
    batch = Batch.from_records(recs)
    path = str(tmp_path / "l.bam")
    bamio.write_bam(path, [("chrA", 4_000_000), ("chrB", 1_000_000)], batch)
    ann = synth.make_annotation(seed=3, contigs=[("chrA", 4_000_000, 50), ("chrB", 1_000_000, 20)])
    e = engine.Engine(abi.default_params())
    e.set_annotation(ann)
    for chunk, max_out in ((48 << 20, 768 << 20), (1 << 17, 300_000)):
        parts, runs, total, info, _n = decode_file(e, path, 2, chunk_bytes=chunk, max_out=max_out)
        check_columns(parts, batch)
        e.wait(); e.reset()
    e.close()
    # ranges
    contigs = [("cA", 2_000_000, 150), ("cB", 1_500_000, 120), ("cC", 900_000, 60)]
    ann = synth.make_annotation(seed=21, contigs=contigs)
    batch = synth.make_reads(ann, 40_000, seed=22, contig_lengths=np.array([c[1] for c in contigs]))
    path = str(tmp_path / "r.bam")
    voff = bamio.write_bam_fast(path, [(c[0], c[1]) for c in contigs], batch, threads=4, seq_mode=1, bai=True)
    e = engine.Engine(abi.default_params())
    e.set_annotation(ann)
    for s in range(len(batch.seg_tid)):
        lo, hi = int(batch.seg_start[s]), int(batch.seg_start[s + 1])
        parts, runs, total, info, _n = decode_file(e, path, 3, chunk_bytes=1 << 18, voff=(int(voff[s]), int(voff[s + 1])), base=s << 36)
        assert parts[0]["base"] == s << 36
        assert total == hi - lo and set(runs) == {int(batch.seg_tid[s])}
        np.testing.assert_array_equal(np.concatenate([p["core"]["pos"] for p in parts]), batch.pos[lo:hi])
        np.testing.assert_array_equal(np.concatenate([p["aux"]["flag"] for p in parts]), batch.flag[lo:hi])     # (the fast writer names records by their hash)
        e.wait()
    e.close()


def test_decode_errors_and_diagnostics(tmp_path):
    contigs = [("chrA", 3_000_000), ("chrB", 1_000_000)]
    ann = synth.make_annotation(seed=3, contigs=[("chrA", 3_000_000, 50), ("chrB", 1_000_000, 20)])
    recs = [dict(tid=0, pos=100 + 10 * i, mpos=0, isize=0, flag=0, cigar=[(abi.CIG_M, 50)], qname="r%d" % i) for i in range(20000)]
    recs[15000]["pos"] = 5                                                      # goes backwards
    recs[700]["tid"] = 7; recs[700]["qname"] = "alien"                          # RefID outside the header
    recs[701]["flag"] = 0x100; recs[701]["pos"] = 1                             # secondary: not judged
    batch = Batch.from_records(recs)
    path = str(tmp_path / "d.bam")
    bamio.write_bam(path, contigs, batch)
    e = engine.Engine(abi.default_params())
    e.set_annotation(ann)
    for chunk in (48 << 20, 1 << 17):
        _p, _r, total, info, _n = decode_file(e, path, 2, chunk_bytes=chunk, collect=False)
        assert total == 20000 and info[1] and info[2] == 1 and info[3] == ["alien"]
        e.wait(); e.reset()
    # a flipped payload bit: the block's CRC (or the decoder) catches it
    data = bytearray(open(path, "rb").read())
    data[len(data) // 2] ^= 0x10
    bad = str(tmp_path / "bad.bam"); open(bad, "wb").write(bytes(data))
    with pytest.raises(engine.EngineError) as ei:
        decode_file(e, bad, 2, collect=False)
    assert ei.value.code == abi.ERR_INPUT and "inflate" in str(ei.value)
    e.close()
    # a file cut in the middle of a record: whole blocks, but the last record is incomplete
    e = engine.Engine(abi.default_params())
    e.set_annotation(ann)
    chunks = feed_chunks(path)
    comp, tab, skip, limit, _last = chunks[0]
    e.decode_begin(2)
    e.decode_submit(comp, tab[:3], skip, 0)
    with pytest.raises(engine.EngineError) as ei:
        e.decode_end()
    assert ei.value.code == abi.ERR_INPUT and "truncated" in str(ei.value)
    e.close()


def test_cli_device_decode_equals_host_decode(cli, tmp_path):
    """The command line in both decode modes (and sharded over two contexts): identical output files."""
    contigs = [("cA", 2_000_000, 150), ("cB", 1_500_000, 120), ("cC", 900_000, 60), ("cD", 700_000, 50)]
    ann = synth.make_annotation(seed=21, contigs=contigs)
    batch = synth.make_reads(ann, 60_000, seed=22, dup_frac=0.05, contig_lengths=np.array([c[1] for c in contigs]))
    bed = synth.make_bed(ann, min_len=300)
    gtf, bam, bedp = str(tmp_path / "s.gtf"), str(tmp_path / "s.bam"), str(tmp_path / "s.bed")
    bamio.write_gtf(gtf, ann)
    bamio.write_bam_fast(bam, [(c[0], c[1]) for c in contigs], batch, threads=4, seq_mode=1, bai=True)
    bamio.write_bed(bedp, ann, bed)
    common = [gtf, bam, "--coverage", "--bed", bedp, "--fragment-samples", "300", "-vv"]
    outs = {}
    for name, env, extra in (("host", dict(RSQC_DECODE="host", RSQC_BATCH="9000"), []),
                             ("device", dict(RSQC_DECODE="device"), []),
                             ("device_small", dict(RSQC_DECODE="device", RSQC_DECODE_CHUNK=str(1 << 17), RSQC_DECODE_MAX_OUT="400000"), []),
                             ("device_two", dict(RSQC_DECODE="device", RSQC_GPU_LIST="0,0", RSQC_DECODE_CHUNK=str(1 << 18)), ["--gpus", "2"])):
        r = subprocess.run([cli, *common[:2], str(tmp_path / name), *common[2:], *extra], stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=dict(os.environ, **env))
        assert r.returncode == 0, (name, r.stderr.decode())
        outs[name] = r.stdout.decode()
    assert "on the GPU" in outs["device"] and "on the GPU" not in outs["host"] and "on 2 GPUs" in outs["device_two"]
    for name in ("device", "device_small", "device_two"):
        for f in ("gene_reads.gct", "gene_fragments.gct", "fragmentSizes.txt"):
            assert open(str(tmp_path / "host" / ("s.bam." + f))).read() == open(str(tmp_path / name / ("s.bam." + f))).read(), (name, f)
        for f in ("metrics.tsv", "gene_tpm.gct", "exon_reads.gct", "coverage.tsv", "exon_cv.tsv"):
            _compare_tables(str(tmp_path / "host" / ("s.bam." + f)), str(tmp_path / name / ("s.bam." + f)), 3 if f.endswith(".gct") else 1, tol=1e-6)
    m = dict(read_table(str(tmp_path / "device" / "s.bam.metrics.tsv")))
    assert m["Total Alignments"] == str(batch.n)
