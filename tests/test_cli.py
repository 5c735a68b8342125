"""The `rnaseqc gtf bam output` command line: exit codes (CPU) and end-to-end outputs (GPU)."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from rnaseqc_amd import abi, bamio, synth
from tests import cases
from tests.test_host_cli_pieces import host, load_annotation, _results_struct, read_table  # noqa: F401

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "rnaseqc_amd", "bin", "rnaseqc")


@pytest.fixture(scope="module")
def cli():
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "rnaseqc_amd", "csrc"), "cli"])
    return BIN


def run(cli, *args):
    p = subprocess.run([cli, *args], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    return p.returncode, p.stdout.decode(), p.stderr.decode()


def test_cli_exit_codes_without_gpu_work(cli, tmp_path):
    # reference behaviour: SURVEY.md Appendix B "CLI behaviour", src/RNASeQC.cpp:678-766
    rc, out, _ = run(cli, "--version")
    assert rc == 0 and out.strip() == "RNASeQC 2.4.3"                  # python/rnaseqc/run.py:25 needs the prefix
    assert run(cli, "-h")[0] == 4
    assert run(cli)[0] == 6 and run(cli, "a.gtf", "b.bam")[0] == 6
    assert run(cli, "a", "b", "c", "--stranded", "xx")[0] == 6
    assert run(cli, "a", "b", "c", "--nope")[0] == 5 and run(cli, "a", "b", "c", "-q", "abc")[0] == 5
    assert run(cli, str(tmp_path / "missing.gtf"), "b.bam", str(tmp_path / "o"))[0] == 10
    ann, batch = cases.quirk_case()
    gtf, bam = str(tmp_path / "q.gtf"), str(tmp_path / "q.bam")
    bamio.write_gtf(gtf, ann)
    assert run(cli, gtf, str(tmp_path / "missing.bam"), str(tmp_path / "o"))[0] == 10
    assert os.path.isdir(tmp_path / "o")                               # the output dir is created before the BAM is opened
    bamio.write_bam(bam, [("other1", 1000), ("other2", 1000)], batch.slice(0, 0))
    assert run(cli, gtf, bam, str(tmp_path / "o"))[0] == 11            # BAM shares no contigs with the GTF
    fa = tmp_path / "r.fa"; fa.write_text(">chr1\nACGT\n")
    assert run(cli, gtf, bam, str(tmp_path / "o"), "--fasta", str(tmp_path / "missing.fa"))[0] == 10   # fileException
    assert run(cli, gtf, bam, str(tmp_path / "o"), "--fasta", str(fa))[0] == 10                        # no .fai beside it
    empty = tmp_path / "e.gtf"; empty.write_text('c\tx\ttranscript\t1\t5\t.\t+\t.\tgene_id "A"; transcript_id "T";\n')
    assert run(cli, str(empty), bam, str(tmp_path / "o"))[0] == 11     # no genes / exons


def _compare_tables(a, b, skip, tol=1e-6):
    """test_data/approx_diff.py semantics: join on the first column, same NaN pattern, |a-b| <= tol."""
    ta = {r[0]: r[1:] for r in read_table(a, skip)}
    tb = {r[0]: r[1:] for r in read_table(b, skip)}
    assert ta.keys() == tb.keys(), (a, set(ta) ^ set(tb))
    for k in ta:
        for x, y in zip(ta[k], tb[k]):
            try:
                fx, fy = float(x), float(y)
            except ValueError:
                assert x == y, (a, k, x, y)
                continue
            assert np.isnan(fx) == np.isnan(fy), (a, k, x, y)
            if not np.isnan(fx):
                assert abs(fx - fy) <= tol + 1e-9 * abs(fy), (a, k, x, y)


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["plain", "bed", "legacy", "fasta"])
def test_cli_end_to_end_matches_oracle_outputs(cli, host, oracle_lib, tmp_path, mode):
    with_bed, legacy, with_fasta = mode == "bed", mode == "legacy", mode == "fasta"
    contigs = [("chrA", 900_000, 70), ("chrB", 500_000, 40)]
    ann = synth.make_annotation(seed=41, contigs=contigs)
    batch = synth.make_reads(ann, 40000, seed=42, keep_qnames=True, dup_frac=0.1, frac=(0.85, 0.06, 0.05, 0.04), expr_sigma=1.2,
                             contig_lengths=np.array([c[1] for c in contigs]))
    gtf, bam, bedp = str(tmp_path / "s.gtf"), str(tmp_path / "s.bam"), str(tmp_path / "s.bed")
    bamio.write_gtf(gtf, ann)
    bamio.write_bam(bam, [(c[0], c[1]) for c in contigs], batch)
    bed = synth.make_bed(ann, min_len=250) if with_bed else None
    args = [gtf, bam, str(tmp_path / "cli"), "--coverage", "-vv"]
    if with_bed:
        bamio.write_bed(bedp, ann, bed)
        args += ["--bed", bedp]
    ref = None
    if with_fasta:                                               # chrA only: chrB is not in the FASTA index
        ref = synth.make_reference([contigs[0][1]], seed=43)
        bamio.write_fasta(str(tmp_path / "ref.fa"), ["chrA"], ref, index_path=str(tmp_path / "ref.fai"))   # <stem>.fai is looked up first
        args += ["--fasta", str(tmp_path / "ref.fa")]
    if legacy:
        args.append("--legacy")                                  # counting rules + the -q 4 default (src/RNASeQC.cpp:90)
    env = dict(os.environ, RSQC_BATCH="30000")                   # several batches
    p = subprocess.run([cli, *args], stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env)
    assert p.returncode == 0, p.stderr.decode()
    assert "Average Reads/Sec" in p.stdout.decode()
    # expected files: the same C++ report writer fed with the ORACLE's results on the same inputs
    want = oracle_lib.run_oracle(abi.default_params(**(dict(legacy=1, mapq_threshold=4) if legacy else {})), ann, [batch], bed=bed, reference=ref)
    h, err = load_annotation(host, gtf, [c[0] for c in contigs], bedp if with_bed else None)
    assert err == 0
    if legacy:
        assert want.counter("Split Reads") > 100 and "Split Reads\t" in open(os.path.join(str(tmp_path / "cli"), "s.bam.metrics.tsv")).read()
    rs, keep = _results_struct(want)
    exp = str(tmp_path / "exp"); os.makedirs(exp)
    visit = (C.c_int * 2)(0, 1)
    assert host.host_write_reports(h, C.byref(rs), exp.encode(), b"s.bam", 0, 0, 1, 5, None, 0, visit, 2) == 0
    files = ["metrics.tsv", "gene_reads.gct", "gene_tpm.gct", "gene_fragments.gct", "exon_reads.gct", "coverage.tsv", "exon_cv.tsv"]
    if with_bed:
        files.append("fragmentSizes.txt")
        assert want.fragment_count.sum() > 50
    if with_fasta:
        files.append("gc_content.tsv")
        assert int(want.gc_bins.sum()) > 500
        assert "Fragment GC Content Kurtosis" in open(os.path.join(str(tmp_path / "cli"), "s.bam.metrics.tsv")).read()
        assert open(os.path.join(str(tmp_path / "cli"), "s.bam.exon_cv.tsv")).readline() == "Exon ID\tExon CV\tGC Content\n"
    for f in files:
        skip = 3 if f.endswith(".gct") else 1
        _compare_tables(os.path.join(str(tmp_path / "cli"), "s.bam." + f), os.path.join(exp, "s.bam." + f), skip, tol=1e-5)
    # integer tables must be byte-identical
    for f in ["gene_reads.gct", "gene_fragments.gct"] + (["fragmentSizes.txt"] if with_bed else []) + (["gc_content.tsv"] if with_fasta else []):
        assert open(os.path.join(str(tmp_path / "cli"), "s.bam." + f)).read() == open(os.path.join(exp, "s.bam." + f)).read()
    host.host_annotation_free(h)


@pytest.mark.gpu
def test_cli_reads_a_bam_from_a_fifo(cli, tmp_path):
    """A BAM that is not a regular file (FIFO, /dev/stdin, process substitution): the block feeder of the device decode needs
    pread, so such an input is streamed by the host reader -- same report files as the run on the file itself."""
    import threading
    contigs = [("chrA", 900_000, 70), ("chrB", 500_000, 40)]
    ann = synth.make_annotation(seed=41, contigs=contigs)
    batch = synth.make_reads(ann, 8000, seed=42, keep_qnames=True, contig_lengths=np.array([c[1] for c in contigs]))
    gtf, bam, fifo = str(tmp_path / "s.gtf"), str(tmp_path / "s.bam"), str(tmp_path / "s.fifo")
    bamio.write_gtf(gtf, ann)
    bamio.write_bam(bam, [(c[0], c[1]) for c in contigs], batch)
    os.mkfifo(fifo)
    def writer():
        with open(fifo, "wb") as w, open(bam, "rb") as r:
            w.write(r.read())
    t = threading.Thread(target=writer); t.start()
    a = subprocess.run([cli, gtf, fifo, str(tmp_path / "from_fifo"), "-s", "x", "-vv"], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    t.join()
    assert a.returncode == 0, a.stderr.decode()
    assert "on the GPU" not in a.stdout.decode()                 # host decode
    b = subprocess.run([cli, gtf, bam, str(tmp_path / "from_file"), "-s", "x", "-vv"], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert b.returncode == 0, b.stderr.decode()
    assert "on the GPU" in b.stdout.decode()
    for f in ("metrics.tsv", "gene_reads.gct", "gene_fragments.gct", "exon_reads.gct", "gene_tpm.gct"):
        assert open(str(tmp_path / "from_fifo" / ("x." + f))).read() == open(str(tmp_path / "from_file" / ("x." + f))).read(), f


@pytest.mark.gpu
def test_cli_stderr_of_the_reference_loop(cli, tmp_path):
    """src/RNASeQC.cpp:354-355: the sort warning (positions going backwards inside a contig, or a contig visited twice);
    :333-337: under -v, the names of primary mapped records whose RefID the header does not define.  A sorted file with
    a complete header prints neither."""
    contigs = [("chrA", 900_000, 70), ("chrB", 500_000, 40)]
    ann = synth.make_annotation(seed=41, contigs=contigs)
    batch = synth.make_reads(ann, 6000, seed=42, keep_qnames=True, contig_lengths=np.array([c[1] for c in contigs]))
    gtf = str(tmp_path / "s.gtf")
    bamio.write_gtf(gtf, ann)
    WARN = "Warning: The input bam does not appear to be sorted. An unsorted bam will yield incorrect results"

    def go(b, hdr, *flags):
        bam = str(tmp_path / "t.bam")
        bamio.write_bam(bam, hdr, b)
        p = subprocess.run([cli, gtf, bam, str(tmp_path / "o"), *flags], stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                           env=dict(os.environ, RSQC_BATCH="2500"))
        assert p.returncode == 0, p.stderr.decode()
        return p.stderr.decode()

    hdr = [(c[0], c[1]) for c in contigs]
    err = go(batch, hdr, "-v")
    assert WARN not in err and "Unrecognized RefID" not in err
    tid = batch.tid_per_record()
    ok = np.flatnonzero((tid == 1) & ((batch.flag & 0x904) == 0))
    sw = batch.slice(0, batch.n)
    i, j = int(ok[5]), int(ok[900])
    sw.pos[i], sw.pos[j] = batch.pos[j], batch.pos[i]
    assert go(sw, hdr).count(WARN) == 1
    # the second contig's records come first: chrB, chrA, then the unmapped tail -> still sorted per contig, no warning;
    # a contig that comes back after another one is a revisit -> warning
    a_lo, a_hi = int(batch.seg_start[0]), int(batch.seg_start[1])
    from rnaseqc_amd.model import Batch
    back = Batch.concat([batch.slice(a_lo, a_lo + 1000), batch.slice(a_hi, int(batch.seg_start[2])), batch.slice(a_lo + 1000, a_hi)])
    assert go(back, hdr).count(WARN) == 1
    err = go(batch, hdr[:1], "-v")                                # chrB's records now carry a RefID the header lacks
    assert err.count("Unrecognized RefID on alignment: SYN:") >= 10 and WARN not in err
    assert "Unrecognized RefID" not in go(batch, hdr[:1])         # only under -v, like the reference


@pytest.mark.gpu
def test_cli_sharded_by_contig_matches_single_gpu(cli, tmp_path):
    """`--gpus N`: the BAM sharded by contig through its index, one context per GPU (here both on the box's one GPU,
    RSQC_GPU_LIST=0,0), results summed with rsqc_reduce_peer and the order-dependent outputs composed from the shard
    summaries -- every output file must equal the single-GPU run's, incl. fragmentSizes.txt under a tight --fragment-samples
    cut-off and Read Length on input with mixed read lengths."""
    contigs = [("cA", 2_000_000, 150), ("cB", 1_500_000, 120), ("cC", 900_000, 60), ("cD", 700_000, 50)]
    ann = synth.make_annotation(seed=21, contigs=contigs)
    batch = synth.make_reads(ann, 30000, seed=22, dup_frac=0.05, contig_lengths=np.array([c[1] for c in contigs]))
    rng = np.random.default_rng(5)
    batch.l_qseq = rng.choice(np.array([36, 50, 76, 101, 150], np.uint16), size=batch.n).astype(np.uint16)
    bed = synth.make_bed(ann, min_len=300)
    gtf, bam, bedp = str(tmp_path / "s.gtf"), str(tmp_path / "s.bam"), str(tmp_path / "s.bed")
    bamio.write_gtf(gtf, ann)
    bamio.write_bam_fast(bam, [(c[0], c[1]) for c in contigs], batch, threads=4, bai=True)
    bamio.write_bed(bedp, ann, bed)
    common = [gtf, bam, "--coverage", "--bed", bedp, "--fragment-samples", "300", "-vv"]
    env = dict(os.environ, RSQC_BATCH="9000")
    one = subprocess.run([cli, *common[:2], str(tmp_path / "one"), *common[2:]], stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env)
    assert one.returncode == 0, one.stderr.decode()
    two = subprocess.run([cli, *common[:2], str(tmp_path / "two"), *common[2:], "--gpus", "2"], stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                         env=dict(env, RSQC_GPU_LIST="0,0"))
    assert two.returncode == 0, two.stderr.decode()
    assert "on 2 GPUs" in two.stdout.decode()
    m1 = dict(read_table(str(tmp_path / "one" / "s.bam.metrics.tsv"))); m2 = dict(read_table(str(tmp_path / "two" / "s.bam.metrics.tsv")))
    assert m1.keys() == m2.keys()
    assert m1["Read Length"] == m2["Read Length"] and m1["Total Alignments"] == m2["Total Alignments"] == str(batch.n)
    for f in ("gene_reads.gct", "gene_fragments.gct", "fragmentSizes.txt"):
        assert open(str(tmp_path / "one" / ("s.bam." + f))).read() == open(str(tmp_path / "two" / ("s.bam." + f))).read(), f
    assert sum(int(r[1]) for r in read_table(str(tmp_path / "two" / "s.bam.fragmentSizes.txt"), 1)) == 300
    for f in ("metrics.tsv", "gene_tpm.gct", "exon_reads.gct", "coverage.tsv", "exon_cv.tsv"):
        _compare_tables(str(tmp_path / "one" / ("s.bam." + f)), str(tmp_path / "two" / ("s.bam." + f)), 3 if f.endswith(".gct") else 1, tol=1e-6)
    # without the index the run falls back to one GPU with a notice
    os.remove(bam + ".bai")
    three = subprocess.run([cli, *common[:2], str(tmp_path / "three"), *common[2:], "--gpus", "2"], stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                           env=dict(env, RSQC_GPU_LIST="0,0"))
    assert three.returncode == 0 and "needs the BAM index" in three.stderr.decode()
    assert open(str(tmp_path / "three" / "s.bam.gene_reads.gct")).read() == open(str(tmp_path / "one" / "s.bam.gene_reads.gct")).read()
