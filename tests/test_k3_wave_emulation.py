"""No GPU needed: the end-of-file coverage kernel itself (rnaseqc_amd/csrc/rsqc_k3.h, unmodified source: coalesced block scan
into LDS, per-exon CV with the rows gathered per wave, argmax + gate, radix select of the 5th percentile, trim, window medians,
gene mean / std / CV) on the 64-lane SIMT emulation of tests/hostemu/wavemu.h against the oracle
(src/Metrics.cpp:132-151,160-235,265-337), in every workgroup class incl. the in-memory mode."""
import numpy as np
import pytest

from rnaseqc_amd import abi, synth
from tests import cases, hostemu
from tests.compare import FLOAT_ATOL, FLOAT_RTOL


def _compare(got, want):
    assert got.rc == 0
    np.testing.assert_array_equal(got.gene_cov_valid, want.gene_cov_valid)
    v = want.gene_cov_valid.astype(bool)
    np.testing.assert_allclose(got.gene_cov_mean[v], want.gene_cov_mean[v], rtol=FLOAT_RTOL, atol=FLOAT_ATOL)
    np.testing.assert_allclose(got.gene_cov_std[v], want.gene_cov_std[v], rtol=FLOAT_RTOL, atol=FLOAT_ATOL)
    nan_g, nan_w = np.isnan(got.gene_cov_cv[v]), np.isnan(want.gene_cov_cv[v])
    np.testing.assert_array_equal(nan_g, nan_w)
    np.testing.assert_allclose(got.gene_cov_cv[v][~nan_w], want.gene_cov_cv[v][~nan_w], rtol=1e-8, atol=FLOAT_ATOL)
    np.testing.assert_array_equal(got.exon_cv_valid, want.exon_cv_valid)
    ev = want.exon_cv_valid.astype(bool)
    np.testing.assert_allclose(got.exon_cv[ev], want.exon_cv[ev], rtol=1e-8, atol=FLOAT_ATOL)
    np.testing.assert_array_equal(got.bias_three, want.bias_three)
    np.testing.assert_array_equal(got.bias_five, want.bias_five)


def _pass(p, ann, batch):
    ref = hostemu.run(p, ann, batch, mode=1, want_cov=True)          # the per-record code on the host: difference array + gene counts
    return ref.cov, ref.gene_reads


@pytest.mark.parametrize("kw", [dict(), dict(coverage_mask=100), dict(coverage_mask=0, bias_window=50, bias_offset=10)])
def test_shallow_coverage_every_class(oracle_lib, kw):
    ann = synth.make_annotation(seed=3, contigs=[("chrA", 1_500_000, 110), ("chrB", 800_000, 50)])
    batch = synth.make_reads(ann, 5000, seed=4, contig_lengths=np.array([1_500_000, 800_000]))
    p = abi.default_params(**kw)
    want = oracle_lib.run_oracle(p, ann, [batch])
    assert want.gene_cov_valid.sum() > 50 and want.exon_cv_valid.sum() > 100
    cov, gr = _pass(p, ann, batch)
    for force in ((0, 2, 3, 4) if not kw else (0, 4)):              # the library's classes; 1024 / 256 threads and one wave for every gene
        _compare(hostemu.run_k3(p, ann, cov, gr, force=force), want)


@pytest.mark.parametrize("kw", [dict(), dict(bias_window=100, bias_offset=25, bias_gene_length=700), dict(coverage_mask=50, bias_window=120)])
def test_deep_coverage_bias_path(oracle_lib, kw):
    """Few genes, most reads on them: depth in the hundreds, so the bias gate (>= 100) opens: radix select, trim, window medians."""
    ann = synth.make_annotation(seed=8, contigs=[("chrA", 400_000, 40)])
    batch = synth.make_reads(ann, 60000, seed=9, frac=(0.97, 0.01, 0.01, 0.01), expr_sigma=1.0, contig_lengths=np.array([400_000]))
    p = abi.default_params(**kw)
    want = oracle_lib.run_oracle(p, ann, [batch])
    assert int(((want.bias_three + want.bias_five) > 0).sum()) >= 1
    cov, gr = _pass(p, ann, batch)
    for force in (0, 1, 4):
        _compare(hostemu.run_k3(p, ann, cov, gr, force=force), want)


def test_depth_beyond_16_bits_takes_the_in_memory_mode(oracle_lib):
    """The 1024-thread instances keep 16-bit depths in LDS; a base covered >= 65 536 times sends the gene through the in-memory
    mode of the same kernel (one deep and one shallow gene of 20 kb coding length)."""
    from rnaseqc_amd.model import Annotation, Batch
    rows = []
    for gid, base in (("deep", 10_000), ("shallow", 200_000)):
        rows.append(dict(contig="c", type="gene", start=base, end=base + 60_000, strand="+", gene_id=gid, gene_name=gid))
        for k in range(5):
            rows.append(dict(contig="c", type="exon", start=base + k * 10_000, end=base + k * 10_000 + 3_999, strand="+", gene_id=gid, exon_id="%s_e%d" % (gid, k)))
    ann = Annotation.from_rows(["c"], rows)
    n_deep = 66_000
    n = n_deep + 2_000
    pos = np.concatenate([np.full(n_deep, 10_000 + 10_100, np.int32) - 1,
                          (200_000 + np.sort(np.random.default_rng(5).integers(0, 3_900, 2_000))).astype(np.int32) - 1])
    qh = abi.qname_hash_bytes(np.frombuffer(b"".join(b"%015d" % i for i in range(n)), np.uint8).reshape(n, 15))
    batch = Batch(pos=pos, mpos=pos.copy(), isize=np.zeros(n, np.int32), qhash=qh, cigar_off=np.arange(n, dtype=np.uint32),
                  flag=np.zeros(n, np.uint16), l_qseq=np.full(n, 100, np.uint16), mapq=np.full(n, 255, np.uint8),
                  nm=np.zeros(n, np.uint8), tagbits=np.full(n, abi.TB_HAS_NM | abi.TB_MTID_SAME, np.uint8),
                  n_cigar=np.ones(n, np.uint8), cigar=np.full(n, (100 << 4) | abi.CIG_M, np.uint32),
                  seg_tid=np.array([0], np.int32), seg_start=np.array([0, n], np.uint64),
                  wide_index=np.zeros(0, np.uint64), wide_nm=np.zeros(0, np.int32), wide_l_qseq=np.zeros(0, np.int32),
                  wide_n_cigar=np.zeros(0, np.uint32))
    p = abi.default_params(unpaired=1)
    want = oracle_lib.run_oracle(p, ann, [batch])
    cov, gr = _pass(p, ann, batch)
    got = hostemu.run_k3(p, ann, cov, gr, force=0)
    _compare(got, want)
    assert got.classes[1] == 2 and got.gene_cov_mean[0] > 100      # both genes in the 1024-thread / 64 KB instance


def test_gene_with_more_exons_than_lanes(oracle_lib):
    """150 short exons in one gene: a wave's exon rows are gathered 64 at a time (one-wave class: three rounds; 256 threads: one
    round of 38 per wave), plus a second, ordinary gene."""
    ann, batch = cases.many_exon_case()
    p = abi.default_params(unpaired=1, coverage_mask=0)
    want = oracle_lib.run_oracle(p, ann, [batch])
    assert want.exon_cv_valid.sum() >= 140
    cov, gr = _pass(p, ann, batch)
    for force in (0, 3, 4, 2):
        _compare(hostemu.run_k3(p, ann, cov, gr, force=force), want)


def test_chained_with_the_emulated_per_read_kernels(oracle_lib):
    """The difference array and the gene counts come from the EMULATED per-read kernels (classify_ei_kernel + classify_slow_kernel,
    tests/hostemu/k1_emu.cpp) instead of the per-record host code: per-read kernels -> coverage kernel, all on the fiber emulation."""
    ann = synth.make_annotation(seed=8, contigs=[("chrA", 400_000, 40)])
    batch = synth.make_reads(ann, 30000, seed=9, frac=(0.97, 0.01, 0.01, 0.01), expr_sigma=1.0, contig_lengths=np.array([400_000]))
    p = abi.default_params()
    want = oracle_lib.run_oracle(p, ann, [batch])
    k1 = hostemu.run_k1(p, ann, batch, grid=3, want_cov=True)
    np.testing.assert_array_equal(k1.gene_reads, want.gene_reads)
    _compare(hostemu.run_k3(p, ann, k1.cov, k1.gene_reads, force=0), want)
