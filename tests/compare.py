"""Result comparison shared by the CPU and GPU parity tests.

Bar (BASELINE.json north_star): integer outputs bit-exact; floating outputs within 1e-6
(the reference's own test tolerance, test_data/approx_diff.py:47).  We hold the GPU to a
tighter relative tolerance where the only difference is summation order.
"""
import numpy as np

from rnaseqc_amd import abi

FLOAT_ATOL = 1e-6      # the tolerance north_star states for floating ratios
FLOAT_RTOL = 1e-9      # what we actually expect from re-ordered f64 sums


def assert_results_match(got, want, check_coverage=True, check_fragments=True):
    for i, n in enumerate(abi.COUNTER_NAMES):
        assert int(got.counters[i]) == int(want.counters[i]), (n, int(got.counters[i]), int(want.counters[i]))
    np.testing.assert_array_equal(got.gene_reads, want.gene_reads)
    np.testing.assert_array_equal(got.gene_unique, want.gene_unique)
    np.testing.assert_array_equal(got.gene_fragments, want.gene_fragments)
    np.testing.assert_array_equal(got.exon_hit, want.exon_hit)
    np.testing.assert_allclose(got.exon_reads, want.exon_reads, rtol=FLOAT_RTOL, atol=FLOAT_ATOL)
    assert got.read_length == want.read_length
    if check_coverage:
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
    if check_fragments:
        np.testing.assert_array_equal(got.fragment_size, want.fragment_size)
        np.testing.assert_array_equal(got.fragment_count, want.fragment_count)
    # --fasta: the histogram is integer work; exon GC values come out of the same sequence of additions (bit-exact)
    assert got.have_reference == want.have_reference
    if want.have_reference:
        np.testing.assert_array_equal(got.gc_bins, want.gc_bins)
        assert got.gc_out_of_range == want.gc_out_of_range
        ev = want.exon_cv_valid.astype(bool)
        np.testing.assert_array_equal(got.exon_gc[ev], want.exon_gc[ev])
