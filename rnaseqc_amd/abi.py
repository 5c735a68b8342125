"""ctypes mirror of include/rnaseqc_amd.h (the C ABI of the hot path).

Nothing here computes: it only describes the structs and turns numpy arrays
into the plain pointers the boundary takes.  Field order must match the header.
"""
from __future__ import annotations

import ctypes as C
import numpy as np

ABI_VERSION = 5

# error codes
OK, ERR_ARG, ERR_HIP, ERR_BAD_CIGAR, ERR_CAPACITY, ERR_NO_DEVICE, ERR_EMPTY_MEDIAN, ERR_INPUT = 0, -1, -2, -3, -4, -5, -6, -7

# BAM flags
FPAIRED, FPROPER, FUNMAP, FMUNMAP, FREVERSE, FMREVERSE = 0x1, 0x2, 0x4, 0x8, 0x10, 0x20
FREAD1, FREAD2, FSECONDARY, FQCFAIL, FDUP, FSUPP = 0x40, 0x80, 0x100, 0x200, 0x400, 0x800

TB_HAS_NM, TB_HAS_CH, TB_MTID_SAME, TB_FILTER0 = 0x01, 0x02, 0x04, 0x08
MAX_FILTER_TAGS = 5
NM_ESCAPE, LQSEQ_ESCAPE, NCIGAR_ESCAPE = 0xFF, 0xFFFF, 0xFF

STRAND_FORWARD, STRAND_REVERSE, STRAND_UNKNOWN = 0, 1, 2
FF_STRAND_MASK, FF_RIBOSOMAL = 0x03, 0x04

# CIGAR op codes (BAM): MIDNSHP=XB
CIG_M, CIG_I, CIG_D, CIG_N, CIG_S, CIG_H, CIG_P, CIG_EQ, CIG_X, CIG_B = range(10)

COUNTER_NAMES = [
    "Alternative Alignments", "Supplementary Alignments", "Failed Vendor QC", "Low Mapping Quality",
    "Chimeric Fragments_auto", "Chimeric Fragments_tag", "Unique Mapping, Vendor QC Passed Reads",
    "Unpaired Reads", "Mapped Reads", "Mapped Duplicate Reads", "Mapped Unique Reads",
    "Total Mapped Pairs", "End 1 Mapped Reads", "End 1 Mismatches", "End 1 Bases", "Duplicate Pairs",
    "Unique Fragments", "End 2 Mapped Reads", "End 2 Mismatches", "End 2 Bases", "Mismatched Bases",
    "Total Bases", "High Quality Reads", "Low Quality Reads", "Reads used for Intron/Exon counts",
    "Alignment Blocks", "Non-Globin Reads", "Non-Globin Duplicate Reads", "Intronic Reads",
    "Intragenic Reads", "HQ Intronic Reads", "HQ Intragenic Reads", "Intergenic Reads",
    "HQ Intergenic Reads", "Exonic Reads", "HQ Exonic Reads", "Ambiguous Reads", "HQ Ambiguous Reads",
    "rRNA Reads", "End 1 Sense", "End 1 Antisense", "End 2 Sense", "End 2 Antisense",
    "Total Alignments", "Filtered by tag: 0", "Filtered by tag: 1", "Filtered by tag: 2",
    "Filtered by tag: 3", "Filtered by tag: 4", "Split Reads",
]
N_COUNTERS = len(COUNTER_NAMES)
COUNTER_INDEX = {n: i for i, n in enumerate(COUNTER_NAMES)}

_P = C.c_void_p

# 16-byte half-records of rsqc_batch (rsqc_rec_core / rsqc_rec_aux)
REC_CORE = np.dtype([("pos", "<i4"), ("mpos", "<i4"), ("isize", "<i4"), ("cigar_off", "<u4")])
REC_AUX = np.dtype([("qhash", "<u8"), ("flag", "<u2"), ("l_qseq", "<u2"), ("mapq", "u1"), ("nm", "u1"),
                    ("tagbits", "u1"), ("n_cigar", "u1")])
assert REC_CORE.itemsize == 16 and REC_AUX.itemsize == 16


class ShardInfoStruct(C.Structure):
    _fields_ = [("n_batches", C.c_uint32), ("batch_file_index", C.c_void_p), ("batch_records", C.c_void_p),
                ("rl_offset", C.c_void_p), ("rl_span", C.c_void_p), ("rl_state", C.c_void_p),
                ("n_samples", C.c_uint32), ("sample_file_index", C.c_void_p), ("sample_size", C.c_void_p)]


class DeviceRange(C.Structure):
    _fields_ = [("base", C.c_void_p), ("count", C.c_uint64)]


class Params(C.Structure):
    _fields_ = [
        ("abi_version", C.c_uint32), ("device", C.c_int32),
        ("mapq_threshold", C.c_uint32), ("base_mismatch", C.c_uint32),
        ("chimeric_distance", C.c_int32), ("fragment_samples", C.c_uint32),
        ("bias_offset", C.c_int32), ("bias_window", C.c_int32),
        ("bias_gene_length", C.c_uint64), ("coverage_mask", C.c_uint32),
        ("stranded", C.c_int32), ("unpaired", C.c_int32), ("exclude_chimeric", C.c_int32),
        ("n_filter_tags", C.c_int32), ("legacy", C.c_int32), ("reserved", C.c_int32 * 6),
    ]


def default_params(**kw) -> Params:
    """Reference defaults, src/RNASeQC.cpp:87-100."""
    p = Params()
    p.abi_version = ABI_VERSION
    p.device = 0
    p.mapq_threshold = 255
    p.base_mismatch = 6
    p.chimeric_distance = 2000000
    p.fragment_samples = 1000000
    p.bias_offset = 0
    p.bias_window = 100
    p.bias_gene_length = 200
    p.coverage_mask = 500
    p.stranded = STRAND_UNKNOWN
    p.unpaired = 0
    p.exclude_chimeric = 0
    p.n_filter_tags = 0
    p.legacy = 0
    for k, v in kw.items():
        if not hasattr(p, k):
            raise AttributeError(k)
        setattr(p, k, v)
    return p


class AnnotationStruct(C.Structure):
    _fields_ = [
        ("n_ref", C.c_int32), ("n_contigs", C.c_int32),
        ("n_genes", C.c_int32), ("n_genes_listed", C.c_int32), ("n_exons", C.c_int32),
        ("gene_row_contig", _P), ("gene_row_start", _P), ("gene_row_end", _P),
        ("gene_row_flags", _P), ("gene_row_id", _P),
        ("exon_row_contig", _P), ("exon_row_start", _P), ("exon_row_end", _P),
        ("exon_row_flags", _P), ("exon_row_id", _P), ("exon_row_gene", _P),
        ("gene_is_globin", _P), ("gene_exon_off", _P), ("gene_exon_row", _P),
        ("gene_row_order", _P), ("exon_row_order", _P),      # optional, legacy rules only
    ]


class BedStruct(C.Structure):
    _fields_ = [("n_intervals", C.c_int32), ("contig", _P), ("start", _P), ("end", _P)]


class BatchStruct(C.Structure):
    _fields_ = [
        ("n", C.c_uint64), ("file_index_base", C.c_uint64),
        ("core", _P), ("aux", _P),
        ("cigar", _P), ("n_cigar_total", C.c_uint64),
        ("n_seg", C.c_uint32), ("seg_tid", _P), ("seg_start", _P),
        ("n_wide", C.c_uint32), ("wide_index", _P), ("wide_nm", _P), ("wide_l_qseq", _P),
        ("wide_n_cigar", _P),
        ("qname_off", _P), ("qname", _P),
        ("qhash2", _P),
        ("seg_file_index", _P),
    ]


BGZF_INFLATED = 1


class BgzfBlock(C.Structure):
    _fields_ = [("in_offset", C.c_uint64), ("in_bytes", C.c_uint32), ("out_bytes", C.c_uint32), ("crc32", C.c_uint32), ("flags", C.c_uint32)]


class DecodeParams(C.Structure):
    _fields_ = [("n_ref", C.c_int32), ("has_chimeric_tag", C.c_int32), ("chimeric_tag", C.c_char * 2),
                ("filter_tag", (C.c_char * 2) * MAX_FILTER_TAGS), ("file_index_base", C.c_uint64), ("pipelined", C.c_int32), ("reserved", C.c_int32),
                ("reserve_inflated_bytes", C.c_uint64)]


class DecodeWindow(C.Structure):
    _fields_ = [("n_records", C.c_uint64), ("n_runs", C.c_uint32), ("run_tid", _P), ("device_batch", BatchStruct)]


class DecodeInfo(C.Structure):
    _fields_ = [("last", DecodeWindow), ("records", C.c_uint64), ("unsorted", C.c_int32), ("n_bad_refid", C.c_int32), ("bad_refid", _P)]


class ResultsStruct(C.Structure):
    _fields_ = [
        ("n_genes_listed", C.c_int32), ("n_exons", C.c_int32),
        ("gene_reads", _P), ("gene_unique", _P), ("gene_fragments", _P),
        ("exon_reads", _P), ("exon_hit", _P),
        ("counters", C.c_uint64 * N_COUNTERS),
        ("read_length", C.c_int32),
        ("gene_cov_mean", _P), ("gene_cov_std", _P), ("gene_cov_cv", _P), ("gene_cov_valid", _P),
        ("exon_cv", _P), ("exon_cv_valid", _P),
        ("bias_three", _P), ("bias_five", _P),
        ("n_fragment_sizes", C.c_uint32), ("fragment_size", _P), ("fragment_count", _P),
        ("fragment_samples_remaining", C.c_uint32),
        ("have_reference", C.c_int32), ("gc_bins", _P), ("gc_out_of_range", C.c_uint64), ("exon_gc", _P),
        ("exons_outside_gene_row", C.c_uint32),
    ]


class ReferenceStruct(C.Structure):
    _fields_ = [("n", C.c_int32), ("contig", _P), ("length", _P), ("sequence", _P)]


GC_BINS = 100


class TimingStruct(C.Structure):
    _fields_ = [
        ("classify_ms", C.c_double), ("classify_launches", C.c_uint64),
        ("classify_records", C.c_uint64), ("classify_bytes", C.c_uint64),
        ("finalize_ms", C.c_double), ("h2d_ms", C.c_double), ("slow_records", C.c_uint64), ("fragment_sizes_ms", C.c_double),
        ("classify_long_ms", C.c_double),
    ]


def ptr(a: np.ndarray | None):
    """numpy array -> void* (None -> NULL).  The caller keeps `a` alive."""
    if a is None:
        return None
    assert a.flags["C_CONTIGUOUS"], "boundary arrays must be contiguous"
    return a.ctypes.data_as(C.c_void_p)


def _view(p, n, dtype):
    if not p or n == 0:
        return np.zeros(0, dtype=dtype)
    buf = (C.c_char * (n * np.dtype(dtype).itemsize)).from_address(p)
    return np.frombuffer(buf, dtype=dtype, count=n).copy()


class Results:
    """Host view of an rsqc_results struct.  The vectors live in the library's host buffers (already on the
    host after rsqc_finalize); each is copied out the first time it is read, so holding a Results costs nothing
    until it is inspected.  Call materialise() to detach everything before the next finalize/reset of the context
    (Engine.finalize does that unless lazy=True)."""

    _VEC = {  # name -> (struct field, length selector, dtype)
        "gene_reads": ("gene_reads", "G", np.uint64), "gene_unique": ("gene_unique", "G", np.uint64),
        "gene_fragments": ("gene_fragments", "G", np.uint64), "exon_reads": ("exon_reads", "E", np.float64),
        "exon_hit": ("exon_hit", "E", np.uint8), "gene_cov_mean": ("gene_cov_mean", "G", np.float64),
        "gene_cov_std": ("gene_cov_std", "G", np.float64), "gene_cov_cv": ("gene_cov_cv", "G", np.float64),
        "gene_cov_valid": ("gene_cov_valid", "G", np.uint8), "exon_cv": ("exon_cv", "E", np.float64),
        "exon_cv_valid": ("exon_cv_valid", "E", np.uint8), "bias_three": ("bias_three", "G", np.uint64),
        "bias_five": ("bias_five", "G", np.uint64), "fragment_size": ("fragment_size", "F", np.int64),
        "fragment_count": ("fragment_count", "F", np.uint64),
        "gc_bins": ("gc_bins", "B", np.uint64), "exon_gc": ("exon_gc", "R", np.float64),
    }

    def __init__(self, rs: ResultsStruct):
        self._rs = rs
        self.have_reference = int(rs.have_reference)
        self.gc_out_of_range = int(rs.gc_out_of_range)
        self.exons_outside_gene_row = int(rs.exons_outside_gene_row)     # (ABI 5: non-zero flags the divergence DESIGN.md 5 describes)
        self._n = {"G": rs.n_genes_listed, "E": rs.n_exons, "F": rs.n_fragment_sizes,
                   "B": GC_BINS if rs.have_reference else 0, "R": rs.n_exons if rs.have_reference else 0}
        self.read_length = int(rs.read_length)
        self.fragment_samples_remaining = int(rs.fragment_samples_remaining)

    def __getattr__(self, name):
        # only called for attributes not materialised yet
        if name == "counters":
            v = np.frombuffer(self._rs.counters, dtype=np.uint64, count=N_COUNTERS).copy()
        elif name in Results._VEC:
            field, sel, dt = Results._VEC[name]
            v = _view(getattr(self._rs, field), self._n[sel], dt)
        else:
            raise AttributeError(name)
        self.__dict__[name] = v
        return v

    def materialise(self) -> "Results":
        for name in list(Results._VEC) + ["counters"]:
            getattr(self, name)
        return self

    def counter(self, name: str) -> int:
        return int(self.counters[COUNTER_INDEX[name]])

    def counter_dict(self) -> dict:
        return {n: int(self.counters[i]) for i, n in enumerate(COUNTER_NAMES)}


def qname_hash_bytes(names: np.ndarray) -> np.ndarray:
    """rsqc_qname_hash over a [n, width] uint8 matrix of fixed-width names
    (FNV-1a 64 followed by the murmur3 fmix64 finaliser), vectorised."""
    names = np.ascontiguousarray(names, dtype=np.uint8)
    h = np.full(names.shape[0], 0xCBF29CE484222325, dtype=np.uint64)
    prime = np.uint64(0x100000001B3)
    with np.errstate(over="ignore"):
        for j in range(names.shape[1]):
            h ^= names[:, j].astype(np.uint64)
            h *= prime
        h ^= h >> np.uint64(33)
        h *= np.uint64(0xFF51AFD7ED558CCD)
        h ^= h >> np.uint64(33)
        h *= np.uint64(0xC4CEB9FE1A85EC53)
        h ^= h >> np.uint64(33)
    return h


def qname_hash(name: bytes) -> int:
    return int(qname_hash_bytes(np.frombuffer(name, dtype=np.uint8)[None, :])[0])


def qname_hash2_bytes(names: np.ndarray) -> np.ndarray:
    """rsqc_qname_hash2 (the second, independent name hash: rsqc_batch.qhash2) over a [n, width] uint8 matrix, vectorised:
    per byte h = (h + b) * 0xCC9E2D51, h ^= h >> 15; then murmur3's fmix32 of (h ^ width)."""
    names = np.ascontiguousarray(names, dtype=np.uint8)
    M = np.uint64(0xFFFFFFFF)
    h = np.full(names.shape[0], 0x2F0B4A87, dtype=np.uint64)
    for j in range(names.shape[1]):
        h = ((h + names[:, j].astype(np.uint64)) * np.uint64(0xCC9E2D51)) & M
        h ^= h >> np.uint64(15)
    h ^= np.uint64(names.shape[1])
    h ^= h >> np.uint64(16); h = (h * np.uint64(0x85EBCA6B)) & M
    h ^= h >> np.uint64(13); h = (h * np.uint64(0xC2B2AE35)) & M
    h ^= h >> np.uint64(16)
    return h.astype(np.uint32)


def qname_hash2(name: bytes) -> int:
    return int(qname_hash2_bytes(np.frombuffer(name, dtype=np.uint8)[None, :])[0])
