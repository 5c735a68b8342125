"""CPU: the C-ABI library loads, exports every symbol include/rnaseqc_amd.h declares, and
refuses to run without a GPU (there is no CPU path)."""
import ctypes as C
import os
import re
import subprocess

import pytest

from rnaseqc_amd import abi, engine

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_header_symbols_are_exported():
    lib = engine.load_library()
    hdr = open(os.path.join(ROOT, "include", "rnaseqc_amd.h")).read()
    declared = sorted(set(re.findall(r"RSQC_API[^;(]*?\b(rsqc_\w+)\s*\(", hdr)))
    assert declared == sorted(engine.EXPORTED_SYMBOLS)
    for sym in declared:
        assert hasattr(lib, sym), sym


def test_struct_sizes_match_header():
    # compile a tiny C program against the header and compare sizeof with the ctypes mirrors
    import subprocess, tempfile
    src = r'''
    #include <stdio.h>
    #include "rnaseqc_amd.h"
    int main(void){ printf("%zu %zu %zu %zu %zu %zu %d %zu %zu %zu %zu\n", sizeof(rsqc_params), sizeof(rsqc_annotation), sizeof(rsqc_bed),
      sizeof(rsqc_batch), sizeof(rsqc_results), sizeof(rsqc_timing), (int)RSQC_N_COUNTERS,
      sizeof(rsqc_bgzf_block), sizeof(rsqc_decode_params), sizeof(rsqc_decode_window), sizeof(rsqc_decode_info)); return 0; }'''
    with tempfile.TemporaryDirectory() as d:
        open(os.path.join(d, "s.c"), "w").write(src)
        subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), os.path.join(d, "s.c"), "-o", os.path.join(d, "s")])
        out = subprocess.check_output([os.path.join(d, "s")]).decode().split()
    sizes = [int(x) for x in out]
    assert sizes[:6] == [C.sizeof(abi.Params), C.sizeof(abi.AnnotationStruct), C.sizeof(abi.BedStruct),
                         C.sizeof(abi.BatchStruct), C.sizeof(abi.ResultsStruct), C.sizeof(abi.TimingStruct)]
    assert sizes[6] == abi.N_COUNTERS
    assert sizes[7:11] == [C.sizeof(abi.BgzfBlock), C.sizeof(abi.DecodeParams), C.sizeof(abi.DecodeWindow), C.sizeof(abi.DecodeInfo)]


def test_counter_names_and_version():
    lib = engine.load_library()
    for i, n in enumerate(abi.COUNTER_NAMES):
        assert lib.rsqc_counter_name(i).decode() == n
    assert lib.rsqc_version().decode().startswith("RNASeQC 2")      # python/rnaseqc/run.py:25
    for name in [b"", b"a", b"SYN:000000000042", b"HWI-ST1234:100:C0ABCACXX:1:1101:1234:5678"]:
        assert lib.rsqc_qname_hash(name, len(name)) == abi.qname_hash(name)
        assert lib.rsqc_qname_hash2(name, len(name)) == abi.qname_hash2(name)


def test_no_cpu_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(engine.EngineError) as e:
        engine.Engine(abi.default_params())
    assert e.value.code == abi.ERR_NO_DEVICE


def test_product_library_has_no_skip_knobs_or_debug_exports():
    """The shipped library cannot be told to skip work and exports nothing outside include/rnaseqc_amd.h: the ablation masks,
    RSQC_DIAG_* / RSQC_K3_FORCE environment knobs and rsqc_debug_* functions exist only in the diagnostic build (`make prof`)."""
    path = os.path.join(ROOT, "rnaseqc_amd", "lib", "librnaseqc_amd.so")
    blob = open(path, "rb").read()
    for needle in (b"RSQC_DEBUG_MASK", b"RSQC_DIAG_", b"RSQC_K3_FORCE", b"rsqc_debug_"):
        assert needle not in blob, needle
    exported = subprocess.check_output(["nm", "-D", "--defined-only", path]).decode()
    names = {line.split()[-1] for line in exported.splitlines() if " T " in line}
    declared = set(re.findall(r"\b(rsqc_[a-z0-9_]+)\s*\(", open(os.path.join(ROOT, "include", "rnaseqc_amd.h")).read()))
    assert {n for n in names if n.startswith("rsqc_")} <= declared, sorted(n for n in names if n.startswith("rsqc_") and n not in declared)
