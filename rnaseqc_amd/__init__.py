"""rnaseqc_amd -- MI355X-native RNA-SeQC per-read hot path (host-side Python mirror).

The compute lives in rnaseqc_amd/csrc (HIP kernels behind the C ABI declared in
include/rnaseqc_amd.h); this package only describes the boundary (abi, model),
drives it (engine) and generates synthetic inputs (synth).
"""
from . import abi, model  # noqa: F401

__version__ = "0.1.0"
