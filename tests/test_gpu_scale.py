"""Whole-boundary checks against the oracle at full size (GPU): BED intervals, three host-fed batches that do not end on
tile borders, hot genes spanning hundreds of fragment partitions, the longest genes of a GENCODE-sized annotation."""
import numpy as np
import pytest

from rnaseqc_amd import abi, engine, synth
from tests.compare import assert_results_match

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("genome", [False, True])
def test_ten_million_records_with_bed(oracle_lib, genome):
    ann = synth.make_annotation(seed=1, contigs=synth.human_contigs() if genome else None)
    bed = synth.make_bed(ann)
    batch = synth.make_reads(ann, 5_000_000 if not genome else 2_500_000, seed=2)
    cuts = [0, batch.n // 3 + 5, 2 * batch.n // 3 + 11, batch.n]
    parts = [batch.slice(cuts[k], cuts[k + 1]) for k in range(3)]
    p = abi.default_params()
    got = engine.run_engine(p, ann, parts, bed=bed)
    want = oracle_lib.run_oracle(p, ann, parts, bed=bed)
    assert_results_match(got, want)
    assert len(got.fragment_size) > 100 and got.counter("Total Alignments") == batch.n
