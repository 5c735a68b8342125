"""-m gpu: the device-side BGZF inflate on what other writers produce -- stored blocks (samtools -u / level 0), zlib's
default and best levels (longer matches, more long codes, several DEFLATE blocks per BGZF block).  The other GPU tests and the
bench write level 1.  (Named to run last: it covers input shapes, not a row of the contract.)"""
import numpy as np
import pytest

from rnaseqc_amd import abi, bamio, engine, synth
from tests.compare import assert_results_match
from tests.test_gpu_decode import check_columns, decode_file

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("level", [0, 6, 9])
def test_decode_other_compression_levels(tmp_path, level):
    contigs = [("chrA", 3_000_000), ("chrB", 1_000_000), ("chrC", 500_000)]
    ann = synth.make_annotation(seed=35, contigs=[("chrA", 3_000_000, 120), ("chrB", 1_000_000, 40), ("chrC", 500_000, 10)])
    batch = synth.make_reads(ann, 30_000, seed=37 + level, keep_qnames=True, chimeric_tag_frac=0.02, filter_tag_frac=0.03,
                             contig_lengths=np.array([3_000_000, 1_000_000, 500_000]))
    path = str(tmp_path / "l.bam")
    bamio.write_bam(path, contigs, batch, level=level)
    p = abi.default_params(); p.n_filter_tags = 1
    e = engine.Engine(p)
    e.set_annotation(ann)
    parts, runs, total, info, n_calls = decode_file(e, path, 3, "ch", ("XF",), 1 << 20, 1 << 40)
    assert total == batch.n and info[0] == batch.n and not info[1] and info[2] == 0
    check_columns(parts, batch)
    got = e.finalize()
    e.close()
    assert_results_match(got, engine.run_engine(p, ann, [batch]))
