"""No GPU needed: the fragment-counting kernels themselves (rnaseqc_amd/csrc/rsqc_k4.h, unmodified source: partition layout,
chunk-wide de-dup window, ranks + list reservations, the two counting instances) on the 64-lane SIMT emulation of
tests/hostemu/wavemu.h, against a std::set of names per gene (src/Expression.cpp:383-387: geneFragmentCounts = distinct read
names among the records counted to a gene)."""
import pytest

from tests import hostemu

CASES = [
    # seed, genes, chunks, names, records of the hot gene, arena form
    (1, 40, 6, 20000, 0, False),            # a handful of genes, one partition each or a few
    (2, 300, 5, 30000, 9000, False),        # + a gene with a dozen partitions
    (3, 40, 0, 25000, 6000, True),          # the dense list of retired batches
    (4, 2500, 9, 60000, 12000, False),      # more genes than one layout workgroup holds
    (5, 20, 4, 4000, 45000, False),         # partitions beyond 1024 keys: the second counting instance
    (6, 20, 0, 4000, 30000, True),          # the same through the dense list
    (7, 3, 2, 0, 0, False),                 # nothing counted at all
    (8, 2, 1, 300, 0, False),               # one gene, one short chunk
    (11, 20000, 6, 100000, 0, False),       # a pass touches ~1 700 partitions: the rank table runs crowded (direct reservations)
]


@pytest.mark.parametrize("seed,genes,chunks,names,hot,arena", CASES)
def test_fragment_kernels_against_name_sets(seed, genes, chunks, names, hot, arena):
    rc, st = hostemu.run_k4(seed, genes, chunks, names, hot, arena)
    assert rc == 0, (rc, st)
    assert st["kept"] <= st["pairs"] and st["kept"] >= st["distinct"]        # the window only ever drops repeats
    if hot >= 30000:
        assert st["fuller"] > 0, st                                         # the case does reach the 32 KB instance
    if names >= 20000 and not arena:
        assert st["kept"] < st["pairs"], st                                 # the window does find mates


@pytest.mark.parametrize("seed,genes,chunks,names,hot,arena", [CASES[0], CASES[1], CASES[2], CASES[4], CASES[5], CASES[7]])
def test_fragment_kernels_with_the_96_bit_identity(seed, genes, chunks, names, hot, arena):
    """The same with second name hashes (rsqc_batch.qhash2): one name in 53 shares its 64-bit key -- and its gene -- with an earlier,
    different name and must be counted on its own (window: both words compared; counting sets: the owner's second hash is compared
    after the barrier, unequal ones are set aside and counted by one thread)."""
    rc, st = hostemu.run_k4(seed + 100, genes, chunks, names, hot, arena, wide=True)
    assert rc == 0, (rc, st)
    assert st["kept"] <= st["pairs"] and st["kept"] >= st["distinct"]
