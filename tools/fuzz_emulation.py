#!/usr/bin/env python
"""Offline fuzz of the device kernels under the 64-lane host emulation (tests/hostemu/wavemu.h) -- no GPU needed.
  python tools/fuzz_emulation.py k1-hostile <seed0> <n>   hostile / stacked annotations (tests/test_legacy_rules.py), random parameters
  python tools/fuzz_emulation.py k1-dense   <seed>  <n>   random contigs / genes / read sets of the synthetic generator, 1-8 workgroups
  python tools/fuzz_emulation.py k4         <seed>  <n>   random (gene, name) pair streams: chunk and dense-list form
Every case compares the UNMODIFIED kernel source (rsqc_k1.h / rsqc_k4.h) with the oracle (K1: counters, gene tables, exon values,
Read Length, the coverage difference array) or with a std::set of names per gene (K4).  Round 3: 3 000 + 2 500 + 400 cases, 0 mismatches; round 4 (lane-mask gate, 96-bit name identity): see profiles/r4_fuzz_emulation.txt."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from rnaseqc_amd import abi, synth
from tests import hostemu


def compare(o, r, cov):
    bad = [n for i, n in enumerate(abi.COUNTER_NAMES) if int(o.counters[i]) != int(r.counters[i])]
    for f in ("gene_reads", "gene_unique", "gene_fragments"):
        if not np.array_equal(getattr(o, f), getattr(r, f)): bad.append(f)
    if not np.allclose(o.exon_reads, r.exon_reads, rtol=0, atol=1e-9): bad.append("exon_reads")
    if o.read_length != r.read_length: bad.append("read_length")
    if cov is not None and not np.array_equal(o.cov, cov): bad.append("cov")
    return bad


def random_params(rng, tags=False):
    kw = dict(mapq_threshold=int(rng.integers(0, 10)))
    st = int(rng.integers(0, 3))
    if st == 1: kw["stranded"] = abi.STRAND_FORWARD
    if st == 2: kw["stranded"] = abi.STRAND_REVERSE
    if rng.random() < 0.3: kw["unpaired"] = 1
    if rng.random() < 0.3: kw["exclude_chimeric"] = 1
    if rng.random() < 0.3: kw["base_mismatch"] = int(rng.integers(0, 4))
    if tags and rng.random() < 0.2: kw["n_filter_tags"] = 1
    return kw


def one_k1(ann, batch, kw, grid):
    from oracle import binding
    p = abi.default_params(**kw)
    r = binding.run_oracle(p, ann, [batch])
    ref = hostemu.run(p, ann, batch, mode=1, want_cov=True)
    return compare(hostemu.run_k1(p, ann, batch, grid=grid, want_cov=True), r, ref.cov)


def main():
    what, seed, n = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
    rng = np.random.default_rng(seed)
    fails = done = refused = 0
    t0 = time.time()
    if what == "k1-hostile":
        from tests.test_legacy_rules import hostile_case, stacked_case
        for s in range(seed, seed + n):
            for maker in (hostile_case, stacked_case):
                ann, batch = maker(s)
                kw, grid = random_params(rng), int(rng.integers(1, 5))
                bad = one_k1(ann, batch, kw, grid); done += 1
                if bad: fails += 1; print("MISMATCH", s, maker.__name__, kw, grid, bad, flush=True)
    elif what == "k1-dense":
        for it in range(n):
            contigs = [("c%d" % i, int(rng.integers(200_000, 3_000_000)), int(rng.integers(0, 300))) for i in range(int(rng.integers(1, 5)))]
            if all(c[2] == 0 for c in contigs): contigs[0] = (contigs[0][0], contigs[0][1], 50)
            try:
                ann = synth.make_annotation(seed=int(rng.integers(1, 10**6)), contigs=contigs)
                batch = synth.make_reads(ann, int(rng.integers(50, 9000)), seed=int(rng.integers(1, 10**6)), dup_frac=float(rng.random() * 0.3),
                                         chimeric_tag_frac=0.01, filter_tag_frac=0.02, contig_lengths=np.array([c[1] for c in contigs]),
                                         read_len=int(rng.choice([100, 150])))
            except (ValueError, OverflowError):
                continue                                   # (the generator refuses some shapes: contig too small for its genes, ...)
            kw, grid = random_params(rng, tags=True), int(rng.integers(1, 9))
            bad = one_k1(ann, batch, kw, grid); done += 1
            if bad: fails += 1; print("MISMATCH", it, contigs, kw, grid, bad, flush=True)
    elif what == "k4":
        for it in range(n):
            s = int(rng.integers(1, 2**40)); G = int(rng.integers(2, 600)); arena = bool(rng.random() < 0.4)
            nch = 0 if arena else int(rng.integers(1, 12)); names = int(rng.integers(0, 40000)); hot = int(rng.choice([0, 0, 500, 3000, 20000, 50000]))
            wide = bool(rng.random() < 0.5)                # second name hashes, some names sharing their 64-bit key
            rc, st = hostemu.run_k4(s, G, nch, names, hot, arena, wide=wide); done += 1
            # (the generator makes one name in 53 share its 64-bit key with another of its gene: a partition that collects more than 32
            #  such entries reports RSQC_ERR_CAPACITY by design -- a refusal, not a miscount; real names never get near it)
            if wide and rc == 4: refused += 1
            elif rc != 0: fails += 1; print("MISMATCH", s, G, nch, names, hot, arena, wide, rc, st, flush=True)
    else:
        sys.exit(__doc__)
    print("%s: %d cases, %d mismatches%s, %.0f s" % (what, done, fails, (", %d capacity refusals of the overflow list" % refused) if refused else "", time.time() - t0))
    sys.exit(1 if fails else 0)


if __name__ == "__main__":
    main()
