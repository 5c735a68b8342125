#!/usr/bin/env python
"""bench.py -- reads/sec of the RNA-SeQC per-read hot path on MI355X.

One "step" is one complete pass of the hot path over one device-resident batch of
synthetic alignment records: zero the accumulators, K1 classify/count over every
record, the Read-Length scan, the end-of-file stage (fragment de-dup, coverage scan,
per-gene coverage statistics and bias windows), read-back of the result vectors and,
for N > 1, the RCCL sum-reduction of the count vectors.  Nothing is skipped or cached
between steps.

Workload (BASELINE.json configs[1]): chr1-like collapsed GTF (5 234 genes) + 10 M
synthetic 2x150 bp coordinate-sorted records, inputs resident in HBM before the timed
region.  With --gpus N each rank owns one chr1-like contig of an N-contig annotation
and processes its own 10 M records (weak scaling, sharded by contig; the only
exchange is the end-of-file reduction of the count vectors).

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0      # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--pairs", type=int, default=5_000_000, help="read pairs per GPU (2 records each)")
    ap.add_argument("--cpu-sample", type=int, default=10_000_000, help="records timed on the CPU oracle (0 = skip)")
    ap.add_argument("--no-finalize", action="store_true", help="(diagnostic) time K1 only; output marked invalid")
    ap.add_argument("--host-fed", action="store_true",
                    help="(diagnostic) every step uploads the batch from host memory through rsqc_submit: the "
                         "PCIe-inclusive rate noted in DESIGN.md; not the bench line")
    ap.add_argument("--dist-selftest", action="store_true",
                    help="(diagnostic) single process, but through the N > 1 code path: 1-rank RCCL group, finalize_device, "
                         "all_reduce of the device accumulators, refresh_results")
    ap.add_argument("--fasta", action="store_true", help="(diagnostic) with a reference sequence: GC statistics on; output marked invalid")
    ap.add_argument("--legacy", action="store_true", help="(diagnostic) the --legacy counting rules; output marked invalid")
    ap.add_argument("--genome", action="store_true",
                    help="BASELINE.json configs[2] shape on ONE GPU: GENCODE-sized annotation (25 contigs, 56 202 genes); "
                         "not the default bench line")
    args = ap.parse_args()

    # RCCL prints a version banner on stdout when the communicator comes up; the contract is ONE JSON line there,
    # so everything else this process (and the libraries it loads) writes to fd 1 is sent to stderr.
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    import torch
    from rnaseqc_amd import abi, engine, synth

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node %d" % args.gpus)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the hot path has no CPU fallback)")
    torch.cuda.set_device(local_rank)
    dist = None
    reduce_path = world > 1 or args.dist_selftest
    if reduce_path:
        import torch.distributed as dist_mod
        dist = dist_mod
        if world == 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
            os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))

    # ---- synthetic inputs: N chr1-like contigs, rank k owns contig k -----------------------------
    chr1 = synth.HUMAN_CONTIGS[0]
    contigs = [("chr1_%d" % k, chr1[1], chr1[2]) for k in range(world)] if world > 1 else [chr1]
    if args.genome:
        if world > 1:
            raise SystemExit("--genome is a single-GPU diagnostic")
        contigs = synth.human_contigs()
    ann = synth.make_annotation(seed=1, contigs=contigs)
    # every rank generates the records of ITS contig only (same annotation everywhere)
    t_gen = time.time()
    batch = synth.make_reads(ann, args.pairs, seed=2 + rank, only_contig=rank if world > 1 else None)
    t_gen = time.time() - t_gen

    p = abi.default_params(device=local_rank, **(dict(legacy=1, mapq_threshold=4) if args.legacy else {}))
    e = engine.Engine(p)
    owned = None
    if world > 1:
        owned = np.zeros(ann.n_contigs, np.uint8); owned[rank] = 1
    e.set_annotation(ann, owned)
    if args.fasta:
        t_ref = time.time()
        e.set_reference(synth.make_reference([c[1] for c in contigs], seed=7, uniform=True))
        t_ref = time.time() - t_ref
    h = e.upload(batch)                      # inputs resident in HBM before the timed region
    host_struct = None
    if args.host_fed:                        # the same batch, packed once, in page-locked host memory
        host_struct = batch.to_struct()
        keep = []
        for f, n_items, dt in (("core", batch.n, abi.REC_CORE), ("aux", batch.n, abi.REC_AUX), ("cigar", len(batch.cigar), np.uint32)):
            src = np.frombuffer((__import__("ctypes").c_char * (n_items * np.dtype(dt).itemsize)).from_address(getattr(host_struct, f)), dtype=dt, count=n_items)
            pin = e.pinned_copy(src); keep.append(pin)
            setattr(host_struct, f, pin.ctypes.data)

    u64_t = f64_t = None
    if reduce_path:
        u64_d, f64_d = e.device_accumulators()
        u64_t = torch.as_tensor(u64_d, device="cuda")
        f64_t = torch.as_tensor(f64_d, device="cuda")

    def step():
        e.reset()
        if args.host_fed:
            e.submit_struct(host_struct)     # H2D (DMA from page-locked memory) + K1, as the CLI does per batch
        else:
            e.submit_resident(h)
        if args.no_finalize:
            e.wait()
            return None
        if not reduce_path:
            return e.finalize(lazy=True)     # the vectors are on the host (library buffers); Python copies are made on access
        e.finalize_device()                  # results stay on the device until the counts are reduced
        dist.all_reduce(u64_t)               # RCCL over xGMI: gene reads/unique/fragments + scalar counters (as i64)
        dist.all_reduce(f64_t)               # exon fractions (torch's coalescing context needs one dtype: two launches)
        torch.cuda.synchronize()
        return e.refresh_results(lazy=True)

    for _ in range(args.warmup):
        step()
    e.reset_timing()
    if dist: dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    res = None
    for _ in range(args.steps):
        res = step()
    torch.cuda.synchronize()
    if dist: dist.barrier()
    elapsed = time.perf_counter() - t0
    if dist:
        t = torch.tensor([elapsed], device="cuda", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        nrec = torch.tensor([batch.n], device="cuda", dtype=torch.int64)
        dist.all_reduce(nrec)
        total_records = int(nrec.item())
    else:
        total_records = batch.n
    tm = e.timing()

    if rank == 0:
        # HBM traffic of K1 per launch comes from separate rocprofv3 --pmc passes of this same command
        # (FETCH_SIZE and WRITE_SIZE cannot share a pass); tools/pmc.sh stores them in profiles/k1_traffic.json
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "k1_traffic.json")
        if os.path.exists(tpath) and world == 1 and args.pairs == 5_000_000 and not args.genome:
            tj = json.load(open(tpath))
            traffic = tj.get("hbm_bytes_per_launch")
        k1_ms = tm["classify_ms"] / max(tm["classify_launches"], 1)
        bytes_per_launch = tm["classify_bytes"] / max(tm["classify_launches"], 1)
        achieved = bytes_per_launch / (k1_ms * 1e-3) / 1e9 if k1_ms > 0 else 0.0
        cpu = None
        if args.cpu_sample > 0 and world == 1:          # the CPU baseline is a single-GPU-run item (rank 0, N = 1)
            from oracle import binding
            ns = min(args.cpu_sample, batch.n)
            sample = batch.slice(0, ns) if ns < batch.n else batch
            o = binding.Oracle(abi.default_params(**(dict(legacy=1, mapq_threshold=4) if args.legacy else {})))
            o.set_annotation(ann)
            tc = time.perf_counter()
            o.submit(sample)
            o.finalize()
            tc = time.perf_counter() - tc
            o.close()
            cpu = {"value": ns / tc, "unit": "reads/s", "cores": 1, "kind": "port",
                   "sample": "first %d records of the rank-0 workload, oracle/rsqc_oracle.c (single thread, "
                             "SoA input, no BAM decode)" % ns, "seconds": round(tc, 3)}
        out = {
            "metric": "reads/sec whole-node (synthetic coordinate-sorted 2x150 records, collapsed GTF)",
            "value": total_records * args.steps / elapsed,
            "unit": "reads/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None,
            "dtype": "i32/u64 counters, f64 exon fractions",
            "data": "synthetic (seeded generator rnaseqc_amd/synth.py; no real GENCODE/BAM offline)",
            "config": {"workload": ("configs[2] shape: GENCODE-sized collapsed GTF (%d genes, %d exons, 25 contigs) + %d records, "
                                    "device-resident SoA, full pass incl. end-of-file stage" % (ann.n_genes, ann.n_exons, batch.n))
                                   if args.genome else
                                   "configs[1]: chr1-like collapsed GTF (%d genes, %d exons per contig) + %d records/GPU, "
                                   "device-resident SoA, full pass incl. end-of-file stage" %
                                   (chr1[2], ann.n_exons // max(world, 1), batch.n),
                       "records_per_gpu": int(batch.n), "contigs": world, "sharding": "by contig",
                       "collective": "RCCL all_reduce(sum) of u64[3G+%d] + f64[E] per step" % abi.N_COUNTERS if world > 1 else "none"},
            "roofline": {"bound": "hbm", "kernel": "classify_count_kernel_w4r1", "achieved": achieved, "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "algorithmic_bytes_per_launch": bytes_per_launch, "kernel_ms": k1_ms},
            "cpu_baseline": cpu,
            "stage_ms": {"classify_k1": k1_ms, "finalize_kernels": tm["finalize_ms"] / max(args.steps, 1)},
            "checks": None if res is None else {"gene_reads_sum": int(res.gene_reads.sum()),
                                                "total_alignments": res.counter("Total Alignments")},
            "input_generation_s": round(t_gen, 1),
        }
        if args.fasta:
            out["invalid"] = "diagnostic run: --fasta GC statistics on (extra candidate pass + mate pairing per step)"
            out["gc_fragments"] = None if res is None else int(res.gc_bins.sum())
            out["reference_setup_s"] = round(t_ref, 2)
        if args.legacy:
            out["invalid"] = "diagnostic run: --legacy counting rules (general per-record kernel, not the headline path)"
        if args.no_finalize:
            out["invalid"] = "diagnostic run: end-of-file stage skipped"
        if args.dist_selftest:
            out["invalid"] = "diagnostic run: the N > 1 code path on one rank"
        if args.host_fed:
            out["invalid"] = "diagnostic run: PCIe-inclusive (inputs uploaded from host memory inside the timed region)"
            out["h2d_ms_per_step"] = tm["h2d_ms"] / max(args.steps, 1)
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    e.close()
    if dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
