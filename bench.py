#!/usr/bin/env python
"""bench.py -- reads/sec of the RNA-SeQC per-read hot path on MI355X.

Workload (the one BASELINE.json's metric is quoted on, configs[2] / configs[3]): a GENCODE-sized collapsed
annotation (25 contigs, 56 202 genes, ~323 k exons) + ~100 M synthetic 2x150 bp coordinate-sorted alignment
records (50 M pairs + secondary / supplementary copies + the unmapped tail), resident in HBM as the boundary's
SoA batch before the timed region.

One "step" is one complete pass of the hot path over those records: zero the accumulators, K1
classify/count over every record, the slow-path and Read-Length kernels, the end-of-file stage (fragment
de-dup, coverage scan, per-gene coverage statistics and bias windows), and the read-back of every result
vector; for N > 1 also the RCCL sum-reduction.  Nothing is skipped or cached between steps.

--gpus N (strong scaling): the SAME 100 M records, sharded by contig -- contigs are packed onto the N ranks
by longest-processing-time on their record counts (rnaseqc_amd/distributed.assign_contigs), every rank
generates and owns only its contigs, and at end of file the ranks sum-reduce the three device ranges of
rsqc_device_vectors over RCCL (counts + exon sums are additive, per-gene statistics are owner-only).

Three tiers, never mixed:
  value / roofline   device-resident kernel tier (what the contract's `value` is); roofline.kernel_ms comes from hipEvents
                     around the per-read kernel on the context's stream (rsqc_get_timing)
  whole_node /       the CLI's `Average Reads/Sec` window (src/RNASeQC.cpp:240-241,389-394) on a BAM of the same records:
  end_to_end         BGZF inflate + BAM parse on the GPU + the same kernels + the end-of-file stage, report files written;
                     at every N (rank 0 writes the BAM, `rnaseqc --gpus N` reads it by contig).  Its report files are
                     compared with the kernel tier's results (whole_node.parity) -- the two tiers measure the same job
  cpu_baseline       the C oracle on one host core, bounded sample

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import re
import shutil
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0      # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md


def _kv(path):
    out = {}
    for line in open(path):
        f = line.rstrip("\n").split("\t")
        if len(f) >= 2:
            out[f[0]] = f[1]
    return out


def _gct(path):
    """(row count of the header line, last column of every row) of a GCT file"""
    with open(path) as fh:
        lines = fh.read().split("\n")
    n_hdr = int(lines[1].split("\t")[0])
    vals = np.array([float(x.rsplit("\t", 1)[1]) for x in lines[3:] if x], np.float64)
    return n_hdr, vals


def e2e_parity(odir, sample, res, read_length):
    """The CLI's report files against the kernel tier's result of the SAME records (one engine, resident input): every
    integer counter the metrics file prints, Read Length, gene_reads / gene_fragments row by row, the exon_reads header
    count (src/RNASeQC.cpp:513) and exon values within the contract's 1e-6."""
    from rnaseqc_amd import abi
    bad = []
    m = _kv(os.path.join(odir, sample + ".metrics.tsv"))
    checked = 0
    for name in abi.COUNTER_NAMES:
        if name in m:
            checked += 1
            if int(float(m[name])) != int(res.counter(name)):
                bad.append("%s: file %s, kernel tier %d" % (name, m[name], res.counter(name)))
    if int(float(m.get("Read Length", -1))) != int(read_length):
        bad.append("Read Length: file %s, kernel tier %d" % (m.get("Read Length"), read_length))
    n, v = _gct(os.path.join(odir, sample + ".gene_reads.gct"))
    if n != len(res.gene_reads) or not np.array_equal(v.astype(np.int64), np.asarray(res.gene_reads).astype(np.int64)):
        bad.append("gene_reads.gct differs (file sum %d, kernel tier %d)" % (int(v.sum()), int(res.gene_reads.sum())))
    n, v = _gct(os.path.join(odir, sample + ".gene_fragments.gct"))
    if not np.array_equal(v.astype(np.int64), np.asarray(res.gene_fragments).astype(np.int64)):
        bad.append("gene_fragments.gct differs (file sum %d, kernel tier %d)" % (int(v.sum()), int(res.gene_fragments.sum())))
    n, v = _gct(os.path.join(odir, sample + ".exon_reads.gct"))
    if n != int(np.count_nonzero(res.exon_hit)):
        bad.append("exon_reads.gct header count %d, kernel tier %d" % (n, int(np.count_nonzero(res.exon_hit))))
    if len(v) != len(res.exon_reads) or not np.allclose(v, np.asarray(res.exon_reads), rtol=0, atol=1.5e-6):   # (the file has 6 decimals)
        bad.append("exon_reads.gct values differ")
    return {"parity": not bad, "counters_checked": checked + 1, "gene_reads_sum": int(res.gene_reads.sum()),
            "against": "kernel tier of this run (same records resident in HBM, one engine): integer counters + Read Length exact, "
                       "gene_reads / gene_fragments row by row, exon_reads header count and values within 1e-6",
            "mismatches": bad[:8]}


def end_to_end(args, ann, contigs, batch, st, log, res=None, read_length=None, gpus=1, gpu_list=None):
    """The whole-node tier: `rnaseqc gtf bam out -vv [--gpus N]` on a BAM of the bench records.  Driver-timed here; the
    reads/s is the CLI's own `Average Reads/Sec` line (BAM loop + end-of-file stage, GTF load and report writing
    excluded, exactly the reference's window, src/RNASeQC.cpp:240-241,389-394).  The report files of the timed run are
    compared with the kernel tier's result before they are deleted."""
    from rnaseqc_amd import bamio, hostinfo
    cores = hostinfo.effective_cpus()
    d = tempfile.mkdtemp(prefix="rsqc_e2e_", dir=args.tmp or None)
    out = {"cores": cores, "hardware_threads": os.cpu_count(), "gpus": gpus,
           "cores_note": "CPUs the process may use = affinity mask capped by the cgroup CPU quota (cpu.max); the decode pools are sized to it"}
    try:
        bam, gtf = os.path.join(d, "s.bam"), os.path.join(d, "s.gtf")
        t = time.time()
        bamio.write_bam_fast(bam, [(c[0], c[1]) for c in contigs], batch, threads=min(cores, 96), struct=st, bai=gpus > 1)
        out["bam_write_s"] = round(time.time() - t, 1)
        out["bam_bytes"] = os.path.getsize(bam)
        bamio.write_gtf(gtf, ann)
        exe = os.path.join(ROOT, "rnaseqc_amd", "bin", "rnaseqc")
        env = dict(os.environ)
        if args.e2e_threads:
            env["RSQC_HOST_THREADS"] = str(args.e2e_threads)
        if gpu_list:
            env["RSQC_GPU_LIST"] = gpu_list
        extra = ["--gpus", str(gpus)] if gpus > 1 else []
        def cli_runs(bam_path, mode, tag, reps=2):
            """`rnaseqc gtf bam out -vv` with RSQC_DECODE=mode; the best of reps runs (the second has the file in the page cache
            and the GPU driver warm).  Every run writes its own output directory."""
            runs = []
            for rep in range(reps):
                odir = os.path.join(d, "out_%s_%d" % (tag, rep))
                t = time.time()
                p = subprocess.run([exe, gtf, bam_path, odir, "-vv"] + extra, env=dict(env, RSQC_DECODE=mode, RSQC_DECODE_PROFILE="1"), capture_output=True, text=True)
                wall = time.time() - t
                # the decode calls of the run (hipEvents around the kernels of every rsqc_decode_submit, summed by the library; one GPU)
                dp = None
                pr = [x for x in re.findall(r"\[decode\] (\d+) calls, ([0-9.]+) MB in, ([0-9.]+) MB inflated: copy ([0-9.]+) ms, inflate ([0-9.]+) ms \(([0-9.]+) GB/s out\), "
                                            r"frame\+parse ([0-9.]+) ms", p.stderr) if int(x[0]) > 0]
                if len(pr) == 1:
                    c, mb_in, mb_out, _cp, inf_ms, gbps, fp_ms = pr[0]
                    sh = re.search(r"CPU share of the inflate work at the end ([0-9.]+); ([0-9.]+) ms waiting for file chunks", p.stderr)
                    dp = {"calls": int(c), "uploaded_MB": float(mb_in), "inflated_MB": float(mb_out), "inflate_kernels_ms": float(inf_ms),
                          "inflate_GBps_of_inflated_bytes": float(gbps), "frame_parse_kernels_ms": float(fp_ms),
                          "bgzf_blocks_per_call": int(float(mb_out) * 1e6 / 65280 / max(int(c), 1)),
                          "cpu_share_of_blocks": float(sh.group(1)) if sh else None, "waiting_for_file_ms": float(sh.group(2)) if sh else None,
                          "note": "one wavefront inflates one BGZF block; the chip holds 5 120 of the kernel's waves (DESIGN.md 6b)"}
                m = re.search(r"Average Reads/Sec: ([0-9.e+]+)", p.stdout)
                e = re.search(r"Time Elapsed: ([0-9.e+-]+); Alignments processed: (\d+)", p.stdout)
                th = re.search(r"decode threads: (\d+) inflate \+ (\d+) parse", p.stdout)
                wb = re.search(r"Wall time: ([0-9.e+-]+) s = GTF ([0-9.e+-]+) \+ waiting for the GPU context ([0-9.e+-]+) \+ index / annotation upload / buffers ([0-9.e+-]+)"
                               r" \+ BAM loop ([0-9.e+-]+) \+ reports ([0-9.e+-]+) \+ release ([0-9.e+-]+)", p.stdout)
                runs.append({"rc": p.returncode, "wall_s": round(wall, 3), "bam_loop_s": float(e.group(1)) if e else None,
                             "wall_breakdown_s": dict(zip(("process", "gtf", "gpu_context_wait", "index_upload_buffers", "bam_loop", "reports", "release"),
                                                          (round(float(x), 3) for x in wb.groups()))) if wb else None,
                             "alignments": int(e.group(2)) if e else None,
                             "reads_per_s": float(m.group(1)) if m else None,
                             "decode": "device" if "on the GPU" in p.stdout else "host", "decode_profile": dp,
                             "decode_threads": [int(th.group(1)), int(th.group(2))] if th else None, "odir": odir})
                if p.returncode:
                    log("end_to_end: rnaseqc (%s decode) exited %d: %s" % (mode, p.returncode, p.stderr[-400:]))
            best = max(runs, key=lambda r: r["reads_per_s"] or 0.0)
            return best, [{k: v for k, v in r.items() if k != "odir"} for r in runs]
        # the product's default: BGZF inflate + record framing + parsing on the GPU (rsqc_decode_*); the host-decode path beside it
        best, runs = cli_runs(bam, "device", "dev")
        host_best, _hr = cli_runs(bam, "host", "host")
        out.update({"value": best["reads_per_s"], "unit": "reads/s", "bam_loop_s": best["bam_loop_s"], "wall_s": best["wall_s"],
                    "alignments": best["alignments"], "decode": best["decode"], "decode_profile": best.get("decode_profile"), "runs": runs, "wall_breakdown_s": best.get("wall_breakdown_s"),
                    "host_decode": {"value": host_best["reads_per_s"], "unit": "reads/s", "decode_threads": host_best["decode_threads"],
                                    "note": "RSQC_DECODE=host: libdeflate inflate + record parsing on the CPU threads, batches over PCIe"},
                    "window": "CLI `Average Reads/Sec` = alignments / (BAM loop incl. end-of-file stage), src/RNASeQC.cpp:240-241,389-394",
                    "bam": "SEQ all 'A', QUAL 0xff, BGZF level 1 (SURVEY.md 8(d)): %.1f B/record compressed" % (out["bam_bytes"] / max(batch.n, 1))})
        if res is not None and best["rc"] == 0:
            try:
                out.update(e2e_parity(best["odir"], "s.bam", res, read_length))
                hp = e2e_parity(host_best["odir"], "s.bam", res, read_length) if host_best["rc"] == 0 else {"parity": False}
                out["host_decode"]["parity"] = hp["parity"]
            except Exception as ex:
                out.update({"parity": False, "mismatches": ["parity check failed to run: %r" % (ex,)]})
        if args.e2e_real:                       # second flavour: the compressibility of a real file, first --e2e-real records
            nr = min(args.e2e_real, batch.n)
            sub = batch.slice(0, nr) if nr < batch.n else batch
            bam2 = os.path.join(d, "r.bam")
            bamio.write_bam_fast(bam2, [(c[0], c[1]) for c in contigs], sub, threads=min(cores, 96), seq_mode=1, bai=gpus > 1)
            b2, _r2 = cli_runs(bam2, "device", "rdev")
            h2, _r3 = cli_runs(bam2, "host", "rhost", reps=1)
            same = None
            if b2["rc"] == 0 and h2["rc"] == 0:     # two decoders (GPU kernels / libdeflate + host parser), one set of report files
                same = all(open(os.path.join(b2["odir"], "r.bam." + f), "rb").read() == open(os.path.join(h2["odir"], "r.bam." + f), "rb").read()
                           for f in ("metrics.tsv", "gene_reads.gct", "gene_fragments.gct", "exon_reads.gct", "gene_tpm.gct"))
            out["realistic_entropy"] = {"value": b2["reads_per_s"], "unit": "reads/s", "records": int(sub.n), "bam_bytes": os.path.getsize(bam2),
                                        "decode": b2["decode"], "decode_profile": b2.get("decode_profile"), "host_decode": h2["reads_per_s"],
                                        "parity": same, "against": "the host-decode run of the same file: report files byte for byte",
                                        "bam": "random bases, binned Phred-like qualities with runs: %.1f B/record compressed" %
                                               (os.path.getsize(bam2) / max(sub.n, 1))}
    finally:
        shutil.rmtree(d, ignore_errors=True)
    return out


def inflate_model(measured_gbps):
    """Ceilings of bgzf_inflate_kernel on the realistic-entropy file from ITS counters (profiles/r5_decode_pmc_realistic_6M.txt: the kernel
    source has not changed since -- rsqc_inflate.h; one wavefront inflates one BGZF block of <= 65 280 bytes in ~2 500 serial rounds):
      scalar_port   one scalar ALU per CU, one scalar instruction per cycle: CUs x clock / (SALU per block) x bytes per block
      issue(W)      a wave issues one instruction per ~4.7 cycles (SQ_ACTIVE_INST_ANY / instructions): 1024 SIMDs x W waves x clock /
                    (instructions per block x 4.7) x bytes per block -- what more waves per SIMD (a smaller decoder state) could lift
    The binding one is the scalar port; occupancy does not move it."""
    path = os.path.join(ROOT, "profiles", "r5_decode_pmc_realistic_6M.txt")
    try:
        cur, c = None, {}
        for line in open(path):
            if not line.startswith(" "):
                cur = line.strip(); continue
            if cur and "bgzf_inflate_kernel" in cur:
                m = re.match(r"\s+(\S+)\s+n=\d+\s+mean=(\S+)", line)
                if m: c[m.group(1)] = float(m.group(2))
        blocks = c["SQ_WAVES"]
        salu, valu, lds = c["SQ_INSTS_SALU"] / blocks, c["SQ_INSTS_VALU"] / blocks, c["SQ_INSTS_LDS"] / blocks
        instr = salu + valu + lds
        cyc_per_instr = 4.0 * c["SQ_ACTIVE_INST_ANY"] / (c["SQ_INSTS_SALU"] + c["SQ_INSTS_VALU"] + c["SQ_INSTS_LDS"])
        bpb, clk = 65280.0, 2.4e9
        out = {"per_block": {"salu": round(salu), "valu": round(valu), "lds": round(lds), "rounds": 2500, "bytes_per_round": 26, "symbols_per_round": 5.8},
               "cycles_per_instruction_of_a_wave": round(cyc_per_instr, 2),
               "scalar_port_ceiling_GBps": round(256 * clk / salu * bpb / 1e9, 1),
               "issue_ceiling_GBps_at_waves_per_simd": {str(w): round(1024 * w * clk / (instr * cyc_per_instr) * bpb / 1e9, 1) for w in (5, 6, 7)},
               "measured_GBps": measured_gbps, "file": "realistic entropy",
               "source": "profiles/r5_decode_pmc_realistic_6M.txt (kernel unchanged since: rsqc_inflate.h)"}
        out["binding"] = "scalar_port"
        out["measured_over_binding"] = (measured_gbps / out["scalar_port_ceiling_GBps"]) if measured_gbps else None
        return out
    except Exception as ex:
        return {"error": repr(ex)}


def k1_floor_model(k1_ms):
    """Lower bounds of classify_ei_kernel on the contract workload from the counters of THIS build (profiles/k1_model.json, written by
    tools/k1_model.py from the rocprofv3 --pmc passes of the same bench command; stamped with the hash of the K1 sources).  Three floors,
    each what the kernel would take if that resource alone were the limit:
      valu_floor_ms    SQ_INSTS_VALU x cycles per wave64 instruction / (1024 SIMDs x clock)
      atomic_floor_ms  memory atomics + scattered stores / the chip's measured rate for that pattern (tools/atomic_bench)
      traffic_floor_ms HBM bytes (FETCH_SIZE / WRITE_SIZE with the calibrated corrections) / 6.3 TB/s achievable"""
    path = os.path.join(ROOT, "profiles", "k1_model.json")
    if not os.path.exists(path):
        return None
    try:
        m = json.load(open(path))
        from rnaseqc_amd.hostinfo import k1_code_hash
        m["current"] = m.get("k1_code_hash") == k1_code_hash()
        m["measured_kernel_ms"] = k1_ms
        return m
    except Exception as ex:
        return {"error": repr(ex)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--pairs", type=int, default=50_000_000, help="read pairs of the whole job (2 records each + 1.5 %% copies + 1 %% unmapped)")
    ap.add_argument("--cpu-sample", type=int, default=60_000_000, help="records timed on the CPU oracle (0 = skip)")
    ap.add_argument("--workers", type=int, default=0, help="generator processes (0 = min(usable CPUs, 24))")
    ap.add_argument("--chr1", action="store_true", help="(diagnostic) BASELINE.json configs[1]: chr1-like GTF + --pairs pairs "
                                                         "(default 5 M) on one GPU; output marked invalid")
    ap.add_argument("--no-e2e", action="store_true", help="skip the end_to_end tier (CLI on a BAM of the same records)")
    ap.add_argument("--e2e-real", type=int, default=50_000_000, help="records of the realistic-entropy BAM flavour (0 = skip)")
    ap.add_argument("--e2e-threads", type=int, default=0, help="RSQC_HOST_THREADS for the CLI runs (0 = its default)")
    ap.add_argument("--tmp", default="", help="directory for the end_to_end files (default: the system temp dir)")
    ap.add_argument("--no-finalize", action="store_true", help="(diagnostic) time K1 only; output marked invalid")
    ap.add_argument("--host-fed", action="store_true",
                    help="(diagnostic) every step uploads the batch from page-locked host memory through rsqc_submit: the "
                         "PCIe-inclusive rate noted in DESIGN.md; not the bench line")
    ap.add_argument("--per-contig-batches", action="store_true", help="sharded runs: one resident batch per owned contig (round 4) instead of one batch of file ranges")
    ap.add_argument("--dist-selftest", action="store_true",
                    help="(diagnostic) single process, but through the N > 1 code path: 1-rank RCCL group, finalize_device, "
                         "all_reduce of the device ranges, refresh_results")
    ap.add_argument("--fasta", action="store_true", help="(diagnostic) with a reference sequence: GC statistics on; output marked invalid")
    ap.add_argument("--legacy", action="store_true", help="(diagnostic) the --legacy counting rules; output marked invalid")
    ap.add_argument("--bed", action="store_true", help="BASELINE configs[4] on the GPUs given: the same workload with BED intervals, i.e. the fragment-size sampler on (per-base coverage and the bias windows are always computed)")
    args = ap.parse_args()

    # RCCL prints a version banner on stdout when the communicator comes up; the contract is ONE JSON line there,
    # so everything else this process (and the libraries it loads) writes to fd 1 is sent to stderr.
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world == 1 and args.gpus > 1:
        raise SystemExit("launch with torch.distributed.run --nproc-per-node %d" % args.gpus)

    def log(msg):
        if rank == 0:
            sys.stderr.write("[bench] %s\n" % msg); sys.stderr.flush()

    # ---- synthetic inputs (generated BEFORE the HIP runtime comes up: the generator forks worker processes) -----
    from rnaseqc_amd import abi, distributed, synth
    if args.chr1:
        if world > 1:
            raise SystemExit("--chr1 is a single-GPU diagnostic")
        contigs = [synth.HUMAN_CONTIGS[0]]
        if args.pairs == 50_000_000:
            args.pairs = 5_000_000
    else:
        contigs = synth.human_contigs()
    t_gen = time.time()
    ann = synth.make_annotation(seed=1, contigs=contigs)
    share = synth.contig_pair_shares(ann, args.pairs)
    rank_of = distributed.assign_contigs(share, world)
    load = np.array([int(share[rank_of == r].sum()) for r in range(world)])
    mine = [int(c) for c in np.flatnonzero(rank_of == rank)]
    tail_rank = int(np.argmin(load))                     # the unmapped tail goes to the lightest shard
    from rnaseqc_amd import hostinfo
    cores = hostinfo.effective_cpus()
    workers = args.workers or max(1, min(cores // max(world, 1), 24))
    if args.chr1:
        batch = synth.make_reads(ann, args.pairs, seed=2)
        parts = [batch]
    elif world > 1 or args.dist_selftest:
        # sharded runs submit one batch per owned contig: a batch is a contiguous range of the file, and the
        # order-dependent outputs are composed per batch in file order (rsqc_shard_info)
        want_e2e = rank == 0 and not args.no_e2e and not (args.no_finalize or args.host_fed or args.fasta or args.legacy or args.bed)
        if want_e2e and world > 1:
            # rank 0 also runs the whole-node tier (`rnaseqc --gpus N` on a BAM of ALL the records): it generates every contig
            # and keeps its own for the kernel tier
            every, _ = synth.make_reads_sharded(ann, args.pairs, seed=2, workers=max(workers, cores // 2), with_unmapped=True, as_parts=True)
            ids = [c for c in range(ann.n_ref) if share[c] > 0]
            parts = [every[k] for k, c in enumerate(ids) if c in mine] + ([every[-1]] if (rank == tail_rank and len(every) > len(ids)) else [])
            full_parts = every
        else:
            parts, _ = synth.make_reads_sharded(ann, args.pairs, seed=2, contigs=mine, workers=workers, with_unmapped=(rank == tail_rank), as_parts=True)
            full_parts = parts if world == 1 else None
        batch = None
    else:
        batch, _ = synth.make_reads_sharded(ann, args.pairs, seed=2, contigs=mine, workers=workers, with_unmapped=(rank == tail_rank))
        parts = [batch]
        full_parts = None
    if args.chr1:
        full_parts = None
    if (world > 1 or args.dist_selftest) and not args.per_contig_batches and len(parts) > 1:
        # a shard's contigs are non-adjacent ranges of the file: ONE batch whose segments carry their own file index
        # (rsqc_batch.seg_file_index), one kernel launch; the order-dependent outputs stay per range (rsqc_shard_info)
        if full_parts is parts:
            full_parts = list(parts)
        from rnaseqc_amd.model import Batch
        parts = [Batch.concat_ranges(parts)]
    structs = [b.to_struct() for b in parts]
    st = structs[0] if len(structs) == 1 else None
    n_local = sum(b.n for b in parts)
    t_gen = time.time() - t_gen
    log("inputs: %d genes, %d exons, %d records in %d batch(es) on this rank, %.1f s (%d generator processes)" % (ann.n_genes, ann.n_exons, n_local, len(parts), t_gen, workers))

    import torch
    from rnaseqc_amd import engine
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

    p = abi.default_params(device=local_rank, **(dict(legacy=1, mapq_threshold=4) if args.legacy else {}))
    e = engine.Engine(p)
    owned = None
    if world > 1:
        owned = distributed.owned_mask(rank_of, rank)
        owned = np.concatenate([owned, np.zeros(ann.n_contigs - len(owned), np.uint8)])
    e.set_annotation(ann, owned)
    if args.bed:
        e.set_bed(synth.make_bed(ann))
    if args.fasta:
        t_ref = time.time()
        e.set_reference(synth.make_reference([c[1] for c in contigs], seed=7, uniform=True))
        t_ref = time.time() - t_ref
    handles = [e.upload_struct(x) for x in structs]      # inputs resident in HBM before the timed region
    if args.host_fed:                        # the same batch in page-locked host memory
        if st is None:
            raise SystemExit("--host-fed is a single-batch diagnostic")
        import ctypes
        keep = []
        for f, n_items, dt in (("core", batch.n, abi.REC_CORE), ("aux", batch.n, abi.REC_AUX), ("cigar", len(batch.cigar), np.uint32)):
            src = np.frombuffer((ctypes.c_char * (n_items * np.dtype(dt).itemsize)).from_address(getattr(st, f)), dtype=dt, count=n_items)
            pin = e.pinned_copy(src); keep.append(pin)
            setattr(st, f, pin.ctypes.data)

    vec = None
    order_dep = [None]
    collective_s = [0.0]
    collective_parts = [0.0, 0.0]              # [issue of the three all_reduce + the wait for them behind the merge, the host-side merge between]
    if reduce_path:
        vec = [torch.as_tensor(v, device="cuda") for v in e.device_vectors()]

    def step():
        e.reset()
        if args.host_fed:
            e.submit_struct(st)              # H2D (DMA from page-locked memory) + K1, as the CLI does per batch
        else:
            for h in handles:
                e.submit_resident(h)
        if args.no_finalize:
            e.wait()
            return None
        if not reduce_path:
            return e.finalize(lazy=True)     # the vectors are on the host (library buffers); Python copies are made on access
        e.finalize_device()                  # results stay on the device until they are reduced (returns with the stream idle)
        tc = time.perf_counter()
        # RCCL over xGMI: u64 counts | f64 sums + owner-only statistics | u8 validity flags -- issued together, waited for after
        # the host-side merge below (the three reductions overlap each other and the two small gathers of the merge)
        pending = [dist.all_reduce(t, async_op=True) for t in vec]
        t1 = time.perf_counter()
        # the order-dependent outputs (Read Length; the fragment-size cut-off with --bed): per-batch transfer functions
        # and kept samples of every rank, composed in file order on the host
        order_dep[0] = distributed.merge_order_dependent(e.shard_summary(), dist, torch.device("cuda", local_rank), p.fragment_samples)
        t2 = time.perf_counter()
        for w in pending:
            w.wait()
        torch.cuda.synchronize()
        t3 = time.perf_counter()
        collective_s[0] += t3 - tc
        collective_parts[0] += (t1 - tc) + (t3 - t2); collective_parts[1] += t2 - t1
        return e.refresh_results(lazy=True)

    for _ in range(args.warmup):
        step()
    e.reset_timing()
    collective_s[0] = 0.0; collective_parts[0] = collective_parts[1] = 0.0
    if dist: dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    res = None
    marks = [t0]
    for _ in range(args.steps):
        res = step()
        marks.append(time.perf_counter())    # (a step ends with its results on the host: the stream is idle here, except with --no-finalize)
    torch.cuda.synchronize()
    if dist: dist.barrier()
    elapsed = time.perf_counter() - t0
    per_rank = [int(n_local)]
    if dist:
        t = torch.tensor([elapsed], device="cuda", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        nrec = torch.zeros(world, device="cuda", dtype=torch.int64)
        nrec[rank] = n_local
        dist.all_reduce(nrec)
        per_rank = [int(x) for x in nrec.tolist()]
    total_records = sum(per_rank)
    tm = e.timing()

    if rank == 0:
        k1_ms = tm["classify_ms"] / max(tm["classify_launches"], 1)
        k1_long_ms = tm.get("classify_long_ms", 0.0) / max(tm["classify_launches"], 1)       # classify_long_kernel: the ~1 % of the records K1 defers
        bytes_per_launch = tm["classify_bytes"] / max(tm["classify_launches"], 1)
        achieved = bytes_per_launch / (k1_ms * 1e-3) / 1e9 if k1_ms > 0 else 0.0
        # HBM traffic of K1 per launch comes from separate rocprofv3 --pmc passes of this same command
        # (FETCH_SIZE and WRITE_SIZE cannot share a pass); tools/pmc.sh stores them in profiles/k1_traffic.json
        traffic = None; traffic_note = None
        tpath = os.path.join(ROOT, "profiles", "k1_traffic.json")
        if os.path.exists(tpath) and world == 1:
            tj = json.load(open(tpath))
            from rnaseqc_amd.hostinfo import k1_code_hash
            # (the figure is only valid for the code it was measured on: the file carries the hash of the kernel's sources)
            if int(tj.get("records") or 0) == int(n_local) and int(tj.get("genes") or 0) == int(ann.n_genes):
                if tj.get("k1_code_hash") == k1_code_hash():
                    traffic = tj.get("hbm_bytes_per_launch")
                else:
                    traffic_note = "profiles/k1_traffic.json was measured on other K1 code (hash %s, now %s): dropped" % (tj.get("k1_code_hash"), k1_code_hash())
        cpu = None
        if args.cpu_sample > 0 and world == 1 and batch is not None:          # the CPU baseline is a single-GPU-run item (rank 0, N = 1)
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
                   "sample": "first %d records (file order) of the bench workload through oracle/rsqc_oracle.c: single thread, "
                             "SoA input, no BAM decode, incl. its end-of-file stage" % ns, "seconds": round(tc, 3)}
            del sample
        e2e = None
        if not args.no_e2e and not (args.no_finalize or args.host_fed or args.fasta or args.legacy or args.bed):
            try:
                rl_now = int(order_dep[0][0]) if order_dep[0] else int(res.read_length)
                if reduce_path:
                    # the N > 1 form: `rnaseqc --gpus N` shards the BAM by contig through its index, one context per GPU, and sums the
                    # shards over xGMI; --dist-selftest (one rank) drives the same code with two contexts on the one GPU
                    from rnaseqc_amd.model import Batch
                    whole = Batch.concat(full_parts)
                    n_cli = world if world > 1 else 2
                    e2e = end_to_end(args, ann, contigs, whole, None, log, res, rl_now, gpus=n_cli,
                                     gpu_list=None if world > 1 else "0,0")
                else:
                    e2e = end_to_end(args, ann, contigs, batch, st, log, res, rl_now)
            except Exception as ex:             # the kernel tier stands on its own
                e2e = {"error": repr(ex)}
        wl = ("configs[1] (diagnostic): chr1-like collapsed GTF (%d genes, %d exons) + %d records" if args.chr1 else
              "GENCODE-sized collapsed GTF (%d genes, %d exons, 25 contigs) + %d synthetic 2x150 coordinate-sorted records") % (
                  ann.n_genes, ann.n_exons, total_records)
        if args.bed:
            wl = "BASELINE configs[4] on %d GPU(s): " % world + wl + " + BED intervals (fragment-size sampler on, --fragment-samples %d; per-base coverage and bias windows as always)" % p.fragment_samples
        out = {
            "metric": "reads/sec whole-node (100 M-read synthetic BAM, GENCODE GTF) at 1/2/4/8 GPU -- `value`: the hot path over the records resident in "
                      "HBM (the bench contract's definition of value); `whole_node.value`: the same job from the BAM file to the reports "
                      "(CLI `Average Reads/Sec` window), checked against `value`'s results",
            "value": total_records * args.steps / elapsed,
            "unit": "reads/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps,
            "ms_per_step_spread": (lambda d: {"min": round(d[0], 4), "median": round(d[len(d) // 2], 4), "max": round(d[-1], 4),
                                              "note": "rank 0's host clock per step; the kernels' own times are in stage_ms"})(
                sorted(1e3 * (b - a) for a, b in zip(marks, marks[1:]))) if not args.no_finalize else None,
            "higher_is_better": True, "scaling": "strong" if not args.chr1 else "weak",
            "vs_baseline": None,
            "dtype": "i32/u64 counters, f64 exon fractions",
            "data": "synthetic (seeded generator rnaseqc_amd/synth.py; no real GENCODE/BAM offline)",
            "config": {"workload": wl + ", SoA resident in HBM, full pass incl. end-of-file stage and read-back",
                       "records": total_records, "genes": int(ann.n_genes), "exons": int(ann.n_exons),
                       "records_per_gpu": per_rank, "load_imbalance": round(max(per_rank) / (sum(per_rank) / len(per_rank)), 4),
                       "sharding": "by contig, LPT on record counts" if world > 1 else "none (one GPU holds every contig)",
                       "collective": ("RCCL all_reduce(sum) of i64[3G+%d+2G] + f64[2E+3G] + u8[G+E] per step" % abi.N_COUNTERS) if reduce_path else "none"},
            "roofline": {"bound": "hbm", "kernel": "classify_ei_kernel" if not args.legacy else "classify_count_kernel_legacy + classify_slow_kernel", "achieved": achieved, "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_note": traffic_note,
                         "algorithmic_bytes_per_launch": bytes_per_launch, "kernel_ms": k1_ms,
                         # the ~1 % of the records the kernel defers (more than eight CIGAR operations / three blocks) take their feature stage in
                         # classify_long_kernel right behind it: its time, and the fraction with it counted in
                         "deferred_kernel_ms": k1_long_ms,
                         "frac_incl_deferred_kernel": (bytes_per_launch / ((k1_ms + k1_long_ms) * 1e-3) / 1e9 / HBM_PEAK_GBS) if k1_ms > 0 else 0.0,
                         "model": k1_floor_model(k1_ms) if not args.legacy else None,
                         "timer": "hipEvents around the launch on the context's stream (rsqc_get_timing), rank 0"},
            "cpu_baseline": cpu,
            "whole_node": None if not e2e or "value" not in e2e else {
                "value": e2e["value"], "unit": "reads/s", "n_gpus": e2e.get("gpus", 1), "parity": e2e.get("parity"),
                # the same run by the driver's clock: records / wall time of the whole `rnaseqc` process (GTF parse, GPU start-up, BAM
                # loop, report files, exit) -- `value` is the reference's own `Average Reads/Sec` window, a part of that process
                "wall_value": (e2e["alignments"] / e2e["wall_s"]) if e2e.get("alignments") and e2e.get("wall_s") else None,
                "wall_s": e2e.get("wall_s"), "bam_loop_s": e2e.get("bam_loop_s"),
                "realistic_entropy_value": (e2e.get("realistic_entropy") or {}).get("value"),
                "realistic_entropy_parity": (e2e.get("realistic_entropy") or {}).get("parity"),
                # the loop's dominant kernel: GB/s of inflated bytes out of bgzf_inflate_kernel, the two files (end_to_end.*.decode_profile)
                "inflate_GBps": [(e2e.get("decode_profile") or {}).get("inflate_GBps_of_inflated_bytes"),
                                 ((e2e.get("realistic_entropy") or {}).get("decode_profile") or {}).get("inflate_GBps_of_inflated_bytes")],
                "inflate_model": inflate_model(((e2e.get("realistic_entropy") or {}).get("decode_profile") or {}).get("inflate_GBps_of_inflated_bytes")),
                "note": "`rnaseqc gtf bam out` on a BAM of the same records: BGZF inflate + BAM parse on the GPU + the hot path + end-of-file stage; details in end_to_end"},
            "end_to_end": e2e,
            "collective_ms": (1e3 * collective_s[0] / max(args.steps, 1)) if reduce_path else None,   # per step: 3 async all_reduce + 2 gathers + host merge
            # ... split: what the three RCCL all_reduce cost by the host's clock (their issue + the wait for them BEHIND the merge: the part
            # the merge does not hide), and the host-side merge of the order-dependent outputs (two packed all_gathers + numpy) between the two
            "collective_rccl_exposed_ms": (1e3 * collective_parts[0] / max(args.steps, 1)) if reduce_path else None,
            "collective_host_merge_ms": (1e3 * collective_parts[1] / max(args.steps, 1)) if reduce_path else None,
            "stage_ms": {"classify_k1": k1_ms * len(handles), "classify_long": k1_long_ms * len(handles), "classify_launches_per_step": len(handles), "finalize_kernels": tm["finalize_ms"] / max(args.steps, 1),
                         "fragment_sizes": (tm.get("fragment_sizes_ms", 0.0) / max(args.steps, 1)) if args.bed else None,
                         "slow_path_records": int(tm["slow_records"])},
            "checks": None if res is None else {"gene_reads_sum": int(res.gene_reads.sum()),
                                                "total_alignments": res.counter("Total Alignments"),
                                                "read_length": int(order_dep[0][0]) if order_dep[0] else int(res.read_length)},
            "input_generation_s": round(t_gen, 1),
        }
        for flag, why in (("chr1", "diagnostic run: configs[1] (chr1), not the workload the metric is quoted on"),
                          ("fasta", "diagnostic run: --fasta GC statistics on"), ("legacy", "diagnostic run: --legacy counting rules"),
                          ("no_finalize", "diagnostic run: end-of-file stage skipped"),
                          ("dist_selftest", "diagnostic run: the N > 1 code path on one rank"),
                          ("host_fed", "diagnostic run: PCIe-inclusive (inputs uploaded from host memory inside the timed region)")):
            if getattr(args, flag):
                out["invalid"] = why
        if args.fasta:
            out["gc_fragments"] = None if res is None else int(res.gc_bins.sum())
            out["reference_setup_s"] = round(t_ref, 2)
        if args.bed and res is not None:
            out["fragment_samples"] = int(np.asarray(res.fragment_count).sum())
        if args.host_fed:
            out["h2d_ms_per_step"] = tm["h2d_ms"] / max(args.steps, 1)
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    if dist and world > 1:
        dist.barrier()                          # (ranks > 0 keep their contexts until rank 0's `rnaseqc --gpus N` run is over)
    e.close()
    if dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
