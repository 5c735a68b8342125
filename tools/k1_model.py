#!/usr/bin/env python3
"""Floor model of classify_ei_kernel on the contract workload: what the kernel would take if ONE resource were
its only limit, from the counters of the build it is run on.

  usage: tools/k1_model.py <pmc summary.txt> [--traffic traffic.txt] [--kernel-ms 2.27] [--out profiles/k1_model.json]

Inputs: the summary tools/pmc.sh writes (SQ_INSTS_VALU / SALU / LDS, FETCH_SIZE, WRITE_SIZE of classify_ei_kernel, mean per launch)
and, optionally, tools/r6_traffic.sh's table of the ablation builds (the per-source split of the traffic).  Chip constants from
/opt/skills/guides/MI355X_MICROARCH.md: 256 CUs x 4 SIMD-32, 2.4 GHz, one scalar ALU per CU (one scalar instruction per cycle per CU),
a wave64 vector instruction occupies its SIMD for 2 cycles (4 for the quarter-rate ones), HBM 6.3 TB/s achievable of 8 TB/s.
FETCH_SIZE corrections from profiles/r6_fetch_write_calibration.txt (tools/fetch_calib.hip): the counter reports HALF of the bytes of
coalesced streams of any width AND of the sorted, duplicate-heavy rank-word gathers; one 64-byte request per lane of a gather that
shares nothing; reads that hit the Infinity Cache ARE counted; memory atomics show up in WRITE_SIZE only (2.0 x the bytes touched)."""
import argparse, json, os, re, sys

CUS, SIMDS, CLK = 256, 1024, 2.4e9
HBM_ACHIEVABLE = 6.3e12


def kernel_counters(path, frag="classify_ei_kernel"):
    cur, out = None, {}
    for line in open(path):
        if not line.startswith(" "):
            cur = line.strip()
            continue
        if cur and frag in cur:
            m = re.match(r"\s+(\S+)\s+n=\d+\s+mean=(\S+)", line)
            if m:
                out[m.group(1)] = float(m.group(2))
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("summary")
    ap.add_argument("--traffic")
    ap.add_argument("--kernel-ms", type=float, default=None)
    ap.add_argument("--records", type=int, default=102499973)
    ap.add_argument("--algorithmic-bytes", type=float, default=4.093e9)
    ap.add_argument("--out", default=None)
    a = ap.parse_args()
    c = kernel_counters(a.summary)
    if not c:
        sys.exit("no classify_ei_kernel counters in " + a.summary)
    tiles = a.records / 64.0
    valu, salu, lds = c.get("SQ_INSTS_VALU", 0.0), c.get("SQ_INSTS_SALU", 0.0), c.get("SQ_INSTS_LDS", 0.0)
    fetch_b, write_b = c.get("FETCH_SIZE", 0.0) * 1024, c.get("WRITE_SIZE", 0.0) * 1024
    m = {
        "kernel": "classify_ei_kernel<false>", "records": a.records,
        "per_tile": {"valu": round(valu / tiles, 1), "salu": round(salu / tiles, 1), "lds": round(lds / tiles, 1)},
        # one scalar ALU per CU, one instruction per cycle
        "scalar_floor_ms": 1e3 * salu / CUS / CLK,
        # a wave64 vector instruction holds a SIMD-32 for 2 cycles (the guide's measured v_fma_f32 rate); 4 if everything ran at quarter rate
        "valu_floor_ms": 1e3 * valu * 2 / SIMDS / CLK,
        "valu_floor_ms_quarter_rate": 1e3 * valu * 4 / SIMDS / CLK,
        "lds_bank_conflict_rate": (c["SQ_LDS_BANK_CONFLICT"] / c["SQ_LDS_IDX_ACTIVE"]) if c.get("SQ_LDS_IDX_ACTIVE") else None,
        "wait_fraction_of_wave_cycles": (c["SQ_WAIT_ANY"] / c["SQ_WAVE_CYCLES"]) if c.get("SQ_WAVE_CYCLES") else None,
    }
    if fetch_b and write_b:
        # lower bound: every fetched byte reported at face value except the coalesced streams (x2: the algorithmic bytes + the qhash2 column
        # are streamed exactly once, so their share of the counter is half their size); upper bound: x2 on everything
        streamed = a.algorithmic_bytes + 4.0 * a.records
        lo = streamed + max(fetch_b - streamed / 2, 0.0) + write_b
        hi = 2 * fetch_b + write_b
        m.update({"FETCH_SIZE_bytes_reported": fetch_b, "WRITE_SIZE_bytes_reported": write_b,
                  "fabric_bytes_low": lo, "fabric_bytes_high": hi,
                  "traffic_floor_ms": 1e3 * lo / HBM_ACHIEVABLE, "traffic_floor_ms_high": 1e3 * hi / HBM_ACHIEVABLE,
                  "traffic_over_algorithmic": [round(lo / a.algorithmic_bytes, 3), round(hi / a.algorithmic_bytes, 3)]})
    if a.traffic and os.path.exists(a.traffic):
        rows = {}
        for line in open(a.traffic):
            p = line.split()
            if len(p) >= 4 and p[1].startswith("classify_ei_kernel"):
                try:
                    rows[p[0]] = (float(p[-2]), float(p[-1]))
                except ValueError:
                    pass
        if "tree" in rows:
            t = rows["tree"]
            split = {"all": {"fetch": t[0], "write": t[1]}}
            for name, what in (("abl4", "coverage_atomics"), ("abl8", "pairs"), ("abl1", "one_and_two_block_feature_stages"), ("abl17", "all_feature_stages")):
                if name in rows:
                    split[what] = {"fetch": t[0] - rows[name][0], "write": t[1] - rows[name][1]}
            if "abl17" in rows:
                split["phase_A_records_cigar_qhash2_tile_span"] = {"fetch": rows["abl17"][0], "write": rows["abl17"][1]}
            m["traffic_by_source_reported_bytes"] = split
    floors = [m["scalar_floor_ms"], m["valu_floor_ms"], m.get("traffic_floor_ms", 0.0)]
    m["binding_floor_ms"] = max(floors)
    m["frac_at_binding_floor"] = a.algorithmic_bytes / (max(floors) * 1e-3) / 8e12
    if a.kernel_ms:
        m["measured_kernel_ms_rocprof"] = a.kernel_ms
        m["measured_over_binding_floor"] = a.kernel_ms / max(floors)
    try:
        sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
        from rnaseqc_amd.hostinfo import k1_code_hash
        m["k1_code_hash"] = k1_code_hash()
    except Exception:
        pass
    s = json.dumps(m, indent=1)
    if a.out:
        open(a.out, "w").write(s + "\n")
    print(s)


if __name__ == "__main__":
    main()
