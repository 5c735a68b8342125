#!/usr/bin/env python3
"""Static view of classify_ei_kernel in an assembly listing (hipcc -S --cuda-device-only): registers, scalar spills, and the
instruction mix of the tile loop (outermost back edge).  usage: tools/k1_static.py a.s [b.s ...]"""
import re, sys, collections
def kernel(lines, frag="classify_ei_kernel"):
    if any(re.match(r"^_ZN4rsqc\d+" + frag + r"ILb0E", l) for l in lines): frag += "ILb0E"      # (the instance of runs without a BED)
    start = next(i for i, l in enumerate(lines) if re.match(r"^_ZN4rsqc\d+" + frag + r"[A-Za-z0-9_]*:", l))
    end = next(i for i in range(start, len(lines)) if re.match(r"^\.Lfunc_end", lines[i]))
    return lines[start:end + 1]
isin = lambda l: re.match(r"\s*(s_|v_|global_|ds_|scratch_|buffer_|flat_)", l)
def report(path):
    lines = open(path).read().split("\n")
    k = kernel(lines)
    labels = {m.group(1): i for i, l in enumerate(k) for m in [re.match(r"^(\.LBB\d+_\d+):", l)] if m}
    back = [(i - labels[m.group(1)], labels[m.group(1)], i) for i, l in enumerate(k)
            for m in [re.match(r"\s*s_c?branch\S*\s+(\.LBB\d+_\d+)", l)] if m and m.group(1) in labels and labels[m.group(1)] < i]
    _, a, b = max(back)
    body = k[a:b + 1]
    c = collections.Counter()
    for l in body:
        if not isin(l): continue
        op = l.split()[0]
        if op.startswith("v_readlane") or op.startswith("v_writelane"): c["lane"] += 1
        if op.startswith("v_"): c["valu"] += 1
        elif op.startswith("s_waitcnt"): c["wait"] += 1
        elif op.startswith("s_"): c["salu"] += 1
        elif op.startswith("ds_"): c["lds"] += 1
        elif op.startswith("global_") or op.startswith("flat_") or op.startswith("buffer_"): c["vmem"] += 1
        if op.startswith("flat_"): c["flat"] += 1
        if op.startswith("s_load"): c["s_load"] += 1
    meta = {}
    for i, l in enumerate(lines):
        if ".amdhsa_kernel" in l and "classify_ei_kernel" in l and ("ILb1E" not in l):
            for x in lines[i:i + 80]:
                m = re.search(r"\.amdhsa_(next_free_vgpr|next_free_sgpr|private_segment_fixed_size|group_segment_fixed_size)\s+(\d+)", x)
                if m: meta[m.group(1)] = int(m.group(2))
            break
    sp = None
    for l in lines:
        m = re.match(r";\s*SGPRSpill:\s*(\d+)", l) if False else None
    # the metadata comment block after the kernel
    txt = "\n".join(lines)
    m = re.search(r"; Function info:.*?classify_ei", txt, re.S)
    kend = txt.find(".Lfunc_end", txt.find("classify_ei_kernel"))
    tail = txt[kend:kend + 3000]
    sg = re.search(r"; sgpr_spill_count:\s*(\d+)|; SGPRSpill.*?(\d+)", tail)
    vg = re.search(r"; vgpr_spill_count:\s*(\d+)", tail)
    occ = re.search(r"; Occupancy:\s*(\d+)", tail)
    print("%-22s vgpr %s sgpr %s lds %s scratch %s | sgpr_spill %s vgpr_spill %s occ %s | loop: valu %d (lane %d) salu %d wait %d lds %d vmem %d flat %d s_load %d" % (
        path.split("/")[-1], meta.get("next_free_vgpr"), meta.get("next_free_sgpr"), meta.get("group_segment_fixed_size"), meta.get("private_segment_fixed_size"),
        sg and (sg.group(1) or sg.group(2)), vg and vg.group(1), occ and occ.group(1), c["valu"], c["lane"], c["salu"], c["wait"], c["lds"], c["vmem"], c["flat"], c["s_load"]))
for p in sys.argv[1:]: report(p)
