#!/bin/bash
# Static view of a K1 instantiation (no GPU needed): registers, scratch, and the instruction mix of its main loop.
# Usage: tools/k1_isa.sh [kernel-name-fragment]   (default: classify_count_kernel_w4r1E)
K=${1:-classify_count_kernel_w4r1E}
D=$(mktemp -d)
(cd "$(dirname "$0")/../rnaseqc_amd/csrc" && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -munsafe-fp-atomics -S --cuda-device-only rsqc_kernels.hip -o $D/k.s 2>/dev/null)
python3 - "$D/k.s" "$K" <<'PY'
import re, sys
lines = open(sys.argv[1]).read().split("\n")
name = sys.argv[2]
start = next(i for i, l in enumerate(lines) if re.match(r"^_ZN4rsqc\d+" + name + r"[A-Za-z0-9_]*:", l))
end = next(i for i in range(start, len(lines)) if re.match(r"^\.Lfunc_end", lines[i]))   # (the kernel has early s_endpgm exits)
k = lines[start:end + 1]
meta = [l.strip() for l in lines if False]
labels = {m.group(1): i for i, l in enumerate(k) for m in [re.match(r"^(\.LBB\d+_\d+):", l)] if m}
back = []
for i, l in enumerate(k):
    m = re.match(r"\s*s_c?branch\S*\s+(\.LBB\d+_\d+)", l)
    if m and m.group(1) in labels and labels[m.group(1)] < i:
        back.append((i - labels[m.group(1)], labels[m.group(1)], i))
span, a, b = max(back)
body = k[a:b + 1]
isin = lambda l: re.match(r"\s*(s_|v_|global_|ds_|scratch_|buffer_|flat_)", l)
print("kernel instr %d, main loop instr %d" % (sum(1 for l in k if isin(l)), sum(1 for l in body if isin(l))))
for p in ["v_readlane", "v_writelane", "scratch_", "global_load", "global_atomic", "ds_", "s_waitcnt", "v_cmp", "v_cndmask", "s_cbranch"]:
    print("  %-14s %d" % (p, sum(1 for l in body if re.match(r"\s*" + p, l))))
for i, l in enumerate(lines):
    if ".amdhsa_kernel" in l and name in l:
        for x in lines[i:i + 60]:
            if re.search(r"next_free_vgpr|next_free_sgpr|private_segment_fixed_size", x): print(" ", x.strip())
        break
PY
rm -rf $D
