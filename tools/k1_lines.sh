#!/bin/bash
# Which source lines the main loop of a K1 instantiation was compiled from (static instruction counts; no GPU needed).
# Line 0 = compiler-generated code without a source position (spill restores, exec-mask bookkeeping).
# Usage: tools/k1_lines.sh [kernel-name-fragment] [top-N]
K=${1:-classify_count_kernel_w4r1E}; TOP=${2:-40}
D=$(mktemp -d)
(cd "$(dirname "$0")/../rnaseqc_amd/csrc" && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -munsafe-fp-atomics -gline-tables-only -S --cuda-device-only rsqc_kernels.hip -o $D/k.s 2>/dev/null)
python3 - "$D/k.s" "$K" "$TOP" <<'PY'
import collections, re, sys
lines = open(sys.argv[1]).read().split("\n")
files = {}
for l in lines:
    m = re.match(r'\s*\.file\s+(\d+)\s+"([^"]*)"(?:\s+"([^"]*)")?', l)
    if m: files[int(m.group(1))] = (m.group(3) or m.group(2)).split("/")[-1]
start = next(i for i, l in enumerate(lines) if re.match(r"^_ZN4rsqc\d+" + sys.argv[2] + r"[A-Za-z0-9_]*:", l))
end = next(i for i in range(start, len(lines)) if re.match(r"^\.Lfunc_end", lines[i]))   # (the kernel has early s_endpgm exits)
k = lines[start:end + 1]
labels = {m.group(1): i for i, l in enumerate(k) for m in [re.match(r"^(\.LBB\d+_\d+):", l)] if m}
back = [(i - labels[m.group(1)], labels[m.group(1)], i) for i, l in enumerate(k)
        for m in [re.match(r"\s*s_c?branch\S*\s+(\.LBB\d+_\d+)", l)] if m and m.group(1) in labels and labels[m.group(1)] < i]
_, a, b = max(back)
cur, cnt = None, collections.Counter()
for l in k[a:b + 1]:
    m = re.match(r"\s*\.loc\s+(\d+)\s+(\d+)", l)
    if m: cur = (files.get(int(m.group(1)), m.group(1)), int(m.group(2))); continue
    if re.match(r"\s*(s_|v_|global_|ds_|scratch_|buffer_|flat_)", l): cnt[cur] += 1
print("main loop: %d instructions" % sum(cnt.values()))
by_file = collections.Counter()
for (f, _), c in cnt.items(): by_file[f] += c
print("by file:", dict(by_file.most_common(6)))
for (f, ln), c in cnt.most_common(int(sys.argv[3])): print("%5d  %s:%d" % (c, f, ln))
PY
rm -rf $D
