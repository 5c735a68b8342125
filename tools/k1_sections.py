#!/usr/bin/env python3
"""Static instruction mix of classify_ei_kernel between its RSQC_MARK section marks (no GPU needed).
Compiles rsqc_kernels.hip with the marks turned into assembler comments and counts, in text order, the instructions
between consecutive marks by issue class.  Basic blocks that the compiler moved out of line are attributed to the
section whose mark precedes them in the text -- good enough to see where phase A's instructions are."""
import re, subprocess, sys, os, collections, tempfile
root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "rnaseqc_amd", "csrc")
kern = sys.argv[1] if len(sys.argv) > 1 else "classify_ei_kernel"
extra = sys.argv[2:]
out = os.path.join(tempfile.gettempdir(), "k1_sections.s")
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-munsafe-fp-atomics", "-S", "--cuda-device-only",
                       '-DRSQC_MARK(sec)=asm volatile("; MARK_" #sec)', *extra, "rsqc_kernels.hip", "-o", out], cwd=root, stderr=subprocess.DEVNULL)
lines = open(out).read().split("\n")
start = next(i for i, l in enumerate(lines) if re.match(r"^_ZN4rsqc\d+" + kern + r"[A-Za-z0-9_]*:", l))
end = next(i for i in range(start, len(lines)) if re.match(r"^\.Lfunc_end", lines[i]))
def cls(op):
    if op.startswith("v_readlane") or op.startswith("v_writelane") or op.startswith("v_readfirstlane"): return "vlane"
    if op.startswith("v_cmp"): return "vcmp"
    if op.startswith("v_cndmask"): return "vcnd"
    if op.startswith("v_"): return "valu"
    if op.startswith("s_waitcnt"): return "wait"
    if op.startswith("s_cbranch") or op.startswith("s_branch"): return "br"
    if op.startswith("s_"): return "salu"
    if op.startswith("global_") or op.startswith("flat_") or op.startswith("buffer_") or op.startswith("scratch_"): return "vmem"
    if op.startswith("ds_"): return "lds"
    return None
sec = "pre"; order = []; tab = collections.defaultdict(collections.Counter)
for l in lines[start:end]:
    m = re.search(r"; MARK_(\d+)", l)
    if m:
        sec = "after_%s" % m.group(1)
        if sec not in order: order.append(sec)
        continue
    m = re.match(r"\s+([a-z_0-9]+)", l)
    if not m: continue
    c = cls(m.group(1))
    if c: tab[sec][c] += 1
cols = ["valu", "vcmp", "vcnd", "vlane", "salu", "br", "wait", "vmem", "lds"]
print("%-10s" % "section" + "".join("%7s" % c for c in cols) + "   VALU_all")
for s in ["pre"] + order:
    t = tab[s]
    print("%-10s" % s + "".join("%7d" % t[c] for c in cols) + "   %7d" % (t["valu"] + t["vcmp"] + t["vcnd"] + t["vlane"]))
for l in lines[start:end + 40]:
    if ".amdhsa_next_free_vgpr" in l or "private_segment_fixed_size" in l or "next_free_sgpr" in l: print(l.strip())
for l in lines:
    if re.search(r"; (ScratchSize|Occupancy|NumVgprs|NumSgprs|LDSByteSize)", l) and False: print(l)
