"""Extracts the figures tests/test_golden_reference.py needs from the reference's golden OUTPUT files
(/root/reference/test_data/*.output/).  Run in the build container; commits data only."""
import gzip
import json
import os

REF = "/root/reference/test_data"
out = {}
for name, prefix in [("chr1", "chr1.output/chr1.bam"), ("downsampled", "downsampled.output/downsampled.bam"),
                     ("single_pair", "single_pair.output/single_pair.bam")]:
    d = {}
    m = {}
    for line in open(os.path.join(REF, prefix + ".metrics.tsv")):
        k, v = line.rstrip("\n").split("\t")
        m[k] = v
    d["metrics"] = m
    cov = os.path.join(REF, prefix + ".coverage.tsv")
    if os.path.exists(cov):
        means, stds, cvs, zero = [], [], [], 0
        for i, line in enumerate(open(cov)):
            if i == 0:
                continue
            g, a, s, c = line.rstrip("\n").split("\t")
            if a == "0" and s == "0" and c in ("nan", "-nan"):
                zero += 1
                continue
            means.append(float(a)); stds.append(float(s))
            if c not in ("nan", "-nan", "inf", "-inf"):
                cvs.append(float(c))
        d["coverage_mean_nonzero"], d["coverage_std_nonzero"], d["coverage_cv_finite"], d["n_zero_rows"] = means, stds, cvs, zero
    fs = os.path.join(REF, prefix + ".fragmentSizes.txt")
    if os.path.exists(fs):
        d["fragment_sizes"] = {l.split("\t")[0]: int(l.split("\t")[1]) for i, l in enumerate(open(fs)) if i}
    ex = os.path.join(REF, prefix + ".exon_reads.gct.gz")
    if os.path.exists(ex) and name != "single_pair":
        lines = gzip.open(ex, "rt").read().split("\n")
        vals = [float(l.split("\t")[2]) for l in lines[3:] if l]
        d["exon_gct_header_rows"] = int(lines[1].split("\t")[0])
        d["exon_gct_rows"] = len(vals)
        d["exon_gct_nonzero_rows"] = sum(1 for v in vals if v > 0)
        d["exon_reads_sum"] = sum(vals)
        g = gzip.open(os.path.join(REF, prefix + ".gene_reads.gct.gz"), "rt").read().split("\n")
        d["gene_reads_sum"] = sum(int(l.split("\t")[2]) for l in g[3:] if l)
        f = gzip.open(os.path.join(REF, prefix + ".gene_fragments.gct.gz"), "rt").read().split("\n")
        d["gene_fragments_sum"] = sum(int(l.split("\t")[2]) for l in f[3:] if l)
    out[name] = d
# --fasta run of the chr1 case (CRAM input): the GC histogram and the metrics.tsv it belongs to
d = {"metrics_keys": [], "metrics": {}}
for line in open(os.path.join(REF, "chr1.output/chr1.cram.metrics.tsv")):
    k, v = line.rstrip("\n").split("\t")
    d["metrics"][k] = v; d["metrics_keys"].append(k)
d["gc_bins"] = [int(l.split("\t")[1]) for i, l in enumerate(open(os.path.join(REF, "chr1.output/chr1.cram.gc_content.tsv"))) if i]
d["gc_bin_labels"] = [l.split("\t")[0] for i, l in enumerate(open(os.path.join(REF, "chr1.output/chr1.cram.gc_content.tsv"))) if i]
out["chr1_cram"] = d
# --legacy run of the downsampled case
d = {"metrics_keys": [], "metrics": {}}
for line in open(os.path.join(REF, "legacy.output/downsampled.bam.metrics.tsv")):
    k, v = line.rstrip("\n").split("\t")
    d["metrics"][k] = v; d["metrics_keys"].append(k)
out["legacy"] = d
json.dump(out, open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "reference_known_answers.json"), "w"))
print({k: list(v.keys()) for k, v in out.items()})

# The chr1 golden GCT tables themselves (ids, descriptions, values as printed): data for the byte-for-byte check of the GCT
# writers and of "Genes Detected" (tests/test_golden_reference.py::test_gct_writers_reproduce_chr1_golden).  ~120 KB gzipped.
tables = {}
for f in ("gene_reads", "gene_fragments", "exon_reads", "gene_tpm"):
    t = gzip.open(os.path.join(REF, "chr1.output/chr1.bam.%s.gct.gz" % f), "rt").read().split("\n")
    rows = [l.split("\t") for l in t[3:] if l]
    tables[f] = {"header": t[:3], "id": [r[0] for r in rows], "desc": [r[1] for r in rows], "value": [r[2] for r in rows]}
with gzip.open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "chr1_gct_tables.json.gz"), "wt") as fh:
    json.dump(tables, fh)
