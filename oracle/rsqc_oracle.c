/*
 * rsqc_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * A single-threaded CPU restatement, in plain C, of the per-read hot path of
 * getzlab/rnaseqc 2.4.3 (default and --legacy rules, --fasta GC statistics).  It deliberately keeps the
 * reference's *streaming* algorithm -- one start-sorted feature list per
 * contig, destructively front-trimmed as the coordinate-sorted input advances,
 * linearly scanned per CIGAR block -- whereas the HIP product queries a static
 * index; agreement between the two is therefore also a test of the
 * "static query == trimmed window" argument.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library.  The product (rnaseqc_amd/csrc) never does.
 *
 * PARITY PINNING STATUS
 *   - coverage/bias/statistics part (oracle_gene_exit, bias, medians): pinned
 *     against the reference's own src/Metrics.cpp compiled unmodified
 *     (oracle/Makefile -> oracle/_ref/libref_metrics.so) and against the
 *     reference's golden outputs (tests/test_golden_reference.py).
 *   - per-read classification part (oracle_process_record): PARITY UNPINNED by
 *     execution.  src/RNASeQC.cpp and src/Expression.cpp need SeqLib@7e1f982 +
 *     htslib + boost, none of which exist in this image, and writing stand-ins
 *     for them is not allowed; every step below cites the reference line it
 *     restates, and invariants of the reference's golden outputs are checked.
 *
 * Each function cites the reference file:line it follows (paths relative to
 * the reference root).
 */
#include "rnaseqc_amd.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define ORACLE_API __attribute__((visibility("default")))

/* ------------------------------------------------------------------ types */

typedef struct {
    int32_t  start, end;      /* 1-based closed (src/GTF.h:29-37)            */
    uint8_t  flags;           /* RSQC_FF_*                                   */
    uint8_t  is_gene;         /* FeatureType::Gene vs Exon                   */
    uint32_t id;              /* gene id (gene row) / exon id (exon row)     */
    uint32_t gene;            /* gene id named by the row's gene_id          */
    uint32_t row;             /* index in the sorted gene/exon row arrays    */
    uint32_t order;           /* position in the GTF (rsqc_annotation.*_row_order), when given */
} feat_t;

typedef struct {
    feat_t *f;
    size_t  n, head;          /* head = front of the std::list after trimming */
} flist_t;

typedef struct {              /* fragmentTracker[gene]: unordered_set<string> */
    uint64_t *h;              /* 0 = empty slot; hashes are forced non-zero   */
    char    **s;              /* exact names when the batch carries them      */
    uint32_t *h2;             /* second hash (rsqc_batch.qhash2) for the hash-only mode */
    size_t    cap, n;
} nameset_t;

typedef struct {              /* Collector entry / CoverageEntry              */
    uint32_t gene, exon_row;
    double   frac;
    int64_t  offset;
    uint32_t length;
} staged_t;

typedef struct {              /* fragmentSizeMetrics pending mate             */
    uint64_t h; uint32_t h2 /* second hash of the name when the batch carries one and no strings (hash-only mode) */; char *s; int32_t bed; int64_t endpos; int used;
} pending_t;

typedef struct oracle_ctx {
    rsqc_params p;
    /* annotation (deep copy) */
    int32_t n_ref, n_contigs, n_genes, n_listed, n_exons;
    flist_t *feat;            /* [n_contigs] merged gene+exon rows            */
    uint8_t *owned;
    int32_t *ex_start, *ex_end; uint32_t *ex_id, *ex_gene; uint8_t *ex_flags;
    uint8_t *g_globin;        /* [n_genes]                                    */
    uint8_t *g_row_flags;     /* [n_listed] flags of the gene row by gene id  */
    uint32_t *ge_off, *ge_row;/* exonsForGene CSR                             */
    /* BED */
    flist_t *bed;             /* [n_contigs]                                  */
    int have_bed;
    uint32_t frag_remaining;
    pending_t *pend; size_t pend_cap, pend_n;
    int64_t *fs_size; uint64_t *fs_count; size_t fs_n, fs_cap;
    /* --fasta */
    int have_ref;
    uint8_t **ref_seq; uint64_t *ref_len;     /* [n_contigs], NULL = the FASTA index lacks the contig */
    pending_t *gcp; size_t gcp_cap, gcp_n;    /* gcContentFragmentTracker, src/RNASeQC.cpp:169 */
    uint64_t gc_bins[RSQC_GC_BINS]; uint64_t gc_oob;
    double *exon_gc;                          /* by exon id */
    /* streaming state */
    int32_t current_contig;   /* current_chrom (src/RNASeQC.cpp:210), -1 = none */
    int32_t read_length;      /* readLength (src/RNASeQC.cpp:205)             */
    /* accumulators */
    uint64_t counters[RSQC_N_COUNTERS];
    double  *gene_reads, *gene_unique, *gene_frag; /* map<string,double>      */
    double  *exon_reads; uint8_t *exon_hit;        /* by exon id              */
    nameset_t *tracker;       /* [n_genes]                                    */
    uint64_t **cov;           /* [n_exons rows] lazily allocated vectors      */
    uint8_t *seen;            /* BaseCoverage::seen by gene id                */
    /* per-gene outputs */
    double *cov_mean, *cov_std, *cov_cv; uint8_t *cov_valid;
    double *exon_cv; uint8_t *exon_cv_valid;       /* by exon id              */
    uint64_t *bias3, *bias5;
    uint32_t *exit_order; uint32_t n_exit;         /* coverage.tsv row order  */
    /* results mirrors */
    uint64_t *r_reads, *r_unique, *r_frag;
    /* test trace (oracle_enable_trace): what a contig-sharded run needs from a shard for the host-side merge -- the
     * (span, l_qseq) of every record that reaches src/RNASeQC.cpp:275, the record counts at the end of every submit,
     * and every fragment-size sample with the file index of the record that completed it                          */
    int trace;
    /* Collector trace (oracle_enable_collector_trace): every Collector::add / queryGene / collect this restatement
     * performs, in order, so that the REFERENCE's own class (compiled in oracle/_ref) can be driven with the same calls */
    int ctrace; uint32_t ctr_read;
    uint8_t *ct_kind; uint32_t *ct_read, *ct_gene, *ct_exon; double *ct_frac; uint8_t *ct_query; size_t ct_n, ct_cap;
    uint64_t cur_file_index;
    uint32_t *tr_span; int32_t *tr_lq; size_t tr_n, tr_cap;
    uint64_t *tr_batch_end, *tr_batch_file; size_t tr_nb, tr_bcap;
    uint64_t *tr_sample_file; uint32_t *tr_sample_size; size_t tr_ns, tr_scap;
    int error;
    char errmsg[256];
} oracle_ctx;

/* ------------------------------------------------------------ small utils */
static void *xrealloc(void *q, size_t sz);
static void ct_push(struct oracle_ctx *c, int kind, uint32_t gene, uint32_t exon, double frac, int query) {
    if (c->ct_n == c->ct_cap) {
        c->ct_cap = c->ct_cap ? c->ct_cap * 2 : 4096;
        c->ct_kind = xrealloc(c->ct_kind, c->ct_cap); c->ct_read = xrealloc(c->ct_read, c->ct_cap * 4);
        c->ct_gene = xrealloc(c->ct_gene, c->ct_cap * 4); c->ct_exon = xrealloc(c->ct_exon, c->ct_cap * 4);
        c->ct_frac = xrealloc(c->ct_frac, c->ct_cap * 8); c->ct_query = xrealloc(c->ct_query, c->ct_cap);
    }
    c->ct_kind[c->ct_n] = (uint8_t)kind; c->ct_read[c->ct_n] = c->ctr_read; c->ct_gene[c->ct_n] = gene; c->ct_exon[c->ct_n] = exon;
    c->ct_frac[c->ct_n] = frac; c->ct_query[c->ct_n] = (uint8_t)query; c->ct_n++;
}

static void *xcalloc(size_t n, size_t sz) {
    void *p = calloc(n ? n : 1, sz ? sz : 1);
    if (!p) { fprintf(stderr, "oracle: out of memory\n"); abort(); }
    return p;
}
static void *xrealloc(void *q, size_t sz) {
    void *p = realloc(q, sz ? sz : 1);
    if (!p) { fprintf(stderr, "oracle: out of memory\n"); abort(); }
    return p;
}
static void *dup_array(const void *src, size_t n, size_t sz) {
    void *p = xcalloc(n, sz);
    if (n && src) memcpy(p, src, n * sz);
    return p;
}

/* ------------------------------------------------------------- name sets */

static int nameset_insert(nameset_t *t, uint64_t h, uint32_t h2, int has2, const char *s, size_t len) {
    /* returns 1 if newly inserted (fragmentTracker[gene].count(qname) == 0,
       src/Expression.cpp:383-387).  With the names: string comparison, the reference's.  Without: the identity the device works
       with -- the 64-bit hash, and the second hash as well when the batch carries one (rsqc_batch.qhash2)                       */
    if (h == 0) h = 0x9e3779b97f4a7c15ull;
    if ((t->n + 1) * 2 > t->cap) {
        size_t ncap = t->cap ? t->cap * 2 : 16;
        uint64_t *nh = xcalloc(ncap, sizeof(uint64_t));
        char **ns = xcalloc(ncap, sizeof(char *));
        uint32_t *n2 = xcalloc(ncap, sizeof(uint32_t));
        for (size_t i = 0; i < t->cap; ++i) if (t->h[i]) {
            size_t j = (size_t)(t->h[i] * 0x9e3779b97f4a7c15ull >> 17) & (ncap - 1);
            while (nh[j]) j = (j + 1) & (ncap - 1);
            nh[j] = t->h[i]; ns[j] = t->s[i]; n2[j] = t->h2[i];
        }
        free(t->h); free(t->s); free(t->h2);
        t->h = nh; t->s = ns; t->h2 = n2; t->cap = ncap;
    }
    size_t j = (size_t)(h * 0x9e3779b97f4a7c15ull >> 17) & (t->cap - 1);
    while (t->h[j]) {
        if (t->h[j] == h) {
            if (!s) { if (!has2 || t->h2[j] == h2) return 0; }      /* hash-only mode */
            else if (t->s[j] && strlen(t->s[j]) == len && memcmp(t->s[j], s, len) == 0) return 0;
        }
        j = (j + 1) & (t->cap - 1);
    }
    t->h[j] = h; t->h2[j] = h2;
    if (s) { t->s[j] = xcalloc(len + 1, 1); memcpy(t->s[j], s, len); }
    t->n++;
    return 1;
}
static void nameset_clear(nameset_t *t) {
    if (t->s) for (size_t i = 0; i < t->cap; ++i) free(t->s[i]);
    free(t->h); free(t->s); free(t->h2);
    memset(t, 0, sizeof(*t));
}

/* ------------------------------------------- quirky median and statistics */

/* computeMedian, src/Metrics.h:147-160: advance (size-1)/2; ODD size -> mean
 * of [mid],[mid+1]; EVEN size -> [mid]; size 1 -> [0]; size 0 -> range_error */
static int median_u64(const uint64_t *v, unsigned long size, double *out) {
    if (size == 0) return RSQC_ERR_EMPTY_MEDIAN;
    if (size == 1) { *out = (double)v[0]; return 0; }
    unsigned long mid = (size - 1) / 2;
    if (size % 2) *out = ((double)v[mid] + (double)v[mid + 1]) / 2.0;
    else *out = (double)v[mid];
    return 0;
}
static int median_f64(const double *v, unsigned long size, double *out) {
    if (size == 0) return RSQC_ERR_EMPTY_MEDIAN;
    if (size == 1) { *out = v[0]; return 0; }
    unsigned long mid = (size - 1) / 2;
    if (size % 2) *out = (v[mid] + v[mid + 1]) / 2.0;
    else *out = v[mid];
    return 0;
}
ORACLE_API int oracle_median_f64(const double *v, uint64_t n, double *out) {
    return median_f64(v, (unsigned long)n, out);
}
static int cmp_u64(const void *a, const void *b) {
    uint64_t x = *(const uint64_t *)a, y = *(const uint64_t *)b;
    return x < y ? -1 : x > y;
}
static int cmp_f64(const void *a, const void *b) {
    double x = *(const double *)a, y = *(const double *)b;
    return x < y ? -1 : x > y;
}
static int cmp_i64(const void *a, const void *b) {
    int64_t x = *(const int64_t *)a, y = *(const int64_t *)b;
    return x < y ? -1 : x > y;
}

/* getStatistics, src/Metrics.h:166-186: sorts in place; avg = sum d/n
 * (sequential); median (quirky); std = sqrt(sum (d-avg)^2/n); MAD*1.4826 on
 * sorted deviations.  out = {avg, median, std, mad}; empty -> NaNs          */
ORACLE_API void oracle_statistics(double *d, uint64_t n, double out[4]) {
    if (!n) { out[0] = out[1] = out[2] = out[3] = NAN; return; }
    qsort(d, n, sizeof(double), cmp_f64);
    const double size = (double)n;
    double median; median_f64(d, n, &median);
    double avg = 0.0, sd = 0.0;
    double *dev = xcalloc(n, sizeof(double));
    for (uint64_t i = 0; i < n; ++i) { avg += d[i] / size; dev[i] = fabs(d[i] - median); }
    qsort(dev, n, sizeof(double), cmp_f64);
    double mad; median_f64(dev, n, &mad); mad *= 1.4826;
    for (uint64_t i = 0; i < n; ++i) sd += pow(d[i] - avg, 2.0) / size;
    sd = pow(sd, 0.5);
    free(dev);
    out[0] = avg; out[1] = median; out[2] = sd; out[3] = mad;
}

/* Fasta::getSeq(contig, start, end) for 0-based half-open [start, end), src/Fasta.cpp:104-140: whole 1e6-base pages
 * are fetched with bioio::read_fasta_contig (bioio.hpp:298-331, which clips a page at the contig's length) and the
 * requested range is cut out with substr -- i.e. the range clipped at the contig end.  Ranges that start before 0 or
 * at/after the contig end leave the reference in error paths (page of another contig / std::out_of_range) and are not
 * defined here: *len = 0.                                                                                            */
static const uint8_t *get_seq(const oracle_ctx *c, int contig, int64_t start, int64_t end, size_t *len) {
    *len = 0;
    if (!c->ref_seq[contig] || start < 0 || (uint64_t)start >= c->ref_len[contig] || end <= start) return NULL;
    if ((uint64_t)end > c->ref_len[contig]) end = (int64_t)c->ref_len[contig];
    *len = (size_t)(end - start);
    return c->ref_seq[contig] + start;
}
/* gc(), src/Fasta.cpp:67-74: adds 1.0/size once per G/g/C/c base, in sequence order */
static double gc_content(const uint8_t *seq, size_t len) {
    if (len == 0) return -1;
    double content = 0.0, size = (double)len;
    for (size_t i = 0; i < len; ++i)
        if (seq[i] == 'G' || seq[i] == 'g' || seq[i] == 'C' || seq[i] == 'c') content += 1.0 / size;
    return content;
}


/* ---------------------------------------------------- coverage and bias */

/* BiasCounter::computeBias, src/Metrics.cpp:160-235.  Mutates cov/len (the
 * in-place trim that then feeds the gene statistics, quirk Q14).           */
static int compute_bias(oracle_ctx *c, uint32_t gene, uint8_t gene_flags,
                        uint64_t *cov, size_t *plen) {
    size_t len = *plen;
    const int W = c->p.bias_window, OFF = c->p.bias_offset;
    if (len < c->p.bias_gene_length) return 0;                     /* :163 */
    uint64_t peak = 0; unsigned peak_pos = 0;
    for (unsigned i = 0; i < len; ++i) if (cov[i] > peak) { peak_pos = i; peak = cov[i]; } /* :166-170 */
    size_t cur = peak_pos;                                         /* :171 */
    for (int i = 0; i < W / 2 && cur != len; ++i) ++cur;           /* :174 */
    unsigned long n = 0;
    for (int i = 0; i < W && cur != 0; ++i) { --cur; ++n; }        /* :176 (values discarded) */
    double gate;
    int rc = median_u64(cov + cur, n, &gate);                      /* :178 positional, unsorted */
    if (rc) return rc;
    if (gate >= 100) {                                             /* :181 */
        uint64_t *pc = dup_array(cov, len, sizeof(uint64_t));      /* :182-183 */
        qsort(pc, len, sizeof(uint64_t), cmp_u64);
        size_t z = 0; while (z < len && pc[z] == 0) ++z;           /* :185-187 */
        size_t nnz = len - z;
        uint64_t lower = pc[z + (size_t)((double)nnz * 0.05)];     /* :189 */
        free(pc);
        size_t lead = 0; while (lead < len && cov[lead] <= lower) ++lead;   /* :193-199 */
        memmove(cov, cov + lead, (len - lead) * sizeof(uint64_t));
        len -= lead;
        while (len > 0 && cov[len - 1] <= lower) --len;            /* :202-205 */
        if (len >= c->p.bias_gene_length) {                        /* :208 */
            double *l = xcalloc((size_t)(W > 0 ? W : 0) + 1, sizeof(double));
            double *r = xcalloc((size_t)(W > 0 ? W : 0) + 1, sizeof(double));
            size_t nl = 0, nr = 0;
            /* :214  for (unsigned int i = offset; i < offset + windowSize && i < size; ++i) */
            for (unsigned int i = (unsigned int)OFF; i < (unsigned int)(OFF + W) && i < len; ++i)
                l[nl++] = (double)cov[i];
            /* :216  for (int i = size - (windowSize + offset); i >= 0 && i < size - offset; ++i) */
            for (int i = (int)(len - (size_t)(W + OFF)); i >= 0 && (size_t)i < len - (size_t)OFF; ++i)
                r[nr++] = (double)cov[i];
            qsort(l, nl, sizeof(double), cmp_f64);
            qsort(r, nr, sizeof(double), cmp_f64);
            double ml, mr;
            rc = median_f64(r, nr, &mr); if (!rc) rc = median_f64(l, nl, &ml);
            free(l); free(r);
            if (rc) return rc;
            if (gene < (uint32_t)c->n_listed) {
                /* unsigned long += double: truncation (src/Metrics.h:76-77) */
                if ((gene_flags & RSQC_FF_STRAND_MASK) == RSQC_STRAND_FORWARD) {  /* :220-228 */
                    c->bias3[gene] = (uint64_t)((double)c->bias3[gene] + mr);
                    c->bias5[gene] = (uint64_t)((double)c->bias5[gene] + ml);
                } else {
                    c->bias3[gene] = (uint64_t)((double)c->bias3[gene] + ml);
                    c->bias5[gene] = (uint64_t)((double)c->bias5[gene] + mr);
                }
            }
        }
    }
    *plen = len;
    return 0;
}

/* BaseCoverage::compute + computeCoverage, src/Metrics.cpp:132-151,265-337 */
static int gene_exit(oracle_ctx *c, const feat_t *g, int contig) {
    const uint32_t gene = g->id;
    if (!c->owned[contig]) { c->seen[gene] = 1; return 0; }  /* multi-GPU shard: another rank owns this contig */
    const uint32_t e0 = c->ge_off[gene], e1 = c->ge_off[gene + 1], ne = e1 - e0;
    const unsigned mask_size = c->p.coverage_mask;
    size_t total = 0;
    for (uint32_t k = e0; k < e1; ++k) {
        uint32_t row = c->ge_row[k];
        size_t elen = (size_t)(c->ex_end[row] - c->ex_start[row] + 1);
        if (!c->cov[row]) c->cov[row] = xcalloc(elen, sizeof(uint64_t));   /* :137-138 */
        total += elen;
    }
    /* masks :267-279 */
    uint8_t **mask = xcalloc(ne, sizeof(uint8_t *));
    unsigned rem = mask_size;
    for (uint32_t k = 0; k < ne; ++k) {
        uint32_t row = c->ge_row[e0 + k];
        size_t elen = (size_t)(c->ex_end[row] - c->ex_start[row] + 1);
        mask[k] = xcalloc(elen, 1); memset(mask[k], 1, elen);
        for (size_t j = 0; j < elen && rem; ++j, --rem) mask[k][j] = 0;
    }
    rem = mask_size;
    for (int k = (int)ne - 1; k >= 0 && rem; --k) {
        uint32_t row = c->ge_row[e0 + (uint32_t)k];
        long elen = c->ex_end[row] - c->ex_start[row] + 1;
        for (long j = elen - 1; j >= 0 && rem; --j, --rem) mask[k][j] = 0;
    }
    uint64_t *gc = xcalloc(total, sizeof(uint64_t));
    size_t glen = 0;
    for (uint32_t k = 0; k < ne; ++k) {                                     /* :280-309 */
        uint32_t row = c->ge_row[e0 + k];
        size_t elen = (size_t)(c->ex_end[row] - c->ex_start[row] + 1);
        const uint64_t *ec = c->cov[row];
        double mean = 0.0, sd = 0.0, size = 0.0;
        for (size_t j = 0; j < elen; ++j) if (mask[k][j]) size += 1.0;
        if (size > 0) {
            for (size_t j = 0; j < elen; ++j) if (mask[k][j]) mean += (double)ec[j] / size;
            for (size_t j = 0; j < elen; ++j) if (mask[k][j]) sd += pow((double)ec[j] - mean, 2.0) / size;
            sd = pow(sd, 0.5);
            sd /= mean;
            if (!(isnan(sd) || isinf(sd))) {
                c->exon_cv[c->ex_id[row]] = sd; c->exon_cv_valid[c->ex_id[row]] = 1;
                if (c->have_ref) {                                          /* :299-303: getSeq(chr, start, start + length) with the
                                                                               1-BASED start used as a 0-based offset */
                    size_t slen = 0; const uint8_t *sq = NULL;
                    if (c->ref_seq[contig]) sq = get_seq(c, contig, c->ex_start[row], (int64_t)c->ex_start[row] + (int64_t)elen, &slen);
                    c->exon_gc[c->ex_id[row]] = c->ref_seq[contig] ? gc_content(sq, slen) : -1.0;
                }
            }
        }
        memcpy(gc + glen, ec, elen * sizeof(uint64_t));
        glen += elen;
        free(mask[k]);
    }
    free(mask);
    int rc = compute_bias(c, gene, g->flags, gc, &glen);                    /* :311 */
    if (rc) { free(gc); return rc; }
    /* :314-322 mask the (possibly trimmed) gene vector */
    size_t lo = 0, hi = glen;
    if (mask_size) {
        hi = (mask_size > glen) ? 0 : glen - mask_size;
        size_t sz = hi;
        if (sz) lo = (mask_size > sz) ? sz : mask_size;
    }
    double size = (double)(hi - lo);
    if (size > 0) {                                                         /* :325-333 */
        double avg = 0.0, sd = 0.0;
        for (size_t j = lo; j < hi; ++j) avg += (double)gc[j] / size;
        for (size_t j = lo; j < hi; ++j) sd += pow((double)gc[j] - avg, 2.0) / size;
        sd = pow(sd, 0.5);
        if (gene < (uint32_t)c->n_listed) {
            c->cov_mean[gene] = avg; c->cov_std[gene] = sd; c->cov_cv[gene] = sd / avg;
            c->cov_valid[gene] = 1;
        }
    }
    free(gc);
    for (uint32_t k = e0; k < e1; ++k) { free(c->cov[c->ge_row[k]]); c->cov[c->ge_row[k]] = NULL; } /* :148-149 */
    c->seen[gene] = 1;                                                      /* :150 */
    if (gene < (uint32_t)c->n_listed) c->exit_order[c->n_exit++] = gene;
    return 0;
}

/* trimFeatures(alignment, features, coverage), src/Expression.cpp:80-93 */
static int trim_features(oracle_ctx *c, flist_t *fl, int32_t pos0) {
    const int contig = (int)(fl - c->feat);
    while (fl->head < fl->n && fl->f[fl->head].end < pos0) {
        const feat_t *f = &fl->f[fl->head];
        if (f->is_gene) {
            int rc = gene_exit(c, f, contig); if (rc) return rc;
            nameset_clear(&c->tracker[f->id]);
        }
        fl->head++;
    }
    return 0;
}
/* dropFeatures, src/Expression.cpp:96-103 */
static int drop_features(oracle_ctx *c, flist_t *fl) {
    const int contig = (int)(fl - c->feat);
    for (size_t i = fl->head; i < fl->n; ++i) if (fl->f[i].is_gene) {
        int rc = gene_exit(c, &fl->f[i], contig); if (rc) return rc;
        nameset_clear(&c->tracker[fl->f[i].id]);
    }
    fl->head = fl->n;
    return 0;
}

/* intersectInterval, src/GTF.cpp:171-179 (block = [bs, be] with be used inclusively) */
static int intersects(int64_t bs, int64_t be, const feat_t *f) {
    return (f->start >= bs && f->start <= be) || (f->end >= bs && f->end <= be) ||
           (bs >= f->start && bs <= f->end);
}
/* partialIntersect, src/GTF.cpp:181-186 */
static int64_t partial_intersect(const feat_t *t, int64_t bs, int64_t be) {
    if (!intersects(bs, be, t)) return 0;
    int64_t a = t->end < be - 1 ? t->end : be - 1, b = t->start > bs ? t->start : bs;
    return 1 + a - b;
}

/* ----------------------------------------------------------- record view */

typedef struct {
    int32_t tid, pos, mpos, isize, l_qseq, nm;
    uint32_t flag, mapq, tagbits, n_cigar;
    const uint32_t *cigar;
    uint64_t qhash; uint32_t qhash2; int has_qhash2;
    const char *qname; size_t qname_len;
} rec_t;

typedef struct { int64_t start, end; } block_t;

/* bam_endpos as used by SeqLib::BamRecord::PositionEnd (htslib, not in tree):
 * pos + reference length of the CIGAR; pos + 1 for unmapped / CIGAR-less
 * records or when the CIGAR consumes no reference base (htslib >= 1.10).    */
static int32_t end_position(const rec_t *r) {
    if ((r->flag & RSQC_FUNMAP) || r->n_cigar == 0) return r->pos + 1;
    int64_t rl = 0;
    for (uint32_t i = 0; i < r->n_cigar; ++i) {
        uint32_t op = r->cigar[i] & 0xf, len = r->cigar[i] >> 4;
        if (op == 0 || op == 2 || op == 3 || op == 7 || op == 8) rl += len;
    }
    if (rl == 0) rl = 1;
    return (int32_t)(r->pos + rl);
}

/* fragmentSizeMetrics, src/Expression.cpp:482-540 */
static void fragment_size(oracle_ctx *c, const rec_t *r, const block_t *blocks, size_t nb) {
    flist_t *fl = &c->bed[r->tid];
    int first = 1, same = 1; int32_t name = -1;
    while (fl->head < fl->n && fl->f[fl->head].end < r->pos) fl->head++;   /* :489, Expression.cpp:69-78 */
    for (size_t b = 0; same && b < nb; ++b) {
        int64_t bs = blocks[b].start, be = blocks[b].end;
        size_t hits = 0; const feat_t *hit = NULL;
        for (size_t i = fl->head; i < fl->n && fl->f[i].start <= be; ++i)  /* intersectBlock :106-117 */
            if (intersects(bs, be, &fl->f[i])) { if (!hits) hit = &fl->f[i]; ++hits; }
        if (hits == 1 && partial_intersect(hit, bs, be) == be - bs) {       /* :494 */
            if (first) name = (int32_t)hit->id;
            else if (name != (int32_t)hit->id) { same = 0; break; }
        } else same = 0;
        first = 0;
    }
    if (!(same && name >= 0)) return;                                       /* :508 */
    /* fragments.find(Qname) :511 */
    pending_t *found = NULL;
    uint64_t h = r->qhash ? r->qhash : 0x9e3779b97f4a7c15ull;
    if (c->pend_cap) {
        size_t j = (size_t)(h * 0x9e3779b97f4a7c15ull >> 17) & (c->pend_cap - 1);
        while (c->pend[j].used) {
            if (c->pend[j].used == 1 && c->pend[j].h == h &&
                (r->qname ? (c->pend[j].s && strlen(c->pend[j].s) == r->qname_len && memcmp(c->pend[j].s, r->qname, r->qname_len) == 0)
                          : (!r->has_qhash2 || c->pend[j].h2 == r->qhash2))) { found = &c->pend[j]; break; }   /* names; else the 96-bit identity */
            j = (j + 1) & (c->pend_cap - 1);
        }
    }
    int32_t endpos = end_position(r);
    if (!found) {                                                           /* :512-516 */
        if ((c->pend_n + 1) * 2 > c->pend_cap) {
            size_t ncap = c->pend_cap ? c->pend_cap * 2 : 1024;
            pending_t *np = xcalloc(ncap, sizeof(pending_t));
            for (size_t i = 0; i < c->pend_cap; ++i) if (c->pend[i].used == 1) {
                size_t j = (size_t)(c->pend[i].h * 0x9e3779b97f4a7c15ull >> 17) & (ncap - 1);
                while (np[j].used) j = (j + 1) & (ncap - 1);
                np[j] = c->pend[i];
            }
            free(c->pend); c->pend = np; c->pend_cap = ncap;
            /* tombstones were dropped by the rehash */
        }
        size_t j = (size_t)(h * 0x9e3779b97f4a7c15ull >> 17) & (c->pend_cap - 1);
        while (c->pend[j].used == 1) j = (j + 1) & (c->pend_cap - 1);
        int was_tomb = c->pend[j].used == 2;
        c->pend[j].used = 1; c->pend[j].h = h; c->pend[j].h2 = r->has_qhash2 ? r->qhash2 : 0u; c->pend[j].bed = name; c->pend[j].endpos = endpos;
        c->pend[j].s = NULL;
        if (r->qname) { c->pend[j].s = xcalloc(r->qname_len + 1, 1); memcpy(c->pend[j].s, r->qname, r->qname_len); }
        if (!was_tomb) c->pend_n++;
    } else if (found->bed == name) {                                        /* :517 */
        if ((r->flag & RSQC_FMREVERSE) || !(r->flag & RSQC_FREVERSE) ||
            endpos <= found->endpos || r->pos == r->mpos) return;           /* :528 */
        int64_t key = r->isize < 0 ? -(int64_t)r->isize : (int64_t)r->isize;/* :530 */
        size_t k = 0; while (k < c->fs_n && c->fs_size[k] != key) ++k;
        if (k == c->fs_n) {
            if (c->fs_n == c->fs_cap) {
                c->fs_cap = c->fs_cap ? c->fs_cap * 2 : 256;
                c->fs_size = xrealloc(c->fs_size, c->fs_cap * sizeof(int64_t));
                c->fs_count = xrealloc(c->fs_count, c->fs_cap * sizeof(uint64_t));
            }
            c->fs_size[k] = key; c->fs_count[k] = 0; c->fs_n++;
        }
        c->fs_count[k]++;
        if (c->trace) {
            if (c->tr_ns == c->tr_scap) {
                c->tr_scap = c->tr_scap ? c->tr_scap * 2 : 4096;
                c->tr_sample_file = xrealloc(c->tr_sample_file, c->tr_scap * 8); c->tr_sample_size = xrealloc(c->tr_sample_size, c->tr_scap * 4);
            }
            c->tr_sample_file[c->tr_ns] = c->cur_file_index; c->tr_sample_size[c->tr_ns] = (uint32_t)key; c->tr_ns++;
        }
        free(found->s); found->s = NULL; found->used = 2;                   /* erase :531 (tombstone) */
        --c->frag_remaining;                                                /* :532 */
    }
}


/* ------------------------------------------------------------- --fasta */

/* The GC branch of exonAlignmentMetrics, src/Expression.cpp:459-477, reached with `exon_row` = the single element of
 * alignedExons.  Returns the fragment's GC content or -1.                                                            */
static double fragment_gc(oracle_ctx *c, const rec_t *r, uint32_t exon_row) {
    pending_t *found = NULL;
    uint64_t h = r->qhash ? r->qhash : 0x9e3779b97f4a7c15ull;
    if (c->gcp_cap) {                                                       /* fragments.find(Qname) :461 */
        size_t j = (size_t)(h * 0x9e3779b97f4a7c15ull >> 17) & (c->gcp_cap - 1);
        while (c->gcp[j].used) {
            if (c->gcp[j].used == 1 && c->gcp[j].h == h &&
                (r->qname ? (c->gcp[j].s && strlen(c->gcp[j].s) == r->qname_len && memcmp(c->gcp[j].s, r->qname, r->qname_len) == 0)
                          : (!r->has_qhash2 || c->gcp[j].h2 == r->qhash2))) { found = &c->gcp[j]; break; }   /* names; else the 96-bit identity */
            j = (j + 1) & (c->gcp_cap - 1);
        }
    }
    const int32_t endpos = end_position(r);
    if (!found) {                                                           /* :462-466 */
        if ((c->gcp_n + 1) * 2 > c->gcp_cap) {
            size_t ncap = c->gcp_cap ? c->gcp_cap * 2 : 1024;
            pending_t *np = xcalloc(ncap, sizeof(pending_t));
            for (size_t i = 0; i < c->gcp_cap; ++i) if (c->gcp[i].used == 1) {
                size_t j = (size_t)(c->gcp[i].h * 0x9e3779b97f4a7c15ull >> 17) & (ncap - 1);
                while (np[j].used) j = (j + 1) & (ncap - 1);
                np[j] = c->gcp[i];
            }
            free(c->gcp); c->gcp = np; c->gcp_cap = ncap;
        }
        size_t j = (size_t)(h * 0x9e3779b97f4a7c15ull >> 17) & (c->gcp_cap - 1);
        while (c->gcp[j].used == 1) j = (j + 1) & (c->gcp_cap - 1);
        int was_tomb = c->gcp[j].used == 2;
        c->gcp[j].used = 1; c->gcp[j].h = h; c->gcp[j].h2 = r->has_qhash2 ? r->qhash2 : 0u; c->gcp[j].bed = (int32_t)exon_row; c->gcp[j].endpos = endpos;
        c->gcp[j].s = NULL;
        if (r->qname) { c->gcp[j].s = xcalloc(r->qname_len + 1, 1); memcpy(c->gcp[j].s, r->qname, r->qname_len); }
        if (!was_tomb) c->gcp_n++;
        return -1;
    }
    if (found->bed != (int32_t)exon_row) return -1;                         /* :467 */
    if (endpos <= found->endpos || r->pos == r->mpos) return -1;            /* :471 */
    size_t len;
    const uint8_t *seq = get_seq(c, r->tid, found->endpos - (int64_t)r->l_qseq, endpos, &len);   /* :473 */
    free(found->s); found->s = NULL; found->used = 2;                       /* erase :474 */
    return len > 0 ? gc_content(seq, len) : -1;                             /* :475 */
}

#define INC(k) (c->counters[(k)]++)

/* exonAlignmentMetrics, src/Expression.cpp:308-458 (GC branch :459-477 is
 * --fasta only and out of scope)                                           */
static double exon_alignment_metrics(oracle_ctx *c, const rec_t *r, const block_t *blocks,
                                     size_t nb, unsigned length, int hq) {
    uint32_t aligned_exon = 0; size_t n_aligned_exons = 0;                  /* set<string> alignedExons :318 */
    flist_t *fl = &c->feat[r->tid];
    int intragenic = 0, plus = 0, minus = 0, ribosomal = 0, do_exon = 0, exonic = 0; /* :321 */
    /* feature_strand, src/Expression.cpp:119-125 */
    int read_strand = RSQC_STRAND_UNKNOWN;
    if (c->p.stranded != RSQC_STRAND_UNKNOWN) {
        int target = (r->flag & RSQC_FREVERSE) != 0;
        if ((c->p.stranded == RSQC_STRAND_FORWARD) ^ ((r->flag & RSQC_FREAD1) != 0)) target = !target;
        read_strand = target ? RSQC_STRAND_REVERSE : RSQC_STRAND_FORWARD;
    }
    /* genes[b] = set of gene ids of block b; staged = Collector + BaseCoverage cache */
    uint32_t **gset = xcalloc(nb, sizeof(uint32_t *));
    size_t *gn = xcalloc(nb, sizeof(size_t));
    staged_t *st = NULL; size_t nst = 0, cst = 0;
    for (size_t b = 0; b < nb; ++b) {                                       /* :325 */
        int64_t bs = blocks[b].start, be = blocks[b].end;
        size_t gcap = 0;
        for (size_t i = fl->head; i < fl->n && fl->f[i].start <= be; ++i) { /* intersectBlock :111 */
            const feat_t *f = &fl->f[i];
            if (!intersects(bs, be, f)) continue;
            int fstrand = f->flags & RSQC_FF_STRAND_MASK;
            if (read_strand != RSQC_STRAND_UNKNOWN && read_strand != fstrand) continue; /* :331 */
            if (fstrand == RSQC_STRAND_FORWARD) plus = 1;                   /* :332-333 */
            else if (fstrand == RSQC_STRAND_REVERSE) minus = 1;
            if (!f->is_gene) {                                              /* :335 */
                exonic = 1;
                int64_t isz = partial_intersect(f, bs, be);
                if (isz == be - bs) {                                       /* :341 */
                    size_t k = 0; while (k < gn[b] && gset[b][k] != f->gene) ++k;
                    if (k == gn[b]) {
                        if (gn[b] == gcap) { gcap = gcap ? gcap * 2 : 4; gset[b] = xrealloc(gset[b], gcap * sizeof(uint32_t)); }
                        gset[b][gn[b]++] = f->gene;
                    }
                    if (nst == cst) { cst = cst ? cst * 2 : 8; st = xrealloc(st, cst * sizeof(staged_t)); }
                    st[nst].gene = f->gene; st[nst].exon_row = f->row;
                    st[nst].frac = (double)isz / length;                    /* :345 */
                    if (c->ctrace) ct_push(c, 0, f->gene, c->ex_id[f->row], st[nst].frac, 0);   /* Collector::add :346 */
                    st[nst].offset = bs - f->start;                         /* Metrics.cpp:99-100 */
                    st[nst].length = (uint32_t)(be - bs);
                    {                                                       /* alignedExons.insert(feature_id) :348 */
                        int seen_exon = 0;
                        for (size_t k2 = 0; k2 < nst; ++k2) if (st[k2].exon_row == f->row) seen_exon = 1;
                        if (!seen_exon) { aligned_exon = f->row; n_aligned_exons++; }
                    }
                    nst++;
                }
            } else intragenic = 1;                                          /* :352-354 */
            if (f->flags & RSQC_FF_RIBOSOMAL) ribosomal = 1;                /* :358 */
        }
    }
    if (nb >= 1) {                                                          /* :363 */
        uint32_t *last = dup_array(gset[0], gn[0], sizeof(uint32_t)); size_t nl = gn[0];
        for (size_t b = 1; b < nb; ++b) {                                   /* :369-374 */
            size_t w = 0;
            for (size_t i = 0; i < nl; ++i) {
                size_t k = 0; while (k < gn[b] && gset[b][k] != last[i]) ++k;
                if (k < gn[b]) last[w++] = last[i];
            }
            nl = w;
        }
        int globin = 0;
        for (size_t i = 0; i < nl; ++i) {                                   /* :377-394 */
            uint32_t gene = last[i];
            if (hq) {
                int query = 0;                                              /* Collector::queryGene: entries with coverage > 0 */
                for (size_t k = 0; k < nst; ++k) if (st[k].gene == gene && st[k].frac > 0) query = 1;
                if (c->ctrace) { ct_push(c, 1, gene, 0, 0.0, query); ct_push(c, 2, gene, 0, 0.0, 0); }  /* queryGene :380, collect :390 */
                if (query) {
                    c->gene_reads[gene] += 1.0;                             /* :382 */
                    if (nameset_insert(&c->tracker[gene], r->qhash, r->qhash2, r->has_qhash2, r->qname, r->qname_len))
                        c->gene_frag[gene] += 1.0;                          /* :383-387 */
                    if (!(r->flag & RSQC_FDUP)) c->gene_unique[gene] += 1.0;/* :388 */
                }
                for (size_t k = 0; k < nst; ++k) if (st[k].gene == gene && st[k].frac > 0) { /* collect :390, Metrics.cpp:59-66 */
                    uint32_t eid = c->ex_id[st[k].exon_row];
                    c->exon_reads[eid] += st[k].frac; c->exon_hit[eid] = 1;
                }
                if (c->seen[gene]) {                                        /* commit :391, Metrics.cpp:106-124 */
                    fprintf(stderr, "Gene encountered after computing coverage %u\n", gene);
                } else for (size_t k = 0; k < nst; ++k) if (st[k].gene == gene) {
                    uint32_t row = st[k].exon_row;
                    size_t elen = (size_t)(c->ex_end[row] - c->ex_start[row] + 1);
                    if (!c->cov[row]) c->cov[row] = xcalloc(elen, sizeof(uint64_t));
                    for (int64_t j = st[k].offset; j < st[k].offset + (int64_t)st[k].length && (size_t)j < elen; ++j)
                        c->cov[row][j] += 1;                                /* add_range, Metrics.cpp:257-262 */
                }
            }
            do_exon = 1;                                                    /* :393 */
            if (c->g_globin[gene]) globin = 1;                              /* :396-398 */
        }
        if (!globin) {                                                      /* :399-404 */
            INC(RSQC_C_NON_GLOBIN_READS);
            if (r->flag & RSQC_FDUP) INC(RSQC_C_NON_GLOBIN_DUPLICATE_READS);
        }
        free(last);
    }
    if (!exonic) {                                                          /* :407-423 */
        if (intragenic) {
            INC(RSQC_C_INTRONIC_READS); INC(RSQC_C_INTRAGENIC_READS);
            if (hq) { INC(RSQC_C_HQ_INTRONIC_READS); INC(RSQC_C_HQ_INTRAGENIC_READS); }
        } else {
            INC(RSQC_C_INTERGENIC_READS);
            if (hq) INC(RSQC_C_HQ_INTERGENIC_READS);
        }
    } else if (do_exon) {                                                   /* :424-433 */
        INC(RSQC_C_EXONIC_READS); INC(RSQC_C_INTRAGENIC_READS);
        if (hq) { INC(RSQC_C_HQ_EXONIC_READS); INC(RSQC_C_HQ_INTRAGENIC_READS); }
    } else {                                                                /* :434-441 */
        INC(RSQC_C_AMBIGUOUS_READS);
        if (hq) INC(RSQC_C_HQ_AMBIGUOUS_READS);
    }
    if (ribosomal) INC(RSQC_C_RRNA_READS);                                  /* :442 */
    if ((minus ^ plus) && (c->p.unpaired || (r->flag & RSQC_FPAIRED))) {    /* :445-457 */
        int rev = (r->flag & RSQC_FREVERSE) != 0;
        int sense = rev ? minus : plus;
        if (c->p.unpaired || (r->flag & RSQC_FREAD1)) INC(sense ? RSQC_C_END1_SENSE : RSQC_C_END1_ANTISENSE);
        else INC(sense ? RSQC_C_END2_SENSE : RSQC_C_END2_ANTISENSE);
    }
    for (size_t b = 0; b < nb; ++b) free(gset[b]);
    free(gset); free(gn); free(st);
    /* :459 (the |InsertSize| window is open on both sides) */
    const double isz = fabs((double)r->isize);
    if (c->have_ref && c->ref_seq[r->tid] && hq && exonic && do_exon && n_aligned_exons == 1 && nb == 1 && isz > 100 && isz < 1000)
        return fragment_gc(c, r, aligned_exon);
    return -1;
}


/* legacyExonAlignmentMetrics, src/Expression.cpp:129-304 (--legacy).  `length` (extractBlocks' return value) is
 * not used by the reference here: the split dosage divides by alignment.Length() = l_qseq (:201).               */
static void legacy_exon_alignment_metrics(oracle_ctx *c, const rec_t *r, const block_t *blocks, size_t nb, int hq) {
    flist_t *fl = &c->feat[r->tid];
    const int64_t SPLIT_DISTANCE = 100;                                     /* LEGACY_SPLIT_DISTANCE, src/RNASeQC.cpp:28 */
    int split = 0; int64_t last_end = -1;                                   /* :135-141 */
    for (size_t b = 0; b < nb; ++b) {
        if (last_end > 0 && !split) split = (blocks[b].start - last_end) > SPLIT_DISTANCE - 1;
        last_end = blocks[b].end;
    }
    const int64_t cs = (int64_t)r->pos + 1, ce = end_position(r);           /* :145-146 (1-based closed span) */
    /* intersectBlock(current, features[chr]) :148, src/Expression.cpp:106-117 */
    size_t nres = 0, cres = 0; const feat_t **res = NULL;
    for (size_t i = fl->head; i < fl->n && fl->f[i].start <= ce; ++i)
        if (intersects(cs, ce, &fl->f[i])) {
            if (nres == cres) { cres = cres ? cres * 2 : 16; res = xrealloc(res, cres * sizeof(*res)); }
            res[nres++] = &fl->f[i];
        }
    int intragenic = 0, plus = 0, minus = 0, ribosomal = 0, do_exon = 0, exonic = 0, junction = 0, not_exonic = 0; /* :151 */
    int not_split = 0;                                                      /* :152 */
    int read_strand = RSQC_STRAND_UNKNOWN;                                  /* :153, feature_strand :119-125 */
    if (c->p.stranded != RSQC_STRAND_UNKNOWN) {
        int target = (r->flag & RSQC_FREVERSE) != 0;
        if ((c->p.stranded == RSQC_STRAND_FORWARD) ^ ((r->flag & RSQC_FREAD1) != 0)) target = !target;
        read_strand = target ? RSQC_STRAND_REVERSE : RSQC_STRAND_FORWARD;
    }
    staged_t *st = NULL; size_t cst = 0;                                    /* BaseCoverage cache of this read */
    uint32_t *dose_row = xcalloc(nb + 1, sizeof(uint32_t)); float *dose = xcalloc(nb + 1, sizeof(float));
    for (size_t ri = 0; ri < nres; ++ri) {                                  /* :154 */
        const feat_t *g = res[ri];
        const feat_t *exon = NULL;                                          /* :156 */
        int found_exon = 0, t_intron = 0, t_exon = 0;                       /* :157 */
        size_t ndose = 0;                                                   /* legacySplitDosage :158 */
        size_t nst = 0;                                                     /* cache[gene] (one gene row per gene id) */
        not_split = 0;                                                      /* :159 */
        if (!g->is_gene) continue;                                          /* :160 */
        const int gstrand = g->flags & RSQC_FF_STRAND_MASK;
        if (gstrand == RSQC_STRAND_FORWARD) plus = 1;                       /* :163-164 */
        else if (gstrand == RSQC_STRAND_REVERSE) minus = 1;
        for (size_t b = 0; b < nb; ++b) {                                   /* :165 */
            const int64_t bs = blocks[b].start, be = blocks[b].end;
            if (read_strand != RSQC_STRAND_UNKNOWN && read_strand != gstrand) continue; /* :167 */
            intragenic = 1;                                                 /* :168 */
            if (bs > g->end) not_exonic = 1;                                /* :170 */
            int first_exon = 0;                                             /* :172 */
            found_exon = 0;                                                 /* :173 */
            for (size_t ei = 0; ei < nres && !first_exon; ++ei) {           /* :178 */
                const feat_t *ex = res[ei];
                if (ex->is_gene || ex->gene != g->id || !intersects(bs, be, ex)) continue; /* :180 */
                if (g->flags & RSQC_FF_RIBOSOMAL) ribosomal = 1;            /* :182 */
                const int64_t pi = partial_intersect(ex, bs, be);
                if (pi == be - bs) {                                        /* :183-190 */
                    exon = ex; t_exon = 1; first_exon = 1; found_exon = 1;
                    if (nst == cst) { cst = cst ? cst * 2 : 8; st = xrealloc(st, cst * sizeof(staged_t)); }
                    st[nst].gene = ex->gene; st[nst].exon_row = ex->row;    /* baseCoverage.add, Metrics.cpp:96-103 */
                    st[nst].offset = bs - ex->start; st[nst].length = (uint32_t)(be - bs); st[nst].frac = 0;
                    nst++;
                } else if (pi > 0) t_intron = 1;                            /* :191-194 */
            }
            if (split && !not_split) {                                      /* :198-205 */
                if (found_exon) {
                    size_t k = 0; while (k < ndose && dose_row[k] != exon->row) ++k;
                    if (k == ndose) { dose_row[k] = exon->row; dose[k] = 0.0f; ndose++; }
                    dose[k] += (float)(be - bs) / (float)r->l_qseq;        /* :201 */
                } else not_split = 1;
            }
        }
        if (found_exon) {                                                   /* :211 */
            if (hq) {
                if (split && !not_split) {                                  /* :215-222 */
                    for (size_t k = 0; k < ndose; ++k) {
                        const uint32_t eid = c->ex_id[dose_row[k]];
                        c->exon_reads[eid] += dose[k]; c->exon_hit[eid] = 1;
                    }
                } else {                                                    /* :223-227 */
                    const uint32_t eid = c->ex_id[exon->row];
                    c->exon_reads[eid] += 1.0; c->exon_hit[eid] = 1;
                }
                const uint32_t gene = exon->gene;
                c->gene_reads[gene] += 1.0;                                 /* :229 */
                if (nameset_insert(&c->tracker[gene], r->qhash, r->qhash2, r->has_qhash2, r->qname, r->qname_len))
                    c->gene_frag[gene] += 1.0;                              /* :230-234 */
                if (!(r->flag & RSQC_FDUP)) c->gene_unique[gene] += 1.0;    /* :235 */
                if (c->seen[gene]) {                                        /* commit :236, Metrics.cpp:106-124 */
                    fprintf(stderr, "Gene encountered after computing coverage %u\n", gene);
                } else for (size_t k = 0; k < nst; ++k) {
                    const uint32_t row = st[k].exon_row;
                    const size_t elen = (size_t)(c->ex_end[row] - c->ex_start[row] + 1);
                    if (!c->cov[row]) c->cov[row] = xcalloc(elen, sizeof(uint64_t));
                    for (int64_t j = st[k].offset; j < st[k].offset + (int64_t)st[k].length && (size_t)j < elen; ++j)
                        c->cov[row][j] += 1;
                }
            }
            do_exon = 1;                                                    /* :238 */
        }
        if (t_intron && t_exon) junction = 1;                               /* :240 */
        if (t_exon) exonic = 1;                                             /* :241 */
    }
    free(res); free(st); free(dose_row); free(dose);
    if (not_exonic || junction || !exonic) {                                /* :248-263 */
        if (intragenic) {
            INC(RSQC_C_INTRONIC_READS); INC(RSQC_C_INTRAGENIC_READS);
            if (hq) { INC(RSQC_C_HQ_INTRONIC_READS); INC(RSQC_C_HQ_INTRAGENIC_READS); }
        } else {
            INC(RSQC_C_INTERGENIC_READS);
            if (hq) INC(RSQC_C_HQ_INTERGENIC_READS);
        }
    } else if (do_exon && !junction && !not_exonic) {                       /* :265-275 */
        INC(RSQC_C_EXONIC_READS); INC(RSQC_C_INTRAGENIC_READS);
        if (hq) { INC(RSQC_C_HQ_EXONIC_READS); INC(RSQC_C_HQ_INTRAGENIC_READS); }
        if (split && !not_split) INC(RSQC_C_SPLIT_READS);
    } else if (intragenic) {                                                /* :276-287 */
        INC(RSQC_C_EXONIC_READS); INC(RSQC_C_INTRAGENIC_READS);
        if (hq) { INC(RSQC_C_HQ_EXONIC_READS); INC(RSQC_C_HQ_INTRAGENIC_READS); }
    }
    if (ribosomal) INC(RSQC_C_RRNA_READS);                                  /* :288 */
    if ((minus ^ plus) && (c->p.unpaired || (r->flag & RSQC_FPAIRED))) {    /* :290-302 */
        int rev = (r->flag & RSQC_FREVERSE) != 0;
        int sense = rev ? minus : plus;
        if (c->p.unpaired || (r->flag & RSQC_FREAD1)) INC(sense ? RSQC_C_END1_SENSE : RSQC_C_END1_ANTISENSE);
        else INC(sense ? RSQC_C_END2_SENSE : RSQC_C_END2_ANTISENSE);
    }
}

/* The body of `while (bam.next(alignment))`, src/RNASeQC.cpp:242-382 */
static int process_record(oracle_ctx *c, const rec_t *r) {
    const uint32_t fl = r->flag;
    INC(RSQC_C_TOTAL_ALIGNMENTS);                                           /* :245,397 */
    if (fl & RSQC_FSECONDARY) INC(RSQC_C_ALTERNATIVE_ALIGNMENTS);           /* :254 */
    if (fl & RSQC_FSUPP) INC(RSQC_C_SUPPLEMENTARY_ALIGNMENTS);              /* :255 */
    else if (fl & RSQC_FQCFAIL) INC(RSQC_C_FAILED_VENDOR_QC);               /* :256 */
    else if (r->mapq < c->p.mapq_threshold) INC(RSQC_C_LOW_MAPPING_QUALITY);/* :257 */
    const int legacy = c->p.legacy != 0;
    const int has_ch = (r->tagbits & RSQC_TB_HAS_CH) != 0;
    if ((fl & RSQC_FSUPP) && !(legacy || has_ch)) {                         /* :258-262 */
        INC(RSQC_C_CHIMERIC_AUTO);
        if (c->p.exclude_chimeric) return 0;
    }
    if (fl & (RSQC_FSECONDARY | RSQC_FQCFAIL | RSQC_FSUPP)) return 0;       /* :263 */
    INC(RSQC_C_UNIQUE_VENDOR_PASSED);                                       /* :265 */
    if (!(fl & RSQC_FPAIRED)) INC(RSQC_C_UNPAIRED_READS);                   /* :267 */
    if (fl & RSQC_FUNMAP) return 0;                                         /* :268 */
    INC(RSQC_C_MAPPED_READS);
    if (fl & RSQC_FDUP) INC(RSQC_C_MAPPED_DUPLICATE_READS); else INC(RSQC_C_MAPPED_UNIQUE_READS); /* :272-273 */
    const int32_t endpos = end_position(r);
    unsigned int alignment_size = (unsigned int)(endpos - r->pos);          /* :275 */
    if (legacy && alignment_size > 100000u) return 0;                       /* :276, LEGACY_MAX_READ_LENGTH :27 */
    if (alignment_size > (unsigned int)c->read_length) c->read_length = r->l_qseq; /* :278 */
    if (c->trace) {
        if (c->tr_n == c->tr_cap) {
            c->tr_cap = c->tr_cap ? c->tr_cap * 2 : 4096;
            c->tr_span = xrealloc(c->tr_span, c->tr_cap * 4); c->tr_lq = xrealloc(c->tr_lq, c->tr_cap * 4);
        }
        c->tr_span[c->tr_n] = alignment_size; c->tr_lq[c->tr_n] = r->l_qseq; c->tr_n++;
    }
    if (!legacy && has_ch) {                                                /* :279-283 */
        if (fl & RSQC_FREAD1) INC(RSQC_C_CHIMERIC_TAG);
        if (c->p.exclude_chimeric) return 0;
    }
    if ((fl & RSQC_FPAIRED) && !(fl & RSQC_FMUNMAP)) {                      /* :284-292 */
        if (fl & RSQC_FREAD1) INC(RSQC_C_TOTAL_MAPPED_PAIRS);
        if (!(r->tagbits & RSQC_TB_MTID_SAME) || abs(r->pos - r->mpos) > c->p.chimeric_distance ||
            (legacy && r->tid > 127)) {                                     /* :287 */
            if (fl & RSQC_FREAD1) INC(RSQC_C_CHIMERIC_AUTO);
            if (c->p.exclude_chimeric) return 0;
        }
    }
    int32_t mismatches = 0;                                                 /* :294 */
    if (r->tagbits & RSQC_TB_HAS_NM) {                                      /* :295-316 */
        mismatches = r->nm;
        if (fl & RSQC_FPAIRED) {
            if (fl & RSQC_FREAD1) {
                INC(RSQC_C_END1_MAPPED_READS);
                c->counters[RSQC_C_END1_MISMATCHES] += (uint64_t)(int64_t)mismatches;
                c->counters[RSQC_C_END1_BASES] += (uint64_t)(int64_t)r->l_qseq;
                if (fl & RSQC_FDUP) INC(RSQC_C_DUPLICATE_PAIRS); else INC(RSQC_C_UNIQUE_FRAGMENTS);
            } else {
                INC(RSQC_C_END2_MAPPED_READS);
                c->counters[RSQC_C_END2_MISMATCHES] += (uint64_t)(int64_t)mismatches;
                c->counters[RSQC_C_END2_BASES] += (uint64_t)(int64_t)r->l_qseq;
            }
        }
        c->counters[RSQC_C_MISMATCHED_BASES] += (uint64_t)(int64_t)mismatches;
    }
    c->counters[RSQC_C_TOTAL_BASES] += (uint64_t)(int64_t)r->l_qseq;        /* :317 */
    int discard = 0;                                                        /* :319-328 */
    for (int t = 0; t < c->p.n_filter_tags; ++t) if (r->tagbits & (RSQC_TB_FILTER0 << t)) {
        discard = 1; INC(RSQC_C_FILTERED_TAG0 + t);
    }
    if (discard) return 0;
    const int hq = ((uint32_t)mismatches <= c->p.base_mismatch) &&
                   (c->p.unpaired || (fl & RSQC_FPROPER)) && (r->mapq >= c->p.mapq_threshold); /* :330 */
    if (r->tid < 0 || r->tid >= c->n_ref) return 0;                         /* :333-337 */
    if (hq) INC(RSQC_C_HIGH_QUALITY_READS); else INC(RSQC_C_LOW_QUALITY_READS); /* :340-341 */
    INC(RSQC_C_READS_USED);                                                 /* :342 */
    if (r->tid != c->current_contig) {                                      /* :346-353 */
        if (c->current_contig >= 0) { int rc = drop_features(c, &c->feat[c->current_contig]); if (rc) return rc; }
        c->current_contig = r->tid;
    }
    /* extractBlocks, src/Expression.cpp:26-67 */
    block_t *blocks = xcalloc(r->n_cigar, sizeof(block_t)); size_t nb = 0;
    int64_t start = (int64_t)r->pos + 1; unsigned int aligned = 0;
    for (uint32_t i = 0; i < r->n_cigar; ++i) {
        uint32_t op = r->cigar[i] & 0xf, len = r->cigar[i] >> 4;
        switch (op) {
        case 0: case 7: case 8:                         /* M = X */
            blocks[nb].start = start; blocks[nb].end = start + len; nb++; aligned += len;
            /* fall through */
        case 3: case 2:                                 /* N D */
            start += len; break;
        case 5: case 6: case 1: case 4:                 /* H P I S */
            break;
        default:                                        /* :61-63 */
            free(blocks);
            snprintf(c->errmsg, sizeof c->errmsg, "Bad cigar operation: %u", op);
            return RSQC_ERR_BAD_CIGAR;
        }
    }
    c->counters[RSQC_C_ALIGNMENT_BLOCKS] += nb;                             /* :360 */
    int rc = trim_features(c, &c->feat[r->tid], r->pos);                    /* :361 */
    if (rc) { free(blocks); return rc; }
    if (legacy) legacy_exon_alignment_metrics(c, r, blocks, nb, hq);        /* :364 */
    else {
        const double gcv = exon_alignment_metrics(c, r, blocks, nb, aligned, hq);   /* :366 */
        if (gcv != -1) {                                                    /* :368 (gcBins has 100 slots: 100 % GC is out of range) */
            const unsigned int bin = (unsigned int)(gcv * 100.0);
            if (bin < RSQC_GC_BINS) c->gc_bins[bin]++; else c->gc_oob++;
        }
    }
    if (hq && c->frag_remaining && (fl & RSQC_FPAIRED) && c->have_bed &&
        c->bed[r->tid].n)                                                   /* :372 */
        fragment_size(c, r, blocks, nb);
    free(blocks);
    return 0;
}

/* -------------------------------------------------------------- public API */

ORACLE_API int oracle_create(const rsqc_params *p, oracle_ctx **out) {
    if (!p || !out || p->abi_version != RSQC_ABI_VERSION) return RSQC_ERR_ARG;
    if (p->n_filter_tags < 0 || p->n_filter_tags > RSQC_MAX_FILTER_TAGS) return RSQC_ERR_ARG;
    oracle_ctx *c = xcalloc(1, sizeof(*c));
    c->p = *p; c->current_contig = -1;
    *out = c;
    return 0;
}

static int cmp_use_order;                               /* the annotation carries GTF positions */
static int cmp_feat(const void *a, const void *b) {     /* compIntervalStart + stable order */
    const feat_t *x = a, *y = b;
    if (x->start != y->start) return x->start < y->start ? -1 : 1;
    if (cmp_use_order && x->order != y->order) return x->order < y->order ? -1 : 1;
    if (x->is_gene != y->is_gene) return x->is_gene ? -1 : 1;   /* gene row before its exons */
    return x->row < y->row ? -1 : x->row > y->row;
}

ORACLE_API int oracle_set_annotation(oracle_ctx *c, const rsqc_annotation *a, const uint8_t *owned) {
    if (!c || !a) return RSQC_ERR_ARG;
    c->n_ref = a->n_ref; c->n_contigs = a->n_contigs; c->n_genes = a->n_genes;
    c->n_listed = a->n_genes_listed; c->n_exons = a->n_exons;
    const int nc = a->n_contigs, G = a->n_genes, L = a->n_genes_listed, E = a->n_exons;
    c->feat = xcalloc(nc, sizeof(flist_t));
    c->owned = xcalloc(nc, 1);
    for (int i = 0; i < nc; ++i) c->owned[i] = owned ? owned[i] : 1;
    size_t *cnt = xcalloc(nc, sizeof(size_t));
    for (int i = 0; i < L; ++i) cnt[a->gene_row_contig[i]]++;
    for (int i = 0; i < E; ++i) cnt[a->exon_row_contig[i]]++;
    for (int i = 0; i < nc; ++i) { c->feat[i].f = xcalloc(cnt[i], sizeof(feat_t)); c->feat[i].n = 0; }
    c->g_row_flags = xcalloc(L, 1);
    for (int i = 0; i < L; ++i) {
        flist_t *fl = &c->feat[a->gene_row_contig[i]];
        feat_t *f = &fl->f[fl->n++];
        f->start = a->gene_row_start[i]; f->end = a->gene_row_end[i]; f->flags = a->gene_row_flags[i];
        f->is_gene = 1; f->id = a->gene_row_id[i]; f->gene = f->id; f->row = (uint32_t)i;
        f->order = a->gene_row_order ? a->gene_row_order[i] : 0;
        if (f->id < (uint32_t)L) c->g_row_flags[f->id] = f->flags;
    }
    for (int i = 0; i < E; ++i) {
        flist_t *fl = &c->feat[a->exon_row_contig[i]];
        feat_t *f = &fl->f[fl->n++];
        f->start = a->exon_row_start[i]; f->end = a->exon_row_end[i]; f->flags = a->exon_row_flags[i];
        f->is_gene = 0; f->id = a->exon_row_id[i]; f->gene = a->exon_row_gene[i]; f->row = (uint32_t)i;
        f->order = a->exon_row_order ? a->exon_row_order[i] : 0;
    }
    cmp_use_order = a->gene_row_order && a->exon_row_order;
    for (int i = 0; i < nc; ++i) qsort(c->feat[i].f, c->feat[i].n, sizeof(feat_t), cmp_feat);
    free(cnt);
    c->ex_start = dup_array(a->exon_row_start, E, 4); c->ex_end = dup_array(a->exon_row_end, E, 4);
    c->ex_id = dup_array(a->exon_row_id, E, 4); c->ex_gene = dup_array(a->exon_row_gene, E, 4);
    c->ex_flags = dup_array(a->exon_row_flags, E, 1);
    c->g_globin = dup_array(a->gene_is_globin, G, 1);
    c->ge_off = dup_array(a->gene_exon_off, (size_t)G + 1, 4);
    c->ge_row = dup_array(a->gene_exon_row, E, 4);
    c->gene_reads = xcalloc(G, 8); c->gene_unique = xcalloc(G, 8); c->gene_frag = xcalloc(G, 8);
    c->exon_reads = xcalloc(E, 8); c->exon_hit = xcalloc(E, 1);
    c->tracker = xcalloc(G, sizeof(nameset_t));
    c->cov = xcalloc(E, sizeof(uint64_t *)); c->seen = xcalloc(G, 1);
    c->cov_mean = xcalloc(L, 8); c->cov_std = xcalloc(L, 8); c->cov_cv = xcalloc(L, 8); c->cov_valid = xcalloc(L, 1);
    c->exon_cv = xcalloc(E, 8); c->exon_cv_valid = xcalloc(E, 1);
    c->bias3 = xcalloc(L, 8); c->bias5 = xcalloc(L, 8);
    c->exit_order = xcalloc(L, 4);
    c->r_reads = xcalloc(L, 8); c->r_unique = xcalloc(L, 8); c->r_frag = xcalloc(L, 8);
    c->bed = xcalloc(nc, sizeof(flist_t));
    return 0;
}

ORACLE_API int oracle_set_bed(oracle_ctx *c, const rsqc_bed *b) {
    if (!c || !b || !c->bed) return RSQC_ERR_ARG;
    size_t *cnt = xcalloc(c->n_contigs, sizeof(size_t));
    for (int i = 0; i < b->n_intervals; ++i) cnt[b->contig[i]]++;
    for (int i = 0; i < c->n_contigs; ++i) { free(c->bed[i].f); c->bed[i].f = xcalloc(cnt[i], sizeof(feat_t)); c->bed[i].n = 0; c->bed[i].head = 0; }
    for (int i = 0; i < b->n_intervals; ++i) {          /* file order, src/RNASeQC.cpp:185 */
        flist_t *fl = &c->bed[b->contig[i]];
        feat_t *f = &fl->f[fl->n++];
        f->start = b->start[i]; f->end = b->end[i]; f->id = (uint32_t)i; f->is_gene = 0;
    }
    free(cnt);
    c->have_bed = 1; c->frag_remaining = c->p.fragment_samples;             /* :176 */
    return 0;
}

/* Fasta::open, src/Fasta.cpp:77-98 (the index decides which contigs exist) */
ORACLE_API int oracle_set_reference(oracle_ctx *c, const rsqc_reference *ref) {
    if (!c || !ref || !c->feat) return RSQC_ERR_ARG;
    c->ref_seq = xcalloc(c->n_contigs, sizeof(uint8_t *)); c->ref_len = xcalloc(c->n_contigs, sizeof(uint64_t));
    for (int i = 0; i < ref->n; ++i) {
        const int k = ref->contig[i];
        if (k < 0 || k >= c->n_contigs) return RSQC_ERR_ARG;
        c->ref_seq[k] = dup_array(ref->sequence[i], ref->length[i], 1); c->ref_len[k] = ref->length[i];
    }
    c->exon_gc = xcalloc(c->n_exons, 8);
    c->have_ref = 1;
    return 0;
}

ORACLE_API int oracle_submit(oracle_ctx *c, const rsqc_batch *b) {
    if (!c || !b || !c->feat) return RSQC_ERR_ARG;
    if (c->error) return c->error;
    uint32_t w = 0;
    for (uint32_t s = 0; s < b->n_seg; ++s) {
        for (uint64_t i = b->seg_start[s]; i < b->seg_start[s + 1]; ++i) {
            rec_t r;
            const rsqc_rec_core *rcore = &b->core[i];
            const rsqc_rec_aux *ra = &b->aux[i];
            r.tid = b->seg_tid[s]; r.pos = rcore->pos; r.mpos = rcore->mpos; r.isize = rcore->isize;
            r.flag = ra->flag; r.mapq = ra->mapq; r.tagbits = ra->tagbits;
            r.l_qseq = ra->l_qseq; r.nm = ra->nm; r.n_cigar = ra->n_cigar;
            if (ra->l_qseq == RSQC_LQSEQ_ESCAPE || ra->nm == RSQC_NM_ESCAPE || ra->n_cigar == RSQC_NCIGAR_ESCAPE) {
                while (w < b->n_wide && b->wide_index[w] < i) ++w;
                if (w >= b->n_wide || b->wide_index[w] != i) return c->error = RSQC_ERR_ARG;
                r.l_qseq = b->wide_l_qseq[w]; r.nm = b->wide_nm[w]; r.n_cigar = b->wide_n_cigar[w];
            }
            r.cigar = b->cigar + rcore->cigar_off;
            r.qhash = ra->qhash; r.has_qhash2 = b->qhash2 != NULL; r.qhash2 = b->qhash2 ? b->qhash2[i] : 0u;
            r.qname = NULL; r.qname_len = 0;
            if (b->qname && b->qname_off) { r.qname = b->qname + b->qname_off[i]; r.qname_len = b->qname_off[i + 1] - b->qname_off[i]; }
            c->cur_file_index = b->file_index_base + i;
            c->ctr_read++;
            int rc = process_record(c, &r);
            if (rc) return c->error = rc;
        }
    }
    if (c->trace) {
        if (c->tr_nb == c->tr_bcap) {
            c->tr_bcap = c->tr_bcap ? c->tr_bcap * 2 : 64;
            c->tr_batch_end = xrealloc(c->tr_batch_end, c->tr_bcap * 8); c->tr_batch_file = xrealloc(c->tr_batch_file, c->tr_bcap * 8);
        }
        c->tr_batch_end[c->tr_nb] = c->tr_n; c->tr_batch_file[c->tr_nb] = b->file_index_base; c->tr_nb++;
    }
    return 0;
}

/* test trace: see oracle_ctx.trace */
typedef struct oracle_trace {
    uint64_t n_eligible; const uint32_t *span; const int32_t *l_qseq;
    uint64_t n_batches; const uint64_t *batch_end, *batch_file_index;
    uint64_t n_samples; const uint64_t *sample_file_index; const uint32_t *sample_size;
} oracle_trace;
ORACLE_API int oracle_enable_trace(oracle_ctx *c) { if (!c) return RSQC_ERR_ARG; c->trace = 1; return 0; }
typedef struct oracle_collector_trace {
    uint64_t n; const uint8_t *kind;        /* 0 add, 1 queryGene, 2 collect */
    const uint32_t *read, *gene, *exon; const double *frac; const uint8_t *query;
} oracle_collector_trace;
ORACLE_API int oracle_enable_collector_trace(oracle_ctx *c) { if (!c) return RSQC_ERR_ARG; c->ctrace = 1; return 0; }
ORACLE_API int oracle_get_collector_trace(oracle_ctx *c, oracle_collector_trace *t) {
    if (!c || !t) return RSQC_ERR_ARG;
    t->n = c->ct_n; t->kind = c->ct_kind; t->read = c->ct_read; t->gene = c->ct_gene; t->exon = c->ct_exon; t->frac = c->ct_frac; t->query = c->ct_query;
    return 0;
}
ORACLE_API int oracle_get_trace(oracle_ctx *c, oracle_trace *t) {
    if (!c || !t) return RSQC_ERR_ARG;
    t->n_eligible = c->tr_n; t->span = c->tr_span; t->l_qseq = c->tr_lq;
    t->n_batches = c->tr_nb; t->batch_end = c->tr_batch_end; t->batch_file_index = c->tr_batch_file;
    t->n_samples = c->tr_ns; t->sample_file_index = c->tr_sample_file; t->sample_size = c->tr_sample_size;
    return 0;
}

ORACLE_API int oracle_finalize(oracle_ctx *c, rsqc_results *out) {
    if (!c || !out || !c->feat) return RSQC_ERR_ARG;
    if (c->error) return c->error;
    for (int i = 0; i < c->n_contigs; ++i)                                  /* src/RNASeQC.cpp:385-386 */
        if (c->feat[i].head < c->feat[i].n) { int rc = drop_features(c, &c->feat[i]); if (rc) return c->error = rc; }
    memset(out, 0, sizeof(*out));
    for (int g = 0; g < c->n_listed; ++g) {                                 /* static_cast<long>, :441-442 */
        c->r_reads[g] = (uint64_t)(long)c->gene_reads[g];
        c->r_unique[g] = (uint64_t)(long)c->gene_unique[g];
        c->r_frag[g] = (uint64_t)(long)c->gene_frag[g];
    }
    /* fragment sizes ascending (std::map<long long, unsigned long>) */
    for (size_t i = 1; i < c->fs_n; ++i) for (size_t j = i; j > 0 && c->fs_size[j - 1] > c->fs_size[j]; --j) {
        int64_t ts = c->fs_size[j]; c->fs_size[j] = c->fs_size[j - 1]; c->fs_size[j - 1] = ts;
        uint64_t tc = c->fs_count[j]; c->fs_count[j] = c->fs_count[j - 1]; c->fs_count[j - 1] = tc;
    }
    (void)cmp_i64;
    out->n_genes_listed = c->n_listed; out->n_exons = c->n_exons;
    out->gene_reads = c->r_reads; out->gene_unique = c->r_unique; out->gene_fragments = c->r_frag;
    out->exon_reads = c->exon_reads; out->exon_hit = c->exon_hit;
    memcpy(out->counters, c->counters, sizeof(c->counters));
    out->read_length = c->read_length;
    out->gene_cov_mean = c->cov_mean; out->gene_cov_std = c->cov_std; out->gene_cov_cv = c->cov_cv;
    out->gene_cov_valid = c->cov_valid; out->exon_cv = c->exon_cv; out->exon_cv_valid = c->exon_cv_valid;
    out->bias_three = c->bias3; out->bias_five = c->bias5;
    out->n_fragment_sizes = (uint32_t)c->fs_n; out->fragment_size = c->fs_size; out->fragment_count = c->fs_count;
    out->fragment_samples_remaining = c->frag_remaining;
    out->have_reference = c->have_ref; out->gc_bins = c->gc_bins; out->gc_out_of_range = c->gc_oob; out->exon_gc = c->exon_gc;
    out->exons_outside_gene_row = 0;          /* (the restatement streams like the reference: such rows get the reference's treatment) */
    return 0;
}

/* coverage.tsv row order = gene exit order (tests only) */
ORACLE_API int oracle_exit_order(oracle_ctx *c, const uint32_t **order, uint32_t *n) {
    if (!c) return RSQC_ERR_ARG;
    *order = c->exit_order; *n = c->n_exit; return 0;
}

ORACLE_API const char *oracle_last_error(oracle_ctx *c) { return c ? c->errmsg : ""; }

ORACLE_API void oracle_destroy(oracle_ctx *c) {
    if (!c) return;
    for (int i = 0; i < c->n_contigs; ++i) { if (c->feat) free(c->feat[i].f); if (c->bed) free(c->bed[i].f); }
    if (c->tracker) for (int g = 0; g < c->n_genes; ++g) nameset_clear(&c->tracker[g]);
    if (c->cov) for (int e = 0; e < c->n_exons; ++e) free(c->cov[e]);
    for (size_t i = 0; i < c->pend_cap; ++i) free(c->pend[i].s);
    for (size_t i = 0; i < c->gcp_cap; ++i) free(c->gcp[i].s);
    if (c->ref_seq) for (int i = 0; i < c->n_contigs; ++i) free(c->ref_seq[i]);
    free(c->gcp); free(c->ref_seq); free(c->ref_len); free(c->exon_gc);
    free(c->pend); free(c->fs_size); free(c->fs_count);
    free(c->feat); free(c->bed); free(c->owned); free(c->ex_start); free(c->ex_end); free(c->ex_id);
    free(c->ex_gene); free(c->ex_flags); free(c->g_globin); free(c->g_row_flags); free(c->ge_off); free(c->ge_row);
    free(c->gene_reads); free(c->gene_unique); free(c->gene_frag); free(c->exon_reads); free(c->exon_hit);
    free(c->tracker); free(c->cov); free(c->seen); free(c->cov_mean); free(c->cov_std); free(c->cov_cv);
    free(c->cov_valid); free(c->exon_cv); free(c->exon_cv_valid); free(c->bias3); free(c->bias5);
    free(c->exit_order); free(c->r_reads); free(c->r_unique); free(c->r_frag);
    free(c->ct_kind); free(c->ct_read); free(c->ct_gene); free(c->ct_exon); free(c->ct_frac); free(c->ct_query);
    free(c->tr_span); free(c->tr_lq); free(c->tr_batch_end); free(c->tr_batch_file); free(c->tr_sample_file); free(c->tr_sample_size);
    free(c);
}

/* Library-complexity search, src/RNASeQC.cpp:398-415 (report tail; restated
 * here so the product's bracketed search can be checked against the literal
 * 1e9-step loop on small inputs).  `limit` replaces the 1e9 bound.          */
ORACLE_API unsigned int oracle_library_complexity(double duplicates, double unique, double limit) {
    double numReads = duplicates + unique;
    unsigned int minReads = 0u, minError = 0xffffffffu;
    if (duplicates > 0) {
        for (double x = unique; x < limit; ++x) {
            double estimate = x * (1.0 - exp(-1.0 * numReads / x));
            unsigned int error = (unsigned int)fabs(estimate - unique);
            if (error < minError) { minError = error; minReads = (unsigned int)x; }
        }
    }
    return minReads;
}
