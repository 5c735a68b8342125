// rsqc_read.h -- per-record semantics of the RNA-SeQC hot path, written once as
// __host__ __device__ code and parameterised on an accumulator policy `Acc`
// (HIP kernels scatter with atomics / LDS; nothing else changes).
//
// What one call of classify_record() covers in the reference:
//   gate cascade + scalar counters      src/RNASeQC.cpp:254-342
//   extractBlocks                       src/Expression.cpp:26-67
//   trimFeatures + intersectBlock       src/Expression.cpp:80-117   (as a STATIC overlap
//       query on a start-sorted row table with a prefix-max-of-end column; equivalent
//       on coordinate-sorted input)
//   exonAlignmentMetrics                src/Expression.cpp:308-458
// It does not keep any per-record heap state: the CIGAR is walked twice (once to
// find the gene set common to all blocks, once to commit), so a lane needs only a
// handful of registers.
#pragma once

#include <stdint.h>
#include "../../include/rnaseqc_amd.h"

#if defined(__HIPCC__)
#define RSQC_HD __host__ __device__ __forceinline__
#else
#define RSQC_HD inline
#endif

namespace rsqc {

// ---- device-resident annotation index ------------------------------------------
struct DevAnnotation {
    int32_t n_ref, n_contigs, n_genes, n_listed, n_exons;
    // exon rows (sorted by contig,start)
    const int32_t  *ex_start, *ex_end, *ex_pmax;   // pmax = running max of end inside the contig
    const uint32_t *ex_gene;                       // gene id
    const uint8_t  *ex_flags;
    const uint32_t *ex_cov;                        // offset of the row's per-base coverage
    const uint32_t *ex_range;                      // [n_contigs+1] row range of a contig
    // gene rows
    const int32_t  *g_start, *g_end, *g_pmax;
    const uint8_t  *g_flags;
    const uint32_t *g_range;
    // per gene id
    const uint8_t  *gene_globin;
    // coarse position bins: first row with start >= bin*2^shift (per contig, concatenated)
    const uint32_t *ex_bin, *g_bin;                // [bin_off[n_contigs]] + 1 sentinel per contig
    const uint64_t *bin_off;                       // [n_contigs+1]
    int32_t bin_shift;
    // BED rows (sorted by contig,start), optional
    const int32_t  *bed_start, *bed_end, *bed_pmax;
    const uint32_t *bed_range;                     // [n_contigs+1]
    int32_t have_bed;
};

struct DevParams {
    uint32_t mapq_threshold, base_mismatch;
    int32_t  chimeric_distance;
    int32_t  stranded, unpaired, exclude_chimeric, n_filter_tags;
};

// one record, already widened
struct Record {
    int32_t tid, pos, mpos, isize, l_qseq, nm;
    uint32_t flag, mapq, tagbits, n_cigar;
    const uint32_t *cigar;
    uint64_t qhash;
};

// per-record scalar outputs; the kernel reduces them across the wave
struct RecordCounters {
    uint64_t bits;          // bit c set -> counter c += 1
    uint32_t e1_mm, e1_bases, e2_mm, e2_bases, mm, bases, blocks;   // sum-type counters
    // Read-Length state machine inputs (src/RNASeQC.cpp:275-278)
    uint32_t rl_eligible;   // record reaches :275
    uint32_t rl_span;       // PositionEnd() - Position()
    int32_t  rl_lqseq;
    int32_t  error;         // RSQC_ERR_BAD_CIGAR or 0
    // fragment-size candidate (src/RNASeQC.cpp:372): record passed HQ && PAIRED
    uint32_t frag_candidate;
    int32_t  endpos;
};

#define RSQC_BIT(c) (1ull << (c))

constexpr int FAST_SET = 4;    // genes per block handled on the fast path (registers)
constexpr int SLOW_SET = 128;  // ... on the exact slow path (scratch); more -> RSQC_ERR_CAPACITY

RSQC_HD bool cigar_is_ref(uint32_t op) { return op == 0 || op == 2 || op == 3 || op == 7 || op == 8; }
RSQC_HD bool cigar_is_block(uint32_t op) { return op == 0 || op == 7 || op == 8; }

// first row in [lo,hi) with start > x
RSQC_HD uint32_t upper_bound_rows(const int32_t *start, uint32_t lo, uint32_t hi, int32_t x) {
    while (lo < hi) {
        uint32_t mid = lo + ((hi - lo) >> 1);
        if (start[mid] <= x) lo = mid + 1; else hi = mid;
    }
    return lo;
}

// Row range [lo, ub) of contig `tid` whose start <= be, using the coarse bins.
RSQC_HD uint32_t rows_upto(const int32_t *start, const uint32_t *range, const uint32_t *bins,
                           const uint64_t *bin_off, int shift, int32_t tid, int32_t be) {
    uint32_t lo = range[tid], hi = range[tid + 1];
    if (be < 0) return lo;
    if (bins) {
        uint64_t nb = bin_off[tid + 1] - bin_off[tid] - 1;     // bins of this contig (last = sentinel)
        uint64_t b = (uint64_t)(uint32_t)be >> shift;
        if (b >= nb) return hi;                                // beyond the last feature start
        uint32_t l2 = bins[bin_off[tid] + b], h2 = bins[bin_off[tid] + b + 1];
        return upper_bound_rows(start, l2, h2, be);
    }
    return upper_bound_rows(start, lo, hi, be);
}

// feature_strand, src/Expression.cpp:119-125
RSQC_HD int read_strand_of(const DevParams &p, uint32_t flag) {
    if (p.stranded == RSQC_STRAND_UNKNOWN) return RSQC_STRAND_UNKNOWN;
    bool target = (flag & RSQC_FREVERSE) != 0;
    if ((p.stranded == RSQC_STRAND_FORWARD) ^ ((flag & RSQC_FREAD1) != 0)) target = !target;
    return target ? RSQC_STRAND_REVERSE : RSQC_STRAND_FORWARD;
}

struct ClassFlags { bool intragenic, plus, minus, ribosomal, exonic; };

// Gene rows overlapping [bs, be] (be inclusive: the reference's intersectInterval on a
// block whose end is exclusive, src/GTF.cpp:171-179, Expression.cpp:111) -> flags only.
RSQC_HD void scan_gene_rows(const DevAnnotation &a, int32_t tid, int32_t bs, int32_t be, int rstrand,
                            ClassFlags &f) {
    const uint32_t lo = a.g_range[tid];
    uint32_t ub = rows_upto(a.g_start, a.g_range, a.g_bin, a.bin_off, a.bin_shift, tid, be);
    for (uint32_t i = ub; i > lo;) {
        --i;
        if (a.g_pmax[i] < bs) break;
        if (a.g_end[i] < bs) continue;
        const uint32_t fl = a.g_flags[i];
        const int fs = fl & RSQC_FF_STRAND_MASK;
        if (rstrand != RSQC_STRAND_UNKNOWN && rstrand != fs) continue;     // Expression.cpp:331
        if (fs == RSQC_STRAND_FORWARD) f.plus = true; else if (fs == RSQC_STRAND_REVERSE) f.minus = true;
        f.intragenic = true;                                               // :352-354
        if (fl & RSQC_FF_RIBOSOMAL) f.ribosomal = true;                    // :358
    }
}

// Exon rows overlapping the block.  Visit(row, contained) for every strand-compatible hit.
template <class Visit>
RSQC_HD void scan_exon_rows(const DevAnnotation &a, int32_t tid, int32_t bs, int32_t be, int rstrand,
                            ClassFlags *f, Visit &&visit) {
    const uint32_t lo = a.ex_range[tid];
    uint32_t ub = rows_upto(a.ex_start, a.ex_range, a.ex_bin, a.bin_off, a.bin_shift, tid, be);
    for (uint32_t i = ub; i > lo;) {
        --i;
        if (a.ex_pmax[i] < bs) break;
        const int32_t fe = a.ex_end[i];
        if (fe < bs) continue;
        const uint32_t fl = a.ex_flags[i];
        const int fs = fl & RSQC_FF_STRAND_MASK;
        if (rstrand != RSQC_STRAND_UNKNOWN && rstrand != fs) continue;
        if (f) {
            if (fs == RSQC_STRAND_FORWARD) f->plus = true; else if (fs == RSQC_STRAND_REVERSE) f->minus = true;
            f->exonic = true;                                              // :337 (even for the phantom base)
            if (fl & RSQC_FF_RIBOSOMAL) f->ribosomal = true;
        }
        // partialIntersect == end - start  <=>  fs <= bs && fe >= be - 1   (src/GTF.cpp:181-186)
        const bool contained = a.ex_start[i] <= bs && fe >= be - 1;
        visit(i, contained);
    }
}

// ---- stage 1: the gate cascade and scalar counters, src/RNASeQC.cpp:254-342,359-360 -----
// Returns true when the record reaches the feature stage; `hq` = highQuality (:330).
RSQC_HD bool gate_cascade(const DevAnnotation &a, const DevParams &p, const Record &r, RecordCounters &out,
                          bool &hq, uint32_t &aligned) {
    const uint32_t fl = r.flag;
    uint64_t bits = RSQC_BIT(RSQC_C_TOTAL_ALIGNMENTS);                                     // :245,397
    out.e1_mm = out.e1_bases = out.e2_mm = out.e2_bases = out.mm = out.bases = out.blocks = 0;
    out.rl_eligible = 0; out.rl_span = 0; out.rl_lqseq = 0; out.error = 0; out.frag_candidate = 0; out.endpos = 0;
    hq = false; aligned = 0;
#define RSQC_LEAVE() do { out.bits = bits; return false; } while (0)
    if (fl & RSQC_FSECONDARY) bits |= RSQC_BIT(RSQC_C_ALTERNATIVE_ALIGNMENTS);             // :254
    if (fl & RSQC_FSUPP) bits |= RSQC_BIT(RSQC_C_SUPPLEMENTARY_ALIGNMENTS);                // :255
    else if (fl & RSQC_FQCFAIL) bits |= RSQC_BIT(RSQC_C_FAILED_VENDOR_QC);                 // :256
    else if (r.mapq < p.mapq_threshold) bits |= RSQC_BIT(RSQC_C_LOW_MAPPING_QUALITY);      // :257
    const bool has_ch = (r.tagbits & RSQC_TB_HAS_CH) != 0;
    if ((fl & RSQC_FSUPP) && !has_ch) {                                                    // :258-262
        bits |= RSQC_BIT(RSQC_C_CHIMERIC_AUTO);
        if (p.exclude_chimeric) RSQC_LEAVE();
    }
    if (fl & (RSQC_FSECONDARY | RSQC_FQCFAIL | RSQC_FSUPP)) RSQC_LEAVE();                  // :263
    bits |= RSQC_BIT(RSQC_C_UNIQUE_VENDOR_PASSED);
    if (!(fl & RSQC_FPAIRED)) bits |= RSQC_BIT(RSQC_C_UNPAIRED_READS);
    if (fl & RSQC_FUNMAP) RSQC_LEAVE();                                                    // :268
    bits |= RSQC_BIT(RSQC_C_MAPPED_READS);
    bits |= (fl & RSQC_FDUP) ? RSQC_BIT(RSQC_C_MAPPED_DUPLICATE_READS) : RSQC_BIT(RSQC_C_MAPPED_UNIQUE_READS);

    // one CIGAR walk: reference length (bam_endpos), aligned size, block count
    uint32_t ref_len = 0, nblocks = 0;
    bool bad = false;
    for (uint32_t i = 0; i < r.n_cigar; ++i) {
        const uint32_t c = r.cigar[i], op = c & 0xf, len = c >> 4;
        if (op > 8) bad = true;                                     // Expression.cpp:61-63
        if (cigar_is_ref(op)) ref_len += len;
        if (cigar_is_block(op)) { aligned += len; ++nblocks; }
    }
    // bam_endpos: pos + rlen, rlen = 1 for CIGAR-less records or when no reference base is consumed
    const int32_t endpos = r.pos + (int32_t)((r.n_cigar == 0 || ref_len == 0) ? 1u : ref_len);
    out.endpos = endpos;
    out.rl_eligible = 1; out.rl_span = (uint32_t)(endpos - r.pos); out.rl_lqseq = r.l_qseq;   // :275-278
    if (has_ch) {                                                                          // :279-283
        if (fl & RSQC_FREAD1) bits |= RSQC_BIT(RSQC_C_CHIMERIC_TAG);
        if (p.exclude_chimeric) RSQC_LEAVE();
    }
    if ((fl & RSQC_FPAIRED) && !(fl & RSQC_FMUNMAP)) {                                     // :284-292
        if (fl & RSQC_FREAD1) bits |= RSQC_BIT(RSQC_C_TOTAL_MAPPED_PAIRS);
        int32_t d = r.pos - r.mpos; if (d < 0) d = -d;
        if (!(r.tagbits & RSQC_TB_MTID_SAME) || d > p.chimeric_distance) {
            if (fl & RSQC_FREAD1) bits |= RSQC_BIT(RSQC_C_CHIMERIC_AUTO);
            if (p.exclude_chimeric) RSQC_LEAVE();
        }
    }
    int32_t mismatches = 0;
    if (r.tagbits & RSQC_TB_HAS_NM) {                                                      // :295-316
        mismatches = r.nm;
        if (fl & RSQC_FPAIRED) {
            if (fl & RSQC_FREAD1) {
                bits |= RSQC_BIT(RSQC_C_END1_MAPPED_READS);
                out.e1_mm = (uint32_t)mismatches; out.e1_bases = (uint32_t)r.l_qseq;
                bits |= (fl & RSQC_FDUP) ? RSQC_BIT(RSQC_C_DUPLICATE_PAIRS) : RSQC_BIT(RSQC_C_UNIQUE_FRAGMENTS);
            } else {
                bits |= RSQC_BIT(RSQC_C_END2_MAPPED_READS);
                out.e2_mm = (uint32_t)mismatches; out.e2_bases = (uint32_t)r.l_qseq;
            }
        }
        out.mm = (uint32_t)mismatches;
    }
    out.bases = (uint32_t)r.l_qseq;                                                        // :317
    bool discard = false;                                                                  // :319-328
    for (int t = 0; t < p.n_filter_tags; ++t)
        if (r.tagbits & (RSQC_TB_FILTER0 << t)) { discard = true; bits |= RSQC_BIT(RSQC_C_FILTERED_TAG0 + t); }
    if (discard) RSQC_LEAVE();
    hq = ((uint32_t)mismatches <= p.base_mismatch) && (p.unpaired || (fl & RSQC_FPROPER)) &&
         (r.mapq >= p.mapq_threshold);                                                     // :330
    if (r.tid < 0 || r.tid >= a.n_ref) RSQC_LEAVE();                                       // :333-337
    bits |= hq ? RSQC_BIT(RSQC_C_HIGH_QUALITY_READS) : RSQC_BIT(RSQC_C_LOW_QUALITY_READS);
    bits |= RSQC_BIT(RSQC_C_READS_USED);
    if (bad) { out.error = RSQC_ERR_BAD_CIGAR; RSQC_LEAVE(); }
    out.blocks = nblocks;                                                                  // :360
    out.frag_candidate = (hq && (fl & RSQC_FPAIRED)) ? 1u : 0u;                            // :372
    out.bits = bits;
    return true;
#undef RSQC_LEAVE
}

// ---- stage 2: exonAlignmentMetrics, src/Expression.cpp:308-458 ---------------------------
// Returns the feature-stage counter bits, or sets `overflow` (and scatters nothing) when a
// block is fully inside exons of more than K distinct genes.  `Acc` provides
//   void gene_hit(uint32_t gene, bool not_duplicate, uint64_t qhash);  geneCounts/unique + de-dup key
//   void exon_add(uint32_t row, double frac);                          exonCounts[exon] += frac
//   void cov_range(uint32_t row, uint32_t offset, uint32_t len, uint32_t exon_len);   BaseCoverage commit
template <int K, class Acc>
RSQC_HD uint64_t exon_metrics(const DevAnnotation &a, const DevParams &p, const Record &r, bool hq,
                              uint32_t aligned, Acc &acc, bool &overflow) {
    const uint32_t fl = r.flag;
    uint64_t bits = 0;
    overflow = false;
    const int rstrand = read_strand_of(p, fl);
    ClassFlags f = {false, false, false, false, false};
    uint32_t last[K]; int nlast = 0;
    uint32_t cur[K];
    bool first = true, over = false;
    uint32_t nblocks = 0;
    // pass 1: flags + the gene set common to all blocks (:325-374)
    {
        int32_t start = r.pos + 1;                                                         // Expression.cpp:31
        for (uint32_t i = 0; i < r.n_cigar; ++i) {
            const uint32_t c = r.cigar[i], op = c & 0xf, len = c >> 4;
            if (cigar_is_block(op)) {
                ++nblocks;
                const int32_t bs = start, be = start + (int32_t)len;
                scan_gene_rows(a, r.tid, bs, be, rstrand, f);
                int ncur = 0;
                scan_exon_rows(a, r.tid, bs, be, rstrand, &f, [&](uint32_t row, bool contained) {
                    if (!contained) return;
                    const uint32_t g = a.ex_gene[row];
                    if (first) {                          // genes.front()
                        bool have = false;
                        for (int k = 0; k < nlast; ++k) have |= (last[k] == g);
                        if (!have) { if (nlast < K) last[nlast++] = g; else over = true; }
                    } else {
                        bool have = false;
                        for (int k = 0; k < ncur; ++k) have |= (cur[k] == g);
                        if (!have) { if (ncur < K) cur[ncur++] = g; else over = true; }
                    }
                });
                if (!first) {                             // set_intersection
                    int w = 0;
                    for (int k = 0; k < nlast; ++k) {
                        bool keep = false;
                        for (int j = 0; j < ncur; ++j) keep |= (cur[j] == last[k]);
                        if (keep) last[w++] = last[k];
                    }
                    nlast = w;
                }
                first = false;
            }
            if (cigar_is_ref(op)) start += (int32_t)len;
        }
    }
    if (over) { overflow = true; return 0; }
    const bool do_exon = nlast > 0;                                                        // :393
    if (nblocks >= 1) {                                                                    // :363,395-404
        bool globin = false;
        for (int k = 0; k < nlast; ++k) globin |= (a.gene_globin[last[k]] != 0);
        if (!globin) {
            bits |= RSQC_BIT(RSQC_C_NON_GLOBIN_READS);
            if (fl & RSQC_FDUP) bits |= RSQC_BIT(RSQC_C_NON_GLOBIN_DUPLICATE_READS);
        }
    }
    // pass 2: commit (only HQ records counted to at least one gene, :377-392)
    if (hq && nlast > 0) {
        int32_t start = r.pos + 1;
        for (uint32_t i = 0; i < r.n_cigar; ++i) {
            const uint32_t c = r.cigar[i], op = c & 0xf, len = c >> 4;
            if (cigar_is_block(op)) {
                const int32_t bs = start, be = start + (int32_t)len;
                scan_exon_rows(a, r.tid, bs, be, rstrand, (ClassFlags *)nullptr, [&](uint32_t row, bool contained) {
                    if (!contained) return;
                    const uint32_t g = a.ex_gene[row];
                    bool in_last = false;
                    for (int k = 0; k < nlast; ++k) in_last |= (last[k] == g);
                    if (!in_last) return;
                    if (len > 0) acc.exon_add(row, (double)len / (double)aligned);         // :345, Metrics.cpp:59-66
                    acc.cov_range(row, (uint32_t)(bs - a.ex_start[row]), len,
                                  (uint32_t)(a.ex_end[row] - a.ex_start[row] + 1));        // Metrics.cpp:96-124
                });
            }
            if (cigar_is_ref(op)) start += (int32_t)len;
        }
        if (aligned > 0)                                   // Collector::queryGene, :380
            for (int k = 0; k < nlast; ++k) acc.gene_hit(last[k], !(fl & RSQC_FDUP), r.qhash);
    }
    // classification counters, :407-457
    if (!f.exonic) {
        if (f.intragenic) {
            bits |= RSQC_BIT(RSQC_C_INTRONIC_READS) | RSQC_BIT(RSQC_C_INTRAGENIC_READS);
            if (hq) bits |= RSQC_BIT(RSQC_C_HQ_INTRONIC_READS) | RSQC_BIT(RSQC_C_HQ_INTRAGENIC_READS);
        } else {
            bits |= RSQC_BIT(RSQC_C_INTERGENIC_READS);
            if (hq) bits |= RSQC_BIT(RSQC_C_HQ_INTERGENIC_READS);
        }
    } else if (do_exon) {
        bits |= RSQC_BIT(RSQC_C_EXONIC_READS) | RSQC_BIT(RSQC_C_INTRAGENIC_READS);
        if (hq) bits |= RSQC_BIT(RSQC_C_HQ_EXONIC_READS) | RSQC_BIT(RSQC_C_HQ_INTRAGENIC_READS);
    } else {
        bits |= RSQC_BIT(RSQC_C_AMBIGUOUS_READS);
        if (hq) bits |= RSQC_BIT(RSQC_C_HQ_AMBIGUOUS_READS);
    }
    if (f.ribosomal) bits |= RSQC_BIT(RSQC_C_RRNA_READS);
    if ((f.minus != f.plus) && (p.unpaired || (fl & RSQC_FPAIRED))) {
        const bool sense = (fl & RSQC_FREVERSE) ? f.minus : f.plus;
        if (p.unpaired || (fl & RSQC_FREAD1)) bits |= sense ? RSQC_BIT(RSQC_C_END1_SENSE) : RSQC_BIT(RSQC_C_END1_ANTISENSE);
        else bits |= sense ? RSQC_BIT(RSQC_C_END2_SENSE) : RSQC_BIT(RSQC_C_END2_ANTISENSE);
    }
    return bits;
}

// fragmentSizeMetrics block test (src/Expression.cpp:490-507): every block must hit exactly one
// BED interval, be fully inside it, and all blocks the same interval.  Returns the BED row or -1.
RSQC_HD int32_t bed_interval_of(const DevAnnotation &a, const Record &r) {
    const uint32_t lo = a.bed_range[r.tid], hi = a.bed_range[r.tid + 1];
    if (lo == hi) return -1;
    int32_t name = -1; bool first = true;
    int32_t start = r.pos + 1;
    for (uint32_t i = 0; i < r.n_cigar; ++i) {
        const uint32_t c = r.cigar[i], op = c & 0xf, len = c >> 4;
        if (cigar_is_block(op)) {
            const int32_t bs = start, be = start + (int32_t)len;
            uint32_t ub = upper_bound_rows(a.bed_start, lo, hi, be);
            int hits = 0; uint32_t hit = 0;
            for (uint32_t k = ub; k > lo;) {
                --k;
                if (a.bed_pmax[k] < bs) break;
                if (a.bed_end[k] < bs) continue;
                ++hits; hit = k;
            }
            if (hits == 1 && a.bed_start[hit] <= bs && a.bed_end[hit] >= be - 1) {
                if (first) name = (int32_t)hit;
                else if (name != (int32_t)hit) return -1;
            } else return -1;
            first = false;
        }
        if (cigar_is_ref(op)) start += (int32_t)len;
    }
    return name;
}

}  // namespace rsqc
