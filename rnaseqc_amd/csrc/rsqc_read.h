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

// section marks of the profiling build of K1 (rsqc_kernels.hip, -DRSQC_K1_PROF); nothing otherwise
#ifndef RSQC_MARK
#define RSQC_MARK(sec)
#endif
// (-DK1E_STAGE_MARKS, profiling build only: the marks sit INSIDE the feature stage -- queue read, rank words, entries, flags, commit -- and
//  phase A's own marks are off, see rsqc_k1.h)
#if defined(K1E_STAGE_MARKS)
#define K1E_SMARK(sec) RSQC_MARK(sec)
#else
#define K1E_SMARK(sec)
#endif
#ifndef RSQC_EVENT
#define RSQC_EVENT(id, cond)          // profiling build: counts the tiles in which some lane takes a slow branch
#endif

namespace rsqc {

// base[idx] with the byte offset formed in 32 bits: the device code then addresses the table as
// (scalar base) + (one 32-bit lane offset) instead of a 64-bit address per lane.  Every table loaded
// through this is smaller than 4 GiB by construction (HostIndex::build checks).
template <class T> RSQC_HD T ld32(const T *base, uint32_t idx) {
    return *reinterpret_cast<const T *>(reinterpret_cast<const char *>(base) + (uint32_t)(idx * (uint32_t)sizeof(T)));
}

// ---- device-resident annotation index ------------------------------------------
// One 16-byte row per interval so that a candidate costs one vector load.
struct ExonRow {
    int32_t start, end;      // 1-based closed
    uint32_t cov;            // index of the exon's first base in the per-base coverage array
    uint32_t gf;             // gene id (low 26 bits) | RowFlags << 26
};
// The running max of `end` over the contig's rows up to row i (what bounds the downward walk of a
// query) lives in a separate column, DevAnnotation::ex_pmax: for most rows it equals the row's own
// end, and ROWF_PMAX_EXT in `gf` says when it does not, so the common case never loads it.
// Gene rows only contribute flags (intragenic, strand, rRNA; src/Expression.cpp:331-333,352-358), so they are
// stored as BREAKPOINTS: between two consecutive gene boundaries the set of covering genes is constant
// and so is the union of their flags per strand class.  A block query ORs the masks of the (1-2)
// elementary intervals it touches -- independent of how deeply genes nest.
struct GeneBreak {
    int32_t pos;             // first position of the elementary interval
    uint32_t mask;           // bit s (s = RSQC_STRAND_*): a gene of strand class s covers it; bit 3+s: ... a ribosomal one
};
constexpr uint32_t ROW_GENE_MASK = (1u << 26) - 1u;
constexpr int ROW_FLAG_SHIFT = 26;     // bits 26-27 strand, 28 ribosomal, 29 the row's gene is a globin, 30 pmax > end, 31 closed to the left
constexpr uint32_t ROWF_RIBOSOMAL = 4u, ROWF_GLOBIN = 8u, ROWF_PMAX_EXT = 16u;
// ROWF_LEFT_CLOSED: every row before this one (same contig) ends before this row starts, i.e. running max of end over the
// earlier rows < start.  A query block that starts at or after such a row cannot overlap anything below it: the downward
// walk stops there without looking at the next row (which in a bin with several exon starts is usually the only reason
// to look further down -- measured: 41 % of the lanes of a tile went back to memory for that before the flag existed).
constexpr uint32_t ROWF_LEFT_CLOSED = 32u;

struct ContigInfo {          // 32 bytes per contig
    uint32_t ex_lo, ex_hi;   // exon rows of the contig
    uint32_t gb_lo, gb_hi;   // gene breakpoints of the contig
    uint32_t bin_base;       // first entry of the contig in the two bin tables
    uint32_t n_bins;         // bins of 2^bin_shift bases (0 = contig without features)
    uint32_t rk_base;        // first word of the contig in the rank table (EiRank)
    uint32_t rk_words;       // words of 64 positions the contig has there (0 = contig without features)
};

// ---- elementary intervals (the index of the per-record kernel since round 3) ------------------------------------------
// The start and end + 1 of every gene and exon row of a contig cut it into ELEMENTARY INTERVALS inside which the set of
// covering features is constant.  One 32-byte entry per interval carries everything a block needs from the features
// that cover it: the class bits of the covering genes and exons (per strand class, so that --stranded is a mask) and
// the (at most two) covering exons as ready-made commit operands -- exon id (= accumulator index), gene | flags,
// and `cov - start` so that the coverage index of a block is one addition.  An interval covered by more than two exons
// is marked EIM_DEEP and its records take the general code (classify_slow_kernel).
//   block [bs, be] (be inclusive, src/Expression.cpp:111, src/GTF.cpp:171-179):
//     class flags      = OR of the masks of intervals find(bs) .. find(be)
//     containing exons = refs(find(bs)) that are also refs(find(max(bs, be - 1)))          (src/GTF.cpp:181-186)
// find(x) = the last breakpoint <= x comes from a bit vector with one bit per position and a running count per 64
// positions (EiRank): ONE 16-byte load and a popcount, no walk and no dependent second load.  Every contig with
// features starts with a sentinel interval at position 0 (mask 0, no exons).
struct EiEntry {
    int32_t pos;             // first position of the interval
    uint32_t mask;           // bits 0-5: GeneBreak::mask; bit 6+s: an exon of strand class s covers it; bit 9+s: ... a ribosomal one; EIM_DEEP
    uint32_t eidA, eidB;     // exon ids of the covering exons (higher row first), EI_NONE = none
    uint32_t gfA, cdA;       // ExonRow::gf of A (gene | RowFlags << 26); ExonRow::cov - ExonRow::start (mod 2^32)
    uint32_t gfB, cdB;
};
struct EiRank { uint32_t lo, hi; uint32_t rank; uint32_t pad; };   // bits of 64 positions; global index of the first breakpoint at or after the word
constexpr uint32_t EI_NONE = 0xFFFFFFFFu;
constexpr uint32_t EIM_DEEP = 1u << 31;
constexpr int EIM_EXON_SHIFT = 6;

// Tables read only by the --legacy rules (legacy_metrics below): the gene ROWS themselves (the default rules need
// only the breakpoint masks) and the rank of every row in the reference's one start-sorted list of genes and exons.
struct GeneRow {
    int32_t start, end;      // 1-based closed
    uint32_t gf;             // gene id (low 26 bits) | feature flags << 26
    uint32_t ord;            // rank in the contig's combined (gene + exon) list, src/RNASeQC.cpp:150-152
};
struct LegacyTables {
    const GeneRow *gr;                 // sorted by (contig, start)
    const int32_t *gr_pmax;            // running max of end per contig
    const uint32_t *gr_range;          // [n_contigs + 1]
    const uint32_t *ex_ord;            // per exon row: rank in the combined list
    const uint32_t *gr_binhi;          // per bin (ContigInfo::bin_base / n_bins, DevAnnotation::bin_shift): first gene row with start >= (b + 1) << shift
};

struct DevAnnotation {
    int32_t n_ref, n_contigs, n_genes, n_listed, n_exons;
    int32_t bin_shift;
    const ExonRow *ex;                 // sorted by (contig, start)
    const int32_t *ex_pmax;            // running max of end, per row (see ExonRow)
    const EiEntry *ei;                 // elementary intervals, sorted by (contig, pos)
    const EiRank *ei_rank;             // rank table: ContigInfo::rk_base + (pos >> 6)
    const uint32_t *ei_coarse;         // (ContigInfo::rk_base >> 3) + (pos >> 9): 1 + the interval that covers 1024 breakpoint-free positions, or 0 (rsqc_index.h)
    const GeneBreak *gb;               // sorted by (contig, pos)
    const ContigInfo *contig;          // [n_contigs]
    // bin tables: ex_binhi = first exon row whose start >= (bin + 1) << bin_shift;
    //             gb_bin   = first breakpoint whose pos > bin << bin_shift
    const uint32_t *ex_binhi, *gb_bin;
    // per-base coverage: exons of a gene are contiguous (exonsForGene order) and every gene is
    // followed by one pad slot, so a block's -1 at offset+len always lands inside the array and a
    // plain prefix sum over the gene reproduces BaseCoverage's per-exon vectors
    const uint32_t *ex_cov;            // offset of an exon row's first base (== ExonRow::cov; used by the end-of-file stage)
    const uint32_t *ex_id;             // exon row -> exon id (exonList order): accumulators are indexed by id
    // BED rows (sorted by contig,start), optional
    const int32_t  *bed_start, *bed_end, *bed_pmax;
    const uint32_t *bed_range;         // [n_contigs+1]
    // bins of 2^RSQC_BED_BIN_SHIFT positions per contig: bed_binhi[bed_bin_base[contig] + bin] = first BED row of the contig whose start
    // >= (bin + 1) << shift (one load and a step or two down instead of a 14-step binary search per block; null = search)
    const uint32_t *bed_binhi, *bed_bin_base;
    int32_t have_bed;
    const LegacyTables *legacy;        // device copy of the struct above (always built; read under DevParams::legacy)
};

struct DevParams {
    uint32_t mapq_threshold, base_mismatch;
    int32_t  chimeric_distance;
    int32_t  stranded, unpaired, exclude_chimeric, n_filter_tags;
    int32_t  legacy;   // --legacy rules (rsqc_params.legacy)
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

// Where the one-per-record counter increments go, and the BOOLEAN DOMAIN the conditions live in.
// BitSink: one record, conditions are `bool`, the increments a 64-bit set (general code, host emulation).
// WaveSink (the per-record kernels): the whole wave runs the code converged, and a condition is a 64-bit LANE MASK in
// scalar registers (LaneMask).  A primitive test (`prim`: a compare on a per-lane value) is ONE v_cmp that writes the mask;
// and / or / not are scalar instructions; a counter is s_bcnt1 of the mask dropped into lane `c` of one vector register
// (v_writelane); a select reads the mask back as the condition of v_cndmask (`lane`).  Round 3 kept the conditions as
// per-lane `bool`s and took `__ballot` of every derived one: the compiler materialises such a bool in a VGPR and compares
// it again (v_cndmask + v_cmp per ballot -- 2 of the 3 vector instructions every counter cost; tools/k1_sections.py).
struct BitSink {
    using B = bool;
    uint64_t bits = 0;
    static RSQC_HD bool prim(bool c) { return c; }
    static RSQC_HD bool lane(bool c) { return c; }
    static RSQC_HD bool any(bool c) { return c; }
    template <int C> RSQC_HD void add(bool cond) { bits |= cond ? RSQC_BIT(C) : 0ull; }
};
// a condition of all 64 lanes (every lane of the wave is active where these are used, so `!` is a plain complement)
struct LaneMask { uint64_t m; };
RSQC_HD LaneMask operator&&(LaneMask a, LaneMask b) { return LaneMask{a.m & b.m}; }
RSQC_HD LaneMask operator||(LaneMask a, LaneMask b) { return LaneMask{a.m | b.m}; }
RSQC_HD LaneMask operator!(LaneMask a) { return LaneMask{~a.m}; }
RSQC_HD LaneMask operator!=(LaneMask a, LaneMask b) { return LaneMask{a.m ^ b.m}; }
RSQC_HD LaneMask operator&&(LaneMask a, bool u) { return LaneMask{u ? a.m : 0ull}; }            // (wave-uniform operand)
RSQC_HD LaneMask operator&&(bool u, LaneMask a) { return LaneMask{u ? a.m : 0ull}; }
RSQC_HD LaneMask operator||(LaneMask a, bool u) { return LaneMask{u ? ~0ull : a.m}; }
RSQC_HD LaneMask operator||(bool u, LaneMask a) { return LaneMask{u ? ~0ull : a.m}; }
#if defined(__HIPCC__)
struct WaveSink {
    using B = LaneMask;
    uint32_t vec = 0;                          // lane c: records of this tile that increment counter c
    static __device__ __forceinline__ LaneMask prim(bool c) {
#if defined(__HIP_DEVICE_COMPILE__)
        return LaneMask{__builtin_amdgcn_ballot_w64(c)};
#else
        return LaneMask{c ? ~0ull : 0ull};
#endif
    }
    static __device__ __forceinline__ bool lane(LaneMask x) {
#if defined(__HIP_DEVICE_COMPILE__)
        return __builtin_amdgcn_inverse_ballot_w64(x.m);
#else
        return x.m != 0;
#endif
    }
    static __device__ __forceinline__ bool any(LaneMask x) { return x.m != 0ull; }
    template <int C> __device__ __forceinline__ void add(LaneMask cond) {
        static_assert(C >= 0 && C < 64, "one lane per counter");
#if defined(__HIP_DEVICE_COMPILE__)
        const uint32_t n = (uint32_t)__popcll(cond.m);                   // s_bcnt1_i32_b64 of the condition mask
        asm("v_writelane_b32 %0, %1, %2" : "+v"(vec) : "s"(n), "n"(C));  // (no clang builtin for v_writelane in this toolchain)
#else
        (void)cond;
#endif
    }
    template <int C> __device__ __forceinline__ void add(bool cond) { add<C>(prim(cond)); }
};
#elif defined(RSQC_WAVE_EMU)
struct WaveSink {                              // the same on the host's wave emulation (tests/hostemu/wavemu.h)
    using B = LaneMask;
    uint32_t vec = 0;
    static LaneMask prim(bool c) { return LaneMask{(uint64_t)__ballot(c)}; }
    static bool lane(LaneMask x) { return ((x.m >> (threadIdx.x & 63u)) & 1ull) != 0; }
    static bool any(LaneMask x) { return x.m != 0ull; }
    template <int C> void add(LaneMask cond) {
        const uint32_t n = (uint32_t)__builtin_popcountll(cond.m);
        if ((int)(threadIdx.x & 63u) == C) vec = n;
    }
    template <int C> void add(bool cond) { add<C>(prim(cond)); }
};
#endif
#define RSQC_COUNT(sink, c, cond) (sink).template add<(c)>(cond)

constexpr int RSQC_BED_BIN_SHIFT = 12;
constexpr int FAST_SET = 2;    // genes per block handled on the fast path (registers)
#ifndef RSQC_FAST_BLOCKS
#define RSQC_FAST_BLOCKS 4
#endif
constexpr int FAST_BLOCKS = RSQC_FAST_BLOCKS; // aligned blocks per record on the fast path; more -> slow path
constexpr int FAST_HITS = 2;   // exons fully containing one block on the fast path

// the first FAST_BLOCKS aligned blocks of a record (extractBlocks, src/Expression.cpp:26-67)
struct Blocks { int32_t bs[FAST_BLOCKS]; uint32_t len[FAST_BLOCKS]; uint32_t nb; };

constexpr int MID_SET = 4;     // ... first tier of the slow path (records with many blocks)
constexpr int SLOW_SET = 32;   // ... second tier (scratch); more -> RSQC_ERR_CAPACITY

RSQC_HD bool cigar_is_ref(uint32_t op) { return op == 0 || op == 2 || op == 3 || op == 7 || op == 8; }
RSQC_HD bool cigar_is_block(uint32_t op) { return op == 0 || op == 7 || op == 8; }

// first row in [lo,hi) with start > x  (BED rows only)
RSQC_HD uint32_t upper_bound_rows(const int32_t *start, uint32_t lo, uint32_t hi, int32_t x) {
    while (lo < hi) {
        uint32_t mid = lo + ((hi - lo) >> 1);
        if (start[mid] <= x) lo = mid + 1; else hi = mid;
    }
    return lo;
}

// feature_strand, src/Expression.cpp:119-125
RSQC_HD int read_strand_of(const DevParams &p, uint32_t flag) {
    if (p.stranded == RSQC_STRAND_UNKNOWN) return RSQC_STRAND_UNKNOWN;
    bool target = (flag & RSQC_FREVERSE) != 0;
    if ((p.stranded == RSQC_STRAND_FORWARD) ^ ((flag & RSQC_FREAD1) != 0)) target = !target;
    return target ? RSQC_STRAND_REVERSE : RSQC_STRAND_FORWARD;
}

struct ClassFlags { bool intragenic, plus, minus, ribosomal, exonic; };

// union of the gene masks over positions [bs, be] of a contig
RSQC_HD uint32_t gene_mask(const DevAnnotation &a, const ContigInfo &ci, int32_t bs, int32_t be) {
    if (ci.gb_hi == ci.gb_lo || ci.n_bins == 0 || be < 0) return 0u;
    if (bs < 0) bs = 0;
    uint32_t b = (uint32_t)bs >> a.bin_shift;
    if (b >= ci.n_bins) b = ci.n_bins - 1;
    uint32_t nxt = ld32(a.gb_bin, ci.bin_base + b);         // first breakpoint with pos > bin start
    while (nxt < ci.gb_hi && ld32(a.gb, nxt).pos <= bs) ++nxt;   // ... with pos > bs
    uint32_t mask = nxt > ci.gb_lo ? ld32(a.gb, nxt - 1).mask : 0u;
    while (nxt < ci.gb_hi) {
        const GeneBreak g = ld32(a.gb, nxt);
        if (g.pos > be) break;
        mask |= g.mask; ++nxt;
    }
    return mask;
}
// gene-row part of the feature loop (src/Expression.cpp:331-333,352-358) from a mask
RSQC_HD void apply_gene_mask(uint32_t mask, int rstrand, ClassFlags &f) {
    uint32_t present = mask & 7u, ribo = (mask >> 3) & 7u;
    if (rstrand != RSQC_STRAND_UNKNOWN) { present &= 1u << rstrand; ribo &= 1u << rstrand; }   // :331
    if (present) f.intragenic = true;
    if (present & (1u << RSQC_STRAND_FORWARD)) f.plus = true;
    if (present & (1u << RSQC_STRAND_REVERSE)) f.minus = true;
    if (ribo) f.ribosomal = true;
}

// The overlap query of one block [bs, be] (be inclusive: the reference's intersectInterval on a
// block whose end is exclusive, src/GTF.cpp:171-179, and its scan bound `start <= block.end`,
// src/Expression.cpp:111).  The bin table gives the last row whose start can be <= be; rows are
// walked downwards until the running max of `end` drops below bs.  Equivalent to the reference's
// linear scan of the trimmed, start-sorted list on coordinate-sorted input.
// Gene rows set flags; exon rows call visit(row_index, row, contained).
template <class Visit>
RSQC_HD void query_block(const DevAnnotation &a, const ContigInfo &ci, int32_t bs, int32_t be, int rstrand,
                         ClassFlags *f, Visit &&visit) {
    if (ci.n_bins == 0 || be < 0) return;
    uint32_t b = (uint32_t)be >> a.bin_shift;
    if (b >= ci.n_bins) b = ci.n_bins - 1;
    const uint32_t ehi = a.ex_binhi[ci.bin_base + b];
    if (f) apply_gene_mask(gene_mask(a, ci, bs, be), rstrand, *f);
    for (uint32_t i = ehi; i > ci.ex_lo;) {
        --i;
        const ExonRow row = a.ex[i];
        if (a.ex_pmax[i] < bs) break;
        const bool last = ((row.gf >> ROW_FLAG_SHIFT) & ROWF_LEFT_CLOSED) && row.start <= bs;   // nothing below can reach the block
        if (row.start > be || row.end < bs) { if (last) break; continue; }
        const uint32_t fl = row.gf >> ROW_FLAG_SHIFT;
        const int fs = (int)(fl & RSQC_FF_STRAND_MASK);
        if (rstrand != RSQC_STRAND_UNKNOWN && rstrand != fs) { if (last) break; continue; }
        if (f) {
            if (fs == RSQC_STRAND_FORWARD) f->plus = true; else if (fs == RSQC_STRAND_REVERSE) f->minus = true;
            f->exonic = true;                                                  // :337 (even for the phantom base)
            if (fl & ROWF_RIBOSOMAL) f->ribosomal = true;
        }
        // partialIntersect == end - start  <=>  start <= bs && end >= be - 1   (src/GTF.cpp:181-186)
        visit(i, row, row.start <= bs && row.end >= be - 1);
        if (last) break;
    }
}

// ---- CIGAR walk: extractBlocks (src/Expression.cpp:26-67) and bam_endpos in one pass ----
struct CigarWalk { uint32_t ref_len, nblocks, aligned; bool bad; };
// op classes as bit sets over the BAM op codes MIDNSHP=XB (0..9): aligned block = M,=,X; consumes reference = M,D,N,=,X
constexpr uint32_t CIG_BLOCK_SET = 0x181u, CIG_REF_SET = 0x18Du;
RSQC_HD void cigar_op(uint32_t c, int32_t pos, CigarWalk &w, Blocks &B, bool active = true) {
    const uint32_t op = c & 0xf, len = c >> 4;
    const bool blk = active && ((CIG_BLOCK_SET >> op) & 1u), ref = active && ((CIG_REF_SET >> op) & 1u);
    w.bad = w.bad || (active && op > 8);                            // Expression.cpp:61-63
#pragma unroll
    for (int k = 0; k < FAST_BLOCKS; ++k) {
        const bool here = blk && (uint32_t)k == w.nblocks;
        B.bs[k] = here ? pos + 1 + (int32_t)w.ref_len : B.bs[k];
        B.len[k] = here ? len : B.len[k];
    }
    w.aligned += blk ? len : 0u; w.nblocks += blk ? 1u : 0u;
    w.ref_len += ref ? len : 0u;
}
// `first` holds the record's first 4 CIGAR words, loaded by the caller in one go (words past
// n_cigar are ignored); longer CIGARs continue from memory.
RSQC_HD void walk_cigar(const Record &r, const uint32_t (&first)[4], CigarWalk &w, Blocks &B) {
    w.ref_len = 0; w.nblocks = 0; w.aligned = 0; w.bad = false;
#pragma unroll
    for (int k = 0; k < FAST_BLOCKS; ++k) { B.bs[k] = 0; B.len[k] = 0; }
#pragma unroll
    for (int k = 0; k < 4; ++k) cigar_op(first[k], r.pos, w, B, (uint32_t)k < r.n_cigar);
    for (uint32_t i = 4; i < r.n_cigar; ++i) cigar_op(r.cigar[i], r.pos, w, B);
    B.nb = w.nblocks;
}

// ---- stage 1: the gate cascade and scalar counters, src/RNASeQC.cpp:254-342,359-360 -----
// Pure register arithmetic on the record and its CIGAR summary.  Returns true when the record
// reaches the feature stage; `hq` = highQuality (:330).
// LEGACY: the LegacyMode tests of the loop body (src/RNASeQC.cpp:258,276,279,287) as a compile-time switch, so
// that the default kernel carries none of them.
// LEAN: the caller derives the counters that are differences of others when it flushes its totals (Total Alignments = records
// seen, Mapped Unique = Mapped - Mapped Duplicate, Unique Fragments = End 1 Mapped - Duplicate Pairs, Low Quality = Reads used -
// High Quality: rsqc_k1.h, K1eTables::flush) and the tag filters are skipped as a whole when the run has none.
template <bool LEGACY = false, class Sink = BitSink, bool LEAN = false>
RSQC_HD typename Sink::B gate_cascade_b(const DevAnnotation &a, const DevParams &p, const Record &r, const CigarWalk &w,
                                        RecordCounters &out, typename Sink::B &hq, Sink &cnt, typename Sink::B on) {
    // `on` = the caller's lane holds a record at all (a whole wave runs the cascade converged; see WaveSink)
    // Straight-line form of the cascade: `alive` stays true while the reference's loop body has not hit a
    // `continue`; every counter is added under the conjunction of `alive` and its own condition.  (In a 64-lane
    // wave every early exit is taken by some lane, so branching only adds exec-mask bookkeeping.)
    // Conditions live in the sink's boolean domain B (bool, or a lane mask: and / or / not are scalar instructions there).
    using B = typename Sink::B;
#define RSQC_P(x) Sink::prim(x)
#define RSQC_L(x) Sink::lane(x)
    const uint32_t fl = r.flag;
    const bool excl = p.exclude_chimeric != 0;
    const B paired = RSQC_P((fl & RSQC_FPAIRED) != 0), read1 = RSQC_P((fl & RSQC_FREAD1) != 0), dup = RSQC_P((fl & RSQC_FDUP) != 0);
    const B sec = RSQC_P((fl & RSQC_FSECONDARY) != 0), supp = RSQC_P((fl & RSQC_FSUPP) != 0), qcf = RSQC_P((fl & RSQC_FQCFAIL) != 0);
    if (!LEAN) RSQC_COUNT(cnt, RSQC_C_TOTAL_ALIGNMENTS, on);                                       // :245,397
    RSQC_COUNT(cnt, RSQC_C_ALTERNATIVE_ALIGNMENTS, on && sec);                                     // :254
    RSQC_COUNT(cnt, RSQC_C_SUPPLEMENTARY_ALIGNMENTS, on && supp);                                  // :255
    RSQC_COUNT(cnt, RSQC_C_FAILED_VENDOR_QC, on && !supp && qcf);                                  // :256
    const B lowq = RSQC_P(r.mapq < p.mapq_threshold);
    RSQC_COUNT(cnt, RSQC_C_LOW_MAPPING_QUALITY, on && !supp && !qcf && lowq);                      // :257
    const B has_ch = RSQC_P((r.tagbits & RSQC_TB_HAS_CH) != 0);
    const B supp_auto = on && !LEGACY && supp && !has_ch;                                  // :258-262
    B alive = on && !(supp_auto && excl);
    alive = alive && !(sec || qcf || supp);                                                // :263
    RSQC_COUNT(cnt, RSQC_C_UNIQUE_VENDOR_PASSED, alive);
    RSQC_COUNT(cnt, RSQC_C_UNPAIRED_READS, alive && !paired);
    alive = alive && !RSQC_P((fl & RSQC_FUNMAP) != 0);                                     // :268
    RSQC_COUNT(cnt, RSQC_C_MAPPED_READS, alive);
    RSQC_COUNT(cnt, RSQC_C_MAPPED_DUPLICATE_READS, alive && dup); if (!LEAN) RSQC_COUNT(cnt, RSQC_C_MAPPED_UNIQUE_READS, alive && !dup);
    // bam_endpos: pos + rlen, rlen = 1 for CIGAR-less records or when no reference base is consumed
    // (a record without operations has walked none: ref_len == 0)
    const int32_t endpos = r.pos + (int32_t)(w.ref_len == 0 ? 1u : w.ref_len);
    if (LEGACY) alive = alive && !RSQC_P((uint32_t)(endpos - r.pos) > 100000u);            // :276, LEGACY_MAX_READ_LENGTH
    {
        const bool al = RSQC_L(alive);
        out.endpos = al ? endpos : 0;
        out.rl_eligible = al ? 1u : 0u;                                                    // :275-278
        out.rl_span = al ? (uint32_t)(endpos - r.pos) : 0u; out.rl_lqseq = al ? r.l_qseq : 0;
    }
    const B ch_here = !LEGACY && alive && has_ch;                                          // :279-283
    RSQC_COUNT(cnt, RSQC_C_CHIMERIC_TAG, ch_here && read1);
    alive = alive && !(ch_here && excl);
    const B mate_mapped = alive && paired && !RSQC_P((fl & RSQC_FMUNMAP) != 0);            // :284-292
    RSQC_COUNT(cnt, RSQC_C_TOTAL_MAPPED_PAIRS, mate_mapped && read1);
    int32_t d = r.pos - r.mpos; if (d < 0) d = -d;
    const B far = mate_mapped && (RSQC_P((r.tagbits & RSQC_TB_MTID_SAME) == 0) || RSQC_P(d > p.chimeric_distance) || (LEGACY && RSQC_P(r.tid > 127)));
    RSQC_COUNT(cnt, RSQC_C_CHIMERIC_AUTO, supp_auto || (far && read1));                            // (:258 and :289 exclude each other: :263)
    alive = alive && !(far && excl);
    const B has_nm = alive && RSQC_P((r.tagbits & RSQC_TB_HAS_NM) != 0);                   // :295-316
    const int32_t mismatches = (r.tagbits & RSQC_TB_HAS_NM) ? r.nm : 0;
    const B nm1 = has_nm && paired && read1, nm2 = has_nm && paired && !read1;
    RSQC_COUNT(cnt, RSQC_C_END1_MAPPED_READS, nm1); RSQC_COUNT(cnt, RSQC_C_DUPLICATE_PAIRS, nm1 && dup); if (!LEAN) RSQC_COUNT(cnt, RSQC_C_UNIQUE_FRAGMENTS, nm1 && !dup);
    RSQC_COUNT(cnt, RSQC_C_END2_MAPPED_READS, nm2);
    {
        const bool l1 = RSQC_L(nm1), l2 = RSQC_L(nm2);
        out.e1_mm = l1 ? (uint32_t)mismatches : 0u; out.e1_bases = l1 ? (uint32_t)r.l_qseq : 0u;
        out.e2_mm = l2 ? (uint32_t)mismatches : 0u; out.e2_bases = l2 ? (uint32_t)r.l_qseq : 0u;
        out.mm = RSQC_L(has_nm) ? (uint32_t)mismatches : 0u;
        out.bases = RSQC_L(alive) ? (uint32_t)r.l_qseq : 0u;                               // :317
    }
    B discard = alive && false;                                                            // :319-328
#define RSQC_FILTER_TAG(t) { const B hit = ((t) < p.n_filter_tags) && alive && RSQC_P((r.tagbits & (RSQC_TB_FILTER0 << (t))) != 0); \
                             RSQC_COUNT(cnt, RSQC_C_FILTERED_TAG0 + (t), hit); discard = discard || hit; }
    if (!LEAN || p.n_filter_tags > 0) { RSQC_FILTER_TAG(0) RSQC_FILTER_TAG(1) RSQC_FILTER_TAG(2) RSQC_FILTER_TAG(3) RSQC_FILTER_TAG(4) }
#undef RSQC_FILTER_TAG
    static_assert(RSQC_MAX_FILTER_TAGS == 5, "one line per filter tag above");
    alive = alive && !discard;
    hq = alive && RSQC_P((uint32_t)mismatches <= p.base_mismatch) && ((p.unpaired != 0) || RSQC_P((fl & RSQC_FPROPER) != 0)) &&
         !lowq;                                                                            // :330
    alive = alive && !RSQC_P((uint32_t)r.tid >= (uint32_t)a.n_ref);                        // :333-337 (tid < 0 || tid >= n_ref)
    hq = hq && alive;
    RSQC_COUNT(cnt, RSQC_C_HIGH_QUALITY_READS, hq); if (!LEAN) RSQC_COUNT(cnt, RSQC_C_LOW_QUALITY_READS, alive && !hq); RSQC_COUNT(cnt, RSQC_C_READS_USED, alive);
    const B bad = alive && RSQC_P(w.bad);
    out.error = RSQC_L(bad) ? RSQC_ERR_BAD_CIGAR : 0;
    alive = alive && !bad;
    out.blocks = RSQC_L(alive) ? w.nblocks : 0u;                                           // :360
    out.frag_candidate = RSQC_L(alive && hq && paired) ? 1u : 0u;                          // :372
#undef RSQC_P
#undef RSQC_L
    return alive;
}
template <bool LEGACY = false, class Sink = BitSink, bool LEAN = false>
RSQC_HD bool gate_cascade(const DevAnnotation &a, const DevParams &p, const Record &r, const CigarWalk &w,
                          RecordCounters &out, bool &hq, Sink &cnt, bool on = true) {
    typename Sink::B hqb = Sink::prim(false);
    const typename Sink::B alive = gate_cascade_b<LEGACY, Sink, LEAN>(a, p, r, w, out, hqb, cnt, Sink::prim(on));
    hq = Sink::lane(hqb);
    return Sink::lane(alive);
}
// the same with the counters as a bit set in out.bits
template <bool LEGACY = false>
RSQC_HD bool gate_cascade(const DevAnnotation &a, const DevParams &p, const Record &r, const CigarWalk &w,
                          RecordCounters &out, bool &hq) {
    BitSink sink;
    const bool alive = gate_cascade<LEGACY, BitSink>(a, p, r, w, out, hq, sink);
    out.bits = sink.bits;
    return alive;
}
// convenience form for callers that did not stage the CIGAR words themselves
RSQC_HD bool gate_cascade(const DevAnnotation &a, const DevParams &p, const Record &r, RecordCounters &out,
                          bool &hq, uint32_t &aligned, Blocks &B) {
    uint32_t first[4] = {0, 0, 0, 0};
    for (uint32_t k = 0; k < 4 && k < r.n_cigar; ++k) first[k] = r.cigar[k];
    CigarWalk w;
    walk_cigar(r, first, w, B);
    aligned = w.aligned;
    return p.legacy ? gate_cascade<true>(a, p, r, w, out, hq) : gate_cascade<false>(a, p, r, w, out, hq);
}

// ---- fast path of stage 2 ------------------------------------------------------------------
// Handles the overwhelmingly common records: <= FAST_BLOCKS blocks, every block inside at most
// FAST_HITS exons, gene sets of at most FAST_SET genes, at most NSTAGE commits.  Anything else
// sets `overflow` and is recounted by the general code on the slow path.
//
// The kernel is bound by the LATENCY of dependent loads (a wave waits for the slowest lane of every
// round trip), so the query is organised as two rounds of independent, unconditional loads:
//   round 1  the two bin-table entries of every block                       (fast_load_bins)
//   round 2  per block: the two highest candidate exon rows and the three gene breakpoints
//            around the block start                                         (fast_load_rows)
// and everything else is arithmetic on registers (fast_resolve_block).  Only rows whose running
// max differs from their own end, blocks that still reach a third row, or blocks that span more
// than two gene boundaries go back to memory.  Lanes without a given block load entry 0 of each
// table (one shared line) and ignore it.
struct FastBins { uint32_t ehi[FAST_BLOCKS], nxt[FAST_BLOCKS]; uint32_t have; };   // bit k: block k exists and its contig has features
struct FastRows { ExonRow e0, e1; GeneBreak g0, g1, g2; };

RSQC_HD void fast_load_bins(const DevAnnotation &a, const ContigInfo &ci, const Blocks &B, FastBins &fb) {
    fb.have = 0;
#pragma unroll
    for (int k = 0; k < FAST_BLOCKS; ++k) {
        const int32_t bs = B.bs[k], be = B.bs[k] + (int32_t)B.len[k];
        const bool have = (uint32_t)k < B.nb && ci.n_bins != 0 && be >= 0;
        uint32_t bE = (uint32_t)(be < 0 ? 0 : be) >> a.bin_shift, bG = (uint32_t)(bs < 0 ? 0 : bs) >> a.bin_shift;
        const uint32_t last = ci.n_bins ? ci.n_bins - 1 : 0;
        if (bE > last) bE = last;
        if (bG > last) bG = last;
        const uint32_t vE = ld32(a.ex_binhi, have ? ci.bin_base + bE : 0u);
        const uint32_t vG = ld32(a.gb_bin, have ? ci.bin_base + bG : 0u);
        fb.ehi[k] = have ? vE : ci.ex_lo;
        fb.nxt[k] = have ? vG : ci.gb_hi;
        if (have) fb.have |= 1u << k;
    }
}
RSQC_HD void fast_load_rows(const DevAnnotation &a, const ContigInfo &ci, uint32_t ehi, uint32_t nxt, bool have, FastRows &fr) {
    const uint32_t en = ehi - ci.ex_lo;
    const bool hg = have && ci.gb_hi != ci.gb_lo;
    fr.e0 = ld32(a.ex, en > 0 ? ehi - 1 : 0u);
    fr.e1 = ld32(a.ex, en > 1 ? ehi - 2 : 0u);
    fr.g0 = ld32(a.gb, (hg && nxt > ci.gb_lo) ? nxt - 1 : 0u);
    fr.g1 = ld32(a.gb, (hg && nxt < ci.gb_hi) ? nxt : 0u);
    fr.g2 = ld32(a.gb, (hg && nxt + 1 < ci.gb_hi) ? nxt + 1 : 0u);
}
RSQC_HD int32_t row_pmax(const DevAnnotation &a, const ExonRow &row, uint32_t i) {
    return ((row.gf >> ROW_FLAG_SHIFT) & ROWF_PMAX_EXT) ? ld32(a.ex_pmax, i) : row.end;
}
struct Commit { uint32_t row, cidx, len; };    // exon row, coverage index of the block's first base, block length
constexpr int NSTAGE = 4;

constexpr int SLOW_STAGE = 8;  // staged commits on the slow path (records with many blocks)
template <int K, int NST = NSTAGE>
struct FeatureOut {
    uint64_t bits;              // feature-stage counter bits
    int n_hit; uint32_t hit[K]; // genes to count: geneCounts++, uniqueGeneCounts, (gene, qname) de-dup
    // exonCounts[row] += len/aligned ; coverage[cidx, cidx+len) += 1 for every k with bit k of cmask set
    // (the general code fills commit[0..n_commit) densely; the fast path marks its staged hits in place)
    int n_commit; uint32_t cmask; Commit commit[NST];
};

// small fixed arrays indexed with unrolled compares so that they stay in registers on the GPU
template <int K> RSQC_HD bool set_contains(const uint32_t (&s)[K], int n, uint32_t v) {
    bool have = false;
#pragma unroll
    for (int k = 0; k < K; ++k) have |= (k < n) & (s[k] == v);
    return have;
}
template <int K> RSQC_HD void set_put(uint32_t (&s)[K], int idx, uint32_t v) {
#pragma unroll
    for (int k = 0; k < K; ++k) if (k == idx) s[k] = v;
}

// classification counters of exonAlignmentMetrics, src/Expression.cpp:407-457; `keep` = the record is counted here
// (a record handed to the general code is counted there)
template <class B> struct ClassFlagsT { B intragenic, plus, minus, ribosomal, exonic; };
template <class Sink>
RSQC_HD void class_counts_b(Sink &cnt, const DevParams &p, uint32_t fl, const ClassFlagsT<typename Sink::B> &f,
                            typename Sink::B do_exon, typename Sink::B hq, typename Sink::B keep) {
    using B = typename Sink::B;
    const B intronic = keep && !f.exonic && f.intragenic, intergenic = keep && !f.exonic && !f.intragenic;
    const B exonic = keep && f.exonic && do_exon, ambiguous = keep && f.exonic && !do_exon;
    RSQC_COUNT(cnt, RSQC_C_INTRONIC_READS, intronic); RSQC_COUNT(cnt, RSQC_C_HQ_INTRONIC_READS, intronic && hq);
    RSQC_COUNT(cnt, RSQC_C_INTRAGENIC_READS, intronic || exonic); RSQC_COUNT(cnt, RSQC_C_HQ_INTRAGENIC_READS, (intronic || exonic) && hq);
    RSQC_COUNT(cnt, RSQC_C_INTERGENIC_READS, intergenic); RSQC_COUNT(cnt, RSQC_C_HQ_INTERGENIC_READS, intergenic && hq);
    RSQC_COUNT(cnt, RSQC_C_EXONIC_READS, exonic); RSQC_COUNT(cnt, RSQC_C_HQ_EXONIC_READS, exonic && hq);
    RSQC_COUNT(cnt, RSQC_C_AMBIGUOUS_READS, ambiguous); RSQC_COUNT(cnt, RSQC_C_HQ_AMBIGUOUS_READS, ambiguous && hq);
    RSQC_COUNT(cnt, RSQC_C_RRNA_READS, keep && f.ribosomal);
    const B one_strand = keep && (f.minus != f.plus) && ((p.unpaired != 0) || Sink::prim((fl & RSQC_FPAIRED) != 0));
    const B rev = Sink::prim((fl & RSQC_FREVERSE) != 0);
    const B sense = (rev && f.minus) || (!rev && f.plus);
    const B end1 = (p.unpaired != 0) || Sink::prim((fl & RSQC_FREAD1) != 0);
    RSQC_COUNT(cnt, RSQC_C_END1_SENSE, one_strand && end1 && sense); RSQC_COUNT(cnt, RSQC_C_END1_ANTISENSE, one_strand && end1 && !sense);
    RSQC_COUNT(cnt, RSQC_C_END2_SENSE, one_strand && !end1 && sense); RSQC_COUNT(cnt, RSQC_C_END2_ANTISENSE, one_strand && !end1 && !sense);
}
// the same from the CF_* bits of a record (what the feature stages hold)
template <class Sink>
RSQC_HD void class_counts_cf(Sink &cnt, const DevParams &p, uint32_t fl, uint32_t cf, typename Sink::B do_exon, typename Sink::B hq, typename Sink::B keep) {
    ClassFlagsT<typename Sink::B> f;
    f.intragenic = Sink::prim((cf & 1u /*CF_INTRAGENIC*/) != 0); f.plus = Sink::prim((cf & 2u /*CF_PLUS*/) != 0); f.minus = Sink::prim((cf & 4u /*CF_MINUS*/) != 0);
    f.ribosomal = Sink::prim((cf & 8u /*CF_RIBOSOMAL*/) != 0); f.exonic = Sink::prim((cf & 16u /*CF_EXONIC*/) != 0);
    class_counts_b(cnt, p, fl, f, do_exon, hq, keep);
}
template <class Sink>
RSQC_HD void class_counts(Sink &cnt, const DevParams &p, uint32_t fl, const ClassFlags &f, bool do_exon, bool hq, bool keep) {
    ClassFlagsT<typename Sink::B> fb;
    fb.intragenic = Sink::prim(f.intragenic); fb.plus = Sink::prim(f.plus); fb.minus = Sink::prim(f.minus);
    fb.ribosomal = Sink::prim(f.ribosomal); fb.exonic = Sink::prim(f.exonic);
    class_counts_b(cnt, p, fl, fb, Sink::prim(do_exon), Sink::prim(hq), Sink::prim(keep));
}
RSQC_HD uint64_t class_bits(const DevParams &p, uint32_t fl, const ClassFlags &f, bool do_exon, bool hq) {
    BitSink s;
    class_counts(s, p, fl, f, do_exon, hq, true);
    return s.bits;
}

// ---- the fast feature stage ------------------------------------------------------------------
// Written as straight-line predicated arithmetic: in a 64-lane wave every branch of a per-record
// decision tree is taken by some lane, so divergent control flow only adds exec-mask bookkeeping.
// Branches remain only where a lane must go back to memory (rows beyond the two staged ones, a
// running-max column entry, more than two gene boundaries) and around the second round of loads
// (wave-uniform: skipped when no lane has a third block).
//
// Commit slots are positional: slots 2b and 2b + 1 hold the (at most two) exons that contain block b.
// `cmask` selects the slots whose gene survived the intersection over all blocks.
constexpr int NSLOT = 2 * FAST_BLOCKS;
constexpr uint32_t CF_INTRAGENIC = 1u, CF_PLUS = 2u, CF_MINUS = 4u, CF_RIBOSOMAL = 8u, CF_EXONIC = 16u;

#if defined(__HIP_DEVICE_COMPILE__) || defined(RSQC_WAVE_EMU)
#define RSQC_ANY_LANE(x) (__ballot(x) != 0ull)
#else
#define RSQC_ANY_LANE(x) (x)
#endif

// gene-row part of the feature loop (apply_gene_mask) as flag bits
RSQC_HD uint32_t gene_class_flags(uint32_t mask, int rstrand) {
    uint32_t present = mask & 7u, ribo = (mask >> 3) & 7u;
    const uint32_t sel = rstrand == RSQC_STRAND_UNKNOWN ? 7u : 1u << rstrand;                  // :331
    present &= sel; ribo &= sel;
    return (present ? CF_INTRAGENIC : 0u) | ((present & (1u << RSQC_STRAND_FORWARD)) ? CF_PLUS : 0u) |
           ((present & (1u << RSQC_STRAND_REVERSE)) ? CF_MINUS : 0u) | (ribo ? CF_RIBOSOMAL : 0u);
}
// one exon row against one block: bit 0 = overlaps (strand-compatible), bit 1 = contains the block;
// the class flags of an overlapping row are OR-ed into cf
RSQC_HD uint32_t exon_row_test(const ExonRow &row, bool reach, int32_t bs, int32_t be, int rstrand, uint32_t &cf) {
    const uint32_t fl = row.gf >> ROW_FLAG_SHIFT;
    const int fs = (int)(fl & RSQC_FF_STRAND_MASK);
    const bool ov = reach && row.start <= be && row.end >= bs && (rstrand == RSQC_STRAND_UNKNOWN || rstrand == fs);
    const uint32_t add = CF_EXONIC | (fs == RSQC_STRAND_FORWARD ? CF_PLUS : fs == RSQC_STRAND_REVERSE ? CF_MINUS : 0u) |
                         ((fl & ROWF_RIBOSOMAL) ? CF_RIBOSOMAL : 0u);
    cf |= ov ? add : 0u;
    // partialIntersect == end - start  <=>  start <= bs && end >= be - 1   (src/GTF.cpp:181-186)
    const bool con = ov && row.start <= bs && row.end >= be - 1;
    return (ov ? 1u : 0u) | (con ? 2u : 0u);
}

// What the fast feature stage returns: slot s = 2b + j commits block b (length B.len[b]) to exon `row[s]` at
// coverage index `cidx[s]` when bit s of cmask is set.
struct FastOut {
    int n_hit; uint32_t hit[FAST_SET];
    uint32_t cmask;
    uint32_t row[NSLOT], cidx[NSLOT];
};

// `ci` is the ContigInfo of the record's contig (wave-uniform in the kernel).  ROUND = blocks whose row
// loads are in flight together (registers vs. round trips).
// `cnt` takes the feature-stage counters of the records handled here (`lane_on` = this lane holds such a record; with a
// WaveSink the function is called by the whole wave); a record that sets `overflow` is counted by the general code instead.
template <int ROUND = 2, class Sink = BitSink>
RSQC_HD void exon_metrics_fast(const DevAnnotation &a, const DevParams &p, const ContigInfo &ci, uint32_t fl,
                               const Blocks &B, bool hq, uint32_t aligned, FastOut &out, bool &overflow, Sink &cnt, bool lane_on = true,
                               const FastBins *staged_bins = nullptr) {
    // staged_bins: round 1 (fast_load_bins) already issued by the caller -- the per-record kernel starts it right after the
    // CIGAR walk, so that the bin entries travel while the gate cascade computes
    out.n_hit = 0; out.cmask = 0;
    const int rstrand = read_strand_of(p, fl);
    uint32_t cf = 0;                                          // CF_* class flags of the whole record
    uint32_t la = 0, lb = 0; bool va = false, vb = false, ga = false, gb = false;   // gene set common to all blocks so far
    // bit s of con: slot s holds a containing exon; of ma / mb: its gene is the first / second gene of block 0's
    // set (the final set can only be a subset of that one, so these two bits decide the slot at the end)
    uint32_t con = 0, ma = 0, mb = 0;
    bool over = B.nb > (uint32_t)FAST_BLOCKS;
#pragma unroll
    for (int k = 0; k < NSLOT; ++k) { out.row[k] = 0; out.cidx[k] = 0; }
    FastBins fb;
    if (staged_bins) fb = *staged_bins; else fast_load_bins(a, ci, B, fb);
    RSQC_MARK(3);
#pragma unroll
    for (int b0 = 0; b0 < FAST_BLOCKS; b0 += ROUND) {
        if (b0 > 0 && !RSQC_ANY_LANE(B.nb > (uint32_t)b0)) break;
        RSQC_MARK(4 + b0);
        FastRows fr[ROUND];
#pragma unroll
        for (int j = 0; j < ROUND; ++j) fast_load_rows(a, ci, fb.ehi[b0 + j], fb.nxt[b0 + j], ((fb.have >> (b0 + j)) & 1u) != 0, fr[j]);
#pragma unroll
        for (int j = 0; j < ROUND; ++j) {
            const int b = b0 + j;
            const bool act = (uint32_t)b < B.nb && ((fb.have >> b) & 1u) != 0;
            const int32_t bs = B.bs[b], be = B.bs[b] + (int32_t)B.len[b];
            const uint32_t ehi = fb.ehi[b], nxt = fb.nxt[b];
            const FastRows &r = fr[j];
            // -- genes: gene_mask() on the three staged breakpoints
            {
                const bool hg = act && ci.gb_hi != ci.gb_lo;
                const int32_t bsc = bs < 0 ? 0 : bs;
                const bool v0 = nxt > ci.gb_lo, v1 = nxt < ci.gb_hi, v2 = nxt + 1 < ci.gb_hi, more = nxt + 2 < ci.gb_hi;
                const bool adv = v1 && r.g1.pos <= bsc;                       // the block starts at or after g1
                const bool in1 = v1 && r.g1.pos <= be, in2 = v2 && r.g2.pos <= be;
                uint32_t mask = adv ? r.g1.mask : ((v0 ? r.g0.mask : 0u) | (in1 ? r.g1.mask : 0u));
                mask |= (in2 && (adv || in1)) ? r.g2.mask : 0u;
                const bool deeper = (adv && v2 && r.g2.pos <= bsc) || (in2 && (adv || in1) && more);
                RSQC_EVENT(0, hg && deeper);
                if (hg && deeper) mask = gene_mask(a, ci, bs, be);
                cf |= hg ? gene_class_flags(mask, rstrand) : 0u;
            }
            // -- exons: the two staged rows; further rows only while the running max still reaches the block
            const uint32_t en = ehi - ci.ex_lo;
            int32_t pm0 = r.e0.end, pm1 = r.e1.end;
            {
                const bool x0 = act && en > 0 && ((r.e0.gf >> ROW_FLAG_SHIFT) & ROWF_PMAX_EXT) != 0;
                const bool x1 = act && en > 1 && ((r.e1.gf >> ROW_FLAG_SHIFT) & ROWF_PMAX_EXT) != 0;
                RSQC_EVENT(1, x0 || x1);
                if (x0 || x1) {                                              // both from memory in one round trip
                    const int32_t q0 = ld32(a.ex_pmax, ehi - 1), q1 = ld32(a.ex_pmax, en > 1 ? ehi - 2 : ehi - 1);
                    if (x0) pm0 = q0;
                    if (x1) pm1 = q1;
                }
            }
            // a staged row that is closed to the left and starts at or before the block ends the walk (ROWF_LEFT_CLOSED)
            const bool stop0 = ((r.e0.gf >> ROW_FLAG_SHIFT) & ROWF_LEFT_CLOSED) != 0 && r.e0.start <= bs;
            const bool stop1 = ((r.e1.gf >> ROW_FLAG_SHIFT) & ROWF_LEFT_CLOSED) != 0 && r.e1.start <= bs;
            const bool reach0 = act && en > 0 && pm0 >= bs, reach1 = reach0 && !stop0 && en > 1 && pm1 >= bs;
            const bool reach2 = reach1 && !stop1;
            const uint32_t t0 = exon_row_test(r.e0, reach0, bs, be, rstrand, cf);
            const uint32_t t1 = exon_row_test(r.e1, reach1, bs, be, rstrand, cf);
            const bool k0 = (t0 & 2u) != 0, k1 = (t1 & 2u) != 0;
            // the block's (at most two) containing exons, in row order: A then B
            bool c0 = k0 || k1, c1 = k0 && k1;
            uint32_t rowA = k0 ? ehi - 1 : ehi - 2, gfA = k0 ? r.e0.gf : r.e1.gf, rowB = ehi - 2, gfB = r.e1.gf;
            uint32_t cidxA = k0 ? r.e0.cov + (uint32_t)(bs - r.e0.start) : r.e1.cov + (uint32_t)(bs - r.e1.start);
            uint32_t cidxB = r.e1.cov + (uint32_t)(bs - r.e1.start);
            RSQC_EVENT(2, reach2);
            if (reach2) {
                for (uint32_t i = ehi - 2; i > ci.ex_lo;) {
                    --i;
                    const ExonRow row = ld32(a.ex, i);
                    if (row_pmax(a, row, i) < bs) break;
                    if (exon_row_test(row, true, bs, be, rstrand, cf) & 2u) {
                        const uint32_t cx = row.cov + (uint32_t)(bs - row.start);
                        if (!c0) { c0 = true; rowA = i; gfA = row.gf; cidxA = cx; }
                        else if (!c1) { c1 = true; rowB = i; gfB = row.gf; cidxB = cx; }
                        else over = true;                                                   // a third containing exon
                    }
                    if (((row.gf >> ROW_FLAG_SHIFT) & ROWF_LEFT_CLOSED) && row.start <= bs) break;
                }
            }
            const uint32_t g0 = gfA & ROW_GENE_MASK, g1 = gfB & ROW_GENE_MASK;
            out.row[2 * b] = rowA; out.cidx[2 * b] = cidxA;
            out.row[2 * b + 1] = rowB; out.cidx[2 * b + 1] = cidxB;
            con |= (c0 ? 1u : 0u) << (2 * b) | (c1 ? 1u : 0u) << (2 * b + 1);
            // -- gene set: genes.front() for the first block, set_intersection afterwards (:363-374)
            if (b == 0) {
                la = g0; va = c0; ga = c0 && ((gfA >> ROW_FLAG_SHIFT) & ROWF_GLOBIN) != 0;
                lb = g1; vb = c1 && !(c0 && g1 == g0); gb = vb && ((gfB >> ROW_FLAG_SHIFT) & ROWF_GLOBIN) != 0;
                ma = (c0 ? 1u : 0u) | ((c1 && g1 == g0) ? 2u : 0u); mb = vb ? 2u : 0u;
            } else {
                const bool live = (uint32_t)b < B.nb;            // a block on a feature-less contig intersects with the empty set
                const bool a_in = (c0 && la == g0) || (c1 && la == g1), b_in = (c0 && lb == g0) || (c1 && lb == g1);
                va = live ? (va && a_in) : va; vb = live ? (vb && b_in) : vb;
                ma |= ((c0 && la == g0) ? 1u : 0u) << (2 * b) | ((c1 && la == g1) ? 1u : 0u) << (2 * b + 1);
                mb |= ((c0 && lb == g0) ? 1u : 0u) << (2 * b) | ((c1 && lb == g1) ? 1u : 0u) << (2 * b + 1);
            }
        }
    }
    RSQC_MARK(8);
    over = over || !lane_on;
    overflow = over;
    ga = ga && va; gb = gb && vb;
    const int nlast = (va ? 1 : 0) + (vb ? 1 : 0);
    const bool nonglobin = !over && B.nb >= 1 && !(ga || gb);                              // :363,395-404
    RSQC_COUNT(cnt, RSQC_C_NON_GLOBIN_READS, nonglobin);
    RSQC_COUNT(cnt, RSQC_C_NON_GLOBIN_DUPLICATE_READS, nonglobin && (fl & RSQC_FDUP) != 0);
    if (hq && nlast > 0) {                                                                 // :377-392
        out.cmask = con & ((va ? ma : 0u) | (vb ? mb : 0u));
        if (aligned > 0) {
            out.hit[0] = va ? la : lb; out.hit[1] = lb;
            out.n_hit = nlast;
        }
    }
    ClassFlags f;
    f.intragenic = (cf & CF_INTRAGENIC) != 0; f.plus = (cf & CF_PLUS) != 0; f.minus = (cf & CF_MINUS) != 0;
    f.ribosomal = (cf & CF_RIBOSOMAL) != 0; f.exonic = (cf & CF_EXONIC) != 0;
    class_counts(cnt, p, fl, f, nlast > 0, hq, !over);
}

// ---- the feature stage on elementary intervals (EiEntry / EiRank above) -----------------------------------------------
// find(x): index of the last breakpoint <= x on the contig, and whether x itself is one.  x is clamped into the
// positions the contig's rank words cover (beyond the last breakpoint nothing changes any more).
struct EiFind { uint32_t j; bool at; };
// the rank word that answers find(x) (phase 1 of a block query: the load), and the answer from it (phase 2)
RSQC_HD EiRank ei_find_word(const DevAnnotation &a, const ContigInfo &ci, int32_t x, bool have) {
    const int32_t top = (int32_t)(ci.rk_words << 6) - 1;
    const int32_t xc = x < 0 ? 0 : (x > top ? top : x);
    return ld32(a.ei_rank, have ? ci.rk_base + ((uint32_t)xc >> 6) : 0u);
}
RSQC_HD EiFind ei_find_in(const EiRank &w, const ContigInfo &ci, int32_t x) {
    const int32_t top = (int32_t)(ci.rk_words << 6) - 1;
    const int32_t xc = x < 0 ? 0 : (x > top ? top : x);
    // bits 0..(xc & 63) of the word, moved to the top: their count is the number of breakpoints <= xc inside the word and
    // the top bit says whether xc is one
    const uint32_t sh = 63u - ((uint32_t)xc & 63u);
    const uint64_t t = (((uint64_t)w.hi << 32) | (uint64_t)w.lo) << sh;
    EiFind f;
#if defined(__HIP_DEVICE_COMPILE__)
    f.j = w.rank + (uint32_t)__popcll(t) - 1u;
#else
    f.j = w.rank + (uint32_t)__builtin_popcountll(t) - 1u;
#endif
    f.at = (t >> 63) != 0 && xc == x;
    return f;
}

// class flags (CF_*) of a record from the OR of the interval masks its blocks touch
RSQC_HD uint32_t ei_class_flags(uint32_t mask, int rstrand) {
    const uint32_t sel = rstrand == RSQC_STRAND_UNKNOWN ? 7u : 1u << rstrand;                  // src/Expression.cpp:331
    const uint32_t gp = mask & sel, gr = (mask >> 3) & sel, ep = (mask >> EIM_EXON_SHIFT) & sel, er = (mask >> (EIM_EXON_SHIFT + 3)) & sel;
    return (gp ? CF_INTRAGENIC : 0u) | (((gp | ep) & (1u << RSQC_STRAND_FORWARD)) ? CF_PLUS : 0u) |
           (((gp | ep) & (1u << RSQC_STRAND_REVERSE)) ? CF_MINUS : 0u) | ((gr | er) ? CF_RIBOSOMAL : 0u) | (ep ? CF_EXONIC : 0u);
}

// One block against the index.  Returns the OR of the interval masks of [bs, be] (incl. EIM_DEEP of the two intervals
// the containment test reads) and the block's (at most two) containing exons as commit operands.
struct EiBlock { uint32_t mask; bool cA, cB; uint32_t eidA, gfA, cidxA, eidB, gfB, cidxB; };
// A block query in three phases, so that the loads of SEVERAL blocks of a record travel together (a record of two blocks costs
// two round trips, not four: the latency of dependent loads, not their number, is what a wave waits for):
//   ei_probe    the two rank words (independent loads)
//   ei_fetch    indices from the rank words; the interval entries (one round of independent loads)
//   ei_resolve  masks and containing exons; only a block across more than three intervals goes back to memory
struct EiProbe { EiRank ws, we; int32_t bs, be; uint32_t len; bool have; uint32_t pre; };
struct EiFetch { EiEntry S; uint32_t m1, e1A, e1B, m_je, m_js1; uint32_t js, je, j1; };
// `pre` != 0: the caller already knows the answer from the coarse table (DevAnnotation::ei_coarse) -- the whole block lies
// in interval pre - 1 and no rank word is read (the lane loads word 0 like a lane without a block: one shared line)
RSQC_HD void ei_probe(const DevAnnotation &a, const ContigInfo &ci, int32_t bs, uint32_t len, bool on, EiProbe &q, uint32_t pre = 0u) {
    q.bs = bs; q.len = len; q.be = bs + (int32_t)len;
    q.have = on && ci.rk_words != 0 && q.be >= 0;
    q.pre = pre;
    const bool look = q.have && pre == 0u;
    q.ws = ei_find_word(a, ci, q.bs, look); q.we = ei_find_word(a, ci, q.be, look);
}
RSQC_HD void ei_fetch(const DevAnnotation &a, const ContigInfo &ci, const EiProbe &q, EiFetch &f) {
    const EiFind fs = ei_find_in(q.ws, ci, q.bs), fe = ei_find_in(q.we, ci, q.be);
    const bool known = q.pre != 0u;
    f.js = q.have ? (known ? q.pre - 1u : fs.j) : 0u; f.je = q.have ? (known ? q.pre - 1u : fe.j) : 0u;
    f.j1 = (!known && fe.at && q.len > 0 && f.je > f.js) ? f.je - 1 : f.je;        // find(max(bs, be - 1))
    // (Measured and removed, profiles/r5_k1_variants.txt call r5a: a wave-uniform shortcut HERE -- all lanes' blocks in one interval: one
    //  entry load instead of the five gathers -- made the kernel 9 % SLOWER: a branch between the rounds of a block query puts a join in
    //  front of the gathers of the record's other block, and the wait the compiler places at a join is for everything in flight.)
    f.S = ld32(a.ei, f.js);
    f.m1 = ld32(a.ei, f.j1).mask; f.e1A = ld32(a.ei, f.j1).eidA; f.e1B = ld32(a.ei, f.j1).eidB;
    // intervals js .. je: js, j1 (= je - 1 or je), je and js + 1 are read in this one round
    f.m_je = ld32(a.ei, f.je).mask; f.m_js1 = ld32(a.ei, f.js < f.je ? f.js + 1 : f.je).mask;
}
RSQC_HD void ei_resolve(const DevAnnotation &a, const EiProbe &q, const EiFetch &f, int rstrand, EiBlock &o) {
    uint32_t mask = f.S.mask | f.m1 | f.m_je | f.m_js1;
    if (RSQC_ANY_LANE(q.have && f.je > f.js + 2)) {
#if defined(__clang__)                                     /* (rare path: interleaved by two it took six more registers -- spills at five waves per SIMD) */
#pragma clang loop unroll(disable) vectorize(disable) interleave(disable)
#endif
        for (uint32_t j = f.js + 2; q.have && j < f.je; ++j) mask |= ld32(a.ei, j).mask;
    }
    // EIM_DEEP matters only where the exon lists are read
    mask = (mask & ~EIM_DEEP) | ((f.S.mask | f.m1) & EIM_DEEP);
    const bool same = f.j1 == f.js;
    const bool sA = rstrand == RSQC_STRAND_UNKNOWN || rstrand == (int)((f.S.gfA >> ROW_FLAG_SHIFT) & RSQC_FF_STRAND_MASK);
    const bool sB = rstrand == RSQC_STRAND_UNKNOWN || rstrand == (int)((f.S.gfB >> ROW_FLAG_SHIFT) & RSQC_FF_STRAND_MASK);
    o.mask = q.have ? mask : 0u;
    o.cA = q.have && f.S.eidA != EI_NONE && sA && (same || f.S.eidA == f.e1A || f.S.eidA == f.e1B);
    o.cB = q.have && f.S.eidB != EI_NONE && sB && (same || f.S.eidB == f.e1A || f.S.eidB == f.e1B);
    o.eidA = f.S.eidA; o.gfA = f.S.gfA; o.cidxA = f.S.cdA + (uint32_t)q.bs;
    o.eidB = f.S.eidB; o.gfB = f.S.gfB; o.cidxB = f.S.cdB + (uint32_t)q.bs;
}

// exonAlignmentMetrics (src/Expression.cpp:308-458) for a record of NB <= FAST_BLOCKS blocks, all lanes of a wave
// running the same NB (the per-record kernel sorts records by block count first).  Same contract as exon_metrics_fast:
// slot 2b + j commits block b to exon id `eid[2b + j]` at coverage index `cidx[2b + j]` when bit 2b + j of cmask is set;
// `overflow` hands the record to the general code (more than two exons on an interval, more than FAST_SET genes).
struct EiOut {
    int n_hit; uint32_t hit[FAST_SET];
    uint32_t cmask;
    uint32_t eid[NSLOT], cidx[NSLOT];
};
// `nbv` = the record's own block count (1 .. NB) where a wave mixes records of different counts (the kernel for long CIGARs).
template <int NB, class Sink>
RSQC_HD void exon_metrics_ei(const DevAnnotation &a, const DevParams &p, const ContigInfo &ci, uint32_t fl, const int32_t (&bs)[NB],
                             const uint32_t (&len)[NB], bool hq, EiOut &out, bool &overflow, Sink &cnt, bool lane_on = true,
                             uint32_t nbv = (uint32_t)NB, uint32_t pre0 = 0u /* coarse-table answer for block 0, see ei_probe */) {
    static_assert(NB >= 1 && NB <= FAST_BLOCKS, "blocks per record on the fast path");
    const int rstrand = read_strand_of(p, fl);
    uint32_t mask = 0, con = 0, ma = 0, mb = 0;
    uint32_t la = 0, lb = 0; bool va = false, vb = false, ga = false, gb = false;
    uint32_t aligned = 0;
#pragma unroll
    for (int k = 0; k < NSLOT; ++k) { out.eid[k] = 0; out.cidx[k] = 0; }
    EiBlock qb[NB];
#ifndef RSQC_EI_GROUP3
#define RSQC_EI_GROUP3 2                                   /* blocks in flight at a time in the three-block instance (A/B: 1) */
#endif
    constexpr int G = NB == 3 ? RSQC_EI_GROUP3 : 2;
#pragma unroll
    for (int b0 = 0; b0 < NB; b0 += G) {                  // two blocks' loads in flight at a time
        EiProbe pr[G]; EiFetch fe[G];
#pragma unroll
        for (int j = 0; j < G; ++j) if (b0 + j < NB) ei_probe(a, ci, bs[b0 + j], len[b0 + j], lane_on && (uint32_t)(b0 + j) < nbv, pr[j], (b0 + j == 0) ? pre0 : 0u);
        K1E_SMARK(2);                                      // [2] rank words issued
#pragma unroll
        for (int j = 0; j < G; ++j) if (b0 + j < NB) ei_fetch(a, ci, pr[j], fe[j]);
        K1E_SMARK(3);                                      // [3] rank words landed, entries issued
#pragma unroll
        for (int j = 0; j < G; ++j) if (b0 + j < NB) ei_resolve(a, pr[j], fe[j], rstrand, qb[b0 + j]);
        K1E_SMARK(4);                                      // [4] entries landed, blocks resolved
    }
#pragma unroll
    for (int b = 0; b < NB; ++b) {
        const bool live = (uint32_t)b < nbv;
        const EiBlock &q = qb[b];
        mask |= q.mask;
        aligned += live ? len[b] : 0u;
        const uint32_t g0 = q.gfA & ROW_GENE_MASK, g1 = q.gfB & ROW_GENE_MASK;
        // containing exons in slot order: A first; when only B contains the block it takes slot 2b
        const bool c0 = q.cA || q.cB, c1 = q.cA && q.cB;
        const uint32_t gX = q.cA ? g0 : g1, gfX = q.cA ? q.gfA : q.gfB;
        out.eid[2 * b] = q.cA ? q.eidA : q.eidB; out.cidx[2 * b] = q.cA ? q.cidxA : q.cidxB;
        out.eid[2 * b + 1] = q.eidB; out.cidx[2 * b + 1] = q.cidxB;
        con |= (c0 ? 1u : 0u) << (2 * b) | (c1 ? 1u : 0u) << (2 * b + 1);
        if (b == 0) {                                                      // genes.front(), :363-367
            la = gX; va = c0; ga = c0 && ((gfX >> ROW_FLAG_SHIFT) & ROWF_GLOBIN) != 0;
            lb = g1; vb = c1 && g1 != gX; gb = vb && ((q.gfB >> ROW_FLAG_SHIFT) & ROWF_GLOBIN) != 0;
            ma = (c0 ? 1u : 0u) | ((c1 && g1 == gX) ? 2u : 0u); mb = vb ? 2u : 0u;
        } else {                                                           // set_intersection, :368-374
            const bool a_in = (c0 && la == gX) || (c1 && la == g1), b_in = (c0 && lb == gX) || (c1 && lb == g1);
            va = live ? (va && a_in) : va; vb = live ? (vb && b_in) : vb;
            ma |= ((c0 && la == gX) ? 1u : 0u) << (2 * b) | ((c1 && la == g1) ? 1u : 0u) << (2 * b + 1);
            mb |= ((c0 && lb == gX) ? 1u : 0u) << (2 * b) | ((c1 && lb == g1) ? 1u : 0u) << (2 * b + 1);
        }
    }
    const bool over = (mask & EIM_DEEP) != 0 || !lane_on;
    overflow = over;
    ga = ga && va; gb = gb && vb;
    const int nlast = (va ? 1 : 0) + (vb ? 1 : 0);
    // counters in the sink's boolean domain (lane masks in the per-record kernel: rsqc_read.h, WaveSink)
    using SB = typename Sink::B;
    const SB keep = !Sink::prim(over), hqb = Sink::prim(hq), any_gene = Sink::prim(va || vb);
    const SB nonglobin = keep && !Sink::prim(ga || gb);                                        // :363,395-404
    RSQC_COUNT(cnt, RSQC_C_NON_GLOBIN_READS, nonglobin);
    RSQC_COUNT(cnt, RSQC_C_NON_GLOBIN_DUPLICATE_READS, nonglobin && Sink::prim((fl & RSQC_FDUP) != 0));
    out.n_hit = 0; out.cmask = 0;
    if (Sink::lane(hqb && any_gene && keep)) {                                                 // :377-392
        out.cmask = con & ((va ? ma : 0u) | (vb ? mb : 0u));
        if (aligned > 0) { out.hit[0] = va ? la : lb; out.hit[1] = lb; out.n_hit = nlast; }
    }
    class_counts_cf(cnt, p, fl, ei_class_flags(mask, rstrand), any_gene, hqb, keep);
    K1E_SMARK(5);                                          // [5] gene sets, class flags, counters
}

// ---- stage 2: exonAlignmentMetrics, src/Expression.cpp:308-458 ---------------------------
// Nothing is scattered from inside: the function returns what to count.  Up to NSTAGE
// (block, exon) commits are staged in registers while the gene set common to all blocks is
// being built; only records with more contained hits than that re-walk their CIGAR and go
// through `acc` directly (exon_add / cov_range).
// `Acc` (used only on the re-walk path) provides
//   void exon_add(uint32_t row, double frac);                                      Metrics.cpp:59-66
//   void cov_range(uint32_t cidx, uint32_t len);       (coverage index of the first base)  Metrics.cpp:96-124
// Sets `overflow` (and returns nothing to count) when a block lies inside exons of more than K genes.
template <int K, class Acc>
RSQC_HD void exon_metrics(const DevAnnotation &a, const DevParams &p, const Record &r, bool hq,
                          uint32_t aligned, Acc &acc, FeatureOut<K, SLOW_STAGE> &out, bool &overflow) {
    constexpr int NSTAGE = SLOW_STAGE;      // shadows the fast path's capacity inside this function
    const uint32_t fl = r.flag;
    uint64_t bits = 0;
    overflow = false;
    out.bits = 0; out.n_hit = 0; out.n_commit = 0; out.cmask = 0;
    const int rstrand = read_strand_of(p, fl);
    const ContigInfo ci = a.contig[r.tid];
    ClassFlags f = {false, false, false, false, false};
    uint32_t last[K]; int nlast = 0;
    uint32_t cur[K];
    Commit st[NSTAGE]; uint32_t st_gene[NSTAGE]; int nst = 0; bool st_over = false;
    static_assert(K <= 32, "the globin bit-set below holds 32 slots");
    bool first = true, over = false;
    uint32_t last_globin = 0;      // bit k: last[k] is a globin gene
    uint32_t nblocks = 0;
    // pass 1: flags + the gene set common to all blocks (:325-374)
    {
        int32_t start = r.pos + 1;                                                         // Expression.cpp:31
        for (uint32_t i = 0; i < r.n_cigar; ++i) {
            const uint32_t c = r.cigar[i], op = c & 0xf, len = c >> 4;
            if (cigar_is_block(op)) {
                ++nblocks;
                const int32_t bs = start, be = start + (int32_t)len;
                int ncur = 0;
                query_block(a, ci, bs, be, rstrand, &f, [&](uint32_t row_i, const ExonRow &row, bool contained) {
                    if (!contained) return;
                    const uint32_t g = row.gf & ROW_GENE_MASK;
                    if (nst < NSTAGE) {
                        const uint32_t cidx = row.cov + (uint32_t)(bs - row.start);
#pragma unroll
                        for (int k = 0; k < NSTAGE; ++k) if (k == nst) { st[k].row = row_i; st[k].cidx = cidx; st[k].len = len; st_gene[k] = g; }
                        ++nst;
                    } else st_over = true;
                    if (first) {                          // genes.front()
                        if (!set_contains<K>(last, nlast, g)) {
                            if (nlast < K) {
                                if ((row.gf >> ROW_FLAG_SHIFT) & ROWF_GLOBIN) last_globin |= 1u << nlast;
                                set_put<K>(last, nlast, g); ++nlast;
                            } else over = true;
                        }
                    } else if (!set_contains<K>(cur, ncur, g)) {
                        if (ncur < K) { set_put<K>(cur, ncur, g); ++ncur; } else over = true;
                    }
                });
                if (!first) {                             // set_intersection, :368-374
                    int w = 0; uint32_t wg = 0;
#pragma unroll
                    for (int k = 0; k < K; ++k) {
                        if (k < nlast && set_contains<K>(cur, ncur, last[k])) {
                            if ((last_globin >> k) & 1u) wg |= 1u << w;
                            set_put<K>(last, w, last[k]); ++w;
                        }
                    }
                    nlast = w; last_globin = wg;
                }
                first = false;
            }
            if (cigar_is_ref(op)) start += (int32_t)len;
        }
    }
    if (over) { overflow = true; return; }
    const bool do_exon = nlast > 0;                                                        // :393
    if (nblocks >= 1) {                                                                    // :363,395-404
        const bool globin = last_globin != 0;       // geneNames[gene] in the globin blacklist, :396-398
        if (!globin) {
            bits |= RSQC_BIT(RSQC_C_NON_GLOBIN_READS);
            if (fl & RSQC_FDUP) bits |= RSQC_BIT(RSQC_C_NON_GLOBIN_DUPLICATE_READS);
        }
    }
    // commit (only HQ records counted to at least one gene, :377-392)
    if (hq && nlast > 0) {
        if (!st_over) {
#pragma unroll
            for (int k = 0; k < NSTAGE; ++k) {
                if (k < nst && set_contains<K>(last, nlast, st_gene[k])) {
#pragma unroll
                    for (int j = 0; j < NSTAGE; ++j) if (j == out.n_commit) out.commit[j] = st[k];
                    out.cmask |= 1u << out.n_commit;
                    ++out.n_commit;
                }
            }
        } else {
            // rare: more than NSTAGE contained hits -> pass 2 re-walks the CIGAR
            int32_t start = r.pos + 1;
            for (uint32_t i = 0; i < r.n_cigar; ++i) {
                const uint32_t c = r.cigar[i], op = c & 0xf, len = c >> 4;
                if (cigar_is_block(op)) {
                    const int32_t bs = start, be = start + (int32_t)len;
                    query_block(a, ci, bs, be, rstrand, (ClassFlags *)nullptr, [&](uint32_t row_i, const ExonRow &row, bool contained) {
                        if (!contained) return;
                        const uint32_t g = row.gf & ROW_GENE_MASK;
                        if (!set_contains<K>(last, nlast, g)) return;
                        if (len > 0) acc.exon_add(row_i, (double)len / (double)aligned);   // :345
                        acc.cov_range(row.cov + (uint32_t)(bs - row.start), len);
                    });
                }
                if (cigar_is_ref(op)) start += (int32_t)len;
            }
        }
        if (aligned > 0) {                                 // Collector::queryGene, :380
#pragma unroll
            for (int k = 0; k < K; ++k) out.hit[k] = last[k];
            out.n_hit = nlast;
        }
    }
    bits |= class_bits(p, fl, f, do_exon, hq);
    out.bits = bits;
}


// ---- --legacy: legacyExonAlignmentMetrics, src/Expression.cpp:129-304 --------------------------------------
// The reference intersects the read's whole span [pos + 1, endpos] with its start-sorted feature list once
// (:145-148) and then runs, for every GENE row of that result list, a loop over the read's blocks in which the
// exon rows of the same result list are scanned in list order up to the first exon that contains the block
// (:154-241).  Here the result list is never materialised: gene rows come from LegacyTables (a start-sorted
// table with a running max of end), and "the first containing exon of gene g in list order, and whether a
// partially overlapping exon of g precedes it" is one downward walk of the exon rows (legacy_find).  The CIGAR is
// re-walked per gene instead of being staged, so a lane keeps no per-block state.
struct LegacyFind { uint32_t cmin; bool partial_before, any; };
RSQC_HD LegacyFind legacy_find(const DevAnnotation &a, const ContigInfo &ci, uint32_t g, int32_t bs, int32_t be, int32_t se) {
    LegacyFind f = {0xFFFFFFFFu, false, false};
    // a row of the result list starts at or before the span's end; one that intersects the block starts at or before
    // the block's (exclusive, but compared inclusively: src/GTF.cpp:171-179) end
    const int32_t hi = be < se ? be : se;
    if (ci.n_bins == 0 || hi < 0) return f;
    uint32_t b = (uint32_t)hi >> a.bin_shift;
    if (b >= ci.n_bins) b = ci.n_bins - 1;
    uint32_t pmin = 0xFFFFFFFFu;
    for (uint32_t i = a.ex_binhi[ci.bin_base + b]; i > ci.ex_lo;) {
        --i;
        const ExonRow row = a.ex[i];
        if (a.ex_pmax[i] < bs) break;
        if (row.start > hi || row.end < bs) continue;
        if ((row.gf & ROW_GENE_MASK) != g) continue;                                   // ex->gene_id == result->gene_id, :180
        f.any = true;
        const int32_t lo_e = row.end < be - 1 ? row.end : be - 1, hi_s = row.start > bs ? row.start : bs;
        const int32_t pi = 1 + lo_e - hi_s;                                            // partialIntersect, src/GTF.cpp:181-186
        if (pi == be - bs) f.cmin = i;               // walking down: the last one seen is the first in list order
        else if (pi > 0) pmin = i;
    }
    f.partial_before = pmin < f.cmin;                // :191-194 is reached only before the scan stops at the first container
    return f;
}

template <int K> struct LegacyOut { uint64_t bits; int n_hit; uint32_t hit[K]; };

// `Acc` provides exon_add(row, double), cov_range(cidx, len) and gene_hit(gene, notdup, qhash) (genes beyond the K
// returned in `out`).
template <int K, class Acc>
RSQC_HD void legacy_metrics(const DevAnnotation &a, const DevParams &p, const Record &r, bool hq, Acc &acc, LegacyOut<K> &out) {
    const LegacyTables &T = *a.legacy;
    const uint32_t fl = r.flag;
    out.bits = 0; out.n_hit = 0;
    const int rstrand = read_strand_of(p, fl);
    const ContigInfo ci = a.contig[r.tid];
    // split detection (:135-141, LEGACY_SPLIT_DISTANCE = 100) and the span
    bool split = false; uint32_t nblocks = 0, ref_len = 0;
    {
        int64_t last_end = -1; int32_t start = r.pos + 1;
        for (uint32_t i = 0; i < r.n_cigar; ++i) {
            const uint32_t c = r.cigar[i], op = c & 0xf, len = c >> 4;
            if (cigar_is_block(op)) {
                if (last_end > 0 && !split) split = ((int64_t)start - last_end) > 99;
                last_end = (int64_t)start + (int64_t)len; ++nblocks;
            }
            if (cigar_is_ref(op)) { start += (int32_t)len; ref_len += len; }
        }
    }
    const int32_t ss = r.pos + 1;                                                          // :145
    const int32_t se = r.pos + (int32_t)((r.n_cigar == 0 || ref_len == 0) ? 1u : ref_len); // :146 PositionEnd(), 1-based closed
    bool intragenic = false, plus = false, minus = false, ribosomal = false, do_exon = false, exonic = false,
         junction = false, not_exonic = false;
    uint32_t last_ord = 0; bool have_last = false, last_not_split = false;   // the final value of legacyNotSplit (:159) is
                                                                             // the one of the LAST row of the result list
    const uint32_t glo = T.gr_range[r.tid], ghi = T.gr_range[r.tid + 1];
    uint32_t lo = glo;                                                       // first gene row with start > se: from the bin of se, then down
    if (ci.n_bins != 0 && se >= 0 && ghi > glo) {
        uint32_t b = (uint32_t)se >> a.bin_shift;
        if (b >= ci.n_bins) b = ci.n_bins - 1;
        lo = T.gr_binhi[ci.bin_base + b];                                    // (every row from here on starts behind the bin, i.e. behind se -- or the contig's rows end)
        while (lo > glo && T.gr[lo - 1].start > se) --lo;
    }
    for (uint32_t gi = lo; gi > glo;) {
        --gi;
        if (T.gr_pmax[gi] < ss) break;
        const GeneRow G = T.gr[gi];
        if (G.end < ss) continue;
        const uint32_t g = G.gf & ROW_GENE_MASK, gfl = G.gf >> ROW_FLAG_SHIFT;
        const int gs = (int)(gfl & RSQC_FF_STRAND_MASK);
        if (gs == RSQC_STRAND_FORWARD) plus = true; else if (gs == RSQC_STRAND_REVERSE) minus = true;   // :163-164
        bool not_split = false, found = false, t_intron = false, t_exon = false;
        uint32_t last_row = 0;
        if (rstrand == RSQC_STRAND_UNKNOWN || rstrand == gs) {                             // :167
            int32_t start = r.pos + 1;
            for (uint32_t i = 0; i < r.n_cigar; ++i) {
                const uint32_t c = r.cigar[i], op = c & 0xf, len = c >> 4;
                if (cigar_is_block(op)) {
                    const int32_t bs = start, be = start + (int32_t)len;
                    intragenic = true;                                                     // :168
                    if (bs > G.end) not_exonic = true;                                     // :170
                    const LegacyFind f = legacy_find(a, ci, g, bs, be, se);
                    if (f.any && (gfl & RSQC_FF_RIBOSOMAL)) ribosomal = true;              // :182
                    found = f.cmin != 0xFFFFFFFFu; last_row = f.cmin;                      // :173,186-189
                    if (found) t_exon = true;
                    if (f.partial_before) t_intron = true;
                    if (split && !not_split && !found) not_split = true;                   // :198-205
                }
                if (cigar_is_ref(op)) start += (int32_t)len;
            }
        }
        if (!have_last || G.ord > last_ord) { have_last = true; last_ord = G.ord; last_not_split = not_split; }
        if (found) {                                                                       // :211
            if (hq) {
                const bool dose = split && !not_split;
                if (!dose) acc.exon_add(last_row, 1.0);                                    // :223-227
                uint32_t cur = 0xFFFFFFFFu; float dsum = 0.0f;
                int32_t start = r.pos + 1;
                for (uint32_t i = 0; i < r.n_cigar; ++i) {
                    const uint32_t c = r.cigar[i], op = c & 0xf, len = c >> 4;
                    if (cigar_is_block(op)) {
                        const int32_t bs = start, be = start + (int32_t)len;
                        const LegacyFind f = legacy_find(a, ci, g, bs, be, se);
                        if (f.cmin != 0xFFFFFFFFu) {
                            const ExonRow row = a.ex[f.cmin];
                            acc.cov_range(row.cov + (uint32_t)(bs - row.start), len);      // baseCoverage.add + commit, :190,236
                            if (dose) {                                                    // legacySplitDosage: float sums per exon, :201,217-220
                                if (f.cmin != cur) { if (cur != 0xFFFFFFFFu) acc.exon_add(cur, (double)dsum); cur = f.cmin; dsum = 0.0f; }
                                dsum += (float)len / (float)r.l_qseq;
                            }
                        }
                    }
                    if (cigar_is_ref(op)) start += (int32_t)len;
                }
                if (cur != 0xFFFFFFFFu) acc.exon_add(cur, (double)dsum);
                if (out.n_hit < K) { set_put<K>(out.hit, out.n_hit, g); ++out.n_hit; }    // :229-235
                else acc.gene_hit(g, !(fl & RSQC_FDUP), r.qhash);
            }
            do_exon = true;                                                                // :238
        }
        if (t_intron && t_exon) junction = true;                                           // :240
        if (t_exon) exonic = true;                                                         // :241
    }
    // the last row of the result list may be an exon row: legacyNotSplit was reset for it (:159) and nothing set it
    if (ci.n_bins != 0 && se >= 0) {
        uint32_t b = (uint32_t)se >> a.bin_shift;
        if (b >= ci.n_bins) b = ci.n_bins - 1;
        for (uint32_t i = a.ex_binhi[ci.bin_base + b]; i > ci.ex_lo;) {
            --i;
            if (a.ex_pmax[i] < ss) break;
            const ExonRow row = a.ex[i];
            if (row.start > se || row.end < ss) continue;
            if (!have_last || T.ex_ord[i] > last_ord) last_not_split = false;
            break;                                   // rows further down rank lower
        }
    }
    const bool outside = not_exonic || junction || !exonic;                                // :248
    const bool as_exonic = !outside && (do_exon || intragenic);                            // :265,276
    const bool intronic = outside && intragenic, intergenic = outside && !intragenic;
    uint64_t bits = 0;
    bits |= intronic ? RSQC_BIT(RSQC_C_INTRONIC_READS) | RSQC_BIT(RSQC_C_INTRAGENIC_READS) : 0ull;
    bits |= (intronic && hq) ? RSQC_BIT(RSQC_C_HQ_INTRONIC_READS) | RSQC_BIT(RSQC_C_HQ_INTRAGENIC_READS) : 0ull;
    bits |= intergenic ? RSQC_BIT(RSQC_C_INTERGENIC_READS) : 0ull;
    bits |= (intergenic && hq) ? RSQC_BIT(RSQC_C_HQ_INTERGENIC_READS) : 0ull;
    bits |= as_exonic ? RSQC_BIT(RSQC_C_EXONIC_READS) | RSQC_BIT(RSQC_C_INTRAGENIC_READS) : 0ull;
    bits |= (as_exonic && hq) ? RSQC_BIT(RSQC_C_HQ_EXONIC_READS) | RSQC_BIT(RSQC_C_HQ_INTRAGENIC_READS) : 0ull;
    bits |= (!outside && do_exon && split && !last_not_split) ? RSQC_BIT(RSQC_C_SPLIT_READS) : 0ull;   // :274
    bits |= ribosomal ? RSQC_BIT(RSQC_C_RRNA_READS) : 0ull;                                // :288
    const bool one_strand = (minus != plus) && (p.unpaired || (fl & RSQC_FPAIRED));        // :290-302
    const bool sense = (fl & RSQC_FREVERSE) ? minus : plus;
    const bool end1 = p.unpaired || (fl & RSQC_FREAD1);
    const uint64_t sbit = end1 ? (sense ? RSQC_BIT(RSQC_C_END1_SENSE) : RSQC_BIT(RSQC_C_END1_ANTISENSE))
                               : (sense ? RSQC_BIT(RSQC_C_END2_SENSE) : RSQC_BIT(RSQC_C_END2_ANTISENSE));
    bits |= one_strand ? sbit : 0ull;
    (void)nblocks;
    out.bits = bits;
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
            uint32_t ub;
            if (a.bed_binhi) {
                const uint32_t nb = a.bed_bin_base[r.tid + 1] - a.bed_bin_base[r.tid];
                uint32_t bin = be < 0 ? 0u : (uint32_t)be >> RSQC_BED_BIN_SHIFT;
                if (bin >= nb) bin = nb - 1u;
                ub = a.bed_binhi[a.bed_bin_base[r.tid] + bin];
                while (ub > lo && a.bed_start[ub - 1] > be) --ub;
            } else ub = upper_bound_rows(a.bed_start, lo, hi, be);
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
