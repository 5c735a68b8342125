// rsqc_device.h -- device-side views shared by the kernels and the C-ABI host code.
#pragma once

#if !defined(RSQC_WAVE_EMU)      /* tests/hostemu/wavemu.h stands in for the HIP runtime in the host build of K1 */
#include <hip/hip_runtime.h>
#endif
#include <stdint.h>

#include "rsqc_read.h"

#define RSQC_K1_THREADS 256
#define RSQC_MAX_BIAS_WINDOW 1024
#define RSQC_K3_THREADS 1024
// coding-length classes of the end-of-file coverage stage: one wave / 256 threads with the vector in LDS, 1024 threads in memory
#ifndef RSQC_K3_SMALL_MAX
#define RSQC_K3_SMALL_MAX 4096
#endif
#define RSQC_K3_MEDIUM_MAX 12288
#define RSQC_K3_LARGE_LDS16 73000      /* bases a 1024-thread workgroup keeps in LDS as 16-bit depths (146 KB of the CU's 160 KB) */
#define RSQC_K3_LARGE2_LDS16 32768     /* ... the shorter genes of that class: 64 KB */
#define RSQC_K3_MAX_EXONS 1024

namespace rsqc {

// one uploaded batch (all pointers are device pointers)
struct DevBatch {
    uint64_t n;
    uint64_t record_base;        // file index of record 0
    const rsqc_rec_core *core;   // 16-byte half-records: one dwordx4 load per lane each
    const rsqc_rec_aux *aux;
    const uint32_t *cigar;
    const uint32_t *qhash2;      // second name hash per record (rsqc_batch.qhash2), null = 64-bit identity
    uint32_t n_seg;
    const int32_t *seg_tid;
    const uint64_t *seg_start;
    const uint64_t *seg_file_index;   // file index of every segment's first record (rsqc_batch.seg_file_index), null = record_base + index
    uint32_t n_wide;
    const uint64_t *wide_index;
    const int32_t *wide_nm, *wide_l_qseq;
    const uint32_t *wide_n_cigar;
};

// file index of record i (of segment `seg`) of a batch: per segment for a batch of several file ranges
RSQC_HD uint64_t batch_file_index(const DevBatch &b, uint32_t seg, uint64_t i) {
    return b.seg_file_index ? b.seg_file_index[seg] + (i - b.seg_start[seg]) : b.record_base + i;
}

// fragment-size candidates emitted by K1 (only with a BED): one entry per record that passes
// src/RNASeQC.cpp:372 and the block tests of src/Expression.cpp:490-507
struct FragCandidates {
    uint64_t *file_index;        // record index in the whole file
    uint64_t *qhash;
    int32_t *name;               // BED row
    int32_t *endpos;             // PositionEnd()
    uint32_t *flag_size;         // bit 31: !MateReverse && Reverse && pos != mpos ; low 31 bits |isize|
    uint32_t *count; uint32_t cap;
    // classify_ei_kernel: a workgroup writes into its OWN region -- slot = first record of its range + an LDS counter (a record
    // yields at most one candidate), no memory atomic -- and leaves its count here; frag_compact_kernel then packs the regions
    // into the dense list.  (One shared counter was 70 M memory atomics on one address: 41 of the 44 ms of a --bed pass.)
    uint32_t *chunk_count;       // [workgroups of the per-record kernel]; null: `count` is a plain shared counter
    uint32_t *h2;                // second name hash of the record (rsqc_batch.qhash2; zeros for a batch without): a name is (qhash, h2), as in K4
};

// --fasta: one G/C bit per base (gc(), src/Fasta.cpp:67-74, counts G g C c only); every contig starts on a word
struct DevReference {
    const unsigned long long *bits;
    const unsigned long long *word_off;   // [n_contigs] first word of the contig; ~0 = the FASTA index lacks the contig
    const unsigned long long *length;     // [n_contigs] bases
};
// fragment GC candidates (src/Expression.cpp:459): records that reach the fragments map of the GC branch
struct GcCandidates {
    uint64_t *file_index;        // record index in the whole file
    uint64_t *qhash;
    uint32_t *row;               // the single aligned exon (row index)
    int32_t *endpos;             // PositionEnd()
    uint32_t *flag_lq;           // bit 31: pos != mpos ; low 31 bits alignment.Length()
    int32_t *tid;
    uint32_t *count; uint32_t cap;
    uint32_t *h2;                // second name hash of the record (rsqc_batch.qhash2; zeros for a batch without)
};
#if defined(__HIPCC__)
// bases of [s, e) (0-based, inside the contig) that are G/C
__device__ __forceinline__ uint32_t gc_count(const DevReference &R, int contig, int64_t s, int64_t e) {
    if (e <= s) return 0u;
    const unsigned long long *w = R.bits + R.word_off[contig];
    const int64_t w0 = s >> 6, w1 = (e - 1) >> 6;
    uint32_t n = 0;
    for (int64_t i = w0; i <= w1; ++i) {
        unsigned long long x = w[i];
        if (i == w0) x &= ~0ull << (s & 63);
        if (i == w1) x &= ~0ull >> (63 - ((e - 1) & 63));
        n += (uint32_t)__popcll(x);
    }
    return n;
}
// gc(): 1.0/size added once per G/C base -- k sequential additions, not k/size
__device__ __forceinline__ double gc_value(uint32_t k, uint64_t size) {
    const double inc = 1.0 / (double)size;
    double c = 0.0;
    for (uint32_t i = 0; i < k; ++i) c += inc;
    return c;
}
#endif

// one (gene, read name) pair as the per-record kernels emit it.  Round 6: an array of these 16-byte structures -- ONE
// global_store_dwordx4 per pair in the per-record kernel and one global_load_dwordx4 in frag_local_kernel -- instead of three columns
// (three stores in front of every later vmcnt wait of the emitting wave: the pairs were 11.5 % of K1, profiles/r5_k1_variants.txt)
struct alignas(16) PairRec { uint32_t gene, h2; uint64_t hash; };
// accumulators (device pointers)
struct DevAccum {
    unsigned long long *gene_reads, *gene_unique, *gene_frag, *counters;   // one allocation, in this order
    double *exon_acc;            // by exon ROW
    uint32_t *cov_diff;          // per-base difference array / coverage
    // (gene, qname-hash) pairs of one batch: K1 block k owns [k*pair_chunk_cap, +pair_chunk_count[k]);
    // the slow path appends to [pair_slow_base, +*pair_slow_count)
    PairRec *pairs;              // {gene, second name hash (rsqc_batch.qhash2, 0 without it), name hash}: one 16-byte store per pair
    uint32_t pair_chunk_cap; uint32_t *pair_chunk_count;
    uint32_t pair_slow_base, pair_slow_cap; uint32_t *pair_slow_count;
    uint32_t *ovf_count; uint64_t *ovf_index; uint32_t ovf_cap;
    // records classify_ei_kernel leaves to classify_long_kernel (more than eight operations / three blocks): a K1 workgroup lists them in
    // ITS region of defer_index, from the first record of its range on -- no memory atomic (entry: index | hq << 31) -- and moves them into
    // the dense defer_list when it retires, in whole calls of 64 entries ([0, *defer_total): one memory atomic per workgroup)
    uint32_t *defer_index; uint32_t *defer_list; uint32_t *defer_total;
    uint32_t *tile_span;         // max span per 64-record wave tile (Read-Length fallback scan)
    FragCandidates frag;
    uint32_t *rl_stats;          // [3] batch-level max span, min l_qseq, max l_qseq over eligible records
    uint32_t *rl_seg;            // [3 * n_seg] the same per contig segment (batches with seg_file_index: a Read-Length function per segment), else null
    int32_t *read_length;
    int *error;
};

struct GeneCovArgs {
    const uint32_t *ge_off, *ge_row;        // exonsForGene CSR (gene id -> exon rows)
    const ExonRow *ex;
    const uint32_t *ex_cov;                 // coverage offset of an exon row
    const uint32_t *ex_id;                  // exon row -> exon id
    const uint32_t *gene_cov_off;           // [n_listed]
    const uint32_t *gene_coding;            // [n_listed]
    const uint8_t *gene_flags;              // [n_listed] flags of the gene row
    const uint8_t *gene_owned;              // [n_listed] gene lies on a contig of this shard
    const uint32_t *gene_order;             // [n_listed] gene ids, longest coding length first
    const unsigned long long *gene_reads;   // touched test
    uint32_t *cov;
    int32_t n_listed;
    uint32_t mask; int32_t bias_offset, bias_window; uint64_t bias_gene_length;
    double *g_mean, *g_std, *g_cv; uint8_t *g_valid;
    double *e_cv; uint8_t *e_cv_valid;      // by exon row
    unsigned long long *bias3, *bias5;
    int *error;
};
void launch_gene_coverage(hipStream_t s, hipStream_t s2, hipStream_t s3, const GeneCovArgs &A, uint32_t n_large, uint32_t n_medium, uint32_t n_xlarge,
                          uint32_t n_le6144 = 0, uint32_t n_le3072 = 0, uint32_t n_le2048 = 0, uint32_t n_le1024 = 0);

void launch_reduce_add(hipStream_t s, unsigned long long *du, const unsigned long long *su, size_t nu, double *df, const double *sf, size_t nf,
                       uint8_t *db, const uint8_t *sb, size_t nb);
void launch_pack_results(hipStream_t s, const double *exon_acc, uint8_t *exon_hit, uint32_t n_exons);
void launch_reset(hipStream_t s, void *arena, size_t arena_bytes, void *cov, size_t cov_bytes, uint32_t *rl_min);
void launch_classify(hipStream_t s, int grid, int variant, const DevAnnotation &a, const DevParams &p, const DevBatch &b,
                     const DevAccum &acc);
// workgroups of classify_long_kernel for a K1 grid (each owns a pair chunk of its own behind the K1 grid's)
inline int rsqc_long_grid(int k1_grid) { return k1_grid < 1024 ? k1_grid : 1024; }
void launch_classify_long(hipStream_t s, int k1_grid, const DevAnnotation &a, const DevParams &p, const DevBatch &b, const DevAccum &acc);
void launch_ei_rank(hipStream_t s, const EiEntry *ei, uint32_t ei_lo, uint32_t ei_hi, EiRank *rank, uint32_t n_words);
void launch_classify_slow(hipStream_t s, const DevAnnotation &a, const DevParams &p, const DevBatch &b,
                          const DevAccum &acc);
// summary (may be NULL): RSQC_RL_SUMMARY_WORDS words of the batch's Read-Length transfer function (rsqc_kernels.hip, KR)
#define RSQC_RL_SUMMARY_WORDS (2 + 2 * 128)
void launch_read_length(hipStream_t s, const DevAnnotation &a, const DevParams &p, const DevBatch &b,
                        const DevAccum &acc, uint32_t *summary);
// K4, streaming form (rsqc_kernels.hip): partition tables laid out on the device from the final geneCounts
// fragment partitions (rsqc_kernels.hip, K4): a gene with n counted records owns ceil(n / PART_READS) partitions of capacity
// SUB_CAP keys each (or one partition of capacity n); a partition's keys are counted by one workgroup in an LDS set of PART_SLOTS
// --legacy: every record goes through classify_slow_kernel<true>, whose workgroups (at most RSQC_SLOW_LEGACY_GRID) reserve RSQC_SLOW_RES slots
// of the batch's dense pair region at a time (one returning memory atomic; >= the pairs of one pass of a workgroup) and fill what is left of
// their last block with empty entries: the region is sized for those too (rsqc_api.cpp), and frag_local_kernel shares it among enough workgroups
#define RSQC_SLOW_RES 2048u
#define RSQC_SLOW_LEGACY_GRID 4096u
#ifndef RSQC_K4_PART_READS
#define RSQC_K4_PART_READS 1024
#endif
#define RSQC_K4_SUB_CAP (2 * RSQC_K4_PART_READS)
#define RSQC_K4_PART_SLOTS (4 * RSQC_K4_PART_READS)
#ifndef RSQC_K4_COUNT_THREADS
#define RSQC_K4_COUNT_THREADS 256
#endif
// one entry of a partition's key list: the 96-bit identity of a read name (rsqc_rec_aux::qhash, rsqc_batch.qhash2 -- 0 for a
// caller that has none), written with ONE 12-byte store
struct FragKey { uint32_t lo, hi, h2; };          // (16-byte aligned entries measured the same: profiles/r6_k1_variants.txt, call r6d)
struct FragPlan {
    uint32_t *part_first;          // [G + 1] first partition of a gene
    uint32_t *cursor;              // [parts] keys appended so far
    uint4 *ginfo;                  // [G] {first partition, partitions, capacity of one, offset of the gene's key lists / 16}: what frag_local_kernel gathers per pair
    uint4 *part_info;              // [parts] {owning gene, capacity, list offset lo, hi}
    uint32_t *full_list, *full_n;  // [parts] + counter: the partitions frag_count_kernel's first instance leaves to the second
    FragKey *list;                 // key lists
    unsigned long long *blk_space; uint32_t *blk_parts;   // [ceil(G / 1024)] per-workgroup totals of the layout scan
};
void launch_frag_layout(hipStream_t s, const unsigned long long *gene_reads, uint32_t n_genes, const FragPlan &P, int *error);
void launch_frag_local(hipStream_t s, const DevAccum &acc, uint32_t n_chunks, const FragPlan &P, uint32_t list_blocks);
void launch_pairs_append(hipStream_t s, const PairRec *src, uint32_t chunk_cap, const uint32_t *counts,
                         uint32_t n_chunks, uint32_t slow_base, uint32_t slow_cap, PairRec *dst);
void launch_frag_count(hipStream_t s, uint32_t n_genes, const FragPlan &P, uint32_t parts_bound, unsigned long long *gene_frag, int *error);

}  // namespace rsqc

#include <vector>
namespace rsqc {
// device scratch of the pairing steps (rsqc_fragsize.hip), kept by the context between passes (allocation and release synchronise
// the device): k0 / v0 samples (file index, size), k1 / v1 the kept ones, v2 candidate indices bucket by bucket, tmp the bucket
// counts / offsets / cursors, count the control words, table / out_* the size histogram and its compacted (size, count) pairs
struct SortScratch { void *k0 = nullptr, *k1 = nullptr, *v0 = nullptr, *v1 = nullptr, *v2 = nullptr, *v3 = nullptr, *tmp = nullptr, *count = nullptr;
                     size_t cap_n = 0, tmp_bytes = 0; uint32_t *table = nullptr, *out_size = nullptr, *out_count = nullptr; };
void free_sort_scratch(SortScratch &s);
// leaves the kept samples (first max_samples by file index, unordered) on the device: S.k1 = file index, S.v1 = size, n_kept of them
int run_fragment_sizes(hipStream_t stream, const FragCandidates &c, uint32_t n, uint32_t max_samples,
                       std::vector<int64_t> &sizes, std::vector<uint64_t> &counts, uint32_t &remaining, SortScratch &S, uint32_t &n_kept, int *d_error);
void launch_frag_compact(hipStream_t s, const FragCandidates &src, const FragCandidates &dst, uint64_t n_rec, int k1_grid);
// --fasta (rsqc_kernels.hip / rsqc_fragsize.hip)
void launch_gc_pack(hipStream_t s, const uint8_t *ascii, uint64_t len, unsigned long long *words);
void launch_exon_gc(hipStream_t s, const DevAnnotation &a, const DevReference &R, double *exon_gc);
void launch_gc_candidates(hipStream_t s, const DevAnnotation &a, const DevParams &p, const DevBatch &b, const DevReference &R,
                          const GcCandidates &out, int *error);
// pairs the candidates by QNAME in file order and adds every usable fragment to bins[0..100] (slot 100 = 100 % GC);
// asynchronous on `stream`
int run_gc_content(hipStream_t stream, const GcCandidates &c, uint32_t n, const DevReference &R, unsigned long long *bins, SortScratch &scratch, int *d_error);
}
