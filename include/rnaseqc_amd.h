/*
 * rnaseqc_amd.h -- C ABI of the MI355X-native RNA-SeQC per-read hot path.
 *
 * The reference (getzlab/rnaseqc 2.4.3) has no plugin / FFI interface: its hot
 * path is five C++ free functions called from the BAM loop of main()
 * (src/RNASeQC.cpp:242-382, signatures in src/Expression.h:19-37) that mutate
 * namespace-scope globals (src/Metrics.cpp:20-22, src/GTF.cpp:22-27).  This
 * header is the batch-oriented C boundary that replaces that seam:
 *
 *   reference call site / state                         replaced by
 *   --------------------------------------------------  -------------------------
 *   GTF load into features/geneList/exonList/           rsqc_set_annotation()
 *     exonsForGene/exonLengths  (src/RNASeQC.cpp:108-156,
 *     src/GTF.cpp:30-131)
 *   BED load (src/RNASeQC.cpp:172-187, src/BED.cpp:18)  rsqc_set_bed()
 *   Fasta::open / getSeq (src/Fasta.cpp:77-140,          rsqc_set_reference()
 *     src/RNASeQC.cpp:118-121)
 *   while (bam.next(alignment)) { gate cascade;         rsqc_submit() /
 *     extractBlocks; trimFeatures;                      rsqc_submit_resident()
 *     exonAlignmentMetrics; fragmentSizeMetrics }       (one call per SoA batch
 *     (src/RNASeQC.cpp:242-382)                          of records, file order)
 *   dropFeatures at EOF -> BaseCoverage::compute ->     rsqc_finalize()
 *     computeCoverage/computeBias
 *     (src/RNASeQC.cpp:385-388, src/Metrics.cpp:132-337)
 *   geneCounts/uniqueGeneCounts/geneFragmentCounts/     rsqc_results
 *     exonCounts/Metrics/BiasCounter/BaseCoverage
 *     lists/fragmentSizes/readLength read by the
 *     report writer (src/RNASeQC.cpp:397-676)
 *
 * Conventions: plain C, caller-owned inputs, library-owned outputs (valid
 * until the next rsqc_finalize()/rsqc_destroy()), 0 = success, negative =
 * error (no exceptions cross the boundary).  One context drives one GPU (one
 * shard of contigs); contexts are independent, so a node runs one per GPU.
 * All coordinates are as in the reference: features 1-based closed
 * (src/GTF.h:29-37), record pos/mpos 0-based as in BAM (htslib core.pos).
 */
#ifndef RNASEQC_AMD_H
#define RNASEQC_AMD_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RSQC_ABI_VERSION 5

#if defined(__GNUC__)
#define RSQC_API __attribute__((visibility("default")))
#else
#define RSQC_API
#endif

/* ---- error codes ------------------------------------------------------- */
#define RSQC_OK              0
#define RSQC_ERR_ARG        -1   /* bad argument / call order                           */
#define RSQC_ERR_HIP        -2   /* HIP runtime failure (maps to exit code 10)          */
#define RSQC_ERR_BAD_CIGAR  -3   /* CIGAR op outside MIDNSHP=X: the reference throws
                                    std::invalid_argument (src/Expression.cpp:61-63),
                                    exit code 7                                         */
#define RSQC_ERR_CAPACITY   -4   /* a fixed device-side capacity was exceeded            */
#define RSQC_ERR_NO_DEVICE  -5   /* no HIP device: there is NO CPU fallback              */
#define RSQC_ERR_EMPTY_MEDIAN -6 /* std::range_error of computeMedian (src/Metrics.h:149) */
#define RSQC_ERR_INPUT      -7   /* rsqc_decode_*: corrupt BGZF block / malformed or truncated BAM record (the reference
                                    ends with htslib's read error)                       */

/* ---- BAM flag bits (SAM spec; accessors used at src/RNASeQC.cpp:254-330) */
#define RSQC_FPAIRED   0x1
#define RSQC_FPROPER   0x2
#define RSQC_FUNMAP    0x4
#define RSQC_FMUNMAP   0x8
#define RSQC_FREVERSE  0x10
#define RSQC_FMREVERSE 0x20
#define RSQC_FREAD1    0x40
#define RSQC_FREAD2    0x80
#define RSQC_FSECONDARY 0x100
#define RSQC_FQCFAIL   0x200
#define RSQC_FDUP      0x400
#define RSQC_FSUPP     0x800

/* ---- tagbits byte of a record ------------------------------------------ */
#define RSQC_TB_HAS_NM     0x01  /* GetIntTag("NM") succeeded (src/RNASeQC.cpp:295)      */
#define RSQC_TB_HAS_CH     0x02  /* readStringTag(chimeric_tag): Z tag, or A tag with a
                                    non-NUL char (src/RNASeQC.cpp:780-800)              */
#define RSQC_TB_MTID_SAME  0x04  /* ChrID() == MateChrID() (src/RNASeQC.cpp:287)         */
#define RSQC_TB_FILTER0    0x08  /* --tag k present (GetTag: Z, int or float type,
                                    src/RNASeQC.cpp:320-327); k = 0..4 -> bits 3..7     */
#define RSQC_MAX_FILTER_TAGS 5

/* saturating narrow fields: a record whose nm/l_qseq/n_cigar does not fit is
 * stored with the escape value and listed (index-sorted) in the wide table   */
#define RSQC_NM_ESCAPE     0xFF
#define RSQC_LQSEQ_ESCAPE  0xFFFF
#define RSQC_NCIGAR_ESCAPE 0xFF

/* ---- strandedness (reference enum Strand {Forward, Reverse, Unknown},
 *      src/Fasta.h:41; --stranded FR -> Forward, RF -> Reverse,
 *      src/RNASeQC.cpp:78-85) ------------------------------------------------ */
#define RSQC_STRAND_FORWARD 0
#define RSQC_STRAND_REVERSE 1
#define RSQC_STRAND_UNKNOWN 2

/* ---- feature flag byte --------------------------------------------------- */
#define RSQC_FF_STRAND_MASK 0x03 /* RSQC_STRAND_*                                        */
#define RSQC_FF_RIBOSOMAL   0x04 /* transcript_type contains "rRNA" (src/GTF.cpp:113)    */

/* ---- scalar counters (Metrics keys, src/Metrics.cpp:344-384 and
 *      src/RNASeQC.cpp:254-360, src/Expression.cpp:402-455) ------------------ */
enum rsqc_counter {
    RSQC_C_ALTERNATIVE_ALIGNMENTS = 0,
    RSQC_C_SUPPLEMENTARY_ALIGNMENTS,
    RSQC_C_FAILED_VENDOR_QC,
    RSQC_C_LOW_MAPPING_QUALITY,
    RSQC_C_CHIMERIC_AUTO,
    RSQC_C_CHIMERIC_TAG,
    RSQC_C_UNIQUE_VENDOR_PASSED,
    RSQC_C_UNPAIRED_READS,
    RSQC_C_MAPPED_READS,
    RSQC_C_MAPPED_DUPLICATE_READS,
    RSQC_C_MAPPED_UNIQUE_READS,
    RSQC_C_TOTAL_MAPPED_PAIRS,
    RSQC_C_END1_MAPPED_READS,
    RSQC_C_END1_MISMATCHES,
    RSQC_C_END1_BASES,
    RSQC_C_DUPLICATE_PAIRS,
    RSQC_C_UNIQUE_FRAGMENTS,
    RSQC_C_END2_MAPPED_READS,
    RSQC_C_END2_MISMATCHES,
    RSQC_C_END2_BASES,
    RSQC_C_MISMATCHED_BASES,
    RSQC_C_TOTAL_BASES,
    RSQC_C_HIGH_QUALITY_READS,
    RSQC_C_LOW_QUALITY_READS,
    RSQC_C_READS_USED,
    RSQC_C_ALIGNMENT_BLOCKS,
    RSQC_C_NON_GLOBIN_READS,
    RSQC_C_NON_GLOBIN_DUPLICATE_READS,
    RSQC_C_INTRONIC_READS,
    RSQC_C_INTRAGENIC_READS,
    RSQC_C_HQ_INTRONIC_READS,
    RSQC_C_HQ_INTRAGENIC_READS,
    RSQC_C_INTERGENIC_READS,
    RSQC_C_HQ_INTERGENIC_READS,
    RSQC_C_EXONIC_READS,
    RSQC_C_HQ_EXONIC_READS,
    RSQC_C_AMBIGUOUS_READS,
    RSQC_C_HQ_AMBIGUOUS_READS,
    RSQC_C_RRNA_READS,
    RSQC_C_END1_SENSE,
    RSQC_C_END1_ANTISENSE,
    RSQC_C_END2_SENSE,
    RSQC_C_END2_ANTISENSE,
    RSQC_C_TOTAL_ALIGNMENTS,
    RSQC_C_FILTERED_TAG0,          /* "Filtered by tag: X", one per --tag        */
    RSQC_C_FILTERED_TAG1,
    RSQC_C_FILTERED_TAG2,
    RSQC_C_FILTERED_TAG3,
    RSQC_C_FILTERED_TAG4,
    RSQC_C_SPLIT_READS,            /* --legacy only (src/Expression.cpp:274); printed when non-zero
                                      (src/Metrics.cpp:398)                                          */
    RSQC_N_COUNTERS
};

/* ---- run parameters (flag table src/RNASeQC.cpp:39-65, defaults :87-100) - */
typedef struct rsqc_params {
    uint32_t abi_version;          /* RSQC_ABI_VERSION                                   */
    int32_t  device;               /* HIP device ordinal                                 */
    uint32_t mapq_threshold;       /* -q, 255                                            */
    uint32_t base_mismatch;        /* --base-mismatch, 6                                 */
    int32_t  chimeric_distance;    /* --chimeric-distance, 2000000                       */
    uint32_t fragment_samples;     /* --fragment-samples, 1000000 (used iff BED is set)  */
    int32_t  bias_offset;          /* --offset, 0                                        */
    int32_t  bias_window;          /* --window-size, 100                                 */
    uint64_t bias_gene_length;     /* --gene-length, 200                                 */
    uint32_t coverage_mask;        /* --coverage-mask, 500                               */
    int32_t  stranded;             /* RSQC_STRAND_*; UNKNOWN = unstranded                */
    int32_t  unpaired;             /* -u                                                 */
    int32_t  exclude_chimeric;     /* --exclude-chimeric                                 */
    int32_t  n_filter_tags;        /* number of --tag filters carried in tagbits (<=5)   */
    int32_t  legacy;               /* --legacy: RNA-SeQC 1.1.9 counting rules
                                      (legacyExonAlignmentMetrics, src/Expression.cpp:129-304, and the
                                      LegacyMode tests of src/RNASeQC.cpp:258-287).  The caller applies
                                      the two host-side parts: -q defaults to 4 (src/RNASeQC.cpp:90) and
                                      1-base features are left out of the annotation (:129-135)        */
    int32_t  reserved[6];
} rsqc_params;

/* ---- annotation: the flattened form of the reference's GTF state ----------
 * Contig ids: [0, n_ref) are the BAM @SQ entries in header order (a record's
 * tid); [n_ref, n_contigs) are contigs that only the GTF names (their genes
 * still produce coverage rows at EOF, src/RNASeQC.cpp:385-386).
 * Gene ids: [0, n_genes_listed) = geneList order (GTF order of `gene` rows,
 * src/GTF.cpp:87); [n_genes_listed, n_genes) = gene_ids that only exon rows
 * name (they are counted but never reported).  Exon ids = exonList order.
 * Interval rows are sorted by (contig, start), ties in GTF order -- the
 * reference's per-contig std::list::sort(compIntervalStart), stable
 * (src/RNASeQC.cpp:150-152).  start <= end is required.                      */
typedef struct rsqc_annotation {
    int32_t n_ref, n_contigs;
    int32_t n_genes, n_genes_listed, n_exons;

    /* `gene` rows, sorted; n_gene_rows == n_genes_listed                      */
    const int32_t  *gene_row_contig;   /* [n_genes_listed]                     */
    const int32_t  *gene_row_start;    /* 1-based                              */
    const int32_t  *gene_row_end;      /* 1-based closed                       */
    const uint8_t  *gene_row_flags;    /* RSQC_FF_*                            */
    const uint32_t *gene_row_id;       /* gene id of the row                   */

    /* `exon` rows, sorted                                                      */
    const int32_t  *exon_row_contig;   /* [n_exons]                            */
    const int32_t  *exon_row_start;
    const int32_t  *exon_row_end;
    const uint8_t  *exon_row_flags;
    const uint32_t *exon_row_id;       /* exon id (exonList index)             */
    const uint32_t *exon_row_gene;     /* gene id named by the row's gene_id   */

    /* per gene id                                                              */
    const uint8_t  *gene_is_globin;    /* [n_genes] geneNames[gene] in the 12-name
                                          blacklist (src/Expression.cpp:24,396-398) */
    /* exonsForGene (src/RNASeQC.cpp:153-154): CSR over gene id -> sorted exon ROW
       indices, in sorted order                                                 */
    const uint32_t *gene_exon_off;     /* [n_genes + 1]                        */
    const uint32_t *gene_exon_row;     /* [n_exons]                            */

    /* optional (may both be NULL), read only under rsqc_params.legacy: the position of each kept row in the
       GTF (any strictly increasing key, e.g. the line number).  The legacy rules depend on the order of the
       reference's ONE start-sorted list of genes and exons (stable sort, src/RNASeQC.cpp:150-152), i.e. on how
       a gene row and an exon row with the same start are ordered.  When NULL, gene rows are taken to precede
       exon rows of the same start (a GTF whose gene lines precede their exon lines).                        */
    const uint32_t *gene_row_order;    /* [n_genes_listed]                     */
    const uint32_t *exon_row_order;    /* [n_exons]                            */
} rsqc_annotation;

/* ---- BED intervals for the fragment-size sampler (src/BED.cpp:18-45):
 * stored +1/+1 like the reference (start = bed_start+1, end = bed_end+1),
 * per contig in file order, which must be ascending by start               */
typedef struct rsqc_bed {
    int32_t n_intervals;
    const int32_t *contig;             /* contig id (same id space as above)   */
    const int32_t *start;
    const int32_t *end;
} rsqc_bed;

/* ---- reference sequence for the GC statistics of --fasta (src/Fasta.cpp, bioio.hpp): what the reference reads
 * through its .fai index, handed over as plain per-contig base strings (FASTA text without line ends, any case;
 * G/g/C/c count, everything else does not -- gc(), src/Fasta.cpp:67-74).  Contigs the FASTA index does not name
 * are simply absent (Fasta::hasContig).  The library keeps one bit per base in HBM.                             */
typedef struct rsqc_reference {
    int32_t n;                         /* contigs named by the FASTA index     */
    const int32_t  *contig;            /* [n] boundary contig id               */
    const uint64_t *length;            /* [n] bases                            */
    const uint8_t *const *sequence;    /* [n] `length[i]` ASCII bases each     */
} rsqc_reference;
#define RSQC_GC_BINS 100               /* unsigned long gcBins[100], src/RNASeQC.cpp:106 */

/* ---- one batch of alignment records, file order ----------------------------
 * 32 bytes per record + 4 bytes per CIGAR op, stored as two
 * arrays of 16-byte half-records so that a wavefront reads each with one
 * 16-byte-per-lane vector load (1 KiB per wave instruction).  Records of one
 * contig form a segment; tid itself is not stored per record.               */
typedef struct rsqc_rec_core {         /* 16 bytes                             */
    int32_t  pos;                      /* core.pos (0-based)                   */
    int32_t  mpos;                     /* core.mpos                            */
    int32_t  isize;                    /* core.isize                           */
    uint32_t cigar_off;                /* first op of the record in `cigar`    */
} rsqc_rec_core;

typedef struct rsqc_rec_aux {          /* 16 bytes                             */
    uint64_t qhash;                    /* rsqc_qname_hash(QNAME)               */
    uint16_t flag;
    uint16_t l_qseq;                   /* core.l_qseq, RSQC_LQSEQ_ESCAPE = wide */
    uint8_t  mapq;
    uint8_t  nm;                       /* NM value, RSQC_NM_ESCAPE = wide      */
    uint8_t  tagbits;                  /* RSQC_TB_*                            */
    uint8_t  n_cigar;                  /* RSQC_NCIGAR_ESCAPE = wide            */
} rsqc_rec_aux;

/* Limits of one batch: n < 2^32 - 16 records, n_cigar_total < 2^30 ops (split larger inputs into several
 * batches; 1-4 M records per batch is the intended granularity).                                          */
typedef struct rsqc_batch {
    uint64_t n;                        /* records                              */
    uint64_t file_index_base;          /* index of record 0 in the whole file: a batch is a contiguous range of the
                                          file, batches are submitted in ascending order (gaps allowed: the records
                                          of other shards); decides the fragment-size cut-off and the shard merge */
    const rsqc_rec_core *core;         /* [n]                                  */
    const rsqc_rec_aux  *aux;          /* [n]                                  */
    const uint32_t *cigar;             /* BAM packed ops: len<<4 | op          */
    uint64_t n_cigar_total;

    /* contig segments: records [seg_start[s], seg_start[s+1]) have tid seg_tid[s];
       tid may be -1 (unplaced) or >= n_ref (unrecognised RefID)                */
    uint32_t n_seg;
    const int32_t  *seg_tid;           /* [n_seg]                              */
    const uint64_t *seg_start;         /* [n_seg + 1]                          */

    /* wide table for records carrying an escape value, ascending by index     */
    uint32_t n_wide;
    const uint64_t *wide_index;        /* record index within the batch        */
    const int32_t  *wide_nm;
    const int32_t  *wide_l_qseq;
    const uint32_t *wide_n_cigar;

    /* optional (may be NULL): exact QNAMEs, used only by the oracle to check
       that hashing does not change the fragment de-duplication               */
    const uint32_t *qname_off;         /* [n + 1]                              */
    const char     *qname;

    /* optional (may be NULL): a SECOND, independent hash of every record's QNAME (rsqc_qname_hash2).  With it the hot path
       identifies a read name by 96 bits -- (qhash, qhash2) compared exactly in the fragment de-duplication (src/Expression.cpp:
       383-387 compares the strings): two different names are merged only if BOTH hashes collide.  Without it the identity is the
       64-bit qhash alone.  The 32-byte record itself has no room for it (its layout is the contract's: 32 + 4 n_cigar bytes per
       record); the column is read only for records that are counted to a gene.  Both ingest paths of the library (the device
       decode and the host reader) fill it.  Every QNAME-keyed stage uses the same identity: geneFragmentCounts, the fragment-size
       sampler (src/Expression.cpp:511-531) and the fragment GC pairing (:461-474).  ALL OR NONE: the batches of one pass (between
       two rsqc_reset) either all carry the column or none does -- a submit that disagrees with the pass's first batch fails with
       RSQC_ERR_ARG (the two mates of a fragment would otherwise carry (qhash, h2) and (qhash, 0): two names).
       LIMIT: names that share their 64-bit qhash and differ in qhash2 are counted exactly up to 33 of them per fragment partition
       (a gene's names of one hash stripe, ~1 000 records); more is RSQC_ERR_CAPACITY, never a miscount.  rsqc_qname_hash values of
       real read names do not collide at all at these sizes; a crafted 64-bit collision costs ~5e9 hash evaluations each.     */
    const uint32_t *qhash2;            /* [n]                                  */

    /* optional (may be NULL): the batch is SEVERAL ranges of the file, one per contig segment -- seg_file_index[s] is the file index of
       segment s's first record (ascending, every range behind the one before and behind everything submitted earlier); the
       file index of record i of segment s is seg_file_index[s] + (i - seg_start[s]) and file_index_base is ignored.  A GPU
       that owns a set of NON-ADJACENT contigs of a sorted file (a contig-sharded run) submits them as ONE batch and
       one kernel launch; the order-dependent outputs are then kept per segment (rsqc_shard_summary lists one entry per segment
       instead of one per batch) and the context's own "Read Length" is composed from them in file order.  Since ABI version 4.  */
    const uint64_t *seg_file_index;    /* [n_seg]                              */
} rsqc_batch;

/* ---- results ---------------------------------------------------------------- */
typedef struct rsqc_results {
    int32_t  n_genes_listed, n_exons;
    /* geneCounts / uniqueGeneCounts / geneFragmentCounts (src/Metrics.cpp:20) */
    const uint64_t *gene_reads;        /* [n_genes_listed], geneList order     */
    const uint64_t *gene_unique;
    const uint64_t *gene_fragments;
    /* exonCounts: sum of intersection/alignedLength (src/Expression.cpp:345,
       src/Metrics.cpp:63); exon_hit != 0 iff the exon has a map entry (Q7)    */
    const double   *exon_reads;        /* [n_exons], exonList order            */
    const uint8_t  *exon_hit;
    uint64_t counters[RSQC_N_COUNTERS];
    int32_t  read_length;              /* "Read Length" (src/RNASeQC.cpp:275-278) */

    /* BaseCoverage::compute per listed gene (src/Metrics.cpp:132-151,265-337):
       coverage.tsv row (mean, std, cv); cov_valid == 0 -> row "0 0 nan" and the
       gene is absent from geneMeans/Stds/CVs                                  */
    const double   *gene_cov_mean;     /* [n_genes_listed]                     */
    const double   *gene_cov_std;
    const double   *gene_cov_cv;
    const uint8_t  *gene_cov_valid;
    /* per-exon CV (exon_cv.tsv): exon_cv_valid != 0 iff finite CV was stored  */
    const double   *exon_cv;           /* [n_exons]                            */
    const uint8_t  *exon_cv_valid;
    /* BiasCounter accumulators (src/Metrics.h:76-77)                          */
    const uint64_t *bias_three;        /* [n_genes_listed]                     */
    const uint64_t *bias_five;

    /* fragment size histogram (map<long long, unsigned long>), ascending size */
    uint32_t n_fragment_sizes;
    const int64_t  *fragment_size;
    const uint64_t *fragment_count;
    uint32_t fragment_samples_remaining;

    /* --fasta (valid when have_reference != 0): fragment GC histogram gcBins (src/RNASeQC.cpp:366-369, bin =
       unsigned(gc * 100)); a fragment of 100 % GC indexes past the reference's array (undefined there) and is
       counted in gc_out_of_range instead.  exon_gc = gc() of the exon's sequence as fetched at
       src/Metrics.cpp:301 (-1 when the FASTA lacks the contig); meaningful where exon_cv_valid != 0.              */
    int32_t  have_reference;
    const uint64_t *gc_bins;           /* [RSQC_GC_BINS]                       */
    uint64_t gc_out_of_range;
    const double   *exon_gc;           /* [n_exons], exonList order            */
    /* ABI 5: exon rows of the annotation that lie outside the row of their gene (0 for a well-formed GTF).  Non-zero means that
       gene_fragments and the coverage / bias statistics of THOSE genes are computed "as if the gene stayed in the window" and can
       differ from the reference's streamed result (rsqc_set_annotation's warning): the divergence is flagged in the
       results themselves, not only in rsqc_last_error, which any later failure overwrites.                                    */
    uint32_t exons_outside_gene_row;
} rsqc_results;

/* ---- timing of the device work (HIP events on the context's own stream) --- */
typedef struct rsqc_timing {
    double   classify_ms;              /* sum of K1 classify_count launches since reset */
    uint64_t classify_launches;
    uint64_t classify_records;
    uint64_t classify_bytes;           /* algorithmic bytes: 32*n + 4*n_cigar_total     */
    double   finalize_ms;              /* de-dup + coverage scan/stats + bias           */
    double   h2d_ms;                   /* rsqc_submit host->device copies               */
    uint64_t slow_records;             /* records the general (slow-path) kernel took in the last finalized pass */
    double   fragment_sizes_ms;        /* --bed runs: the fragment-size stage of the end-of-file passes (host clock around its kernels and its
                                          two read-backs; src/Expression.cpp:482-540), summed since reset                              */
    double   classify_long_ms;         /* ABI 5: the launches of classify_long_kernel (the ~1 % of the records the per-record kernel defers: more than
                                          eight CIGAR operations or more than three blocks) since reset; NOT part of classify_ms             */
} rsqc_timing;

typedef struct rsqc_ctx rsqc_ctx;

/* Creates a context on params->device.  Fails with RSQC_ERR_NO_DEVICE when no
 * HIP device is usable -- the product has no CPU path.                        */
RSQC_API int rsqc_create(const rsqc_params *params, rsqc_ctx **out);
RSQC_API void rsqc_destroy(rsqc_ctx *ctx);

/* Copies the annotation into HBM and builds the device index.  `owned_contig`
 * (n_contigs bytes, may be NULL = all) marks the contigs of this shard: only
 * their genes get coverage/bias results (multi-GPU by contig).
 * RSQC_OK may come WITH A WARNING in rsqc_last_error (empty otherwise): an exon row outside the row of its gene.  The
 * reference runs such a GTF and prints "Gene encountered after computing coverage" (src/Metrics.cpp:108-112) when a read
 * reaches the exon after the gene row has left its window; here every record is answered from the rows as given: counters,
 * gene reads / unique reads and exon reads are the reference's, fragment counts and coverage statistics of THAT gene are
 * computed as if the gene had stayed in the window.                                                        */
RSQC_API int rsqc_set_annotation(rsqc_ctx *ctx, const rsqc_annotation *ann,
                        const uint8_t *owned_contig);
RSQC_API int rsqc_set_bed(rsqc_ctx *ctx, const rsqc_bed *bed);
/* --fasta: after rsqc_set_annotation.  Copies the bases to the device, packs them to one G/C bit per base and
 * computes the per-exon GC values; the caller's strings are not referenced afterwards.                          */
RSQC_API int rsqc_set_reference(rsqc_ctx *ctx, const rsqc_reference *ref);

/* Asynchronous: copies the batch H2D on the context's stream and launches the
 * per-read kernels; returns without waiting for either.  The batch memory must
 * stay valid and unmodified until rsqc_wait() (page-locked arrays from
 * rsqc_host_alloc are copied by DMA).  Batches must be submitted in file order. */
RSQC_API int rsqc_submit(rsqc_ctx *ctx, const rsqc_batch *batch);
RSQC_API int rsqc_wait(rsqc_ctx *ctx);

/* Resident variant (what bench.py times): upload once, run many times.       */
RSQC_API int rsqc_upload(rsqc_ctx *ctx, const rsqc_batch *batch, int *handle_out);
RSQC_API int rsqc_submit_resident(rsqc_ctx *ctx, int handle);
RSQC_API int rsqc_release(rsqc_ctx *ctx, int handle);

/* End of file: fragment de-dup, coverage scan, per-gene coverage statistics and
 * bias windows, fragment-size pairing; fills `out`, whose vectors point into a
 * page-locked host mirror owned by the context (valid until the next
 * rsqc_finalize / rsqc_refresh_results / rsqc_destroy).                       */
RSQC_API int rsqc_finalize(rsqc_ctx *ctx, rsqc_results *out);

/* Zeroes every accumulator (keeps annotation/BED and uploaded batches).       */
RSQC_API int rsqc_reset(rsqc_ctx *ctx);

RSQC_API int rsqc_get_timing(rsqc_ctx *ctx, rsqc_timing *out);
RSQC_API int rsqc_reset_timing(rsqc_ctx *ctx);

/* Device-resident raw accumulators for an in-place RCCL reduction by the host
 * (torch.distributed).  Pointers are HIP device pointers owned by the ctx.
 * Layout: u64 gene_reads[n_genes] | gene_unique[n_genes] | gene_fragments[n_genes]
 * | counters[RSQC_N_COUNTERS] in one allocation; f64 exon_reads[n_exons] (row
 * order) in another.  Valid after rsqc_finalize().                            */
RSQC_API int rsqc_device_accumulators(rsqc_ctx *ctx, void **u64_base, uint64_t *u64_count,
                             void **f64_base, uint64_t *f64_count);
/* Everything a contig-sharded run exchanges, as three device ranges of one element type each, to
 * be sum-reduced in place across the ranks (RCCL all_reduce, SUM) between rsqc_finalize_device and
 * rsqc_refresh_results:
 *   [0] u64: gene_reads | gene_unique | gene_fragments | counters | bias_three | bias_five
 *   [1] f64: exon_reads | gene_cov_mean | gene_cov_std | gene_cov_cv | exon_cv
 *   [2] u8 : gene_cov_valid | exon_cv_valid (padded to 8 bytes each)
 * Counts and exon sums are additive; the per-gene / per-exon statistics are written by the owner of the gene's
 * contig only and stay zero elsewhere, so the same sum is their merge (a NaN CV stays NaN).  rsqc_device_accumulators
 * exposes the additive prefixes of [0] and [1] only.                                                              */
typedef struct rsqc_device_range { void *base; uint64_t count; } rsqc_device_range;
RSQC_API int rsqc_device_vectors(rsqc_ctx *ctx, rsqc_device_range out[3]);
/* The order-dependent outputs of a shard, for the host-side merge of a contig-sharded run; valid
 * after rsqc_finalize / rsqc_finalize_device until the next reset.  Pointers are owned by the context.
 *   Read Length (src/RNASeQC.cpp:275-278) is a state machine over the file.  Per submitted batch the library keeps
 *   the batch's transfer function as a short table: entered with state r, the batch leaves rl_state[k] for the first k
 *   in [rl_offset[b], rl_offset[b+1]) with rl_span[k] > r, and r itself when no entry qualifies.  Composing the
 *   batches of ALL shards in ascending batch_file_index from state 0 gives the reference's value.
 *   Fragment sizes (src/Expression.cpp:482-540): mates pair inside one BED interval, hence inside one shard; the
 *   cut-off --fragment-samples applies in FILE order.  The shard hands over the samples it kept (its first N by the file
 *   index of the completing record, in no particular order); the merged histogram holds the N smallest file indices
 *   of the union.                                                                                                   */
typedef struct rsqc_shard_info {
    uint32_t n_batches;
    const uint64_t *batch_file_index;  /* [n_batches] file_index_base of the batch (of the SEGMENT for a batch with seg_file_index: one entry per segment) */
    const uint64_t *batch_records;     /* [n_batches]                               */
    const uint32_t *rl_offset;         /* [n_batches + 1]                           */
    const uint32_t *rl_span;           /* ascending inside a batch                  */
    const int32_t  *rl_state;
    uint32_t n_samples;
    const uint64_t *sample_file_index; /* [n_samples] unordered                     */
    const uint32_t *sample_size;       /* [n_samples] abs(InsertSize)               */
} rsqc_shard_info;
RSQC_API int rsqc_shard_summary(rsqc_ctx *ctx, rsqc_shard_info *out);
/* dst += src over those three ranges, for ONE process that drives several GPUs (the command line with --gpus): the
 * peer's ranges cross xGMI by hipMemcpyPeerAsync and are added on dst's device.  Both contexts must hold the same
 * annotation and be past rsqc_finalize_device.  A process-per-GPU host uses an RCCL all_reduce on the ranges instead.  */
RSQC_API int rsqc_reduce_peer(rsqc_ctx *dst, rsqc_ctx *src);
/* The exchange step of ONE process that drives n GPUs (the command line with --gpus): the three ranges of every context
 * are sum-reduced onto ctxs[0] with one RCCL ncclReduce per range inside one group call, each GPU's part on its context's
 * stream, over communicators made with ncclCommInitAll on the contexts' devices (src/RNASeQC.cpp:385-394 is the reference's
 * end-of-file window).  librccl is bound at run time; without it, or when two contexts share a device
 * (a communicator cannot), the peer-copy path of rsqc_reduce_peer is taken instead.  *used_rccl (may be NULL) says which. */
RSQC_API int rsqc_reduce_group(rsqc_ctx **ctxs, int n, int *used_rccl);
/* The same exchange with the communicators made ONCE, outside the caller's timed region: rsqc_group_create brings RCCL up on
 * the contexts' devices (any time after rsqc_create; ncclCommInitAll over eight GPUs takes longer than the BAM loop of a
 * 100 M-record file, so the command line calls it beside the GTF parse, before the reference's `Average Reads/Sec` window
 * opens); rsqc_group_reduce = what rsqc_reduce_group does at end of file, on the group's communicators.  A group whose
 * communicators cannot be made (no librccl, RSQC_NO_RCCL, two contexts on one device, ncclCommInitAll failing for lack of
 * P2P or shared memory) is still a valid group: it sums the shards by peer copies, and rsqc_group_info says why.  An RCCL
 * failure AFTER reductions were issued is fatal (ctxs[0] may hold partial sums); one before takes the peer path.          */
typedef struct rsqc_group rsqc_group;
RSQC_API int rsqc_group_create(rsqc_ctx **ctxs, int n, rsqc_group **out);
RSQC_API int rsqc_group_reduce(rsqc_group *group, int *used_rccl);
RSQC_API int rsqc_group_info(const rsqc_group *group, int *uses_rccl, double *init_ms, double *last_reduce_ms, const char **note);
RSQC_API void rsqc_group_destroy(rsqc_group *group);
/* Re-reads the (reduced) device accumulators into the results struct.         */
RSQC_API int rsqc_refresh_results(rsqc_ctx *ctx, rsqc_results *out);
/* Page-locked host memory for the arrays of an rsqc_batch: rsqc_submit then copies by DMA and returns without
 * waiting for the transfer (arrays from ordinary memory work too, through the driver's staging copy).
 * NULL when no device is available.                                                                       */
RSQC_API void *rsqc_host_alloc(size_t bytes);
RSQC_API void rsqc_host_free(void *p);
/* rsqc_finalize without the read-back: runs the end-of-file stage and leaves every result on the device
 * (multi-GPU runs reduce the accumulators first and read back once with rsqc_refresh_results).            */
RSQC_API int rsqc_finalize_device(rsqc_ctx *ctx);

/* ---- device-side BAM decode --------------------------------------------------------------
 * The input side of the per-read path, moved to the GPU: the caller reads the file and hops over the BGZF block
 * headers (src/BamReader.cpp:12-20 does this through htslib's bgzf.c, one thread, inflate included); inflating
 * (one wavefront per block), finding the records and parsing them into an rsqc_batch run on the device, and the
 * batch is submitted as by rsqc_submit.  One stream of consecutive blocks between rsqc_decode_begin and
 * rsqc_decode_end; a record that straddles two calls is carried over on the device.                              */
typedef struct rsqc_bgzf_block {
    uint64_t in_offset;                /* first DEFLATE byte of the block, from `compressed`                   */
    uint32_t in_bytes;                 /* DEFLATE bytes (BSIZE + 1 - XLEN - 20)                                */
    uint32_t out_bytes;                /* ISIZE (<= 65536)                                                     */
    uint32_t crc32;                    /* of the inflated bytes (the gzip trailer)                             */
    uint32_t flags;                    /* RSQC_BGZF_*                                                          */
} rsqc_bgzf_block;
/* The block has been inflated by the caller (spare CPU threads sharing the work with the GPU): in_offset locates its
 * INFLATED bytes in `compressed`, in_bytes == out_bytes, the CRC-32 has been checked.  Such blocks form one run at the
 * END of a call's table, their bytes one after the other in `compressed`.                                          */
#define RSQC_BGZF_INFLATED 1u
typedef struct rsqc_decode_params {
    int32_t n_ref;                     /* reference sequences in the BAM header                                */
    int32_t has_chimeric_tag;          /* --chimeric-tag (readStringTag, src/RNASeQC.cpp:780-800)              */
    char    chimeric_tag[2];
    char    filter_tag[RSQC_MAX_FILTER_TAGS][2];   /* params.n_filter_tags names; {0,0} never matches          */
    uint64_t file_index_base;          /* index of the stream's first record in the whole file                 */
    int32_t  pipelined;                /* 1: a call returns once its kernels are enqueued; the NEXT call (or rsqc_decode_end)
                                          completes it, so `out` then describes the call before -- the next chunk of the
                                          file crosses PCIe beside this call's kernels                            */
    int32_t  reserved;
    uint64_t reserve_inflated_bytes;   /* 0, or: size the device buffers at rsqc_decode_begin for calls of up to this many
                                          inflated bytes (they grow on demand otherwise, which costs a re-allocation
                                          of every window buffer each time a larger call arrives)                 */
} rsqc_decode_params;
typedef struct rsqc_decode_window {    /* what one rsqc_decode_submit decoded (arrays owned by the context, valid until its next decode call) */
    uint64_t n_records;
    uint32_t n_runs;                   /* runs of consecutive records on one reference sequence ...            */
    const int32_t *run_tid;            /* ... their RefIDs, in file order (the batch's contig segments)        */
    rsqc_batch device_batch;           /* NON-PIPELINED streams only (all zero otherwise): the decoded records as the boundary's SoA
                                          batch, every pointer a DEVICE pointer into the context's window buffers (valid until its
                                          next decode call; n == n_records): what the per-read kernels were given -- a host that
                                          wants the columns copies them out with hipMemcpy after rsqc_wait.  A pipelined stream
                                          returns window n from submit n + 1, which has already queued window n + 1's kernels
                                          into the same buffers: there is no moment at which the pointers could be read          */
} rsqc_decode_window;
typedef struct rsqc_decode_info {
    rsqc_decode_window last;           /* pipelined streams: what the last rsqc_decode_submit decoded            */
    uint64_t records;                  /* records decoded and submitted since rsqc_decode_begin                */
    int32_t  unsorted;                 /* a record starts before its predecessor on the same contig, judged on primary,
                                          mapped, QC-passed records: the reference's sort warning (src/RNASeQC.cpp:354) */
    int32_t  n_bad_refid;              /* records whose RefID the header does not define (:333-337) ...        */
    const char *const *bad_refid;      /* ... and the first 64 of their names (owned by the context)           */
} rsqc_decode_info;
RSQC_API int rsqc_decode_begin(rsqc_ctx *ctx, const rsqc_decode_params *p);
/* Inflates n_blocks consecutive blocks behind what the stream already holds, decodes every complete record and submits
 * them as one batch (asynchronous like rsqc_submit; `compressed` may be reused when the call returns).
 * skip_bytes: inflated bytes at the start of the first block that precede the first record (the BAM header, or the
 * in-block part of a virtual file offset) -- only in a call that starts on a record boundary.
 * limit_bytes: 0, or the inflated offset (counted from the first block of THIS call) at which the wanted range ends:
 * records that start there or later are left out.  out (may be NULL): what the call decoded.
 * Limits per call: 1.9 GiB of inflated data.  Size the calls by their BLOCKS, not by their file bytes: one wavefront inflates one
 * block and an MI355X holds 5 120 of them, so a call should carry several times that many (the command line: 1 GiB of inflated
 * data, about 16 000 blocks); a call with about as many blocks as wave slots takes as long as one block takes one wave.       */
RSQC_API int rsqc_decode_submit(rsqc_ctx *ctx, const void *compressed, uint64_t compressed_bytes,
                                const rsqc_bgzf_block *blocks, uint32_t n_blocks,
                                uint32_t skip_bytes, uint64_t limit_bytes, rsqc_decode_window *out);
/* End of the stream: RSQC_ERR_INPUT if an incomplete record is left over ("truncated BAM record").             */
RSQC_API int rsqc_decode_end(rsqc_ctx *ctx, rsqc_decode_info *out);

RSQC_API const char *rsqc_strerror(int code);
RSQC_API const char *rsqc_last_error(rsqc_ctx *ctx);
RSQC_API const char *rsqc_counter_name(int counter);   /* the reference's Metrics key  */
RSQC_API const char *rsqc_version(void);               /* "RNASeQC 2.4.3 ..." prefix kept for
                                                 python/rnaseqc/run.py:25     */

/* QNAME hash used at the boundary (host decoders must use exactly this).     */
RSQC_API uint64_t rsqc_qname_hash(const char *name, size_t len);
/* the second hash of a name (rsqc_batch.qhash2): a multiply-xorshift recurrence with its own constants + the murmur3 fmix32
 * finaliser over (state ^ length); it shares no structure with the FNV-1a of rsqc_qname_hash                                  */
RSQC_API uint32_t rsqc_qname_hash2(const char *name, size_t len);

#ifdef __cplusplus
}
#endif
#endif /* RNASEQC_AMD_H */
