// rsqc_decode.h -- device-side BAM decode: what one window of inflated data goes through between
// the BGZF inflate kernel (rsqc_inflate.h) and the per-read kernel, as per-thread bodies shared by the HIP kernels
// (rsqc_decode.hip) and the host emulation of the tests (tests/hostemu/decode_emu.cpp).
//
//   frame    one thread per 8 KiB segment: guess a record start, hop to the segment's end, count records and operations
//   chain    one workgroup: every guess must be where the walk of the segment before it landed (else that segment is
//            walked again from the true position), exclusive sums of the counts
//   offsets  one thread per segment: second walk, start and first-operation slot of every record
//   parse    one thread per record: rsqc_bamrec.h's bam_parse_record -> the batch's 16-byte half-records + operations;
//            contig change / wide / unrecognised-RefID marks; the unsorted-input test of src/RNASeQC.cpp:354-355
//   lists    one workgroup: the marks, in record order, become the batch's segment and wide tables
#pragma once

#include "rsqc_bamrec.h"
#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#endif

namespace rsqc {

constexpr uint32_t DEC_SEG_BYTES = 8192;
constexpr uint32_t DEC_MARK_SEG = 1, DEC_MARK_WIDE = 2, DEC_MARK_BADREF = 4, DEC_MARK_JUDGED = 8;
constexpr uint32_t DEC_ST_INFLATE = 1, DEC_ST_BAD_RECORD = 2;         // DecodeSummary.status bits (inflate: the detail is in inflate_fail)
constexpr uint32_t DEC_MAX_BAD = 64;

struct DevBgzfBlock { uint64_t in_off; uint32_t in_len, out_len, out_off, crc; };

struct DecodeSummary {                       // what the host reads back per window
    uint32_t n_rec, n_ops, n_seg, n_wide, n_bad, unsorted, consumed_end, status;
    uint32_t inflate_fail;                   // (block + 1) << 4 | InflateStatus of the first failing block
    uint32_t next_block;                     // work counter of the inflate kernel
    uint32_t bad_off[DEC_MAX_BAD];           // record offsets (window buffer) of the first records with an unrecognised RefID
};
struct DecodeCarry { int32_t have_q, q_tid, q_pos; };      // last judged record of the windows before this one

struct DecodeWindow {
    const uint8_t *buf; uint32_t start, end, n_seg;        // records live in buf[start, end), cut into n_seg segments
    BamSegment *seg; uint32_t *seg_rec0, *seg_ops0;
    uint32_t *rec_off, *ops_at; uint8_t *mark;
    rsqc_rec_core *core; rsqc_rec_aux *aux; uint32_t *qh2, *cigar;
    int32_t *seg_tid; uint64_t *seg_start;
    uint64_t *wide_index; int32_t *wide_nm, *wide_lq; uint32_t *wide_nc;
    DecodeSummary *sum; DecodeCarry *carry;
    BamTagSpec tags;
};

RSQC_BAM_FN void decode_segment_bounds(const DecodeWindow &W, uint32_t s, uint32_t &lo, uint32_t &hi) {
    lo = W.start + s * DEC_SEG_BYTES;
    hi = (W.end - lo < DEC_SEG_BYTES) ? W.end : lo + DEC_SEG_BYTES;
}
RSQC_BAM_FN void decode_frame_one(const DecodeWindow &W, uint32_t s) {
    uint32_t lo, hi;
    decode_segment_bounds(W, s, lo, hi);
    bam_frame_segment(W.buf, lo, hi, W.end, W.tags.n_ref, s == 0 ? W.start : BAM_SEG_NONE, W.seg[s]);
}
// is segment s's guess where the walk of segment s - 1 landed?
RSQC_BAM_FN bool decode_guess_confirmed(const DecodeWindow &W, uint32_t s) {
    const BamSegment &g = W.seg[s];
    if (g.bad || g.start == BAM_SEG_NONE) return false;
    return s == 0 ? true : g.start == W.seg[s - 1].land;
}
RSQC_BAM_FN void decode_offsets_one(const DecodeWindow &W, uint32_t s) {
    if (W.seg[s].n_rec) bam_segment_offsets(W.buf, W.seg[s], W.seg_rec0[s], W.seg_ops0[s], W.rec_off, W.ops_at);
}

// record i of n; returns DEC_ST_* bits to raise, sets `unsorted` when the record starts before its judged predecessor
RSQC_BAM_FN uint32_t decode_parse_one(const DecodeWindow &W, uint32_t i, bool &unsorted) {
    const uint8_t *rec = W.buf + W.rec_off[i];
    const uint32_t bs = bam_ld32(rec);
    BamRecOut ro;
    if (!bam_parse_record(rec, bs, W.tags, ro)) { W.mark[i] = 0; return DEC_ST_BAD_RECORD; }
    const uint32_t at = W.ops_at[i];
    ro.core.cigar_off = at;
    W.core[i] = ro.core; W.aux[i] = ro.aux; W.qh2[i] = ro.qhash2;
    const uint8_t *ops = rec + ro.ops_off;
    for (uint32_t k = 0; k < ro.n_ops; ++k) W.cigar[at + k] = bam_ld32(ops + 4u * k);
    uint32_t m = 0;
    if (i == 0 || (int32_t)bam_ld32(W.buf + W.rec_off[i - 1] + 4) != ro.tid) m |= DEC_MARK_SEG;
    if (ro.wide) m |= DEC_MARK_WIDE;
    if (bam_flag_judged(ro.aux.flag)) {
        if (ro.tid < 0 || ro.tid >= W.tags.n_ref) m |= DEC_MARK_BADREF;
        else {
            m |= DEC_MARK_JUDGED;
            // the judged record before this one: usually record i - 1
            bool found = false; int32_t ptid = 0, ppos = 0;
            for (uint32_t j = i; j-- > 0;) {
                const uint8_t *q = W.buf + W.rec_off[j] + 4;
                const int32_t t = (int32_t)bam_ld32(q);
                if (!bam_flag_judged(bam_ld16(q + 14)) || t < 0 || t >= W.tags.n_ref) continue;
                found = true; ptid = t; ppos = (int32_t)bam_ld32(q + 4);
                break;
            }
            if (!found && W.carry->have_q) { found = true; ptid = W.carry->q_tid; ppos = W.carry->q_pos; }
            if (found && ptid == ro.tid && ppos > ro.core.pos) unsorted = true;
        }
    }
    W.mark[i] = (uint8_t)m;
    return 0;
}

// ---- lists: records [lo, hi) of one thread ------------------------------------------------------------------
struct DecodeListCounts { uint32_t seg, wide, bad; int32_t last_judged; };
RSQC_BAM_FN void decode_lists_count(const DecodeWindow &W, uint32_t lo, uint32_t hi, DecodeListCounts &c) {
    c.seg = c.wide = c.bad = 0; c.last_judged = -1;
    for (uint32_t i = lo; i < hi; ++i) {
        const uint32_t m = W.mark[i];
        c.seg += m & 1u; c.wide += (m >> 1) & 1u; c.bad += (m >> 2) & 1u;
        if (m & DEC_MARK_JUDGED) c.last_judged = (int32_t)i;
    }
}
RSQC_BAM_FN void decode_lists_write(const DecodeWindow &W, uint32_t lo, uint32_t hi, DecodeListCounts base) {
    for (uint32_t i = lo; i < hi; ++i) {
        const uint32_t m = W.mark[i];
        if (!(m & 7u)) continue;
        const uint8_t *rec = W.buf + W.rec_off[i];
        if (m & DEC_MARK_SEG) { W.seg_tid[base.seg] = (int32_t)bam_ld32(rec + 4); W.seg_start[base.seg] = i; ++base.seg; }
        if (m & DEC_MARK_WIDE) {
            BamRecOut ro;
            (void)bam_parse_record(rec, bam_ld32(rec), W.tags, ro);
            W.wide_index[base.wide] = i; W.wide_nm[base.wide] = ro.nm; W.wide_lq[base.wide] = ro.l_seq; W.wide_nc[base.wide] = ro.n_ops;
            ++base.wide;
        }
        if (m & DEC_MARK_BADREF) { if (base.bad < DEC_MAX_BAD) W.sum->bad_off[base.bad] = W.rec_off[i]; ++base.bad; }
    }
}
// totals of the window; last = index of the last judged record or -1
RSQC_BAM_FN void decode_lists_finish(const DecodeWindow &W, uint32_t n, DecodeListCounts total) {
    W.seg_start[total.seg] = n;
    W.sum->n_seg = total.seg; W.sum->n_wide = total.wide; W.sum->n_bad = total.bad;
    if (total.last_judged >= 0) {
        const uint8_t *q = W.buf + W.rec_off[total.last_judged] + 4;
        W.carry->have_q = 1; W.carry->q_tid = (int32_t)bam_ld32(q); W.carry->q_pos = (int32_t)bam_ld32(q + 4);
    }
}

#if defined(__HIPCC__)
// rsqc_decode.hip
// one_pass: the form of the inflate kernel that commits short rounds in one vector pass (rsqc_inflate.h): pays on streams of short
// matches and literals (a real BAM, ~3x), costs 6 % on streams of long matches (ratio 8x): the caller decides from the call's ratio
void launch_bgzf_inflate(hipStream_t s, const uint8_t *in, const DevBgzfBlock *blk, uint32_t n_blk, uint8_t *out, DecodeSummary *sum, bool one_pass);
// scratch: DEC_SCRATCH_WORDS words of device memory for the scans' per-workgroup sums (window of at most 2 GiB)
constexpr size_t DEC_SCRATCH_WORDS = 16 + 2 * 1024 + 2048 + 4 * ((((size_t)1 << 31) / 36 + 8192) / 8192 + 8);
void launch_decode_window(hipStream_t s, const DecodeWindow &W, uint32_t *scratch);
#endif

}  // namespace rsqc
