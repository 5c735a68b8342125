// rsqc_bamrec.h -- BAM alignment records (SAM spec 4.2) to the boundary's batch columns, written once as
// __host__ __device__ code: the host reader (host/bam.cpp) and the device decode kernels (rsqc_decode.hip) run the
// same functions, tests/hostemu/decode_emu.cpp runs the device side's per-lane bodies on the CPU.
//
// What the reference takes from a record (through SeqLib::BamRecord, i.e. htslib's bam1_t):
//   core fields, QNAME                          src/RNASeQC.cpp:245-330, src/Expression.cpp:383-386
//   NM (GetIntTag), chimeric tag (readStringTag), --tag filters (GetTag)   src/RNASeQC.cpp:258,279,295,319-328,780-800
//   the CIGAR, put back from the CG tag by htslib when it has more than 65535 operations (bam_tag2cigar)
// SEQ and QUAL are never read.
#pragma once

#include <stdint.h>
#include "../../include/rnaseqc_amd.h"

#if defined(__HIPCC__)
#define RSQC_BAM_FN __host__ __device__ __forceinline__
#else
#define RSQC_BAM_FN inline
#endif

namespace rsqc {

// little-endian loads at any alignment
RSQC_BAM_FN uint32_t bam_ld16(const uint8_t *p) { uint16_t v; __builtin_memcpy(&v, p, 2); return v; }
RSQC_BAM_FN uint32_t bam_ld32(const uint8_t *p) { uint32_t v; __builtin_memcpy(&v, p, 4); return v; }

RSQC_BAM_FN uint64_t bam_qname_hash(const uint8_t *name, uint32_t len) {
    uint64_t h = 0xCBF29CE484222325ull;                    // FNV-1a 64, then fmix64 (== rsqc_qname_hash)
    for (uint32_t i = 0; i < len; ++i) { h ^= name[i]; h *= 0x100000001B3ull; }
    h ^= h >> 33; h *= 0xFF51AFD7ED558CCDull; h ^= h >> 33; h *= 0xC4CEB9FE1A85EC53ull; h ^= h >> 33;
    return h;
}

// the SECOND name hash (rsqc_qname_hash2, rsqc_batch.qhash2): per byte  h = (h + b) * 0xCC9E2D51; h ^= h >> 15;  then the murmur3
// fmix32 of (h ^ length).  Its own recurrence and constants: nothing of FNV-1a's structure, so a pair of names crafted to
// collide in the first hash does not collide here
constexpr uint32_t BAM_QH2_SEED = 0x2F0B4A87u;
RSQC_BAM_FN uint32_t bam_qh2_step(uint32_t h, uint32_t b) { h = (h + b) * 0xCC9E2D51u; return h ^ (h >> 15); }
RSQC_BAM_FN uint32_t bam_qh2_finish(uint32_t h, uint32_t len) {
    h ^= len; h ^= h >> 16; h *= 0x85EBCA6Bu; h ^= h >> 13; h *= 0xC2B2AE35u; h ^= h >> 16;
    return h;
}
RSQC_BAM_FN uint32_t bam_qname_hash2(const uint8_t *name, uint32_t len) {
    uint32_t h = BAM_QH2_SEED;
    for (uint32_t i = 0; i < len; ++i) h = bam_qh2_step(h, name[i]);
    return bam_qh2_finish(h, len);
}

struct BamTagSpec {
    int32_t n_ref;
    uint8_t have_ch, ch0, ch1, n_filter;
    uint8_t f0[RSQC_MAX_FILTER_TAGS], f1[RSQC_MAX_FILTER_TAGS];
};

// structural check used to GUESS a record start inside inflated data (a guess is always verified by the true chain)
RSQC_BAM_FN bool bam_plausible(const uint8_t *buf, uint64_t p, uint64_t end, int32_t n_ref) {
    if (p + 36 > end) return false;
    const uint32_t bs = bam_ld32(buf + p);
    if (bs < 32 || bs > (1u << 26)) return false;
    const uint8_t *r = buf + p + 4;
    const int32_t tid = (int32_t)bam_ld32(r), pos = (int32_t)bam_ld32(r + 4), mtid = (int32_t)bam_ld32(r + 20), mpos = (int32_t)bam_ld32(r + 24);
    if (tid < -1 || tid >= n_ref || mtid < -1 || mtid >= n_ref || pos < -1 || mpos < -1) return false;
    const uint32_t l_name = r[8], n_cig = bam_ld16(r + 12);
    const int32_t l_seq = (int32_t)bam_ld32(r + 16);
    if (l_name == 0 || l_seq < 0) return false;
    const uint64_t fixed = 32ull + l_name + 4ull * n_cig + (uint64_t)(l_seq + 1) / 2 + (uint64_t)l_seq;
    if (fixed > bs) return false;
    if (p + 4 + 32 + l_name <= end && r[32 + l_name - 1] != 0) return false;     // QNAME is NUL-terminated
    return true;
}

// size in bytes of the value of an aux field of type `type` at v (end_r = end of the record); ~0u = malformed tail
RSQC_BAM_FN uint32_t bam_aux_size(char type, const uint8_t *v, const uint8_t *end_r) {
    switch (type) {
    case 'A': case 'c': case 'C': return 1;
    case 's': case 'S': return 2;
    case 'i': case 'I': case 'f': return 4;
    case 'd': return 8;                                                             // (htslib skips 8 bytes)
    case 'Z': case 'H': { uint32_t n = 0; const uint32_t room = (uint32_t)(end_r - v); while (n < room && v[n]) ++n; return n + 1; }
    case 'B': {
        if (v + 5 > end_r) return (uint32_t)(end_r - v);
        const char st = (char)v[0]; const uint32_t cnt = bam_ld32(v + 1);
        const uint64_t es = (st == 'c' || st == 'C') ? 1 : (st == 's' || st == 'S') ? 2 : 4;
        const uint64_t n = 5 + es * (uint64_t)cnt;
        return n > 0xFFFFFFF0ull ? 0xFFFFFFF0u : (uint32_t)n;
    }
    default: return (uint32_t)(end_r - v);
    }
}
// SeqLib::BamRecord::GetIntTag: an integer-typed aux field (htslib bam_aux2i)
RSQC_BAM_FN bool bam_aux_int(const uint8_t *v, char type, int32_t &out) {
    switch (type) {
    case 'c': out = (int8_t)v[0]; return true;
    case 'C': out = v[0]; return true;
    case 's': out = (int16_t)bam_ld16(v); return true;
    case 'S': out = (int32_t)bam_ld16(v); return true;
    case 'i': case 'I': out = (int32_t)bam_ld32(v); return true;
    default: return false;
    }
}

// The operations of a record as the reader hands them on.  A CIGAR of more than 65535 operations is stored in the CG:B,I
// tag behind the placeholder <l_seq>S<ref_len>N (SAM spec 4.2.2) and htslib puts it back when it reads the record, so
// the reference sees the real one.  rec points at the block_size field.  Returns the byte offset of the first operation
// from rec; false = the fixed part does not fit the record ("bad BAM record").
RSQC_BAM_FN bool bam_record_ops(const uint8_t *rec, uint32_t block_size, uint32_t &n_ops, uint32_t &ops_off) {
    const uint8_t *r = rec + 4;
    const uint32_t l_name = r[8], n_cigar = bam_ld16(r + 12);
    const int32_t l_seq = (int32_t)bam_ld32(r + 16);
    const uint64_t cig = 32ull + l_name, cig_end = cig + 4ull * n_cigar;
    if (cig_end > block_size) return false;
    n_ops = n_cigar; ops_off = 4u + (uint32_t)cig;
    if (n_cigar != 2) return true;
    const uint32_t op0 = bam_ld32(r + cig), op1 = bam_ld32(r + cig + 4);
    if ((op0 & 0xf) != 4 || (int64_t)(op0 >> 4) != (int64_t)l_seq || (op1 & 0xf) != 3) return true;
    const uint8_t *end_r = r + block_size;
    const uint64_t seq = (uint64_t)((l_seq < 0 ? 0 : l_seq + 1) / 2) + (uint64_t)(l_seq < 0 ? 0 : l_seq);
    if (cig_end + seq > block_size) return true;
    uint32_t cg_n = 0, cg_off = 0;
    for (const uint8_t *q = r + cig_end + seq; q + 3 <= end_r;) {
        const char t0 = (char)q[0], t1 = (char)q[1], type = (char)q[2];
        const uint8_t *v = q + 3;
        const uint32_t vlen = bam_aux_size(type, v, end_r);
        if ((uint64_t)(end_r - v) < vlen) break;
        if (t0 == 'C' && t1 == 'G' && type == 'B' && vlen >= 5 && v[0] == 'I') { cg_n = bam_ld32(v + 1); cg_off = (uint32_t)(v + 5 - rec); }   // (the last CG field counts)
        q = v + vlen;
    }
    if (cg_n > 0) { n_ops = cg_n; ops_off = cg_off; }
    return true;
}

struct BamRecOut {
    rsqc_rec_core core;          // cigar_off is the caller's
    rsqc_rec_aux aux;
    int32_t tid;
    int32_t nm, l_seq;           // full-width values (the wide table's when an escape is set)
    uint32_t n_ops, ops_off;     // operations and where they start, from rec
    uint32_t wide;
    uint32_t qname_len;          // bytes before the NUL
    uint32_t qhash2;             // rsqc_qname_hash2 of the name (the batch's qhash2 column)
};

// one record; false = malformed ("bad BAM record")
RSQC_BAM_FN bool bam_parse_record(const uint8_t *rec, uint32_t block_size, const BamTagSpec &tags, BamRecOut &o) {
    if (block_size < 32) return false;
    const uint8_t *r = rec + 4;
    const int32_t tid = (int32_t)bam_ld32(r), pos = (int32_t)bam_ld32(r + 4);
    const uint32_t l_name = r[8], mapq = r[9];
    const uint32_t n_cigar = bam_ld16(r + 12), flag = bam_ld16(r + 14);
    const int32_t l_seq = (int32_t)bam_ld32(r + 16), mtid = (int32_t)bam_ld32(r + 20), mpos = (int32_t)bam_ld32(r + 24), isize = (int32_t)bam_ld32(r + 28);
    const uint8_t *end_r = r + block_size;
    if (!bam_record_ops(rec, block_size, o.n_ops, o.ops_off)) return false;
    // QNAME up to its NUL, hashed on the way (== bam_qname_hash of those bytes).  Eight bytes per load: on the device a lane's
    // byte loads are a memory round trip each, and the name is the longest run of them in a record
    uint32_t qlen = 0;
    uint64_t qh = 0xCBF29CE484222325ull;
    uint32_t qh2 = BAM_QH2_SEED;
    {
        const uint8_t *qn = r + 32;
        const uint32_t room = (32ull + l_name <= block_size) ? l_name : 0u;
        bool open_ = true;
        while (open_ && qlen + 8u <= room) {
            uint64_t w; __builtin_memcpy(&w, qn + qlen, 8);
            for (uint32_t k = 0; k < 8u; ++k, w >>= 8) {
                const uint32_t b = (uint32_t)w & 0xFFu;
                if (!b) { open_ = false; break; }
                qh ^= b; qh *= 0x100000001B3ull; qh2 = bam_qh2_step(qh2, b); ++qlen;
            }
        }
        while (open_ && qlen < room) {
            const uint32_t b = qn[qlen];
            if (!b) break;
            qh ^= b; qh *= 0x100000001B3ull; qh2 = bam_qh2_step(qh2, b); ++qlen;
        }
        qh ^= qh >> 33; qh *= 0xFF51AFD7ED558CCDull; qh ^= qh >> 33; qh *= 0xC4CEB9FE1A85EC53ull; qh ^= qh >> 33;
    }
    o.qname_len = qlen;
    o.qhash2 = bam_qh2_finish(qh2, qlen);
    o.core.pos = pos; o.core.mpos = mpos; o.core.isize = isize; o.core.cigar_off = 0;
    o.aux.qhash = qh;
    o.aux.flag = (uint16_t)flag; o.aux.mapq = (uint8_t)mapq;
    o.tid = tid;
    uint32_t tagbits = (tid == mtid) ? RSQC_TB_MTID_SAME : 0;
    int32_t nm = 0;
    const uint64_t aux_at = 32ull + l_name + 4ull * n_cigar + (uint64_t)((l_seq < 0 ? 0 : l_seq + 1) / 2) + (uint64_t)(l_seq < 0 ? 0 : l_seq);
    if (aux_at <= block_size) {
        for (const uint8_t *q = r + aux_at; q + 3 <= end_r;) {
            const char t0 = (char)q[0], t1 = (char)q[1], type = (char)q[2];
            const uint8_t *v = q + 3;
            const uint32_t vlen = bam_aux_size(type, v, end_r);
            if ((uint64_t)(end_r - v) < vlen) break;                                // malformed tail: stop scanning
            if (t0 == 'N' && t1 == 'M') { int32_t x; if (bam_aux_int(v, type, x)) { nm = x; tagbits |= RSQC_TB_HAS_NM; } }
            if (tags.have_ch && t0 == (char)tags.ch0 && t1 == (char)tags.ch1) {     // readStringTag, src/RNASeQC.cpp:780-800
                if (type == 'Z' || (type == 'A' && v[0] != 0)) tagbits |= RSQC_TB_HAS_CH;
            }
            for (uint32_t fi = 0; fi < tags.n_filter; ++fi)                         // GetTag: Z, integer or float
                if (t0 == (char)tags.f0[fi] && t1 == (char)tags.f1[fi]) {
                    int32_t x;
                    if (type == 'Z' || type == 'f' || bam_aux_int(v, type, x)) tagbits |= (uint32_t)RSQC_TB_FILTER0 << fi;
                }
            q = v + vlen;
        }
    }
    const bool wide = l_seq >= RSQC_LQSEQ_ESCAPE || l_seq < 0 || nm >= RSQC_NM_ESCAPE || nm < 0 || o.n_ops >= RSQC_NCIGAR_ESCAPE;
    o.aux.l_qseq = (l_seq >= RSQC_LQSEQ_ESCAPE || l_seq < 0) ? (uint16_t)RSQC_LQSEQ_ESCAPE : (uint16_t)l_seq;
    o.aux.nm = (nm >= RSQC_NM_ESCAPE || nm < 0) ? (uint8_t)RSQC_NM_ESCAPE : (uint8_t)nm;
    o.aux.n_cigar = o.n_ops >= RSQC_NCIGAR_ESCAPE ? (uint8_t)RSQC_NCIGAR_ESCAPE : (uint8_t)o.n_ops;
    o.aux.tagbits = (uint8_t)tagbits;
    o.nm = nm; o.l_seq = l_seq; o.wide = wide ? 1u : 0u;
    return true;
}

// what the reference's loop judges its stderr diagnostics on: primary, mapped, QC-passed records (src/RNASeQC.cpp:333-337,354-355)
RSQC_BAM_FN bool bam_flag_judged(uint32_t flag) { return !(flag & (RSQC_FSECONDARY | RSQC_FQCFAIL | RSQC_FSUPP | RSQC_FUNMAP)); }

// ---- framing of a window of inflated data (device decode) -------------------------------------------------------
// A record can only be located by hopping from the previous one.  The window is cut into segments; every segment is
// walked in parallel from a GUESSED record start (the first offset where two consecutive records pass bam_plausible),
// and the guesses are then verified against the true chain: segment s is accepted only if the walk of the segment
// before it lands exactly on its guess; otherwise it is walked again from the true position.  A wrong guess costs
// time, never correctness.  All offsets are relative to the window's buffer.
struct BamSegment {
    uint32_t start;      // first record start at or after the segment's beginning (guess, or the truth once verified); NONE = none found
    uint32_t land;       // where the walk stopped: the first record start at or after the segment's end, or the first record that does not fit the window
    uint32_t n_rec, n_ops;
    uint32_t bad;        // the walk met a record that cannot be one (a wrong guess -- or, from a true start, a corrupt file)
};
constexpr uint32_t BAM_SEG_NONE = 0xFFFFFFFFu;

// walk from p while records start before hi and fit [.., end); counts records and operations
RSQC_BAM_FN void bam_walk(const uint8_t *buf, uint32_t p, uint32_t hi, uint32_t end, BamSegment &s) {
    uint32_t n = 0, ops = 0; s.bad = 0;
    uint64_t q = p;
    while (q < hi && q + 4 <= end) {
        const uint32_t bs = bam_ld32(buf + q);
        if (bs < 32) { s.bad = 1; break; }
        if (q + 4 + (uint64_t)bs > end) break;
        uint32_t k, off;
        if (!bam_record_ops(buf + q, bs, k, off)) { s.bad = 1; break; }
        ++n; ops += k;
        q += 4 + (uint64_t)bs;
    }
    s.land = (uint32_t)q; s.n_rec = n; s.n_ops = ops;
}
// segment [lo, hi) of a window that ends at `end`; true_start != NONE for the window's first segment
RSQC_BAM_FN void bam_frame_segment(const uint8_t *buf, uint32_t lo, uint32_t hi, uint32_t end, int32_t n_ref, uint32_t true_start, BamSegment &s) {
    uint32_t p = true_start;
    if (p == BAM_SEG_NONE) {
        // three plausible records in a row (low-entropy SEQ / QUAL bytes pass two now and then).  A candidate whose record reaches
        // past the window's end cannot be checked against a second one: it is kept as a LAST RESORT only.  (Taken at once, as in
        // round 3, the two bytes in front of a true record start -- the tail of an aux field + the low half of block_size, read as
        // a block_size of a few MB -- won in every segment of the window's last megabytes on RefID 0: hundreds of wrong guesses
        // per call, each repaired by ONE thread: 8 % of the decode time on the realistic-entropy file, profiles/r4_decode_guess.txt.)
        uint32_t last_resort = BAM_SEG_NONE;
        for (uint32_t c = lo; c < hi; ++c) {
            if (!bam_plausible(buf, c, end, n_ref)) continue;
            const uint64_t q = (uint64_t)c + 4 + bam_ld32(buf + c);
            if (q + 36 > end) { if (last_resort == BAM_SEG_NONE) last_resort = c; continue; }
            if (!bam_plausible(buf, q, end, n_ref)) continue;
            const uint64_t r = q + 4 + bam_ld32(buf + q);
            if (r + 36 <= end ? bam_plausible(buf, r, end, n_ref) : true) { p = c; break; }
        }
        if (p == BAM_SEG_NONE) p = last_resort;
    }
    s.start = p;
    if (p == BAM_SEG_NONE) { s.land = hi; s.n_rec = s.n_ops = 0; s.bad = 0; return; }
    bam_walk(buf, p, hi, end, s);
}
// exact, sequential: replaces every segment whose guess the true chain does not confirm.  Returns the first byte that
// is not part of a complete record (the window's unconsumed tail starts there); bad = a true walk met a corrupt record.
RSQC_BAM_FN uint32_t bam_verify_chain(const uint8_t *buf, BamSegment *seg, uint32_t n_seg, uint32_t base, uint32_t seg_bytes, uint32_t end, uint32_t &bad) {
    uint32_t truth = seg[0].start;
    bad = 0;
    uint32_t s = 0;
    for (; s < n_seg; ++s) {
        const uint32_t lo = base + s * seg_bytes, hi = (end - lo < seg_bytes) ? end : lo + seg_bytes;
        if (truth >= hi) { seg[s].start = truth; seg[s].land = truth; seg[s].n_rec = seg[s].n_ops = 0; seg[s].bad = 0; continue; }   // a long record spans the segment
        if (seg[s].start != truth || seg[s].bad) { seg[s].start = truth; bam_walk(buf, truth, hi, end, seg[s]); }
        if (seg[s].bad) { bad = 1; break; }
        truth = seg[s].land;
        if (truth < hi) { ++s; break; }                                      // an incomplete record: the window ends here
    }
    for (; s < n_seg; ++s) { seg[s].start = truth; seg[s].land = truth; seg[s].n_rec = seg[s].n_ops = 0; }
    return truth;
}
// the same result as bam_verify_chain when only a few guesses are wrong: `list` holds, ascending, the segments whose guess is
// not where the walk of the segment before them landed.  Each is walked again from the truth, and so is every segment
// after it for as long as the corrected landing differs from that segment's start.  O(wrong guesses), not O(segments).
// (Re-walking ALL unconfirmed segments in parallel does not converge: a segment with a correct guess behind a wrong one gets
// re-walked from the wrong landing, and the defect moves forward one segment per round.)
RSQC_BAM_FN uint32_t bam_repair_listed(const uint8_t *buf, BamSegment *seg, uint32_t n_seg, uint32_t base, uint32_t seg_bytes, uint32_t end,
                                       const uint32_t *list, uint32_t n_list, uint32_t &bad) {
    bad = seg[0].bad;
    uint32_t done = 0;                                                   // segments below this are settled
    for (uint32_t k = 0; k < n_list && !bad; ++k) {
        uint32_t s = list[k];
        if (s < done || s == 0) continue;
        for (; s < n_seg; ++s) {
            const uint32_t truth = seg[s - 1].land;
            if (seg[s].start == truth && !seg[s].bad) break;
            const uint32_t lo = base + s * seg_bytes, hi = (end - lo < seg_bytes) ? end : lo + seg_bytes;
            seg[s].start = truth;
            if (truth >= hi) { seg[s].land = truth; seg[s].n_rec = seg[s].n_ops = 0; seg[s].bad = 0; }    // a long record spans the segment
            else bam_walk(buf, truth, hi, end, seg[s]);
            if (seg[s].bad) { bad = 1; break; }
        }
        done = s;
    }
    return n_seg ? seg[n_seg - 1].land : base;
}
// second walk of a verified segment: where every record starts and where its operations go
RSQC_BAM_FN void bam_segment_offsets(const uint8_t *buf, const BamSegment &s, uint32_t rec0, uint32_t ops0, uint32_t *rec_off, uint32_t *ops_at) {
    uint64_t q = s.start; uint32_t ops = ops0;
    for (uint32_t k = 0; k < s.n_rec; ++k) {
        const uint32_t bs = bam_ld32(buf + q);
        uint32_t n, off;
        (void)bam_record_ops(buf + q, bs, n, off);
        rec_off[rec0 + k] = (uint32_t)q; ops_at[rec0 + k] = ops;
        ops += n;
        q += 4 + (uint64_t)bs;
    }
}

}  // namespace rsqc
