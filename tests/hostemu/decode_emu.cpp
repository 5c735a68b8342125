// decode_emu.cpp -- TEST HARNESS ONLY (never loaded by the product).
//
// Compiles the product's device-decode cores (rnaseqc_amd/csrc/rsqc_inflate.h: the DEFLATE decoder one wavefront runs
// per BGZF block; rsqc_bamrec.h: BAM record framing and parsing) with g++ as a wave of ONE lane, so that what the HIP
// kernels execute can be diffed against zlib and against the host BAM reader in the GPU-less build container.
#include <cstdio>
#include <cstdlib>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <vector>

#include "../../rnaseqc_amd/csrc/rsqc_inflate.h"

using namespace rsqc;

extern "C" __attribute__((visibility("default")))
int emu_inflate(const uint8_t *in, uint32_t in_len, uint8_t *out, uint32_t out_len, uint32_t crc32) {
    static InflateScratch S;
    inflate_crc_init(S);
    // the decoder may look 16 bytes past the payload
    std::vector<uint8_t> padded((size_t)in_len + 32, 0);
    memcpy(padded.data() + 3, in, in_len);                       // (an odd alignment on purpose)
    return inflate_block(S, padded.data() + 3, in_len, out, out_len, crc32);
}

// the 64-lane form of inflate_flush's CRC step (pieces, right alignment, six-level tree), with the lanes as an array:
// checks the arithmetic the device build uses where the one-lane host build takes a shortcut
extern "C" __attribute__((visibility("default")))
uint32_t emu_crc_wave64(const uint8_t *data, uint32_t n, uint32_t crc_before) {
    static InflateScratch S;
    inflate_crc_init(S);
    const uint32_t W = 64, plen = (n + W - 1u) / W, pad = plen * W - n;
    uint32_t r[64];
    for (uint32_t l = 0; l < W; ++l) {
        uint32_t x = 0;
        for (uint32_t k = 0; k < plen; ++k) {
            const uint32_t v = l * plen + k;
            if (v >= pad) x = S.crc_tab[(x ^ data[v - pad]) & 0xFFu] ^ (x >> 8);
        }
        r[l] = x;
    }
    uint32_t m = crc_xpow(8ull * plen);
    for (uint32_t d = 1; d < 64u; d <<= 1) {
        uint32_t nr[64];
        for (uint32_t l = 0; l < W; ++l) { const uint32_t hi = l + d < W ? r[l + d] : r[l]; nr[l] = crc_mulmod(r[l], m) ^ hi; }
        memcpy(r, nr, sizeof r);
        m = crc_mulmod(m, m);
    }
    return crc_mulmod(crc_before, crc_xpow(8ull * n)) ^ r[0];
}

// ---- one window of inflated data through frame / chain / offsets / parse / lists, the kernels' per-thread bodies run
// serially.  `threads` = how many threads the lists step pretends to have (chunk boundaries, exclusive sums);
// perturb != 0 moves some guesses, which the chain step has to repair.
#include "../../rnaseqc_amd/csrc/rsqc_decode.h"

static unsigned long long g_unconfirmed_total = 0;
// segments whose guess the chain did not confirm, summed over the emu_decode_window calls since the last reset (reset != 0 clears)
extern "C" __attribute__((visibility("default")))
unsigned long long emu_decode_unconfirmed(int reset) { const unsigned long long v = g_unconfirmed_total; if (reset) g_unconfirmed_total = 0; return v; }

extern "C" __attribute__((visibility("default")))
int emu_decode_window(const uint8_t *buf, uint32_t start, uint32_t end, const BamTagSpec *tags, int threads, int32_t *carry3, int perturb,
                      rsqc_rec_core *core, rsqc_rec_aux *aux, uint32_t *qh2, uint32_t *cigar, int32_t *seg_tid, uint64_t *seg_start,
                      uint64_t *wide_index, int32_t *wide_nm, int32_t *wide_lq, uint32_t *wide_nc, uint32_t *summary /* 8 + 64 */) {
    DecodeWindow W{};
    W.buf = buf; W.start = start; W.end = end;
    W.n_seg = end > start ? (end - start + DEC_SEG_BYTES - 1) / DEC_SEG_BYTES : 0;
    std::vector<BamSegment> seg(W.n_seg + 1);
    std::vector<uint32_t> rec0(W.n_seg + 1), ops0(W.n_seg + 1);
    W.seg = seg.data(); W.seg_rec0 = rec0.data(); W.seg_ops0 = ops0.data();
    DecodeSummary sum{}; DecodeCarry carry{carry3[0], carry3[1], carry3[2]};
    W.sum = &sum; W.carry = &carry; W.tags = *tags;
    W.core = core; W.aux = aux; W.qh2 = qh2; W.cigar = cigar; W.seg_tid = seg_tid; W.seg_start = seg_start;
    W.wide_index = wide_index; W.wide_nm = wide_nm; W.wide_lq = wide_lq; W.wide_nc = wide_nc;
    for (uint32_t s = 0; s < W.n_seg; ++s) decode_frame_one(W, s);
    if (perturb == 1) for (uint32_t s = 1; s < W.n_seg; s += 3) { seg[s].start += (s % 2) ? 1 : 40; seg[s].n_rec += 1; }
    if (perturb == 2) {                                            // a third of the guesses somewhere else in their segment, walked from there
        uint32_t x = 12345u;
        for (uint32_t s = 1; s < W.n_seg; ++s) {
            x = x * 1664525u + 1013904223u;
            if ((x >> 8) % 3u) continue;
            uint32_t lo, hi; decode_segment_bounds(W, s, lo, hi);
            seg[s].start = lo + (x >> 12) % (hi - lo);
            bam_walk(buf, seg[s].start, hi, end, seg[s]);
        }
    }
    if (perturb == 3) for (uint32_t s = 2; s < W.n_seg && s < 14; ++s) { seg[s].start += 7; seg[s].land += 3; }      // a run of consecutive wrong guesses
    if (perturb == 4) for (uint32_t s = 1; s < W.n_seg; ++s) { seg[s].start = BAM_SEG_NONE; seg[s].n_rec = 0; seg[s].n_ops = 0; }   // no guess anywhere
    if (perturb == 5) for (uint32_t s = 1; s < W.n_seg; s += 2) seg[s].bad = 1;                                   // walks that ran into garbage
    // the chain step: the segments whose guess is not confirmed are repaired in order (bam_repair_listed); the plain sequential
    // walk (what the device falls back to when the list overflows) must give the same segments
    std::vector<uint32_t> list;
    for (uint32_t s = 0; s < W.n_seg; ++s) if (!decode_guess_confirmed(W, s)) list.push_back(s);
    uint32_t consumed = start, bad = 0;
    if (!perturb) g_unconfirmed_total += list.size();
    if (getenv("DEC_EMU_STATS")) {
        fprintf(stderr, "[decode_emu] %u segments, %zu unconfirmed:", W.n_seg, list.size());
        for (size_t i = 0; i < list.size() && i < 12; ++i) { const uint32_t sg = list[i]; fprintf(stderr, " s%u(start %u, prev land %u, prev bad %u, bad %u)", sg, seg[sg].start, sg ? seg[sg - 1].land : 0u, sg ? seg[sg - 1].bad : 0u, seg[sg].bad); }
        fprintf(stderr, "\n");
    }
    if (W.n_seg) {
        std::vector<BamSegment> ref = seg;
        uint32_t bad_ref = 0;
        const uint32_t c_ref = bam_verify_chain(buf, ref.data(), W.n_seg, start, DEC_SEG_BYTES, end, bad_ref);
        consumed = list.empty() ? seg[W.n_seg - 1].land : bam_repair_listed(buf, seg.data(), W.n_seg, start, DEC_SEG_BYTES, end, list.data(), (uint32_t)list.size(), bad);
        if (bad != bad_ref) return 101;
        if (!bad) {
            if (consumed != c_ref) return 102;
            for (uint32_t s = 0; s < W.n_seg; ++s)
                if (seg[s].n_rec != ref[s].n_rec || seg[s].n_ops != ref[s].n_ops || (seg[s].n_rec && seg[s].start != ref[s].start)) return 103;
        }
    }
    uint32_t n = 0, ops = 0;
    for (uint32_t s = 0; s < W.n_seg; ++s) { rec0[s] = n; ops0[s] = ops; n += seg[s].n_rec; ops += seg[s].n_ops; }
    sum.n_rec = n; sum.n_ops = ops; sum.consumed_end = consumed; sum.status = bad ? DEC_ST_BAD_RECORD : 0;
    std::vector<uint32_t> rec_off(n + 1), ops_at(n + 1); std::vector<uint8_t> mark(n + 1);
    W.rec_off = rec_off.data(); W.ops_at = ops_at.data(); W.mark = mark.data();
    if (!bad) {
        for (uint32_t s = 0; s < W.n_seg; ++s) decode_offsets_one(W, s);
        for (uint32_t i = 0; i < n; ++i) { bool u = false; sum.status |= decode_parse_one(W, i, u); if (u) sum.unsorted = 1; }
        const uint32_t T = (uint32_t)threads, per = (n + T - 1) / T;
        std::vector<DecodeListCounts> cnt(T), base(T);
        DecodeListCounts run{0, 0, 0, -1};
        for (uint32_t t = 0; t < T; ++t) {
            const uint32_t lo = std::min(n, t * per), hi = std::min(n, lo + per);
            decode_lists_count(W, lo, hi, cnt[t]);
            base[t] = run;
            run.seg += cnt[t].seg; run.wide += cnt[t].wide; run.bad += cnt[t].bad;
            if (cnt[t].last_judged >= 0) run.last_judged = cnt[t].last_judged;
        }
        for (uint32_t t = 0; t < T; ++t) { const uint32_t lo = std::min(n, t * per), hi = std::min(n, lo + per); decode_lists_write(W, lo, hi, base[t]); }
        decode_lists_finish(W, n, run);
    }
    memcpy(summary, &sum, 8 * 4);
    memcpy(summary + 8, sum.bad_off, sizeof sum.bad_off);
    carry3[0] = carry.have_q; carry3[1] = carry.q_tid; carry3[2] = carry.q_pos;
    return 0;
}
