// rsqc_kr.h -- KR, "Read Length" (src/RNASeQC.cpp:275-278) as a per-batch transfer function.  Device code only, included by
// rsqc_kernels.hip -- and, unmodified, by the host SIMT emulation of the tests (tests/hostemu/k1_emu.cpp).
#pragma once

namespace rsqc {

// ------------------------------------------------------------------ KR
// "Read Length" (src/RNASeQC.cpp:275-278): readLength = l_qseq of each record whose span exceeds the current value, in
// FILE order.  The kernel computes, per batch, the batch's TRANSFER FUNCTION state-in -> state-out, applies it to the
// context's state, and leaves the function in the batch's summary slot, so that a contig-sharded run can compose the
// batches of all shards in file order on the host (rsqc_shard_info) -- exact for any mix of read lengths.
//
// Shape of the function: entered with state r, the first record that fires is the first record whose span exceeds r,
// which is necessarily a PREFIX MAXIMUM of span over the batch's eligible records; from there the walk no longer
// depends on r.  With the prefix maxima p_1..p_P (spans s_1 < .. < s_P) and g_k = the final state of the walk that
// starts by firing p_k:   f(r) = g_k for the first k with s_k > r,  f(r) = r when no span exceeds r.
// When every eligible record of the batch has the same l_qseq L (the normal case) all g_k equal L:  P = 1,
// (s, g) = (max span, L) -- O(1).  Otherwise one wavefront replays the batch with ALL the walks at once (lane j carries
// the walk started by p_j and p_{64+j}), opening only the 64-record tiles whose max span can still change something.
#define RSQC_RL_MAXP 128
// A batch of several file ranges (rsqc_batch.seg_file_index, DevAccum::rl_seg): one wave PER SEGMENT, each leaves the function of its
// own record range [seg_start[s], seg_start[s + 1]) in summary slot s; the context's state is then composed on the host from all
// the slots in file order (rsqc_api.cpp), not here.
__global__ void __launch_bounds__(64)
read_length_kernel(DevAnnotation a, DevParams p, DevBatch b, DevAccum acc, uint32_t *summary) {
    const int l = lane_id();
    const bool per_seg = acc.rl_seg != nullptr;
    const uint32_t sg = per_seg ? blockIdx.x : 0u;
    const uint64_t rec_lo = per_seg ? b.seg_start[sg] : 0ull, rec_hi = per_seg ? b.seg_start[sg + 1] : b.n;
    if (per_seg && summary) summary += (size_t)sg * RSQC_RL_SUMMARY_WORDS;
    const uint32_t r_in = per_seg ? 0u : (uint32_t)*acc.read_length;
    const uint32_t Smax = per_seg ? acc.rl_seg[3u * sg] : acc.rl_stats[0], Lmin = per_seg ? acc.rl_seg[3u * sg + 1u] : acc.rl_stats[1],
                   Lmax = per_seg ? acc.rl_seg[3u * sg + 2u] : acc.rl_stats[2];
    uint32_t P = 0;
    uint32_t s0 = 0, s1 = 0, v0 = 0xFFFFFFFFu, v1 = 0xFFFFFFFFu;      // lane j: walks j and 64 + j (key span, state)
    bool too_many = false;
    if (Lmin == 0xFFFFFFFFu) {
        P = 0;                                                         // no eligible record: identity
    } else if (Lmin == Lmax) {
        P = 1;
        if (l == 0) { s0 = Smax; v0 = Lmin; }
    } else {
        uint32_t cur_max = 0u, vmin = 0xFFFFFFFFu;                     // prefix max of span so far; smallest live state
        const uint64_t n_tiles = (rec_hi + 63) / 64;
        for (uint64_t t0 = (rec_lo / 64) & ~63ull; t0 < n_tiles; t0 += 64) {
            const uint64_t t = t0 + l;
            uint32_t S = 0;
            if (t < n_tiles && t >= rec_lo / 64) S = acc.tile_span[t];
            uint64_t need = __ballot(S > (cur_max < vmin ? cur_max : vmin));
            while (need) {
                const int tl = __ffsll((unsigned long long)need) - 1;
                need &= need - 1;
                const uint32_t St = __shfl(S, tl, 64);
                if (!(St > (cur_max < vmin ? cur_max : vmin))) continue;   // the thresholds moved since the ballot
                const uint64_t i = (t0 + tl) * 64 + l;                 // replay the tile's 64 records in order
                uint32_t span = 0, lq = 0; bool elig = false;
                if (i >= rec_lo && i < rec_hi) {
                    Record rec;
                    if (load_record(b, i, per_seg ? sg : find_segment(b, (t0 + tl) * 64), rec)) {
                        RecordCounters rc; bool hq; uint32_t aligned; Blocks B;
                        gate_cascade(a, p, rec, rc, hq, aligned, B);
                        elig = rc.rl_eligible != 0; span = rc.rl_span; lq = (uint32_t)rc.rl_lqseq;
                    }
                }
                int from = 0;
                while (true) {
                    const uint32_t thr = cur_max < vmin ? cur_max : vmin;
                    const uint64_t m = __ballot(elig && l >= from && span > thr);
                    if (!m) break;
                    const int w = __ffsll((unsigned long long)m) - 1;
                    const uint32_t sp = __shfl(span, w, 64), q = __shfl(lq, w, 64);
                    if (sp > v0 && v0 != 0xFFFFFFFFu) v0 = q;          // every live walk sees the record
                    if (sp > v1 && v1 != 0xFFFFFFFFu) v1 = q;
                    if (sp > cur_max) {                                // a new prefix maximum starts a walk of its own
                        if (P < RSQC_RL_MAXP) {
                            if (l == (int)(P & 63u)) { if (P < 64) { s0 = sp; v0 = q; } else { s1 = sp; v1 = q; } }
                            ++P;
                        } else too_many = true;
                        cur_max = sp;
                    }
                    const uint32_t lm = v0 < v1 ? v0 : v1;
                    vmin = wave_min_u32(lm);
                    from = w + 1;
                }
                // the tiles of this group still to look at, against the thresholds as they are NOW: a walk's state can go DOWN (a spliced
                // record's span fires it and leaves its shorter l_qseq), and a later tile whose largest span lies between the new state and
                // the old one then counts again -- narrowing the first ballot's mask missed it (found by the round-6 junction test of
                // tests/test_k1_wave_emulation.py: reads of mixed lengths in fewer than 64 tiles)
                need = __ballot(S > (cur_max < vmin ? cur_max : vmin)) & (tl == 63 ? 0ull : ~((2ull << tl) - 1ull));
            }
        }
    }
    // the function applied to the incoming state: the first key above it decides
    uint32_t r = r_in;
    {
        const uint64_t m0 = __ballot(P > (uint32_t)l && s0 > r_in), m1 = __ballot(P > 64u + (uint32_t)l && s1 > r_in);
        if (m0) r = __shfl(v0, __ffsll((unsigned long long)m0) - 1, 64);
        else if (m1) r = __shfl(v1, __ffsll((unsigned long long)m1) - 1, 64);
    }
    if (summary) {                                                     // [0] P, [1] flags, then P x (span, state)
        if (l == 0) { summary[0] = P; summary[1] = too_many ? 1u : 0u; }
        if ((uint32_t)l < P) { summary[2 + 2 * l] = s0; summary[3 + 2 * l] = v0; }
        if (64u + (uint32_t)l < P) { summary[2 + 2 * (64 + l)] = s1; summary[3 + 2 * (64 + l)] = v1; }
    }
    if (l == 0 && too_many) atomicExch(acc.error, RSQC_ERR_CAPACITY);
    if (l == 0 && sg == 0u) {
        if (!per_seg) *acc.read_length = (int32_t)r;
        acc.rl_stats[0] = 0u; acc.rl_stats[1] = 0xFFFFFFFFu; acc.rl_stats[2] = 0u;    // ready for the next batch
        acc.ovf_count[1] += *acc.ovf_count;   // records the general kernel took since the last reset (rsqc_timing.slow_records)
        *acc.ovf_count = 0u;                  // (the slow kernel, this batch's only reader, ran before this kernel)
        if (acc.defer_total) *acc.defer_total = 0u;   // (likewise classify_long_kernel and the deferred list)
    }
}

}  // namespace rsqc
