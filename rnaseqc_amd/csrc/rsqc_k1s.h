// rsqc_k1s.h -- K1s, the general per-record kernel: the records classify_ei_kernel leaves on its overflow list (more than four
// blocks, more than two genes per block set, an interval under more than two exons, long-CIGAR stragglers of boundary tiles)
// and, under --legacy, every record (src/Expression.cpp:129-304,308-458).  Device code only, included by rsqc_kernels.hip --
// and, unmodified, by the host SIMT emulation of the tests (tests/hostemu/k1_emu.cpp).
#pragma once

namespace rsqc {

// accumulator used by the general (slow-path) code when it re-walks a CIGAR
struct DirectAcc {
    double *exon_acc; uint32_t *cov_diff; const uint32_t *ex_id;
    __device__ __forceinline__ void exon_add(uint32_t row, double frac) { atomicAdd(&exon_acc[ex_id[row]], frac); }
    __device__ __forceinline__ void cov_range(uint32_t cidx, uint32_t len) {
        if (len == 0) return;
        atomicAdd(&cov_diff[cidx], 1u);
        atomicAdd(&cov_diff[cidx + len], 0xFFFFFFFFu);            // lands on the next exon / the gene's pad slot
    }
};

// record i of the batch; `seg` is a wave-uniform hint for the contig segment
__device__ __forceinline__ bool load_record(const DevBatch &b, uint64_t i, uint32_t seg, Record &r) {
    const int4 cv = *reinterpret_cast<const int4 *>(&b.core[i]);          // global_load_dwordx4
    const int4 av = *reinterpret_cast<const int4 *>(&b.aux[i]);
    r.pos = cv.x; r.mpos = cv.y; r.isize = cv.z;
    r.cigar = b.cigar + (uint32_t)cv.w;
    r.qhash = (uint64_t)(uint32_t)av.x | ((uint64_t)(uint32_t)av.y << 32);
    r.flag = (uint32_t)av.z & 0xFFFFu; r.l_qseq = (int32_t)((uint32_t)av.z >> 16);
    r.mapq = (uint32_t)av.w & 0xFFu; r.nm = (int32_t)(((uint32_t)av.w >> 8) & 0xFFu);
    r.tagbits = ((uint32_t)av.w >> 16) & 0xFFu; r.n_cigar = (uint32_t)av.w >> 24;
    bool ok = true;
    if (r.l_qseq == RSQC_LQSEQ_ESCAPE || r.nm == RSQC_NM_ESCAPE || r.n_cigar == RSQC_NCIGAR_ESCAPE) {
        uint32_t lo = 0, hi = b.n_wide;                     // wide table is sorted by record index
        while (lo < hi) { uint32_t m = (lo + hi) >> 1; if (b.wide_index[m] < i) lo = m + 1; else hi = m; }
        if (lo >= b.n_wide || b.wide_index[lo] != i) ok = false;
        else { r.l_qseq = b.wide_l_qseq[lo]; r.nm = b.wide_nm[lo]; r.n_cigar = b.wide_n_cigar[lo]; }
    }
    while (seg + 1 < b.n_seg && b.seg_start[seg + 1] <= i) ++seg;          // rarely iterates
    r.tid = b.seg_tid[seg];
    return ok;
}
// ------------------------------------------------------------------ K1s
// Records whose block sits fully inside exons of more than FAST_SET genes (pathological
// annotations).  The gate cascade already counted them; only the feature stage runs here.
// full in-wave aggregation by key (not only neighbouring lanes): the slow-path list is not in file order
template <class F>
__device__ __forceinline__ void wave_by_key(bool valid, uint32_t key, F &&leader) {
    uint64_t todo = __ballot(valid);
    while (todo) {
        const int lead = __ffsll((unsigned long long)todo) - 1;
        const uint32_t k0 = __shfl(key, lead, 64);
        const bool mine = valid && key == k0;
        const uint64_t same = __ballot(mine);
        leader(lead, k0, mine, same);
        todo &= ~same;
    }
}

#define RSQC_SLOW_THREADS 256
#define RSQC_SLOW_SLOTS 1024
#define RSQC_SLOW_CSLOTS 8192
// A single hot address sustains only ~90 M atomics/s on this chip (tools/atomic_bench.hip), and the
// slow-path records concentrate on a few genes: exon fractions are summed per workgroup in an LDS
// hash (row -> f64) and flushed with one global atomic per distinct row.
// Atomics into one cache line serialise at ~5-10 ns each as well, and the coverage slots these records
// touch are few (short exons of a few genes): the +1/-1 events go through an LDS hash too.
// CS: slots of the coverage hash -- RSQC_SLOW_CSLOTS for the listed records of the default rules (few per workgroup).  Under --legacy
// EVERY record comes through the kernel: a workgroup fills the table within a dozen passes, after which each of its coverage events costs
// sixteen failed probes before it goes to memory anyway, and the table's 64 KB held the kernel to two workgroups per CU -- that instance
// has no table (CS = 1, unused): its coverage events are plain memory atomics, as in classify_ei_kernel.
template <int CS>
struct SlowSharedT {
    static constexpr int CSLOTS = CS;
    uint32_t key[RSQC_SLOW_SLOTS];
    double val[RSQC_SLOW_SLOTS];
    uint32_t ckey[CS];
    uint32_t cval[CS];
    // --legacy: the workgroup's reservation in the dense pair region (see the kernel): pairs of each wave in the pass, what is left of
    // the block reserved last, and where the pairs of this pass go
    uint32_t wtot[RSQC_SLOW_THREADS / 64];
    uint32_t res_at, res_left;                  // first unused slot of the workgroup's current block, slots left in it
    uint32_t put_old, put_old_n, put_new;       // this pass: the first put_old_n pairs from put_old on, the others from put_new on
};
// (RSQC_SLOW_RES, rsqc_device.h: slots of the dense pair region a workgroup reserves at a time under --legacy)
// accumulator of the general code inside classify_slow_kernel (the `Acc` of legacy_metrics)
template <class SHARED>
struct SlowAccT {
    SHARED *S; const DevAccum *acc; const uint32_t *ex_id;
    __device__ __forceinline__ void exon_add(uint32_t row, double frac) {
#ifdef RSQC_SLOW_ABL                                                     /* (timing-only ablation builds: 1 exon adds, 2 coverage events, 4 gene counts off) */
        if (RSQC_SLOW_ABL & 1) return;
#endif
        uint32_t slot = (row * 2654435761u) >> 22;                      // 10 bits
        for (int probe = 0; probe < 16; ++probe) {
            const uint32_t old = atomicCAS(&S->key[slot], 0xFFFFFFFFu, row);
            if (old == 0xFFFFFFFFu || old == row) { atomicAdd(&S->val[slot], frac); return; }
            slot = (slot + 1) & (RSQC_SLOW_SLOTS - 1);
        }
        atomicAdd(&acc->exon_acc[ex_id[row]], frac);                    // table crowded: straight to memory
    }
    __device__ __forceinline__ void cov_add(uint32_t idx, uint32_t delta) {
#ifdef RSQC_SLOW_ABL
        if (RSQC_SLOW_ABL & 2) return;
#endif
        if (SHARED::CSLOTS > 1) {
            static_assert(SHARED::CSLOTS == 1 || SHARED::CSLOTS == 8192, "13 bits of the hash");
            uint32_t slot = (idx * 2654435761u) >> 19;                  // 13 bits
            for (int probe = 0; probe < 16; ++probe) {
                const uint32_t old = atomicCAS(&S->ckey[slot], 0xFFFFFFFFu, idx);
                if (old == 0xFFFFFFFFu || old == idx) { atomicAdd(&S->cval[slot], delta); return; }
                slot = (slot + 1) & (SHARED::CSLOTS - 1);
            }
        }
        atomicAdd(&acc->cov_diff[idx], delta);
    }
    // --legacy (no coverage table): the +1 / -1 events of a record's blocks are HELD here, up to SLOW_EV per lane, and go out behind
    // legacy_metrics, where the wave is converged, with identical neighbouring slots merged into one atomic (cov_add_merged, as in
    // classify_ei_kernel): the lanes of a pass are neighbours in the sorted file, so the reads of a deeply covered exon start on the same
    // few bases -- as plain per-lane atomics those same-address events were 9 of the kernel's 17.9 ms (call r6ak: 20.3 -> 11.4 ms per step
    // with the events compiled out).  A record with more blocks than slots sends the surplus straight to memory.
    static constexpr int SLOW_EV = 3;
    uint32_t n_ev = 0, ev_idx[SLOW_EV] = {0u, 0u, 0u}, ev_len[SLOW_EV] = {0u, 0u, 0u};
    __device__ __forceinline__ void cov_range(uint32_t cidx, uint32_t len) {
        if (len == 0) return;
        if (SHARED::CSLOTS <= 1 && n_ev < (uint32_t)SLOW_EV) { set_put<SLOW_EV>(ev_idx, (int)n_ev, cidx); set_put<SLOW_EV>(ev_len, (int)n_ev, len); ++n_ev; return; }
        cov_add(cidx, 1u); cov_add(cidx + len, 0xFFFFFFFFu);
    }
    // (all lanes of the wave, converged)
    __device__ __forceinline__ void flush_events() {
#ifdef RSQC_SLOW_ABL
        if (RSQC_SLOW_ABL & 2) { n_ev = 0; return; }
#endif
#pragma unroll
        for (int e = 0; e < SLOW_EV; ++e) {
            const bool v = n_ev > (uint32_t)e;
            const uint64_t m = WaveSink::prim(v).m;
            if (m == 0ull) break;
            const uint32_t at = v ? ev_idx[e] : 0u, ln = v ? ev_len[e] : 0u;
            cov_add_merged(acc->cov_diff, m, at, 1u);
            cov_add_merged(acc->cov_diff, m, at + ln, 0xFFFFFFFFu);
        }
        n_ev = 0;
    }
    uint32_t qh2 = 0;            // second name hash of the record being counted (0 without rsqc_batch.qhash2)
    __device__ __forceinline__ void gene_hit(uint32_t g, bool notdup, uint64_t qhash) {   // genes beyond the wave-aggregated ones
        atomicAdd(&acc->gene_reads[g], 1ull);
        if (notdup) atomicAdd(&acc->gene_unique[g], 1ull);
        const uint32_t slot = atomicAdd(acc->pair_slow_count, 1u);
        if (slot < acc->pair_slow_cap) acc->pairs[acc->pair_slow_base + slot] = PairRec{g, qh2, qhash};
        else atomicExch(acc->error, RSQC_ERR_CAPACITY);
    }
};

// LEGACY = false: the records K1 listed in ovf_index.  LEGACY = true (--legacy): every record of the batch, in file
// order, through legacy_metrics (rsqc_read.h).
#ifndef RSQC_SLOW_LEGACY_WAVES
#define RSQC_SLOW_LEGACY_WAVES 0              /* > 0: waves per SIMD the --legacy instance's register allocation is held to (A/B builds) */
#endif
#if defined(__HIPCC__) && RSQC_SLOW_LEGACY_WAVES > 0
#define RSQC_SLOW_OCC(L) __attribute__((amdgpu_waves_per_eu((L) ? RSQC_SLOW_LEGACY_WAVES : 1, (L) ? RSQC_SLOW_LEGACY_WAVES : 8)))
#else
#define RSQC_SLOW_OCC(L)
#endif
template <bool LEGACY>
__global__ void __launch_bounds__(RSQC_SLOW_THREADS) RSQC_SLOW_OCC(LEGACY)
classify_slow_kernel(DevAnnotation a, DevParams p, DevBatch b, DevAccum acc) {
    typedef SlowSharedT<LEGACY ? 1 : RSQC_SLOW_CSLOTS> SlowShared;
    typedef SlowAccT<SlowShared> SlowAcc;
    constexpr int CSLOTS = SlowShared::CSLOTS > 1 ? SlowShared::CSLOTS : 0;
    __shared__ SlowShared SH;
    uint32_t *const s_key = SH.key; double *const s_val = SH.val; uint32_t *const s_ckey = SH.ckey, *const s_cval = SH.cval;
    uint64_t n = *acc.ovf_count < acc.ovf_cap ? *acc.ovf_count : acc.ovf_cap;
    if (LEGACY) n = b.n;
    // the list is taken grid-stride (it is short); under --legacy a workgroup takes a CONTIGUOUS range of the file, so that the pairs it
    // writes to its blocks of the dense region are neighbours in the file (frag_local_kernel's window finds the mates among them)
    uint64_t k_beg = (uint64_t)blockIdx.x * blockDim.x, k_end = n, stride = (uint64_t)gridDim.x * blockDim.x;
    if (LEGACY) {
        const uint64_t per = ((n + gridDim.x - 1) / gridDim.x + blockDim.x - 1) / blockDim.x * blockDim.x;     // a multiple of the workgroup's size
        k_beg = (uint64_t)blockIdx.x * per; k_end = k_beg + per < n ? k_beg + per : n; stride = blockDim.x;
    }
    if (k_beg >= k_end) return;  // nothing for this workgroup (the usual case for most of the grid)
    if (LEGACY && threadIdx.x == 0) { SH.res_at = 0u; SH.res_left = 0u; }
    for (int i = threadIdx.x; i < RSQC_SLOW_SLOTS; i += blockDim.x) { s_key[i] = 0xFFFFFFFFu; s_val[i] = 0.0; }
    for (int i = threadIdx.x; i < CSLOTS; i += blockDim.x) { s_ckey[i] = 0xFFFFFFFFu; s_cval[i] = 0u; }
    __syncthreads();
    const int l = lane_id();
    DirectAcc dacc{acc.exon_acc, acc.cov_diff, a.ex_id};
    SlowAcc sacc{&SH, &acc, a.ex_id};
    auto exon_add_lds = [&](uint32_t row, double frac) { sacc.exon_add(row, frac); };
    auto cov_add_lds = [&](uint32_t idx, uint32_t delta) { sacc.cov_add(idx, delta); };
    unsigned long long my_cnt = 0ull;                 // lane c accumulates counter c
    for (uint64_t k0 = k_beg; k0 < k_end; k0 += stride) {
        const uint64_t k = k0 + threadIdx.x;
        uint64_t bits = 0;
        FeatureOut<MID_SET, SLOW_STAGE> fm;
        fm.bits = 0; fm.n_hit = 0; fm.n_commit = 0;
        uint32_t aligned = 1; bool notdup = false; uint64_t qhash = 0; uint32_t qh2 = 0;
        if (k < k_end) {
            Record r;
            // bit 63 of a listed index: classify_ei_kernel did not walk this record's CIGAR to the end (a long-CIGAR straggler of a
            // boundary tile) -- its blocks are counted and its operations checked here
            const uint64_t entry = LEGACY ? k : acc.ovf_index[k];
            const uint64_t i = entry & ~(1ull << 63);
            // (--legacy: the pass's records are neighbours in the file -- the segment of its FIRST record, found with scalar loads, is the
            //  hint load_record advances from; a per-lane binary search is five dependent vector loads in front of every record)
            if (load_record(b, i, LEGACY ? find_segment(b, k0) : find_segment(b, i), r)) {
                RecordCounters rc; bool hq; Blocks B;
                const bool go = gate_cascade(a, p, r, rc, hq, aligned, B);
                if (!LEGACY && (entry >> 63)) {
                    if (rc.error) atomicExch(acc.error, rc.error);
                    if (rc.blocks) atomicAdd(&acc.counters[RSQC_C_ALIGNMENT_BLOCKS], (unsigned long long)rc.blocks);
                }
                if (go) {
                    notdup = !(r.flag & RSQC_FDUP); qhash = r.qhash; qh2 = b.qhash2 ? b.qhash2[i] : 0u; sacc.qh2 = qh2;
                    bool overflow = false;
                    if (LEGACY) {
                        LegacyOut<MID_SET> lo;
                        legacy_metrics<MID_SET>(a, p, r, hq, sacc, lo);
                        bits = lo.bits; fm.n_hit = lo.n_hit;
#pragma unroll
                        for (int j = 0; j < MID_SET; ++j) fm.hit[j] = lo.hit[j];
                    } else {
                    exon_metrics<MID_SET>(a, p, r, hq, aligned, dacc, fm, overflow);
                    if (overflow) {                  // rare second tier: up to 32 genes, plain atomics
                        fm.bits = 0; fm.n_hit = 0; fm.n_commit = 0;
                        FeatureOut<SLOW_SET, SLOW_STAGE> fo;
                        exon_metrics<SLOW_SET>(a, p, r, hq, aligned, dacc, fo, overflow);
                        if (overflow) atomicExch(acc.error, RSQC_ERR_CAPACITY);
                        else {
                            for (int j = 0; j < fo.n_commit; ++j) {
                                const Commit cm = fo.commit[j];
                                if (cm.len > 0) dacc.exon_add(cm.row, (double)cm.len / (double)aligned);
                                dacc.cov_range(cm.cidx, cm.len);
                            }
                            for (int j = 0; j < fo.n_hit; ++j) {
                                const uint32_t g = fo.hit[j];
                                atomicAdd(&acc.gene_reads[g], 1ull);
                                if (notdup) atomicAdd(&acc.gene_unique[g], 1ull);
                                const uint32_t slot = atomicAdd(acc.pair_slow_count, 1u);
                                if (slot < acc.pair_slow_cap) acc.pairs[acc.pair_slow_base + slot] = PairRec{g, qh2, qhash};
                                else atomicExch(acc.error, RSQC_ERR_CAPACITY);
                            }
                            bits = fo.bits;
                        }
                    } else bits = fm.bits;
                    }
                }
            }
        }
        if (LEGACY) sacc.flush_events();                 // (the held coverage events of legacy_metrics, neighbours merged)
        // ---- scatter of the first-tier results: few records, so exon fractions and coverage go out as
        //      plain atomics; gene counts and pair slots are aggregated per wave (same-address traffic)
        for (int j = 0; j < fm.n_commit; ++j) {
            const Commit cm = fm.commit[j];
            if (cm.len > 0) exon_add_lds(cm.row, (double)cm.len / (double)aligned);
            if (cm.len > 0) {
                const uint32_t base = cm.cidx;
                cov_add_lds(base, 1u); cov_add_lds(base + cm.len, 0xFFFFFFFFu);
            }
        }
        {
            const uint64_t nd_mask = __ballot(notdup);
            if (LEGACY) {
                // Every record comes through here: a slot reservation per wave and gene rank on the ONE counter of the dense region is two
                // million returning atomics on one address per 100 M records -- 88 per microsecond chip-wide (MI355X_MICROARCH.md,
                // "dequeue"): 20.8 of the kernel's 21 ms (profiles/r6_kernel_stats_legacy.txt).  The workgroup reserves RSQC_SLOW_RES slots
                // at a time instead and hands them out from LDS: ranks of the pass's pairs by a scan, the block's remainder first, a new
                // block when it runs out (one memory atomic per ~2 000 pairs); what is left of its last block is filled with empty
                // entries when it retires (frag_local_kernel skips them).
                const uint32_t nh = (uint32_t)fm.n_hit;
                const uint32_t inc = wave_inclusive_scan_u32_dpp(nh);               // (every lane of the workgroup is here)
                if (l == 63) SH.wtot[threadIdx.x >> 6] = inc;
                __syncthreads();
                if (threadIdx.x == 0) {
                    uint32_t T = 0;
#pragma unroll
                    for (int w = 0; w < RSQC_SLOW_THREADS / 64; ++w) T += SH.wtot[w];
                    static_assert(RSQC_SLOW_RES >= RSQC_SLOW_THREADS * MID_SET, "a block holds the pairs of a pass");
                    if (T <= SH.res_left) { SH.put_old = SH.res_at; SH.put_old_n = T; SH.put_new = 0u; SH.res_at += T; SH.res_left -= T; }
                    else {
                        const uint32_t nb = atomicAdd(acc.pair_slow_count, RSQC_SLOW_RES), rest = T - SH.res_left;
                        SH.put_old = SH.res_at; SH.put_old_n = SH.res_left; SH.put_new = nb;
                        SH.res_at = nb + rest; SH.res_left = RSQC_SLOW_RES - rest;
                    }
                }
                __syncthreads();
                uint32_t r = inc - nh;
#pragma unroll
                for (int w = 0; w < RSQC_SLOW_THREADS / 64; ++w) r += (uint32_t)w < (threadIdx.x >> 6) ? SH.wtot[w] : 0u;
                const uint32_t po = SH.put_old, pn_old = SH.put_old_n, pnew = SH.put_new;
#pragma unroll
                for (int j = 0; j < MID_SET; ++j)
                    if ((uint32_t)j < nh) {
                        const uint32_t rr = r + (uint32_t)j, slot = rr < pn_old ? po + rr : pnew + (rr - pn_old);
                        if (slot < acc.pair_slow_cap) acc.pairs[acc.pair_slow_base + slot] = PairRec{fm.hit[j], qh2, qhash};
                        else atomicExch(acc.error, RSQC_ERR_CAPACITY);
                    }
            }
#pragma unroll
            for (int j = 0; j < MID_SET; ++j) {
                const bool has = fm.n_hit > j;
                const uint64_t m = __ballot(has);
                if (m == 0ull) break;
                const uint32_t g = fm.hit[j];
                if (!LEGACY) {
                    const int lead0 = __ffsll((unsigned long long)m) - 1;
                    uint32_t base = 0;
                    if (l == lead0) base = atomicAdd(acc.pair_slow_count, (uint32_t)__popcll(m));
                    base = __shfl(base, lead0, 64);
                    if (has) {
                        const uint32_t slot = base + mask_rank(m);
                        if (slot < acc.pair_slow_cap) acc.pairs[acc.pair_slow_base + slot] = PairRec{g, qh2, qhash};
                        else atomicExch(acc.error, RSQC_ERR_CAPACITY);
                    }
                }
                wave_by_key(has, g, [&](int lead, uint32_t gg, bool, uint64_t same) {
#ifdef RSQC_SLOW_ABL
                    if (RSQC_SLOW_ABL & 4) return;
#endif
                    if (l == lead) {
                        atomicAdd(&acc.gene_reads[gg], (unsigned long long)__popcll(same));
                        const uint32_t nd = (uint32_t)__popcll(same & nd_mask);
#ifdef RSQC_SLOW_ABL
                        if (RSQC_SLOW_ABL & 16) return;
#endif
                        if (nd) atomicAdd(&acc.gene_unique[gg], (unsigned long long)nd);
                    }
                });
            }
        }
#pragma unroll
        for (int c = 0; c < RSQC_N_COUNTERS; ++c) {       // the gate cascade was counted by K1; only feature-stage bits here
            const uint64_t m = __ballot((bits >> c) & 1ull);
            if (l == c) my_cnt += (unsigned long long)__popcll(m);
        }
    }
    if (l < RSQC_N_COUNTERS && my_cnt) atomicAdd(&acc.counters[l], my_cnt);
    __syncthreads();
    if (LEGACY) {                                        // the unused slots of the workgroup's last block: empty entries
        const uint32_t at = SH.res_at, left = SH.res_left;
        for (uint32_t i = threadIdx.x; i < left; i += blockDim.x)
            if (at + i < acc.pair_slow_cap) acc.pairs[acc.pair_slow_base + at + i] = PairRec{0xFFFFFFFFu, 0u, 0ull};
    }
    for (int i = threadIdx.x; i < RSQC_SLOW_SLOTS; i += blockDim.x)
        if (s_key[i] != 0xFFFFFFFFu) atomicAdd(&acc.exon_acc[a.ex_id[s_key[i]]], s_val[i]);
    for (int i = threadIdx.x; i < CSLOTS; i += blockDim.x)
        if (s_ckey[i] != 0xFFFFFFFFu && s_cval[i] != 0u) atomicAdd(&acc.cov_diff[s_ckey[i]], s_cval[i]);
}

}  // namespace rsqc
