// rsqc_kernels.hip -- hand-written HIP kernels for gfx950 (MI355X, wave64).
//
//   K1  classify_count_kernel   per-record path: gate cascade, CIGAR blocks, overlap query,
//                               gene/exon/coverage scatter, scalar counters (wave-reduced)
//   K1s classify_slow_kernel    exact slow path for records whose block hits > FAST_SET genes
//   KR  read_length_kernel      order-dependent "Read Length" state machine over tile summaries
//   K4  dedup_insert_kernel     per-gene distinct QNAME count (geneFragmentCounts)
//   K3  gene_coverage_kernel    per-gene: diff->coverage scan, per-exon CV, bias windows,
//                               masked gene mean/std/CV  (one wavefront per gene)
//
// This is integer / byte indexing work bound by HBM and atomics, not a contraction:
// no MFMA.  All wave-level idioms are written for 64-lane wavefronts.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "rsqc_device.h"

namespace rsqc {

// ------------------------------------------------------------------ wave helpers
__device__ __forceinline__ int lane_id() { return (int)(threadIdx.x & 63u); }

__device__ __forceinline__ uint32_t mask_rank(uint64_t m) {      // #set bits below this lane
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
}

template <class T>
__device__ __forceinline__ T wave_sum(T v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ uint32_t wave_max_u32(uint32_t v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { uint32_t t = __shfl_xor(v, o, 64); v = t > v ? t : v; }
    return v;
}
__device__ __forceinline__ uint32_t wave_min_u32(uint32_t v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { uint32_t t = __shfl_xor(v, o, 64); v = t < v ? t : v; }
    return v;
}
__device__ __forceinline__ uint32_t wave_inclusive_scan_u32(uint32_t v) {
    const int l = lane_id();
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { uint32_t t = __shfl_up(v, o, 64); if (l >= o) v += t; }
    return v;
}

// One atomic per distinct key in the wave.  Must be called by all 64 lanes (converged).
template <class F>
__device__ __forceinline__ void wave_aggregate(bool valid, uint32_t key, uint64_t flagmask, F &&leader) {
    uint64_t todo = __ballot(valid);
    const int l = lane_id();
    while (todo) {
        const int lead = __ffsll((unsigned long long)todo) - 1;
        const uint32_t k0 = __shfl(key, lead, 64);
        const uint64_t same = __ballot(valid && key == k0);
        if (l == lead) leader(k0, (uint32_t)__popcll(same), (uint32_t)__popcll(same & flagmask));
        todo &= ~same;
    }
}

// ------------------------------------------------------------------ K1 accumulators
// Fast-path policy: gene hits and the first exon fractions are kept in registers and
// flushed with wave-level aggregation at a converged point; coverage goes straight to
// the difference array (2 atomics per committed block).
struct FastAcc {
    uint32_t hit[FAST_SET]; int nhit; bool notdup; uint64_t qhash;
    uint32_t ex_row[2]; double ex_frac[2]; int nex;
    double *exon_acc; uint32_t *cov_diff; const uint32_t *ex_cov;
    __device__ __forceinline__ void gene_hit(uint32_t g, bool nd, uint64_t qh) {
        if (nhit < FAST_SET) hit[nhit++] = g;
        notdup = nd; qhash = qh;
    }
    __device__ __forceinline__ void exon_add(uint32_t row, double frac) {
        if (nex < 2) { ex_row[nex] = row; ex_frac[nex] = frac; ++nex; }
        else atomicAdd(&exon_acc[row], frac);
    }
    __device__ __forceinline__ void cov_range(uint32_t row, uint32_t off, uint32_t len, uint32_t elen) {
        if (len == 0) return;
        const uint32_t base = ex_cov[row];
        atomicAdd(&cov_diff[base + off], 1u);
        if (off + len < elen) atomicAdd(&cov_diff[base + off + len], 0xFFFFFFFFu);
    }
};

// Slow-path policy: plain atomics.
struct SlowAcc {
    unsigned long long *gene_reads, *gene_unique; double *exon_acc; uint32_t *cov_diff; const uint32_t *ex_cov;
    uint32_t *pair_gene; uint64_t *pair_hash; uint32_t *pair_count; uint32_t pair_cap; int *error;
    __device__ __forceinline__ void gene_hit(uint32_t g, bool nd, uint64_t qh) {
        atomicAdd(&gene_reads[g], 1ull);
        if (nd) atomicAdd(&gene_unique[g], 1ull);
        const uint32_t slot = atomicAdd(pair_count, 1u);
        if (slot < pair_cap) { pair_gene[slot] = g; pair_hash[slot] = qh; }
        else atomicExch(error, RSQC_ERR_CAPACITY);
    }
    __device__ __forceinline__ void exon_add(uint32_t row, double frac) { atomicAdd(&exon_acc[row], frac); }
    __device__ __forceinline__ void cov_range(uint32_t row, uint32_t off, uint32_t len, uint32_t elen) {
        if (len == 0) return;
        const uint32_t base = ex_cov[row];
        atomicAdd(&cov_diff[base + off], 1u);
        if (off + len < elen) atomicAdd(&cov_diff[base + off + len], 0xFFFFFFFFu);
    }
};

__device__ __forceinline__ bool load_record(const DevBatch &b, uint64_t i, Record &r) {
    r.pos = b.pos[i]; r.mpos = b.mpos[i]; r.isize = b.isize[i];
    r.flag = b.flag[i]; r.mapq = b.mapq[i]; r.tagbits = b.tagbits[i];
    r.l_qseq = b.l_qseq[i]; r.nm = b.nm[i]; r.n_cigar = b.n_cigar[i];
    r.qhash = b.qhash[i];
    r.cigar = b.cigar + b.cigar_off[i];
    if (r.l_qseq == RSQC_LQSEQ_ESCAPE || r.nm == RSQC_NM_ESCAPE || r.n_cigar == RSQC_NCIGAR_ESCAPE) {
        uint32_t lo = 0, hi = b.n_wide;                     // wide table is sorted by record index
        while (lo < hi) { uint32_t m = (lo + hi) >> 1; if (b.wide_index[m] < i) lo = m + 1; else hi = m; }
        if (lo >= b.n_wide || b.wide_index[lo] != i) return false;
        r.l_qseq = b.wide_l_qseq[lo]; r.nm = b.wide_nm[lo]; r.n_cigar = b.wide_n_cigar[lo];
    }
    // contig of the record: segment lookup (few segments; sorted input)
    uint32_t lo = 0, hi = b.n_seg;
    while (hi - lo > 1) { uint32_t m = (lo + hi) >> 1; if (b.seg_start[m] <= i) lo = m; else hi = m; }
    r.tid = b.seg_tid[lo];
    return true;
}

// ------------------------------------------------------------------ K1
// grid-stride over tiles of blockDim.x records; one record per lane per iteration.
__global__ void __launch_bounds__(RSQC_K1_THREADS)
classify_count_kernel(DevAnnotation a, DevParams p, DevBatch b, DevAccum acc) {
    __shared__ unsigned long long s_cnt[RSQC_N_COUNTERS];
    __shared__ uint32_t s_span[RSQC_K1_THREADS / 64], s_lmin[RSQC_K1_THREADS / 64], s_lmax[RSQC_K1_THREADS / 64];
    const int l = lane_id();
    const int wave = (int)(threadIdx.x >> 6);
    for (int c = threadIdx.x; c < RSQC_N_COUNTERS; c += blockDim.x) s_cnt[c] = 0ull;
    __syncthreads();

    unsigned long long my_cnt = 0ull;     // lane c of every wave accumulates counter c
    const uint64_t n_tiles = (b.n + blockDim.x - 1) / blockDim.x;
    for (uint64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const uint64_t i = tile * blockDim.x + threadIdx.x;
        const bool valid = i < b.n;
        RecordCounters rc;
        rc.bits = 0; rc.e1_mm = rc.e1_bases = rc.e2_mm = rc.e2_bases = rc.mm = rc.bases = rc.blocks = 0;
        rc.rl_eligible = 0; rc.rl_span = 0; rc.rl_lqseq = 0; rc.error = 0; rc.frag_candidate = 0; rc.endpos = 0;
        FastAcc fa;
        fa.nhit = 0; fa.nex = 0; fa.notdup = false; fa.qhash = 0;
        fa.exon_acc = acc.exon_acc; fa.cov_diff = acc.cov_diff; fa.ex_cov = a.ex_cov;
        bool overflow = false;
        Record r;
        if (valid) {
            if (!load_record(b, i, r)) { atomicExch(acc.error, RSQC_ERR_ARG); }
            else {
                bool hq; uint32_t aligned;
                if (gate_cascade(a, p, r, rc, hq, aligned)) {
                    const uint64_t fbits = exon_metrics<FAST_SET>(a, p, r, hq, aligned, fa, overflow);
                    if (overflow) {
                        const uint32_t slot = atomicAdd(acc.ovf_count, 1u);
                        if (slot < acc.ovf_cap) acc.ovf_index[slot] = i;
                        else atomicExch(acc.error, RSQC_ERR_CAPACITY);
                    } else rc.bits |= fbits;
                }
                if (rc.error) atomicExch(acc.error, rc.error);
            }
        }
        // ---- converged: wave-aggregated flushes ----------------------------------------
        const uint64_t nd_mask = __ballot(fa.notdup);
#pragma unroll
        for (int k = 0; k < FAST_SET; ++k) {
            const bool has = fa.nhit > k;
            const uint64_t m = __ballot(has);
            if (m == 0) break;
            // (gene, qname-hash) pairs for the fragment de-dup: one slot reservation per wave
            uint32_t base = 0;
            if (l == (int)(__ffsll((unsigned long long)m) - 1)) base = atomicAdd(acc.pair_count, (uint32_t)__popcll(m));
            base = __shfl(base, __ffsll((unsigned long long)m) - 1, 64);
            if (has) {
                const uint32_t slot = base + mask_rank(m);
                if (slot < acc.pair_cap) { acc.pair_gene[slot] = fa.hit[k]; acc.pair_hash[slot] = fa.qhash; }
                else atomicExch(acc.error, RSQC_ERR_CAPACITY);
            }
            wave_aggregate(has, fa.hit[k], nd_mask, [&](uint32_t g, uint32_t cnt, uint32_t cnt_nd) {
                atomicAdd(&acc.gene_reads[g], (unsigned long long)cnt);
                if (cnt_nd) atomicAdd(&acc.gene_unique[g], (unsigned long long)cnt_nd);
            });
        }
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const bool has = fa.nex > k;
            if (__ballot(has) == 0) break;
            // exon fractions: sum equal rows inside the wave, one f64 atomic per distinct row
            uint64_t todo = __ballot(has);
            while (todo) {
                const int lead = __ffsll((unsigned long long)todo) - 1;
                const uint32_t r0 = __shfl(fa.ex_row[k], lead, 64);
                const bool mine = has && fa.ex_row[k] == r0;
                const uint64_t same = __ballot(mine);
                const double s = wave_sum(mine ? fa.ex_frac[k] : 0.0);
                if (l == lead) atomicAdd(&acc.exon_acc[r0], s);
                todo &= ~same;
            }
        }
        // ---- scalar counters: ballot + popcount, lane c keeps counter c ----------------
#pragma unroll
        for (int c = 0; c < RSQC_N_COUNTERS; ++c) {
            const uint64_t m = __ballot((rc.bits >> c) & 1ull);
            if (l == c) my_cnt += (unsigned long long)__popcll(m);
        }
        {
            const uint32_t s0 = wave_sum(rc.e1_mm), s1 = wave_sum(rc.e1_bases), s2 = wave_sum(rc.e2_mm),
                           s3 = wave_sum(rc.e2_bases), s4 = wave_sum(rc.mm), s5 = wave_sum(rc.bases),
                           s6 = wave_sum(rc.blocks);
            if (l == RSQC_C_END1_MISMATCHES) my_cnt += s0;
            if (l == RSQC_C_END1_BASES) my_cnt += s1;
            if (l == RSQC_C_END2_MISMATCHES) my_cnt += s2;
            if (l == RSQC_C_END2_BASES) my_cnt += s3;
            if (l == RSQC_C_MISMATCHED_BASES) my_cnt += s4;
            if (l == RSQC_C_TOTAL_BASES) my_cnt += s5;
            if (l == RSQC_C_ALIGNMENT_BLOCKS) my_cnt += s6;
        }
        // ---- Read-Length tile summary (max span, min/max l_qseq over eligible records) ---
        {
            const uint32_t sp = wave_max_u32(rc.rl_eligible ? rc.rl_span : 0u);
            const uint32_t mn = wave_min_u32(rc.rl_eligible ? (uint32_t)rc.rl_lqseq : 0xFFFFFFFFu);
            const uint32_t mx = wave_max_u32(rc.rl_eligible ? (uint32_t)rc.rl_lqseq : 0u);
            if (l == 0) { s_span[wave] = sp; s_lmin[wave] = mn; s_lmax[wave] = mx; }
            __syncthreads();
            if (threadIdx.x == 0) {
                uint32_t S = 0, mn2 = 0xFFFFFFFFu, mx2 = 0;
                for (int w = 0; w < (int)(blockDim.x >> 6); ++w) {
                    S = s_span[w] > S ? s_span[w] : S;
                    mn2 = s_lmin[w] < mn2 ? s_lmin[w] : mn2;
                    mx2 = s_lmax[w] > mx2 ? s_lmax[w] : mx2;
                }
                const uint64_t t = b.tile_base + tile;
                acc.tile_span[t] = S; acc.tile_lmin[t] = mn2; acc.tile_lmax[t] = mx2;
            }
            __syncthreads();
        }
    }
    if (l < RSQC_N_COUNTERS && my_cnt) atomicAdd(&s_cnt[l], my_cnt);
    __syncthreads();
    for (int c = threadIdx.x; c < RSQC_N_COUNTERS; c += blockDim.x)
        if (s_cnt[c]) atomicAdd(&acc.counters[c], s_cnt[c]);
}

// ------------------------------------------------------------------ K1s
// Records whose block sits fully inside exons of more than FAST_SET genes (pathological
// annotations).  The gate cascade already counted them; only the feature stage runs here.
__global__ void classify_slow_kernel(DevAnnotation a, DevParams p, DevBatch b, DevAccum acc) {
    const uint32_t n = *acc.ovf_count < acc.ovf_cap ? *acc.ovf_count : acc.ovf_cap;
    for (uint32_t k = blockIdx.x * blockDim.x + threadIdx.x; k < n; k += gridDim.x * blockDim.x) {
        Record r;
        if (!load_record(b, acc.ovf_index[k], r)) continue;
        RecordCounters rc; bool hq; uint32_t aligned;
        if (!gate_cascade(a, p, r, rc, hq, aligned)) continue;
        SlowAcc sa{acc.gene_reads, acc.gene_unique, acc.exon_acc, acc.cov_diff, a.ex_cov,
                   acc.pair_gene, acc.pair_hash, acc.pair_count, acc.pair_cap, acc.error};
        bool overflow = false;
        const uint64_t bits = exon_metrics<SLOW_SET>(a, p, r, hq, aligned, sa, overflow);
        if (overflow) { atomicExch(acc.error, RSQC_ERR_CAPACITY); continue; }
        for (int c = 0; c < RSQC_N_COUNTERS; ++c) if ((bits >> c) & 1ull) atomicAdd(&acc.counters[c], 1ull);
    }
}

// ------------------------------------------------------------------ KR
// "Read Length" (src/RNASeQC.cpp:275-278): readLength = l_qseq of each record whose span
// exceeds the current value, in FILE order.  One wavefront walks the tile summaries 64 at a
// time; a tile is opened only when some record in it could change the state to a new value.
__global__ void __launch_bounds__(64)
read_length_kernel(DevAnnotation a, DevParams p, DevBatch b, DevAccum acc) {
    const int l = lane_id();
    uint32_t r = (uint32_t)*acc.read_length;
    const uint64_t n_tiles = (b.n + RSQC_K1_THREADS - 1) / RSQC_K1_THREADS;
    for (uint64_t t0 = 0; t0 < n_tiles; t0 += 64) {
        const uint64_t t = t0 + l;
        uint32_t S = 0, mn = 0xFFFFFFFFu, mx = 0;
        if (t < n_tiles) { S = acc.tile_span[b.tile_base + t]; mn = acc.tile_lmin[b.tile_base + t]; mx = acc.tile_lmax[b.tile_base + t]; }
        uint64_t need = __ballot(S > r && !(mn == mx && mn == r));
        while (need) {
            const int tl = __ffsll((unsigned long long)need) - 1;
            need &= need - 1;
            const uint32_t St = __shfl(S, tl, 64), mnt = __shfl(mn, tl, 64), mxt = __shfl(mx, tl, 64);
            if (!(St > r && !(mnt == mxt && mnt == r))) continue;      // state moved since the ballot
            // open tile: replay its records in order, 64 at a time
            const uint64_t base = (t0 + tl) * RSQC_K1_THREADS;
            for (uint32_t j0 = 0; j0 < RSQC_K1_THREADS; j0 += 64) {
                const uint64_t i = base + j0 + l;
                uint32_t span = 0, lq = 0; bool elig = false;
                if (i < b.n) {
                    Record rec;
                    if (load_record(b, i, rec)) {
                        RecordCounters rc; bool hq; uint32_t aligned;
                        gate_cascade(a, p, rec, rc, hq, aligned);
                        elig = rc.rl_eligible != 0; span = rc.rl_span; lq = (uint32_t)rc.rl_lqseq;
                    }
                }
                int from = 0;
                while (true) {
                    const uint64_t m = __ballot(elig && l >= from && span > r && lq != r);
                    if (!m) break;
                    const int w = __ffsll((unsigned long long)m) - 1;
                    r = __shfl(lq, w, 64);
                    from = w + 1;
                }
            }
            // re-evaluate the remaining tiles of this group against the new state
            need &= __ballot(S > r && !(mn == mx && mn == r));
        }
    }
    if (l == 0) *acc.read_length = (int32_t)r;
}

// ------------------------------------------------------------------ K4
// geneFragmentCounts: number of distinct QNAMEs among the records counted to a gene
// (src/Expression.cpp:383-387).  Every gene owns a slice of an open-addressing table sized
// 2 x geneCounts[gene], so the probes of one gene stay inside a small, cache-resident range.
__device__ __forceinline__ uint64_t mix64(uint64_t x) {
    x ^= x >> 33; x *= 0xff51afd7ed558ccdull; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ull; x ^= x >> 33;
    return x;
}

__global__ void __launch_bounds__(256)
dedup_insert_kernel(const uint32_t *pair_gene, const uint64_t *pair_hash, uint32_t n_pairs,
                    const uint64_t *tab_off, const uint32_t *tab_cap, unsigned long long *table,
                    unsigned long long *gene_frag) {
    const uint32_t stride = gridDim.x * blockDim.x;
    const uint32_t rounds = (n_pairs + stride - 1) / stride;
    for (uint32_t it = 0; it < rounds; ++it) {
        const uint32_t i = it * stride + blockIdx.x * blockDim.x + threadIdx.x;
        bool fresh = false; uint32_t g = 0;
        if (i < n_pairs) {
            g = pair_gene[i];
            uint64_t key = pair_hash[i];
            if (key == 0) key = 0x9e3779b97f4a7c15ull;           // 0 marks an empty slot
            const uint32_t cap = tab_cap[g];
            unsigned long long *tab = table + tab_off[g];
            uint32_t slot = (uint32_t)(mix64(key) % cap);
            for (uint32_t probes = 0; probes < cap; ++probes) {
                const unsigned long long old = atomicCAS(&tab[slot], 0ull, (unsigned long long)key);
                if (old == 0ull) { fresh = true; break; }
                if (old == key) break;
                slot = slot + 1 == cap ? 0 : slot + 1;
            }
        }
        wave_aggregate(fresh, g, 0ull, [&](uint32_t gg, uint32_t cnt, uint32_t) {
            atomicAdd(&gene_frag[gg], (unsigned long long)cnt);
        });
    }
}

// ------------------------------------------------------------------ K3
// One wavefront per gene.  cov[] holds the per-base DIFFERENCE array of the gene's exons,
// laid out contiguously in exonsForGene order, so the stitched transcript vector of
// computeCoverage (src/Metrics.cpp:306-308) is simply cov[gene_cov_off .. +coding).


__device__ __forceinline__ double wave_sum_f64(double v) { return wave_sum(v); }
__device__ __forceinline__ unsigned long long wave_sum_u64(unsigned long long v) { return wave_sum(v); }

// quirky computeMedian (src/Metrics.h:147-160) of the k-th order statistics of a window held in
// LDS (unsorted): select by rank counting.  Returns false for an empty window (range_error).
__device__ bool window_median(const uint32_t *w, uint32_t n, double *out) {
    if (n == 0) return false;
    const int l = lane_id();
    if (n == 1) { *out = (double)w[0]; return true; }
    const uint32_t mid = (n - 1) / 2;
    const bool odd = (n & 1u) != 0;
    uint32_t va = 0, vb = 0;       // values with rank mid and mid+1
    for (uint32_t i0 = 0; i0 < n; i0 += 64) {
        const uint32_t i = i0 + l;
        uint32_t rank = 0xFFFFFFFFu, v = 0;
        if (i < n) {
            v = w[i]; rank = 0;
            for (uint32_t j = 0; j < n; ++j) { const uint32_t u = w[j]; rank += (u < v || (u == v && j < i)) ? 1u : 0u; }
        }
        const uint64_t ma = __ballot(rank == mid), mb = __ballot(rank == mid + 1);
        if (ma) va = __shfl(v, __ffsll((unsigned long long)ma) - 1, 64);
        if (mb) vb = __shfl(v, __ffsll((unsigned long long)mb) - 1, 64);
    }
    *out = odd ? ((double)va + (double)vb) / 2.0 : (double)va;
    return true;
}

__global__ void __launch_bounds__(256)
gene_coverage_kernel(GeneCovArgs A) {
    __shared__ uint32_t s_hist[4][256];
    __shared__ uint32_t s_win[4][2][RSQC_MAX_BIAS_WINDOW];
    const int l = lane_id();
    const int wv = (int)(threadIdx.x >> 6);
    const int gene = (int)(blockIdx.x * 4 + wv);
    if (gene >= A.n_listed) return;
    if (!A.gene_owned[gene]) return;
    const uint32_t coding = A.gene_coding[gene];
    const uint32_t e0 = A.ge_off[gene], e1 = A.ge_off[gene + 1];
    uint32_t *C = A.cov + A.gene_cov_off[gene];
    const uint32_t MASK = A.mask;
    const bool touched = A.gene_reads[gene] != 0ull;
    const uint32_t W = (uint32_t)A.bias_window, OFF = (uint32_t)A.bias_offset;

    if (!touched) {
        // all-zero coverage: mean 0, std 0, cv NaN; no exon CV; the bias gate reads zeros.
        if (coding >= A.bias_gene_length) {
            uint32_t cur = W / 2 < coding ? W / 2 : coding;
            if ((W < cur ? W : cur) == 0 && l == 0) atomicExch(A.error, RSQC_ERR_EMPTY_MEDIAN);
        }
        if (l == 0) {
            const bool valid = MASK ? coding > 2 * (uint64_t)MASK : coding > 0;
            A.g_valid[gene] = valid ? 1 : 0;
            A.g_mean[gene] = 0.0; A.g_std[gene] = 0.0; A.g_cv[gene] = __longlong_as_double(0x7ff8000000000000ll);
        }
        return;
    }
    // (1) per-exon inclusive scan: difference array -> coverage, in place
    for (uint32_t k = e0; k < e1; ++k) {
        const uint32_t row = A.ge_row[k];
        const uint32_t len = (uint32_t)(A.ex_end[row] - A.ex_start[row] + 1);
        uint32_t *E = A.cov + A.ex_cov[row];
        uint32_t carry = 0;
        for (uint32_t j0 = 0; j0 < len; j0 += 64) {
            const uint32_t j = j0 + l;
            uint32_t v = j < len ? E[j] : 0u;
            v = wave_inclusive_scan_u32(v) + carry;
            if (j < len) E[j] = v;
            carry = __shfl(v, 63, 64);
        }
    }
    __threadfence_block();
    // (2) per-exon CV over the unmasked part (src/Metrics.cpp:267-305): transcript positions
    //     [MASK, coding-MASK) survive the two mask walks
    {
        uint32_t t0 = 0;
        const uint64_t lo_t = MASK, hi_t = coding > MASK ? coding - MASK : 0;
        for (uint32_t k = e0; k < e1; ++k) {
            const uint32_t row = A.ge_row[k];
            const uint32_t len = (uint32_t)(A.ex_end[row] - A.ex_start[row] + 1);
            const uint64_t a0 = t0 > lo_t ? t0 : lo_t, b0 = (uint64_t)t0 + len < hi_t ? (uint64_t)t0 + len : hi_t;
            if (b0 > a0) {
                const uint32_t a = (uint32_t)(a0 - t0), bnd = (uint32_t)(b0 - t0);
                const double size = (double)(bnd - a);
                const uint32_t *E = A.cov + A.ex_cov[row];
                unsigned long long s = 0;
                for (uint32_t j = a + l; j < bnd; j += 64) s += E[j];
                s = wave_sum_u64(s);
                const double mean = (double)s / size;
                double q = 0.0;
                for (uint32_t j = a + l; j < bnd; j += 64) { const double d = (double)E[j] - mean; q += d * d; }
                q = wave_sum_f64(q);
                const double cv = sqrt(q / size) / mean;
                if (l == 0 && !(isnan(cv) || isinf(cv))) { A.e_cv[row] = cv; A.e_cv_valid[row] = 1; }
            }
            t0 += len;
        }
    }
    // (3) bias (src/Metrics.cpp:160-235) on the stitched, unmasked vector C[0..coding)
    uint32_t v0 = 0, v1 = coding;          // the (possibly trimmed) vector the gene stats use (Q14)
    if (coding >= A.bias_gene_length) {
        uint32_t best = 0, best_i = 0xFFFFFFFFu;
        for (uint32_t j = l; j < coding; j += 64) { const uint32_t v = C[j]; if (v > best) { best = v; best_i = j; } }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const uint32_t ob = __shfl_xor(best, o, 64), oi = __shfl_xor(best_i, o, 64);
            if (ob > best || (ob == best && oi < best_i)) { best = ob; best_i = oi; }
        }
        const uint32_t pp = best == 0 ? 0u : best_i;
        uint32_t cur = pp + W / 2 < coding ? pp + W / 2 : coding;
        const uint32_t n = W < cur ? W : cur;
        cur -= n;
        double gate = 0.0;
        if (n == 0) { if (l == 0) atomicExch(A.error, RSQC_ERR_EMPTY_MEDIAN); }
        else if (n == 1) gate = (double)C[cur];
        else {
            const uint32_t mid = (n - 1) / 2;
            gate = (n & 1u) ? ((double)C[cur + mid] + (double)C[cur + mid + 1]) / 2.0 : (double)C[cur + mid];
        }
        if (n != 0 && gate >= 100.0) {
            // 5th percentile of the non-zero coverage: order statistic R of the whole vector
            unsigned long long nz = 0;
            for (uint32_t j = l; j < coding; j += 64) nz += C[j] != 0u;
            nz = wave_sum_u64(nz);
            const uint32_t nnz = (uint32_t)nz;
            uint32_t R = (coding - nnz) + (uint32_t)((double)nnz * 0.05);
            uint32_t prefix = 0, pmask = 0;
            for (int shift = 24; shift >= 0; shift -= 8) {          // MSB-first radix select
                for (int x = l; x < 256; x += 64) s_hist[wv][x] = 0;
                __threadfence_block();
                for (uint32_t j = l; j < coding; j += 64) {
                    const uint32_t v = C[j];
                    if ((v & pmask) == prefix) atomicAdd(&s_hist[wv][(v >> shift) & 0xFF], 1u);
                }
                __threadfence_block();
                // lane x scans 4 bins; find the bin holding rank R
                uint32_t h0 = s_hist[wv][4 * l], h1 = s_hist[wv][4 * l + 1], h2 = s_hist[wv][4 * l + 2], h3 = s_hist[wv][4 * l + 3];
                const uint32_t tot = h0 + h1 + h2 + h3;
                const uint32_t inc = wave_inclusive_scan_u32(tot);
                const uint32_t exc = inc - tot;
                const uint64_t here = __ballot(R >= exc && R < inc);
                const int wl = __ffsll((unsigned long long)here) - 1;
                uint32_t digit = 0, rbase = 0;
                if (l == wl) {
                    uint32_t c0 = exc;
                    if (R < c0 + h0) { digit = 4 * l; rbase = c0; }
                    else if (R < c0 + h0 + h1) { digit = 4 * l + 1; rbase = c0 + h0; }
                    else if (R < c0 + h0 + h1 + h2) { digit = 4 * l + 2; rbase = c0 + h0 + h1; }
                    else { digit = 4 * l + 3; rbase = c0 + h0 + h1 + h2; }
                }
                digit = __shfl(digit, wl, 64); rbase = __shfl(rbase, wl, 64);
                R -= rbase;
                prefix |= digit << shift; pmask |= 0xFFu << shift;
            }
            const uint32_t lower = prefix;
            // trim leading / trailing entries <= lower (in place in the reference: Q14)
            uint32_t first_gt = 0xFFFFFFFFu, last_gt = 0;
            bool any = false;
            for (uint32_t j = l; j < coding; j += 64) if (C[j] > lower) { if (!any) first_gt = j; last_gt = j; any = true; }
            first_gt = wave_min_u32(first_gt);
            last_gt = wave_max_u32(any ? last_gt + 1 : 0u);
            if (first_gt == 0xFFFFFFFFu) { v0 = coding; v1 = coding; } else { v0 = first_gt; v1 = last_gt; }
            const uint32_t tlen = v1 - v0;
            if (tlen >= A.bias_gene_length) {
                // left window [OFF, min(OFF+W, tlen)), right window [tlen-W-OFF, tlen-OFF)
                const uint32_t lhi = OFF + W < tlen ? OFF + W : tlen;
                const uint32_t nl = OFF < lhi ? lhi - OFF : 0u;
                uint32_t nr = 0, rlo = 0;
                if ((uint64_t)W + OFF <= tlen) { rlo = tlen - W - OFF; nr = W; }
                for (uint32_t j = l; j < nl; j += 64) s_win[wv][0][j] = C[v0 + OFF + j];
                for (uint32_t j = l; j < nr; j += 64) s_win[wv][1][j] = C[v0 + rlo + j];
                __threadfence_block();
                double ml = 0.0, mr = 0.0;
                const bool okl = window_median(s_win[wv][0], nl, &ml);
                const bool okr = window_median(s_win[wv][1], nr, &mr);
                if (!(okl && okr)) { if (l == 0) atomicExch(A.error, RSQC_ERR_EMPTY_MEDIAN); }
                else if (l == 0) {
                    const bool fwd = (A.gene_flags[gene] & RSQC_FF_STRAND_MASK) == RSQC_STRAND_FORWARD;
                    A.bias3[gene] = (unsigned long long)(fwd ? mr : ml);      // unsigned long += double: truncation
                    A.bias5[gene] = (unsigned long long)(fwd ? ml : mr);
                }
            }
        }
    }
    // (4) gene mean / std / CV on V = C[v0..v1) with MASK bases removed at both ends
    {
        const uint32_t len = v1 - v0;
        uint32_t a = v0, bnd = v1;
        if (MASK) {
            if (len > 2 * (uint64_t)MASK) { a = v0 + MASK; bnd = v1 - MASK; } else { a = bnd = v0; }
        }
        if (bnd > a) {
            const double size = (double)(bnd - a);
            unsigned long long s = 0;
            for (uint32_t j = a + l; j < bnd; j += 64) s += C[j];
            s = wave_sum_u64(s);
            const double mean = (double)s / size;
            double q = 0.0;
            for (uint32_t j = a + l; j < bnd; j += 64) { const double d = (double)C[j] - mean; q += d * d; }
            q = wave_sum_f64(q);
            const double sd = sqrt(q / size);
            if (l == 0) { A.g_valid[gene] = 1; A.g_mean[gene] = mean; A.g_std[gene] = sd; A.g_cv[gene] = sd / mean; }
        } else if (l == 0) A.g_valid[gene] = 0;
    }
}

// ------------------------------------------------------------------ launch wrappers
void launch_classify(hipStream_t s, int grid, const DevAnnotation &a, const DevParams &p, const DevBatch &b,
                     const DevAccum &acc) {
    hipLaunchKernelGGL(classify_count_kernel, dim3(grid), dim3(RSQC_K1_THREADS), 0, s, a, p, b, acc);
}
void launch_classify_slow(hipStream_t s, const DevAnnotation &a, const DevParams &p, const DevBatch &b,
                          const DevAccum &acc) {
    hipLaunchKernelGGL(classify_slow_kernel, dim3(64), dim3(64), 0, s, a, p, b, acc);
}
void launch_read_length(hipStream_t s, const DevAnnotation &a, const DevParams &p, const DevBatch &b,
                        const DevAccum &acc) {
    hipLaunchKernelGGL(read_length_kernel, dim3(1), dim3(64), 0, s, a, p, b, acc);
}
void launch_dedup(hipStream_t s, const uint32_t *pair_gene, const uint64_t *pair_hash, uint32_t n_pairs,
                  const uint64_t *tab_off, const uint32_t *tab_cap, unsigned long long *table,
                  unsigned long long *gene_frag) {
    if (!n_pairs) return;
    int grid = (int)((n_pairs + 255) / 256);
    if (grid > 256 * 8) grid = 256 * 8;
    hipLaunchKernelGGL(dedup_insert_kernel, dim3(grid), dim3(256), 0, s, pair_gene, pair_hash, n_pairs, tab_off,
                       tab_cap, table, gene_frag);
}
void launch_gene_coverage(hipStream_t s, const GeneCovArgs &A) {
    if (A.n_listed <= 0) return;
    hipLaunchKernelGGL(gene_coverage_kernel, dim3((A.n_listed + 3) / 4), dim3(256), 0, s, A);
}

}  // namespace rsqc
