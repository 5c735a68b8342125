// rsqc_k3.h -- K3, the end-of-file coverage stage (src/Metrics.cpp:132-151,160-235,265-337): per gene, the difference array
// becomes the stitched coverage vector, then per-exon CV, the 3'/5' bias windows and the gene's mean / std / CV.
// Device code only, included by rsqc_kernels.hip -- and, unmodified, by the host SIMT emulation of the tests
// (tests/hostemu/k3_emu.cpp), which runs it on the CPU against the oracle.
#pragma once

namespace rsqc {

// ------------------------------------------------------------------ K3
// barrier of a T-thread workgroup; a one-wave "workgroup" only needs its LDS traffic ordered (it is executed in order)
template <int T> __device__ __forceinline__ void k3_sync() {
    if constexpr (T > 64) __syncthreads();
    else { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup"); __builtin_amdgcn_wave_barrier(); }
}
// One workgroup per gene, longest genes first.  cov[] holds the per-base DIFFERENCE array of the
// gene's exons, contiguous in exonsForGene order (+1 pad slot), so a plain prefix sum yields the
// stitched transcript vector of computeCoverage (src/Metrics.cpp:306-308).
// The stage is a chain of ~40 short data-parallel passes separated by workgroup barriers, so its
// time is (genes / genes in flight) x (barriers x barrier cost): the workgroup is sized to the gene.
// Most genes have a few thousand coding bases and run as ONE WAVE each (T = 64: barriers degenerate
// to in-order LDS traffic and thousands of genes are in flight); longer ones get 256 or 1024 threads.
template <int T_, int WIN_, class CovT_, int LDSCAP_>
struct K3Shared {
    static constexpr int T = T_, WIN = WIN_, W = T_ / 64, LDSCAP = LDSCAP_;
    CovT_ covbuf[LDSCAP_ > 0 ? LDSCAP_ : 1];         // the gene's coverage vector when it fits (see the kernel)
    unsigned long long u64[W];
    double f64[W];
    uint32_t u32a[W], u32b[W];
    uint32_t hist[256];
    uint32_t win[2][WIN];
    uint32_t bc_u32[4]; double bc_f64[2];
};

template <class SH> __device__ __forceinline__ unsigned long long block_sum_u64(unsigned long long v, SH &S) {
    constexpr int T = SH::T;
    v = wave_sum(v);
    k3_sync<T>();
    if (lane_id() == 0) S.u64[threadIdx.x >> 6] = v;
    k3_sync<T>();
    unsigned long long t = 0;
#pragma unroll
    for (int w = 0; w < (T / 64); ++w) t += S.u64[w];
    return t;
}
template <class SH> __device__ __forceinline__ double block_sum_f64(double v, SH &S) {
    constexpr int T = SH::T;
    v = wave_sum(v);
    k3_sync<T>();
    if (lane_id() == 0) S.f64[threadIdx.x >> 6] = v;
    k3_sync<T>();
    double t = 0;
#pragma unroll
    for (int w = 0; w < (T / 64); ++w) t += S.f64[w];
    return t;
}
template <class SH> __device__ __forceinline__ uint32_t block_min_u32(uint32_t v, SH &S) {
    constexpr int T = SH::T;
    v = wave_min_u32(v);
    k3_sync<T>();
    if (lane_id() == 0) S.u32a[threadIdx.x >> 6] = v;
    k3_sync<T>();
    uint32_t t = 0xFFFFFFFFu;
#pragma unroll
    for (int w = 0; w < (T / 64); ++w) t = S.u32a[w] < t ? S.u32a[w] : t;
    return t;
}
template <class SH> __device__ __forceinline__ uint32_t block_max_u32(uint32_t v, SH &S) {
    constexpr int T = SH::T;
    v = wave_max_u32(v);
    k3_sync<T>();
    if (lane_id() == 0) S.u32a[threadIdx.x >> 6] = v;
    k3_sync<T>();
    uint32_t t = 0;
#pragma unroll
    for (int w = 0; w < (T / 64); ++w) t = S.u32a[w] > t ? S.u32a[w] : t;
    return t;
}

// quirky computeMedian (src/Metrics.h:147-160) of a window held in LDS (unsorted): the two middle
// order statistics are found by rank counting.  Called by the whole block; result broadcast.
template <class SH> __device__ bool window_median(const uint32_t *w, uint32_t n, double *out, SH &S) {
    constexpr int T = SH::T;
    if (n == 0) return false;
    if (n == 1) { *out = (double)w[0]; return true; }
    const uint32_t mid = (n - 1) / 2;
    k3_sync<T>();
    for (uint32_t i = threadIdx.x; i < n; i += T) {
        const uint32_t v = w[i];
        uint32_t rank = 0;
        for (uint32_t j = 0; j < n; ++j) { const uint32_t u = w[j]; rank += (u < v || (u == v && j < i)) ? 1u : 0u; }
        if (rank == mid) S.bc_u32[0] = v;
        if (rank == mid + 1) S.bc_u32[1] = v;
    }
    k3_sync<T>();
    *out = (n & 1u) ? ((double)S.bc_u32[0] + (double)S.bc_u32[1]) / 2.0 : (double)S.bc_u32[0];
    return true;
}

template <int T, int WIN, class CovT, int LDSCAP>
__global__ void __launch_bounds__(T)
gene_coverage_kernel(GeneCovArgs A, uint32_t first) {
    __shared__ K3Shared<T, WIN, CovT, LDSCAP> S;
    const int tid = (int)threadIdx.x;
    const int l = lane_id();
    const int wv = tid >> 6;
    const int gene = (int)A.gene_order[first + blockIdx.x];
    if (!A.gene_owned[gene]) return;
    const uint32_t coding = A.gene_coding[gene];
    const uint32_t e0 = A.ge_off[gene], e1 = A.ge_off[gene + 1], n_ex = e1 - e0;
    // Every later pass re-reads the coverage vector; a gene that fits keeps it in LDS (the difference array is
    // read from memory once and never written back: nothing downstream needs it) -- as 32-bit values, or as
    // 16-bit values for the longest genes (160 KB of LDS hold 73 k bases) as long as no base is covered 65 536
    // times or more.  Otherwise the scan runs in place in memory and the passes rely on unrolled, independent loads.
    uint32_t *const D = A.cov + A.gene_cov_off[gene];
    bool in_lds = LDSCAP > 0 && coding <= (uint32_t)LDSCAP;
    auto Cget = [&](uint32_t j) -> uint32_t { return in_lds ? (uint32_t)S.covbuf[j] : D[j]; };
    const uint32_t MASK = A.mask;
    const uint32_t W = (uint32_t)A.bias_window, OFF = (uint32_t)A.bias_offset;
    const double qnan = __longlong_as_double(0x7ff8000000000000ll);

    if (A.gene_reads[gene] == 0ull) {
        // never counted: all-zero coverage -> mean 0, std 0, cv NaN; no exon CV; the bias gate reads zeros
        if (tid == 0) {
            if (coding >= A.bias_gene_length) {
                const uint32_t cur = W / 2 < coding ? W / 2 : coding;
                if ((W < cur ? W : cur) == 0) atomicExch(A.error, RSQC_ERR_EMPTY_MEDIAN);
            }
            const bool pushed = MASK ? coding > 2 * (uint64_t)MASK : coding > 0;
            A.g_valid[gene] = pushed ? 1 : 0;
            A.g_mean[gene] = 0.0; A.g_std[gene] = 0.0; A.g_cv[gene] = pushed ? qnan : 0.0;
        }
        return;
    }
#ifdef RSQC_K1_PROF
    if (first == 0 && threadIdx.x == 0) for (int k = 0; k < 12; ++k) s_fin_stamp[k] = 0ull;
#endif
    RSQC_FIN_STAMP(0);
    // (1) difference array -> coverage: block-wide inclusive scan.  A round covers T x 16 bases; wave w of the round takes 1024
    //     consecutive bases as 16 rows of 64 (lane l loads base row * 64 + l: one 256-byte line run per instruction -- 16
    //     consecutive bases per LANE cost the texture addresser 64 separate lines per instruction and made the scan of a long gene
    //     the longest stage of the kernel), scans each row across the lanes and chains the rows; the next round's rows are in flight
    //     while this one is scanned.
    auto scan = [&]() -> bool {                        // returns false when a value does not fit the LDS cell type
        constexpr int PER = 16;
        uint32_t carry = 0, vmax = 0;
        uint32_t v[PER], nx[PER];
        auto load_round = [&](uint32_t base, uint32_t (&o)[PER]) {
            const uint32_t j0 = base + (uint32_t)wv * (64u * PER) + (uint32_t)l;
#pragma unroll
            for (int k = 0; k < PER; ++k) { const uint32_t j = j0 + (uint32_t)k * 64u; o[k] = j < coding ? D[j] : 0u; }
        };
        load_round(0u, v);
        for (uint32_t base = 0; base < coding; base += T * PER) {
            if (base + T * PER < coding) load_round(base + T * PER, nx);
            // rows scanned across the lanes, each row offset by the total of the rows before it
            uint32_t run = 0;
#pragma unroll
            for (int k = 0; k < PER; ++k) {
                const uint32_t inc = wave_inclusive_scan_u32(v[k]);
                v[k] = inc + run;
                run += lane_value(inc, 63);
            }
            k3_sync<T>();
            if (l == 0) S.u32b[wv] = run;
            k3_sync<T>();
            uint32_t before = carry, total = 0;
#pragma unroll
            for (int w = 0; w < (T / 64); ++w) { const uint32_t t = S.u32b[w]; if (w < wv) before += t; total += t; }
            const uint32_t j0 = base + (uint32_t)wv * (64u * PER) + (uint32_t)l;
#pragma unroll
            for (int k = 0; k < PER; ++k) {
                const uint32_t j = j0 + (uint32_t)k * 64u;
                if (j < coding) {
                    const uint32_t x = v[k] + before;
                    vmax = x > vmax ? x : vmax;
                    if (in_lds) S.covbuf[j] = (CovT)x; else D[j] = x;
                }
            }
            carry += total;
#pragma unroll
            for (int k = 0; k < PER; ++k) v[k] = nx[k];
        }
        if constexpr (sizeof(CovT) < 4) { if (in_lds) return block_max_u32(vmax, S) <= (uint32_t)(CovT)~(CovT)0; }
        return true;
    };
    if (!scan()) { in_lds = false; RSQC_FIN_STAMP(1); scan(); }           // (uniform: block_max_u32 broadcasts) too deep for 16 bits: in memory
    __threadfence_block();
    k3_sync<T>();
    RSQC_FIN_STAMP(2);
#ifdef RSQC_K1_PROF
    if (first == 0 && threadIdx.x == 0) { s_fin_stamp[9] = coding; s_fin_stamp[10] = n_ex; s_fin_stamp[11] = in_lds; }
#endif
    // (2) per-exon CV over transcript positions [MASK, coding-MASK) (src/Metrics.cpp:267-305): one wave per
    //     exon at a time (an exon's bases are contiguous in C): register sums, no shared accumulators.  The rows of the next
    //     64 exons of a wave are gathered by its lanes in one go (exon k of the wave in lane k: two dependent gathers per 64
    //     exons instead of per exon) and handed to the wave one by one.
    {
        const uint64_t lo_t = MASK, hi_t = coding > MASK ? coding - MASK : 0;
        if (hi_t > lo_t) {
            constexpr uint32_t W = (uint32_t)(T / 64);
            const uint32_t gcov = A.gene_cov_off[gene];
            for (uint32_t k0 = (uint32_t)wv; k0 < n_ex; k0 += 64u * W) {
                const uint32_t mine = k0 + (uint32_t)l * W;
                uint32_t r_t0 = 0, r_len = 0, r_id = 0;
                if (mine < n_ex) {
                    const uint32_t row = A.ge_row[e0 + mine];
                    const ExonRow er = A.ex[row];
                    r_t0 = er.cov - gcov; r_len = (uint32_t)(er.end - er.start + 1); r_id = A.ex_id[row];
                }
                const uint32_t left = (n_ex - k0 + W - 1) / W;
                const uint32_t cnt = left < 64u ? left : 64u;
                for (uint32_t i = 0; i < cnt; ++i) {
                    const uint32_t t0 = lane_value(r_t0, (int)i), len = lane_value(r_len, (int)i), id = lane_value(r_id, (int)i);
                    const uint64_t a0 = t0 > lo_t ? t0 : lo_t, b0 = (uint64_t)t0 + len < hi_t ? (uint64_t)t0 + len : hi_t;
                    if (b0 > a0) {
                        const double size = (double)(b0 - a0);
                        unsigned long long sm = 0;
#pragma unroll 4
                        for (uint32_t j = (uint32_t)a0 + (uint32_t)l; j < b0; j += 64) sm += Cget(j);
                        const double mean = (double)wave_sum(sm) / size;
                        double q = 0.0;
#pragma unroll 4
                        for (uint32_t j = (uint32_t)a0 + (uint32_t)l; j < b0; j += 64) { const double d = (double)Cget(j) - mean; q += d * d; }
                        const double cv = sqrt(wave_sum(q) / size) / mean;
                        if (l == 0 && !(isnan(cv) || isinf(cv))) { A.e_cv[id] = cv; A.e_cv_valid[id] = 1; }
                    }
                }
            }
        }
    }
#ifdef RSQC_K1_PROF
    k3_sync<T>();
    RSQC_FIN_STAMP(3);
#endif
    // (3) bias (src/Metrics.cpp:160-235) on the stitched, unmasked coverage vector [0, coding)
    uint32_t v0 = 0, v1 = coding;          // the (possibly trimmed) vector the gene stats use (Q14)
    if (coding >= A.bias_gene_length) {
        uint32_t best = 0, best_i = 0xFFFFFFFFu;
        unsigned long long nz = 0;
#pragma unroll 8
        for (uint32_t j = tid; j < coding; j += T) { const uint32_t v = Cget(j); nz += v != 0u; if (v > best) { best = v; best_i = j; } }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const uint32_t ob = __shfl_xor(best, o, 64), oi = __shfl_xor(best_i, o, 64);
            if (ob > best || (ob == best && oi < best_i)) { best = ob; best_i = oi; }
        }
        k3_sync<T>();
        if (l == 0) { S.u32a[wv] = best; S.u32b[wv] = best_i; }
        k3_sync<T>();
        best = 0; best_i = 0xFFFFFFFFu;
#pragma unroll
        for (int w = 0; w < (T / 64); ++w) { const uint32_t ob = S.u32a[w], oi = S.u32b[w]; if (ob > best || (ob == best && oi < best_i)) { best = ob; best_i = oi; } }
        const uint32_t pp = best == 0 ? 0u : best_i;
        uint32_t cur = pp + W / 2 < coding ? pp + W / 2 : coding;
        const uint32_t n = W < cur ? W : cur;
        cur -= n;
        double gate = 0.0;
        if (n == 0) { if (tid == 0) atomicExch(A.error, RSQC_ERR_EMPTY_MEDIAN); }
        else if (n == 1) gate = (double)Cget(cur);
        else {
            const uint32_t mid = (n - 1) / 2;
            gate = (n & 1u) ? ((double)Cget(cur + mid) + (double)Cget(cur + mid + 1)) / 2.0 : (double)Cget(cur + mid);
        }
        RSQC_FIN_STAMP(4);
        if (n != 0 && gate >= 100.0) {
            // 5th percentile of the non-zero coverage = order statistic R of the whole vector
            const uint32_t nnz = (uint32_t)block_sum_u64(nz, S);
            uint32_t R = (coding - nnz) + (uint32_t)((double)nnz * 0.05);
            uint32_t prefix = 0, pmask = 0;
            // MSB-first radix select; bytes above the top non-zero byte of the maximum are zero for every entry
            const int top = best >> 24 ? 24 : best >> 16 ? 16 : best >> 8 ? 8 : 0;
            pmask = top == 24 ? 0u : 0xFFFFFFFFu << (top + 8);
            for (int shift = top; shift >= 0; shift -= 8) {
                k3_sync<T>();
                for (int h = tid; h < 256; h += T) S.hist[h] = 0;
                k3_sync<T>();
#pragma unroll 8
                for (uint32_t j = tid; j < coding; j += T) {            // (same-bin LDS atomics of a wave serialise inside ONE
                    const uint32_t v = Cget(j);                            //  instruction, ~1 cycle per lane: cheaper than merging them)
                    if ((v & pmask) == prefix) atomicAdd(&S.hist[(v >> shift) & 0xFF], 1u);
                }
                k3_sync<T>();
                if (wv == 0) {                                       // wave 0: lane x scans 4 bins
                    const uint32_t h0 = S.hist[4 * l], h1 = S.hist[4 * l + 1], h2 = S.hist[4 * l + 2], h3 = S.hist[4 * l + 3];
                    const uint32_t tot = h0 + h1 + h2 + h3;
                    const uint32_t inc = wave_inclusive_scan_u32(tot);
                    const uint32_t exc = inc - tot;
                    if (R >= exc && R < inc) {
                        uint32_t digit, rbase;
                        if (R < exc + h0) { digit = 4 * l; rbase = exc; }
                        else if (R < exc + h0 + h1) { digit = 4 * l + 1; rbase = exc + h0; }
                        else if (R < exc + h0 + h1 + h2) { digit = 4 * l + 2; rbase = exc + h0 + h1; }
                        else { digit = 4 * l + 3; rbase = exc + h0 + h1 + h2; }
                        S.bc_u32[2] = digit; S.bc_u32[3] = rbase;
                    }
                }
                k3_sync<T>();
                R -= S.bc_u32[3];
                prefix |= S.bc_u32[2] << shift; pmask |= 0xFFu << shift;
            }
            const uint32_t lower = prefix;
            RSQC_FIN_STAMP(5);
            // trim leading / trailing entries <= lower (in place in the reference: Q14)
            uint32_t first_gt = 0xFFFFFFFFu, last_gt = 0;
#pragma unroll 8
            for (uint32_t j = tid; j < coding; j += T) if (Cget(j) > lower) { if (first_gt == 0xFFFFFFFFu) first_gt = j; last_gt = j + 1; }
            first_gt = block_min_u32(first_gt, S);
            last_gt = block_max_u32(last_gt, S);
            if (first_gt == 0xFFFFFFFFu) { v0 = coding; v1 = coding; } else { v0 = first_gt; v1 = last_gt; }
            const uint32_t tlen = v1 - v0;
            RSQC_FIN_STAMP(6);
            if (tlen >= A.bias_gene_length) {
                // left window [OFF, min(OFF+W, tlen)), right window [tlen-W-OFF, tlen-OFF)
                const uint32_t lhi = OFF + W < tlen ? OFF + W : tlen;
                const uint32_t nl = OFF < lhi ? lhi - OFF : 0u;
                uint32_t nr = 0, rlo = 0;
                if ((uint64_t)W + OFF <= tlen) { rlo = tlen - W - OFF; nr = W; }
                k3_sync<T>();
                for (uint32_t j = tid; j < nl; j += T) S.win[0][j] = Cget(v0 + OFF + j);
                for (uint32_t j = tid; j < nr; j += T) S.win[1][j] = Cget(v0 + rlo + j);
                k3_sync<T>();
                double ml = 0.0, mr = 0.0;
                const bool okl = window_median(S.win[0], nl, &ml, S);
                const bool okr = window_median(S.win[1], nr, &mr, S);
                if (!(okl && okr)) { if (tid == 0) atomicExch(A.error, RSQC_ERR_EMPTY_MEDIAN); }
                else if (tid == 0) {
                    const bool fwd = (A.gene_flags[gene] & RSQC_FF_STRAND_MASK) == RSQC_STRAND_FORWARD;
                    A.bias3[gene] = (unsigned long long)(fwd ? mr : ml);      // unsigned long += double: truncation
                    A.bias5[gene] = (unsigned long long)(fwd ? ml : mr);
                }
            }
        }
    }
    RSQC_FIN_STAMP(7);
    // (4) gene mean / std / CV on positions [v0, v1) with MASK bases removed at both ends
    {
        const uint32_t len = v1 - v0;
        uint32_t a = v0, bnd = v1;
        if (MASK) {
            if (len > 2 * (uint64_t)MASK) { a = v0 + MASK; bnd = v1 - MASK; } else { a = bnd = v0; }
        }
        if (bnd > a) {
            const double size = (double)(bnd - a);
            unsigned long long sm = 0;
#pragma unroll 8
            for (uint32_t j = a + tid; j < bnd; j += T) sm += Cget(j);
            const double mean = (double)block_sum_u64(sm, S) / size;
            double q = 0.0;
#pragma unroll 8
            for (uint32_t j = a + tid; j < bnd; j += T) { const double d = (double)Cget(j) - mean; q += d * d; }
            const double sd = sqrt(block_sum_f64(q, S) / size);
            if (tid == 0) { A.g_valid[gene] = 1; A.g_mean[gene] = mean; A.g_std[gene] = sd; A.g_cv[gene] = sd / mean; }
        } else if (tid == 0) { A.g_valid[gene] = 0; A.g_mean[gene] = 0.0; A.g_std[gene] = 0.0; A.g_cv[gene] = 0.0; }
    }
    RSQC_FIN_STAMP(8);
#ifdef RSQC_K1_PROF
    if (first == 0 && threadIdx.x == 0) {
        const unsigned long long tot = s_fin_stamp[8] - s_fin_stamp[0];
        if (atomicMax(&g_fin_prof[31], tot) < tot) {
            for (int k = 0; k < 9; ++k) g_fin_prof[k] = s_fin_stamp[k];
            g_fin_prof[28] = s_fin_stamp[9]; g_fin_prof[29] = s_fin_stamp[10]; g_fin_prof[30] = s_fin_stamp[11]; g_fin_prof[27] = blockIdx.x;
        }
    }
#endif
}

}  // namespace rsqc
