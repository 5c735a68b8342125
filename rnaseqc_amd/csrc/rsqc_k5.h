// rsqc_k5.h -- K5, device code only: the kernels of the fragment-size sampler (src/Expression.cpp:482-540) and of the mate pairing
// of the fragment GC statistics (src/Expression.cpp:459-477).  Included by rsqc_fragsize.hip (which holds the description and
// the host side) and, unmodified, by the host SIMT emulation of the tests (tests/hostemu/k5_emu.cpp).
#pragma once

namespace rsqc {

constexpr uint32_t PB_MEAN = 384;            // candidates per bucket on average (the name hashes are fmix64 outputs: Poisson; <= 512, the hashed pairing's limit, in all but one bucket in 10^9)
constexpr uint32_t PB_CAP = 2048;            // LDS slots of the per-bucket sort; a fuller bucket is listed and sorted in memory (pair_bucket_big_*)
constexpr int PB_THREADS = 256;
constexpr uint32_t SIZE_TABLE = 1u << 20;    // direct histogram of |isize| below this; larger values are listed

__device__ __forceinline__ uint32_t pair_bucket_of(uint64_t qhash, uint32_t n_buckets) {
    return (uint32_t)(((qhash >> 32) * (uint64_t)n_buckets) >> 32);
}
__global__ void pair_bucket_count_kernel(const uint64_t *qhash, uint32_t n, uint32_t n_buckets, uint32_t *count) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) atomicAdd(&count[pair_bucket_of(qhash[i], n_buckets)], 1u);
}
// exclusive sums of the bucket counts (one workgroup; n_buckets is a few thousand); off[n_buckets] = n; cursors = offsets
// Buckets fuller than the LDS sort are LISTED (big[0] = how many, big[1 ..] = which; round 4 failed the run on the first one): all
// records of one QNAME share a bucket, so a file whose reads carry one name -- stripped or constant names -- is one bucket;
// pair_bucket_big sorts those in memory.
constexpr uint32_t PB_BIG_MAX = 1024;        // listed buckets; more of them (a caller's degenerate hash) is RSQC_ERR_CAPACITY
__global__ void __launch_bounds__(1024) pair_bucket_scan_kernel(const uint32_t *count, uint32_t n_buckets, uint32_t *off, uint32_t *cursor, uint32_t *big, int *error) {
    __shared__ uint32_t part[1024];
    const uint32_t per = (n_buckets + 1023u) / 1024u, lo = threadIdx.x * per, hi = lo + per < n_buckets ? lo + per : n_buckets;
    uint32_t s = 0;
    for (uint32_t b = lo; b < hi; ++b) {
        s += count[b];
        if (count[b] > PB_CAP) { const uint32_t at = atomicAdd(&big[0], 1u); if (at < PB_BIG_MAX) big[1u + at] = b; else atomicExch(error, RSQC_ERR_CAPACITY); }
    }
    part[threadIdx.x] = s;
    __syncthreads();
    for (uint32_t o = 1; o < 1024u; o <<= 1) {
        const uint32_t t = threadIdx.x >= o ? part[threadIdx.x - o] : 0u;
        __syncthreads();
        part[threadIdx.x] += t;
        __syncthreads();
    }
    uint32_t run = part[threadIdx.x] - s;
    for (uint32_t b = lo; b < hi; ++b) { off[b] = run; cursor[b] = run; run += count[b]; }
    if (threadIdx.x == 1023) off[n_buckets] = part[1023];
}
__global__ void pair_bucket_scatter_kernel(const uint64_t *qhash, uint32_t n, uint32_t n_buckets, uint32_t *cursor, uint32_t *perm) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) perm[atomicAdd(&cursor[pair_bucket_of(qhash[i], n_buckets)], 1u)] = i;
}

// The candidates of one bucket in LDS, ordered by (name hash, file index).  Returns the bucket's size (0 when it overflowed:
// the scan kernel has raised the error).  Slots behind the bucket's entries hold the largest key and sort to the end.
// A NAME is 96 bits here as in the fragment de-duplication (K4): the 64-bit hash the buckets are formed on AND the second hash
// (rsqc_batch.qhash2) -- two names that share the first and differ in the second are two groups of the sort.
struct PairBucket { uint64_t q[PB_CAP], f[PB_CAP]; uint32_t e[PB_CAP], h[PB_CAP]; };
__device__ __forceinline__ uint32_t pair_bucket_sorted(PairBucket &S, const uint64_t *qhash, const uint32_t *h2, const uint64_t *file_index, const uint32_t *off, const uint32_t *perm) {
    const uint32_t lo = off[blockIdx.x], m = off[blockIdx.x + 1] - lo;
    if (m == 0 || m > PB_CAP) return 0u;
    uint32_t slots = 2;
    while (slots < m) slots <<= 1;
    for (uint32_t i = threadIdx.x; i < slots; i += PB_THREADS) {
        if (i < m) { const uint32_t c = perm[lo + i]; S.q[i] = qhash[c]; S.h[i] = h2 ? h2[c] : 0u; S.f[i] = file_index[c]; S.e[i] = c; }
        else { S.q[i] = ~0ull; S.h[i] = 0xFFFFFFFFu; S.f[i] = ~0ull; S.e[i] = 0xFFFFFFFFu; }
    }
    __syncthreads();
    for (uint32_t k = 2; k <= slots; k <<= 1)
        for (uint32_t j = k >> 1; j > 0; j >>= 1) {
            for (uint32_t i = threadIdx.x; i < slots; i += PB_THREADS) {
                const uint32_t x = i ^ j;
                if (x > i) {
                    const bool up = (i & k) == 0;
                    const uint64_t qi = S.q[i], qx = S.q[x], fi = S.f[i], fx = S.f[x];
                    const uint32_t hi = S.h[i], hx = S.h[x];
                    const bool greater = qi > qx || (qi == qx && (hi > hx || (hi == hx && fi > fx)));
                    if (greater == up) { S.q[i] = qx; S.q[x] = qi; S.h[i] = hx; S.h[x] = hi; S.f[i] = fx; S.f[x] = fi; const uint32_t t = S.e[i]; S.e[i] = S.e[x]; S.e[x] = t; }
                }
            }
            __syncthreads();
        }
    return m;
}

// The same grouping WITHOUT the sort, for the bucket every real file produces (round 6, third session): a name has one or two candidates --
// a read and its mate -- and then "the records of a name in file order" is a comparison of two file indices.  The bucket's candidates go
// through an LDS set keyed by a 64-bit mix of the 96-bit name; a slot counts its members and keeps the first two; `pair(first, second)`
// (file order) is then called once per two-member slot by one thread.  The bitonic network above is 55 barrier-separated stages for
// 1 024 slots (58 us per bucket: gc_replay_kernel 5.9 ms per 40 M candidates); this is three barriers.
// EXACT, or not taken: a slot with a third member (a name with three or more candidates, two names whose mixes collide with a third record),
// two members whose second hashes differ (a collision of the mix), or a bucket beyond half the set make the function return false BEFORE any call of
// `pair`, and the caller sorts the bucket as before.
#if defined(RSQC_WAVE_EMU)
static unsigned long long g_k5_hashed_buckets = 0, g_k5_sorted_buckets = 0;   // (test harness: buckets paired through the set / handed to the sort)
#endif
constexpr uint32_t PH_SLOTS = 1024;
// (the members' second hashes and file indices ride in LDS beside their candidate numbers: two members of a slot share the 64-bit mix, so equal
//  second hashes make them one name in all 96 bits, and their order needs no gather -- going back to the candidate arrays for either made the
//  GC statistics' replay slower than the sort, call r6ag)
struct PairHash { unsigned long long key[PH_SLOTS]; uint32_t cnt[PH_SLOTS]; uint32_t mem[PH_SLOTS][2], mh[PH_SLOTS][2]; unsigned long long mf[PH_SLOTS][2]; uint32_t fail; };
union PairScratch { PairBucket S; PairHash H; };
// pair(slot, first, second): the slot's members 0 / 1 in file order (H.mem[slot][first] is the candidate seen first in the file)
template <class F>
__device__ __forceinline__ bool pair_bucket_hashed(PairHash &H, const uint64_t *qhash, const uint32_t *h2, const uint64_t *file_index, const uint32_t *off,
                                                   const uint32_t *perm, F &&pair) {
    const uint32_t lo = off[blockIdx.x], m = off[blockIdx.x + 1] - lo;
    if (m == 0) return true;
#if defined(RSQC_WAVE_EMU)
    if (threadIdx.x == 0 && m > PH_SLOTS / 2) ++g_k5_sorted_buckets;
#endif
    if (m > PH_SLOTS / 2) return false;
    uint32_t slots = 64;
    while (slots < 2 * m) slots <<= 1;
    const uint32_t mask = slots - 1;
    for (uint32_t i = threadIdx.x; i < slots; i += PB_THREADS) { H.key[i] = 0ull; H.cnt[i] = 0u; }
    if (threadIdx.x == 0) H.fail = 0u;
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < m; i += PB_THREADS) {
        const uint32_t c = perm[lo + i];
        const uint32_t h = h2 ? h2[c] : 0u;
        const unsigned long long f = file_index[c];
        unsigned long long k = qhash[c] ^ ((unsigned long long)h * 0x9E3779B97F4A7C15ull);
        if (k == 0ull) k = 1ull;
        uint32_t slot = (uint32_t)((k * 0x9E3779B97F4A7C15ull) >> 40) & mask;
        bool placed = false;
        for (uint32_t probe = 0; probe < slots; ++probe) {
            const unsigned long long old = atomicCAS(&H.key[slot], 0ull, k);
            if (old == 0ull || old == k) {
                const uint32_t r = atomicAdd(&H.cnt[slot], 1u);
                if (r < 2u) { H.mem[slot][r] = c; H.mh[slot][r] = h; H.mf[slot][r] = f; } else H.fail = 1u;
                placed = true;
                break;
            }
            slot = (slot + 1) & mask;
        }
        if (!placed) H.fail = 1u;                                          // (cannot happen at load <= 0.5)
    }
    __syncthreads();
    for (uint32_t sl = threadIdx.x; sl < slots; sl += PB_THREADS)
        if (H.cnt[sl] == 2u && H.mh[sl][0] != H.mh[sl][1]) H.fail = 1u;   // one mix, two second hashes: two names (never seen outside crafted input)
    __syncthreads();
#if defined(RSQC_WAVE_EMU)
    if (threadIdx.x == 0) { if (H.fail) ++g_k5_sorted_buckets; else ++g_k5_hashed_buckets; }
#endif
    if (H.fail) return false;                                              // (uniform: read behind the barrier)
    for (uint32_t sl = threadIdx.x; sl < slots; sl += PB_THREADS)
        if (H.cnt[sl] == 2u) { const uint32_t first = H.mf[sl][0] < H.mf[sl][1] ? 0u : 1u; pair(sl, first, 1u - first); }
    return true;
}

// A listed (oversize) bucket, by ONE workgroup of 1024: its candidate indices are sorted in memory by (name hash, second hash, file
// index) -- a bitonic network over `idx` (the bucket's slice of the scatter, padded to a power of two with 0xFFFFFFFF = larger than
// any key) -- and `replay(j, m, key_of)` then runs for every first record of a name as in the LDS path.  Slow and exact: the
// network is log^2 stages of memory gathers, a name with millions of records is replayed by one lane as the reference walks it.
template <class Cand>
__device__ __forceinline__ bool pb_less(const Cand &c, uint32_t a, uint32_t b) {           // candidate a sorts before candidate b
    if (b == 0xFFFFFFFFu) return a != 0xFFFFFFFFu;
    if (a == 0xFFFFFFFFu) return false;
    const uint64_t qa = c.qhash[a], qb = c.qhash[b];
    if (qa != qb) return qa < qb;
    const uint32_t ha = c.h2 ? c.h2[a] : 0u, hb = c.h2 ? c.h2[b] : 0u;
    if (ha != hb) return ha < hb;
    return c.file_index[a] < c.file_index[b];
}
template <class Cand>
__device__ __forceinline__ void pair_bucket_big_sort(const Cand &c, uint32_t *idx, uint32_t slots) {
    for (uint32_t k = 2; k <= slots; k <<= 1)
        for (uint32_t j = k >> 1; j > 0; j >>= 1) {
            for (uint32_t i = threadIdx.x; i < slots; i += blockDim.x) {
                const uint32_t x = i ^ j;
                if (x > i) {
                    const uint32_t a = idx[i], b = idx[x];
                    const bool up = (i & k) == 0;
                    if (pb_less(c, b, a) == up) { idx[i] = b; idx[x] = a; }
                }
            }
            __syncthreads();
        }
}
// scratch `big_idx`: room for 2 x (candidates) indices; listed bucket t sorts at big_idx + 2 * off[bucket]
template <class Cand>
__device__ __forceinline__ uint32_t pair_bucket_big_prepare(const Cand &c, const uint32_t *off, const uint32_t *perm, const uint32_t *big, uint32_t t, uint32_t *big_idx, uint32_t **sorted) {
    const uint32_t bucket = big[1u + t], lo = off[bucket], m = off[bucket + 1] - lo;
    uint32_t slots = 2;
    while (slots < m) slots <<= 1;
    uint32_t *idx = big_idx + 2u * (size_t)lo;
    for (uint32_t i = threadIdx.x; i < slots; i += blockDim.x) idx[i] = i < m ? perm[lo + i] : 0xFFFFFFFFu;
    __syncthreads();
    pair_bucket_big_sort(c, idx, slots);
    *sorted = idx;
    return m;
}
__global__ void __launch_bounds__(1024)
frag_replay_big_kernel(const FragCandidates c, const uint32_t *off, const uint32_t *perm, const uint32_t *big, uint32_t *big_idx,
                       uint64_t *sample_file, uint32_t *sample_size, uint32_t *n_samples) {
    const uint32_t n_big = big[0] < PB_BIG_MAX ? big[0] : PB_BIG_MAX;
    for (uint32_t t = blockIdx.x; t < n_big; t += gridDim.x) {
        uint32_t *idx;
        const uint32_t m = pair_bucket_big_prepare(c, off, perm, big, t, big_idx, &idx);
        for (uint32_t j = threadIdx.x; j < m; j += blockDim.x) {
            const uint32_t e0 = idx[j];
            const uint64_t q = c.qhash[e0]; const uint32_t h = c.h2 ? c.h2[e0] : 0u;
            if (j > 0) { const uint32_t ep = idx[j - 1]; if (c.qhash[ep] == q && (c.h2 ? c.h2[ep] : 0u) == h) continue; }
            bool pending = false; int32_t p_name = 0, p_end = 0;
            for (uint32_t k = j; k < m; ++k) {
                const uint32_t e = idx[k];
                if (c.qhash[e] != q || (c.h2 ? c.h2[e] : 0u) != h) break;
                const int32_t name = c.name[e], endpos = c.endpos[e];
                if (!pending) { pending = true; p_name = name; p_end = endpos; }
                else if (name == p_name) {
                    const uint32_t fs = c.flag_size[e];
                    if (!(fs >> 31) || endpos <= p_end) continue;
                    const uint32_t slot = atomicAdd(n_samples, 1u);
                    sample_file[slot] = c.file_index[e]; sample_size[slot] = fs & 0x7FFFFFFFu;
                    pending = false;
                }
            }
        }
        __syncthreads();
    }
}

// src/Expression.cpp:511-538 for every name of the bucket.  A name yields at most one sample per two of its records: the samples of
// a bucket are collected in LDS and written behind ONE reservation of the output list (round 4 took a slot of the list per
// sample: millions of atomics on one address).
__global__ void __launch_bounds__(PB_THREADS)
frag_replay_kernel(const FragCandidates c, const uint32_t *off, const uint32_t *perm, uint64_t *sample_file, uint32_t *sample_size, uint32_t *n_samples) {
    __shared__ PairScratch U;
    PairBucket &S = U.S;
    __shared__ uint32_t s_n, s_base;
    __shared__ uint32_t s_stash[PH_SLOTS / 4];                         // (hashed path) slot * 2 + member of the second records that yield a sample
    if (threadIdx.x == 0) s_n = 0u;
    __syncthreads();
    // the bucket of a real file -- one or two candidates per name -- is paired by hashing (pair_bucket_hashed); anything else is sorted
    if (pair_bucket_hashed(U.H, c.qhash, c.h2, c.file_index, off, perm, [&](uint32_t sl, uint32_t i1, uint32_t i2) {
            const uint32_t first = U.H.mem[sl][i1], second = U.H.mem[sl][i2];
            if (c.name[second] != c.name[first]) return;                                // :517
            const uint32_t fs = c.flag_size[second];
            if (!(fs >> 31) || c.endpos[second] <= c.endpos[first]) return;             // :528
            s_stash[atomicAdd(&s_n, 1u)] = sl * 2u + i2;                                // :530 (at most one sample per two candidates: <= PH_SLOTS / 4)
        })) {
        __syncthreads();
        if (threadIdx.x == 0 && s_n) s_base = atomicAdd(n_samples, s_n);
        __syncthreads();
        for (uint32_t k = threadIdx.x; k < s_n; k += PB_THREADS) {
            const uint32_t sl = s_stash[k] >> 1, i2 = s_stash[k] & 1u;
            sample_file[s_base + k] = U.H.mf[sl][i2]; sample_size[s_base + k] = c.flag_size[U.H.mem[sl][i2]] & 0x7FFFFFFFu;
        }
        return;
    }
    __syncthreads();
    const uint32_t m = pair_bucket_sorted(S, c.qhash, c.h2, c.file_index, off, perm);
    __syncthreads();
    uint64_t my_file[2]; uint32_t my_size[2]; uint32_t mine = 0;       // a thread owns the names that START at its slots j, j + 256, ...: a handful of samples
    for (uint32_t j = threadIdx.x; j < m; j += PB_THREADS) {
        const uint64_t q = S.q[j]; const uint32_t h = S.h[j];
        if (j > 0 && S.q[j - 1] == q && S.h[j - 1] == h) continue;   // not the first record of its name
        bool pending = false; int32_t p_name = 0, p_end = 0;
        for (uint32_t k = j; k < m && S.q[k] == q && S.h[k] == h; ++k) {
            const uint32_t e = S.e[k];
            const int32_t name = c.name[e], endpos = c.endpos[e];
            if (!pending) { pending = true; p_name = name; p_end = endpos; }            // :512-516
            else if (name == p_name) {                                                  // :517
                const uint32_t fs = c.flag_size[e];
                if (!(fs >> 31) || endpos <= p_end) continue;                            // :528 (the entry stays)
                if (mine < 2u) { my_file[mine] = S.f[k]; my_size[mine] = fs & 0x7FFFFFFFu; ++mine; }     // :530
                else { const uint32_t slot = atomicAdd(n_samples, 1u); sample_file[slot] = S.f[k]; sample_size[slot] = fs & 0x7FFFFFFFu; }   // (a name with dozens of records)
                pending = false;                                                        // :531
            }
        }
    }
    const uint32_t at = mine ? atomicAdd(&s_n, mine) : 0u;
    __syncthreads();
    if (threadIdx.x == 0 && s_n) s_base = atomicAdd(n_samples, s_n);
    __syncthreads();
    for (uint32_t k = 0; k < mine; ++k) { sample_file[s_base + at + k] = my_file[k]; sample_size[s_base + at + k] = my_size[k]; }
}

#if defined(__HIPCC__)   /* (needs the G/C bit helpers of rsqc_device.h, device build only) */
// src/Expression.cpp:461-476 for every name of the bucket.  Real fragments pile up in a dozen neighbouring bins, i.e. in two
// cache lines: memory-side atomics on them serialise; the histogram is kept per workgroup in LDS and flushed once.
__global__ void __launch_bounds__(PB_THREADS)
gc_replay_kernel(const GcCandidates c, const uint32_t *off, const uint32_t *perm, const DevReference R, unsigned long long *bins) {
    __shared__ PairScratch U;
    PairBucket &S = U.S;
    __shared__ uint32_t hist[RSQC_GC_BINS + 1];
    for (int i = threadIdx.x; i <= RSQC_GC_BINS; i += PB_THREADS) hist[i] = 0u;
    __syncthreads();
    // the bucket of a real file -- one or two candidates per name -- is paired by hashing (pair_bucket_hashed); anything else is sorted
    if (pair_bucket_hashed(U.H, c.qhash, c.h2, c.file_index, off, perm, [&](uint32_t sl, uint32_t i1, uint32_t i2) {
            const uint32_t first = U.H.mem[sl][i1], second = U.H.mem[sl][i2];
            if (c.row[second] != c.row[first]) return;                                  // :467
            const uint32_t fl = c.flag_lq[second];
            const int32_t p_end = c.endpos[first], endpos = c.endpos[second];
            if (endpos <= p_end || !(fl >> 31)) return;                                 // :471
            const int tid = c.tid[second];
            const int64_t L = (int64_t)R.length[tid];
            int64_t s0 = (int64_t)p_end - (int64_t)(fl & 0x7FFFFFFFu), en = endpos;     // :473
            if (s0 < 0 || s0 >= L) return;
            if (en > L) en = L;
            if (en <= s0) return;
            const double v = gc_value(gc_count(R, tid, s0, en), (uint64_t)(en - s0));
            const unsigned int bin = (unsigned int)(v * 100.0);                         // src/RNASeQC.cpp:368
            atomicAdd(&hist[bin < RSQC_GC_BINS ? bin : RSQC_GC_BINS], 1u);
        })) {
        __syncthreads();
        for (int i = threadIdx.x; i <= RSQC_GC_BINS; i += PB_THREADS) if (hist[i]) atomicAdd(&bins[i], (unsigned long long)hist[i]);
        return;
    }
    __syncthreads();
    const uint32_t m = pair_bucket_sorted(S, c.qhash, c.h2, c.file_index, off, perm);  // (ends with a barrier when m > 0)
    __syncthreads();
    for (uint32_t j = threadIdx.x; j < m; j += PB_THREADS) {
        const uint64_t q = S.q[j]; const uint32_t h = S.h[j];
        if (j > 0 && S.q[j - 1] == q && S.h[j - 1] == h) continue;
        bool pending = false; uint32_t p_row = 0; int32_t p_end = 0;
        for (uint32_t k = j; k < m && S.q[k] == q && S.h[k] == h; ++k) {
            const uint32_t e = S.e[k];
            const uint32_t row = c.row[e]; const int32_t endpos = c.endpos[e];
            if (!pending) { pending = true; p_row = row; p_end = endpos; }              // :462-466
            else if (row == p_row) {                                                    // :467
                const uint32_t fl = c.flag_lq[e];
                if (endpos <= p_end || !(fl >> 31)) continue;                            // :471 (the entry stays)
                pending = false;                                                        // erase, :474
                const int tid = c.tid[e];
                const int64_t L = (int64_t)R.length[tid];
                int64_t s = (int64_t)p_end - (int64_t)(fl & 0x7FFFFFFFu), en = endpos;  // getSeq(chr, stored end - Length(), PositionEnd()) :473
                if (s < 0 || s >= L) continue;               // outside the contig: error paths of the reference, no fragment here
                if (en > L) en = L;                          // a page is clipped at the contig end (bioio.hpp:306)
                if (en <= s) continue;
                const double v = gc_value(gc_count(R, tid, s, en), (uint64_t)(en - s));
                const unsigned int bin = (unsigned int)(v * 100.0);                     // src/RNASeQC.cpp:368
                atomicAdd(&hist[bin < RSQC_GC_BINS ? bin : RSQC_GC_BINS], 1u);
            }
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i <= RSQC_GC_BINS; i += PB_THREADS) if (hist[i]) atomicAdd(&bins[i], (unsigned long long)hist[i]);
}
// the same for the listed (oversize) buckets, sorted in memory (pair_bucket_big_sort)
__global__ void __launch_bounds__(1024)
gc_replay_big_kernel(const GcCandidates c, const uint32_t *off, const uint32_t *perm, const uint32_t *big, uint32_t *big_idx, const DevReference R, unsigned long long *bins) {
    const uint32_t n_big = big[0] < PB_BIG_MAX ? big[0] : PB_BIG_MAX;
    for (uint32_t t = blockIdx.x; t < n_big; t += gridDim.x) {
        uint32_t *idx;
        const uint32_t m = pair_bucket_big_prepare(c, off, perm, big, t, big_idx, &idx);
        for (uint32_t j = threadIdx.x; j < m; j += blockDim.x) {
            const uint32_t e0 = idx[j];
            const uint64_t q = c.qhash[e0]; const uint32_t h = c.h2 ? c.h2[e0] : 0u;
            if (j > 0) { const uint32_t ep = idx[j - 1]; if (c.qhash[ep] == q && (c.h2 ? c.h2[ep] : 0u) == h) continue; }
            bool pending = false; uint32_t p_row = 0; int32_t p_end = 0;
            for (uint32_t k = j; k < m; ++k) {
                const uint32_t e = idx[k];
                if (c.qhash[e] != q || (c.h2 ? c.h2[e] : 0u) != h) break;
                const uint32_t row = c.row[e]; const int32_t endpos = c.endpos[e];
                if (!pending) { pending = true; p_row = row; p_end = endpos; }
                else if (row == p_row) {
                    const uint32_t fl = c.flag_lq[e];
                    if (endpos <= p_end || !(fl >> 31)) continue;
                    pending = false;
                    const int tid = c.tid[e];
                    const int64_t L = (int64_t)R.length[tid];
                    int64_t s2 = (int64_t)p_end - (int64_t)(fl & 0x7FFFFFFFu), en = endpos;
                    if (s2 < 0 || s2 >= L) continue;
                    if (en > L) en = L;
                    if (en <= s2) continue;
                    const double v = gc_value(gc_count(R, tid, s2, en), (uint64_t)(en - s2));
                    const unsigned int bin = (unsigned int)(v * 100.0);
                    atomicAdd(&bins[bin < RSQC_GC_BINS ? bin : RSQC_GC_BINS], 1ull);
                }
            }
        }
        __syncthreads();
    }
}

#endif

// ---- the N smallest file indices among the samples: radix select, one 8-bit digit per pass ---------------------------------
// counts, per value of the digit at `shift`, the samples whose higher digits equal those of `prefix`
__global__ void __launch_bounds__(256) sample_digit_hist_kernel(const uint64_t *v, const uint32_t *n_at, int shift, const uint64_t *state, uint32_t *hist) {
    __shared__ uint32_t h[256];
    const uint64_t prefix = state[0]; const uint32_t n = *n_at;
    h[threadIdx.x] = 0u;
    __syncthreads();
    const uint64_t high_mask = shift >= 56 ? 0ull : ~0ull << (shift + 8);
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const uint64_t x = v[i];
        if ((x & high_mask) == (prefix & high_mask)) atomicAdd(&h[(x >> shift) & 255u], 1u);
    }
    __syncthreads();
    if (h[threadIdx.x]) atomicAdd(&hist[threadIdx.x], h[threadIdx.x]);
}
// (one reservation of the output per WORKGROUP of 1024: a slot per kept sample was a million atomics on one address -- 0.28 ms --
//  and one per wave still 55 thousand of them: same-address memory atomics retire one after the other)
__global__ void __launch_bounds__(1024) sample_keep_kernel(const uint64_t *file, const uint32_t *size, const uint32_t *n_at, const uint64_t *last_kept, uint64_t *kept_file, uint32_t *kept_size, uint32_t *n_kept) {
    __shared__ uint32_t s_n, s_base;
    if (threadIdx.x == 0) s_n = 0u;
    __syncthreads();
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    const bool keep = i < *n_at && file[i] <= *last_kept;
    const unsigned long long m = __ballot(keep);
    const int lane = (int)(threadIdx.x & 63u), lead = m ? __ffsll(m) - 1 : 0;
    uint32_t base = 0;
    if (m && lane == lead) base = atomicAdd(&s_n, (uint32_t)__popcll(m));
    base = __shfl(base, lead, 64);
    __syncthreads();
    if (threadIdx.x == 0 && s_n) s_base = atomicAdd(n_kept, s_n);
    __syncthreads();
    if (keep) { const uint32_t s = s_base + base + (uint32_t)__popcll(m & ((1ull << lane) - 1ull)); kept_file[s] = file[i]; kept_size[s] = size[i]; }
}
// the select's start: rank of the last sample to keep = min(samples, --fragment-samples); nothing to keep -> a bound no file index reaches
__global__ void sample_plan_kernel(const uint32_t *n_samples, uint32_t max_samples, uint64_t *state) {
    if (threadIdx.x == 0 && blockIdx.x == 0) { const uint32_t ns = *n_samples; state[0] = 0ull; state[1] = ns < max_samples ? ns : max_samples; }
}
// the radix select's decision on the device (round 4 read 256 counters back per digit: eight synchronous copies): from the digit
// histogram of this pass, the digit that holds the sample of rank `want`; prefix and want move on in `state`, the histogram is cleared
__global__ void __launch_bounds__(256) sample_digit_pick_kernel(uint32_t *hist, int shift, uint64_t *state /*[0] prefix, [1] want*/) {
    __shared__ uint32_t h[256];
    h[threadIdx.x] = hist[threadIdx.x]; hist[threadIdx.x] = 0u;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint64_t want = state[1]; uint32_t d = 0;
        while (d < 255 && want > h[d]) { want -= h[d]; ++d; }
        state[0] |= (uint64_t)d << shift; state[1] = want;
    }
}

// ---- (size, count) pairs, ascending size ------------------------------------------------------------------------------------------
// sizes below SIZE_LDS are counted per workgroup in LDS and flushed once (fragment sizes sit in a few hundred neighbouring cells:
// a memory atomic per sample serialised on them -- 0.65 ms per million samples); `top` = largest size seen below SIZE_TABLE
constexpr uint32_t SIZE_LDS = 4096;
__global__ void __launch_bounds__(256) size_hist_kernel(const uint32_t *size, const uint32_t *n_at, uint32_t *table, uint32_t *big, uint32_t *n_big, uint32_t *top) {
    __shared__ uint32_t h[SIZE_LDS];
    __shared__ uint32_t s_top;
    for (uint32_t k = threadIdx.x; k < SIZE_LDS; k += blockDim.x) h[k] = 0u;
    if (threadIdx.x == 0) s_top = 0u;
    __syncthreads();
    const uint32_t n = *n_at;
    uint32_t mx = 0;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const uint32_t s = size[i];
        if (s < SIZE_LDS) atomicAdd(&h[s], 1u);
        else if (s < SIZE_TABLE) atomicAdd(&table[s], 1u);
        else big[atomicAdd(n_big, 1u)] = s;
        if (s < SIZE_TABLE && s > mx) mx = s;
    }
    if (mx) atomicMax(&s_top, mx);
    __syncthreads();
    for (uint32_t k = threadIdx.x; k < SIZE_LDS; k += blockDim.x) if (h[k]) atomicAdd(&table[k], h[k]);
    if (threadIdx.x == 0 && s_top) atomicMax(top, s_top);
}
// one workgroup: every thread owns a contiguous stretch of the USED part of the table [0, *top], counts its non-empty cells, the
// counts are scanned and the cells written in order
__global__ void __launch_bounds__(1024) size_hist_compact_kernel(const uint32_t *table, const uint32_t *top, uint32_t *out_size, uint32_t *out_count, uint32_t *n_out) {
    __shared__ uint32_t part[1024];
    const uint32_t used = *top + 1u, PER = (used + 1023u) / 1024u;
    const uint32_t lo = threadIdx.x * PER, hi = lo + PER < used ? lo + PER : used;
    uint32_t mine = 0;
    for (uint32_t k = lo; k < hi; ++k) mine += table[k] ? 1u : 0u;
    part[threadIdx.x] = mine;
    __syncthreads();
    for (uint32_t o = 1; o < 1024u; o <<= 1) {
        const uint32_t t = threadIdx.x >= o ? part[threadIdx.x - o] : 0u;
        __syncthreads();
        part[threadIdx.x] += t;
        __syncthreads();
    }
    uint32_t at = part[threadIdx.x] - mine;
    for (uint32_t k = lo; k < hi; ++k) { const uint32_t c = table[k]; if (c) { out_size[at] = k; out_count[at] = c; ++at; } }
    if (threadIdx.x == 1023) *n_out = part[1023];
}


}  // namespace rsqc
