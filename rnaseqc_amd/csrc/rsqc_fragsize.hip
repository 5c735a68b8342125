// rsqc_fragsize.hip -- K5: the fragment-size sampler of --bed runs
// (reference fragmentSizeMetrics, src/Expression.cpp:482-540, called at src/RNASeQC.cpp:372-376), and the mate pairing of the
// fragment GC statistics of --fasta runs (src/Expression.cpp:459-477).
//
// The reference keeps a map QNAME -> (BED interval, end position) and walks the file in order: the first qualifying record
// of a name is stored; a later record of the same name that sits in the same interval either yields a sample |isize| (and
// erases the entry) or leaves the entry untouched; sampling stops after --fragment-samples samples.  Order matters twice:
// inside a QNAME group, and for the cut-off.  Everything here is hand-written (rounds 1-3 ordered the candidates with two
// library radix sorts and replayed a whole group on one thread of a grid over ALL candidates):
//
//   K1 emits one candidate per record that passes the per-record tests (HQ, paired, every block inside one BED interval);
//   pair_bucket_*   the candidates are PARTITIONED by the high word of their name hash into buckets of ~512 (count, scan,
//                   scatter of candidate indices: three streaming passes with one memory atomic per candidate each);
//   *_replay_kernel one workgroup per bucket: the bucket's (name hash, file index, candidate) triples are loaded into LDS and
//                   sorted there (bitonic network, 2048 slots), so that a name's records are adjacent and in file order; the
//                   first lane of every name replays the reference's state machine over its few records;
//   sample_select_* more samples than --fragment-samples: the N smallest file indices are found by a radix SELECT on the
//                   device (eight 8-bit digit histograms, the host only reads 256 counters per pass) and copied out;
//   size_hist_*     the kept sizes are counted in a direct table (sizes below 2^20; the handful above go to a list) and the
//                   non-empty cells are compacted, ascending, by one workgroup: the host receives (size, count) pairs --
//                   the reference's std::map<long long, unsigned long> (src/RNASeQC.cpp:171) in iteration order.
#include <hip/hip_runtime.h>
#include <cstring>
#include <string.h>

#include <algorithm>
#include <cstdint>
#include <map>
#include <vector>

#include "rsqc_device.h"

#include "rsqc_k5.h"

namespace rsqc {

// ---- host side -----------------------------------------------------------------------------------------------------------------
namespace {
// control words on the device (S.count): [0] samples, [1] kept, [2] distinct sizes, [3] sizes beyond the table, [4] largest size in the
// table, [8..263] digit histogram, [264..267] the select's state (two 64-bit words: prefix, rank still wanted)
constexpr size_t CTL_WORDS = 8 + 256 + 4;
constexpr uint32_t PAIRS_FIRST = 2048;        // (size, count) pairs read back with the control words; more only if there are more
int grow_scratch(SortScratch &S, uint32_t n) {
    if (n <= S.cap_n) return 0;
    for (void **q : {&S.k0, &S.k1, &S.v0, &S.v1, &S.v2, &S.v3, &S.tmp}) { if (*q) (void)hipFree(*q); *q = nullptr; }
    S.cap_n = 0;
    const size_t cap = (size_t)n + n / 4 + 1024;
    const size_t buckets = cap / PB_MEAN + 2;
    // k0 samples' file index, k1 kept file index, v0 samples' size, v1 kept size, v2 candidate indices by bucket,
    // tmp: bucket counts | offsets | cursors (3 x (buckets + 1) words)
    if (hipMalloc(&S.k0, cap * 8) != hipSuccess || hipMalloc(&S.k1, cap * 8) != hipSuccess || hipMalloc(&S.v0, cap * 4) != hipSuccess ||
        hipMalloc(&S.v1, cap * 4) != hipSuccess || hipMalloc(&S.v2, cap * 4) != hipSuccess || hipMalloc(&S.v3, 2 * cap * 4 + 64) != hipSuccess ||
        hipMalloc(&S.tmp, (3 * (buckets + 1) + PB_BIG_MAX + 1) * 4) != hipSuccess) return RSQC_ERR_HIP;     // v3: the in-memory sort of oversize buckets (2 x their candidates)
    S.tmp_bytes = (3 * (buckets + 1) + PB_BIG_MAX + 1) * 4;
    S.cap_n = cap;
    return 0;
}
struct Buckets { uint32_t n_buckets; uint32_t *count, *off, *cursor, *big; };
// partitions `n` candidates by name hash: S.v2 = candidate indices bucket by bucket, offsets in the returned arrays
int partition_by_name(hipStream_t stream, const uint64_t *qhash, uint32_t n, SortScratch &S, int *d_error, Buckets &B) {
    B.n_buckets = std::max<uint32_t>(1u, n / PB_MEAN);
    B.count = (uint32_t *)S.tmp; B.off = B.count + (B.n_buckets + 1); B.cursor = B.off + (B.n_buckets + 1); B.big = B.cursor + (B.n_buckets + 1);
    if (hipMemsetAsync(B.count, 0, (size_t)(B.n_buckets + 1) * 4, stream) != hipSuccess || hipMemsetAsync(B.big, 0, 4, stream) != hipSuccess) return RSQC_ERR_HIP;
    const int T = 256, G = (int)((n + T - 1) / T);
    hipLaunchKernelGGL(pair_bucket_count_kernel, dim3(G), dim3(T), 0, stream, qhash, n, B.n_buckets, B.count);
    hipLaunchKernelGGL(pair_bucket_scan_kernel, dim3(1), dim3(1024), 0, stream, B.count, B.n_buckets, B.off, B.cursor, B.big, d_error);
    hipLaunchKernelGGL(pair_bucket_scatter_kernel, dim3(G), dim3(T), 0, stream, qhash, n, B.n_buckets, B.cursor, (uint32_t *)S.v2);
    return hipGetLastError() == hipSuccess ? 0 : RSQC_ERR_HIP;
}
}  // namespace

// Runs K5 over `n` candidates (device arrays in `c`): fills the histogram (ascending size) and the number of samples
// left, and leaves the kept samples (the first max_samples by file index, in no particular order) on the device in
// S.k1 (file index) / S.v1 (size), `n_kept` of them, for rsqc_shard_summary.  The scratch arrays are kept by the context
// between passes; `d_error` is the context's device error word (a bucket beyond the LDS sort's capacity).
int run_fragment_sizes(hipStream_t stream, const FragCandidates &c, uint32_t n, uint32_t max_samples,
                       std::vector<int64_t> &sizes, std::vector<uint64_t> &counts, uint32_t &remaining, SortScratch &S, uint32_t &n_kept, int *d_error) {
    sizes.clear(); counts.clear(); remaining = max_samples; n_kept = 0;
    if (n == 0) return 0;
#define FS_TRY(e) do { if ((e) != hipSuccess) return RSQC_ERR_HIP; } while (0)
    if (grow_scratch(S, n)) return RSQC_ERR_HIP;
    if (!S.count) FS_TRY(hipMalloc(&S.count, CTL_WORDS * 4));
    if (!S.table) {
        FS_TRY(hipMalloc((void **)&S.table, (size_t)SIZE_TABLE * 4));
        FS_TRY(hipMalloc((void **)&S.out_size, (size_t)SIZE_TABLE * 4)); FS_TRY(hipMalloc((void **)&S.out_count, (size_t)SIZE_TABLE * 4));
    }
    uint64_t *s_file = (uint64_t *)S.k0, *k_file = (uint64_t *)S.k1;
    uint32_t *s_size = (uint32_t *)S.v0, *k_size = (uint32_t *)S.v1, *ctl = (uint32_t *)S.count;
    FS_TRY(hipMemsetAsync(ctl, 0, CTL_WORDS * 4, stream));
    // (1) names together, file order inside a name; (2) replay: samples (file index of the completing record, |isize|)
    Buckets B;
    if (partition_by_name(stream, c.qhash, n, S, d_error, B)) return RSQC_ERR_HIP;
    hipLaunchKernelGGL(frag_replay_kernel, dim3(B.n_buckets), dim3(PB_THREADS), 0, stream, c, B.off, (const uint32_t *)S.v2, s_file, s_size, ctl + 0);
    // (the buckets the LDS sort cannot hold -- none with ordinary read names: the kernel then finds an empty list)
    hipLaunchKernelGGL(frag_replay_big_kernel, dim3(64), dim3(1024), 0, stream, c, B.off, (const uint32_t *)S.v2, B.big, (uint32_t *)S.v3, s_file, s_size, ctl + 0);
    // (3) the first max_samples samples in file order: a radix select of the sample of rank min(samples, max_samples) among the file
    //     indices, decided ON THE DEVICE digit by digit (round 4 read 256 counters back per digit and the sample count before: nine
    //     synchronous copies per pass); with fewer samples than the limit it selects the last one and everything is kept
    if (max_samples == 0) return 0;
    uint64_t *state = (uint64_t *)(ctl + 264);
    hipLaunchKernelGGL(sample_plan_kernel, dim3(1), dim3(64), 0, stream, ctl + 0, max_samples, state);
    const uint32_t ns_bound = n / 2u + 1u;                              // (a sample takes two candidates)
    const uint32_t sel_grid = std::min<uint32_t>(1024u, (ns_bound + 255u) / 256u);
    for (int shift = 56; shift >= 0; shift -= 8) {
        hipLaunchKernelGGL(sample_digit_hist_kernel, dim3(sel_grid), dim3(256), 0, stream, s_file, ctl + 0, shift, state, ctl + 8);
        hipLaunchKernelGGL(sample_digit_pick_kernel, dim3(1), dim3(256), 0, stream, ctl + 8, shift, state);
    }
    hipLaunchKernelGGL(sample_keep_kernel, dim3((ns_bound + 1023) / 1024), dim3(1024), 0, stream, s_file, s_size, ctl + 0, state, k_file, k_size, ctl + 1);
    // (4) the histogram of the kept sizes as (size, count) pairs, ascending
    FS_TRY(hipMemsetAsync(S.table, 0, (size_t)SIZE_TABLE * 4, stream));
    uint32_t *big = (uint32_t *)S.v2;                                  // (the bucket order is no longer needed)
    hipLaunchKernelGGL(size_hist_kernel, dim3(std::min<uint32_t>(512u, (std::min(ns_bound, max_samples) + 255u) / 256u)), dim3(256), 0, stream, k_size, ctl + 1, S.table, big, ctl + 3, ctl + 4);
    hipLaunchKernelGGL(size_hist_compact_kernel, dim3(1), dim3(1024), 0, stream, S.table, ctl + 4, S.out_size, S.out_count, ctl + 2);
    // ONE read-back: the control words and the first pairs (a fragment-size histogram has a few hundred distinct sizes)
    uint32_t tail[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    std::vector<uint32_t> hs(PAIRS_FIRST), hc(PAIRS_FIRST);
    FS_TRY(hipMemcpyAsync(tail, ctl, sizeof tail, hipMemcpyDeviceToHost, stream));
    FS_TRY(hipMemcpyAsync(hs.data(), S.out_size, PAIRS_FIRST * 4, hipMemcpyDeviceToHost, stream));
    FS_TRY(hipMemcpyAsync(hc.data(), S.out_count, PAIRS_FIRST * 4, hipMemcpyDeviceToHost, stream));
    FS_TRY(hipStreamSynchronize(stream));
    const uint32_t ns = tail[0], keep = std::min(ns, max_samples);
    if (!ns) return 0;
    if (tail[1] != keep) return RSQC_ERR_HIP;                           // (file indices are unique: the selection keeps exactly `keep`)
    n_kept = keep; remaining = max_samples - keep;
    const uint32_t nd = tail[2], nb = tail[3];
    std::vector<uint32_t> hb(nb);
    hs.resize(std::max<uint32_t>(nd, 1)); hc.resize(std::max<uint32_t>(nd, 1));
    if (nd > PAIRS_FIRST) { FS_TRY(hipMemcpyAsync(hs.data(), S.out_size, (size_t)nd * 4, hipMemcpyDeviceToHost, stream)); FS_TRY(hipMemcpyAsync(hc.data(), S.out_count, (size_t)nd * 4, hipMemcpyDeviceToHost, stream)); }
    if (nb) FS_TRY(hipMemcpyAsync(hb.data(), big, (size_t)nb * 4, hipMemcpyDeviceToHost, stream));
    if (nd > PAIRS_FIRST || nb) FS_TRY(hipStreamSynchronize(stream));
    for (uint32_t i = 0; i < nd; ++i) { sizes.push_back((int64_t)hs[i]); counts.push_back((uint64_t)hc[i]); }
    std::sort(hb.begin(), hb.end());                                    // insert sizes of 2^20 and more (a handful, if any): all behind the table's
    for (uint32_t i = 0; i < nb;) {
        uint32_t j = i + 1;
        while (j < nb && hb[j] == hb[i]) ++j;
        sizes.push_back((int64_t)hb[i]); counts.push_back((uint64_t)(j - i));
        i = j;
    }
#undef FS_TRY
    return 0;
}

void free_sort_scratch(SortScratch &s) {
    for (void *p : {s.k0, s.k1, s.v0, s.v1, s.v2, s.v3, s.tmp, s.count, (void *)s.table, (void *)s.out_size, (void *)s.out_count}) if (p) (void)hipFree(p);
    s = SortScratch{};
}

int run_gc_content(hipStream_t stream, const GcCandidates &c, uint32_t n, const DevReference &R, unsigned long long *bins, SortScratch &S, int *d_error) {
    if (n == 0) return 0;
    if (grow_scratch(S, n)) return RSQC_ERR_HIP;
    Buckets B;
    if (partition_by_name(stream, c.qhash, n, S, d_error, B)) return RSQC_ERR_HIP;
    hipLaunchKernelGGL(gc_replay_kernel, dim3(B.n_buckets), dim3(PB_THREADS), 0, stream, c, B.off, (const uint32_t *)S.v2, R, bins);
    hipLaunchKernelGGL(gc_replay_big_kernel, dim3(64), dim3(1024), 0, stream, c, B.off, (const uint32_t *)S.v2, B.big, (uint32_t *)S.v3, R, bins);
    return hipGetLastError() == hipSuccess ? 0 : RSQC_ERR_HIP;
}

}  // namespace rsqc
