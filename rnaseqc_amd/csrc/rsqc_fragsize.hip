// rsqc_fragsize.hip -- K5: the fragment-size sampler of --bed runs
// (reference fragmentSizeMetrics, src/Expression.cpp:482-540, called at src/RNASeQC.cpp:372-376).
//
// The reference keeps a map QNAME -> (BED interval, end position) and walks the file in order:
// the first qualifying record of a name is stored; a later record of the same name that sits in
// the same interval either yields a sample |isize| (and erases the entry) or leaves the entry
// untouched; sampling stops after --fragment-samples samples.  Order matters twice: inside a
// QNAME group, and for the cut-off.  On the device:
//   K1 emits one candidate per record that passes the per-record tests (HQ, paired, every block
//      inside one and the same BED interval);
//   candidates are ordered by (qname hash, file index) with two stable radix sorts (rocPRIM,
//      a library primitive -- the state machine and everything else is ours);
//   one thread per QNAME group replays the reference's state machine;
//   samples are ordered by file index and the first N are kept.
//
// The fragment GC statistics of --fasta runs (src/Expression.cpp:459-477) pair mates the same way -- a map
// QNAME -> (exon, end position), first record stored, a later one in the same exon either yields a fragment or leaves
// the entry -- without a cut-off: run_gc_content below shares the ordering steps and replays the groups with the
// G/C bit mask of the reference.
#include <hip/hip_runtime.h>
#include <cstring>
#include <string.h>
#include <rocprim/rocprim.hpp>

#include <algorithm>
#include <cstdint>
#include <map>
#include <vector>

#include "rsqc_device.h"

namespace rsqc {

__global__ void frag_iota_kernel(uint32_t *idx, uint32_t n) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) idx[i] = i;
}
__global__ void frag_gather_u64_kernel(const uint64_t *src, const uint32_t *idx, uint64_t *dst, uint32_t n) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = src[idx[i]];
}

// One thread per candidate position in (qhash, file index) order; the thread that starts a QNAME
// group replays the group (src/Expression.cpp:511-538).
__global__ void frag_groups_kernel(const uint64_t *sorted_q, const uint32_t *order, const FragCandidates c, uint32_t n,
                                   uint64_t *sample_file, uint32_t *sample_size, uint32_t *n_samples) {
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    const uint64_t q = sorted_q[j];
    if (j > 0 && sorted_q[j - 1] == q) return;           // not a group start
    bool pending = false; int32_t p_name = 0, p_end = 0;
    for (uint32_t k = j; k < n && sorted_q[k] == q; ++k) {
        const uint32_t e = order[k];
        const int32_t name = c.name[e], endpos = c.endpos[e];
        if (!pending) { pending = true; p_name = name; p_end = endpos; }            // :512-516
        else if (name == p_name) {                                                  // :517
            const uint32_t fs = c.flag_size[e];
            if (!(fs >> 31) || endpos <= p_end) continue;                            // :528 (entry stays)
            const uint32_t slot = atomicAdd(n_samples, 1u);
            sample_file[slot] = c.file_index[e]; sample_size[slot] = fs & 0x7FFFFFFFu;   // :530
            pending = false;                                                        // :531
        }
    }
}

static hipError_t sort_pairs_u64(void *tmp, size_t &tmp_bytes, const uint64_t *kin, uint64_t *kout, const uint32_t *vin,
                                 uint32_t *vout, uint32_t n, hipStream_t s) {
    return rocprim::radix_sort_pairs(tmp, tmp_bytes, kin, kout, vin, vout, n, 0, 64, s);
}

static int grow_scratch(SortScratch &S, uint32_t n) {
    if (n <= S.cap_n) return 0;
    for (void **q : {&S.k0, &S.k1, &S.v0, &S.v1, &S.v2}) { if (*q) (void)hipFree(*q); *q = nullptr; }
    S.cap_n = 0;
    const size_t cap = (size_t)n + n / 4 + 1024;
    if (hipMalloc(&S.k0, cap * 8) != hipSuccess || hipMalloc(&S.k1, cap * 8) != hipSuccess || hipMalloc(&S.v0, cap * 4) != hipSuccess ||
        hipMalloc(&S.v1, cap * 4) != hipSuccess || hipMalloc(&S.v2, cap * 4) != hipSuccess) return RSQC_ERR_HIP;
    S.cap_n = cap;
    return 0;
}
static int grow_tmp(SortScratch &S, size_t tmp_bytes) {
    if (tmp_bytes + 256 <= S.tmp_bytes) return 0;
    if (S.tmp) (void)hipFree(S.tmp);
    S.tmp = nullptr; S.tmp_bytes = tmp_bytes + tmp_bytes / 4 + 4096;
    return hipMalloc(&S.tmp, S.tmp_bytes) == hipSuccess ? 0 : RSQC_ERR_HIP;
}

// Runs K5 over `n` candidates (device arrays in `c`): fills the histogram (ascending size) and the number of samples
// left, and leaves the kept samples (the first max_samples by file index, in no particular order) on the device in
// S.k1 (file index) / S.v1 (size), `n_kept` of them, for rsqc_shard_summary.  Everything but the final run-length pass over
// the sorted sizes happens on the device; the scratch arrays are kept by the context between passes.
int run_fragment_sizes(hipStream_t stream, const FragCandidates &c, uint32_t n, uint32_t max_samples,
                       std::vector<int64_t> &sizes, std::vector<uint64_t> &counts, uint32_t &remaining, SortScratch &S, uint32_t &n_kept) {
    sizes.clear(); counts.clear(); remaining = max_samples; n_kept = 0;
    if (n == 0) return 0;
#define FS_TRY(e) do { if ((e) != hipSuccess) return RSQC_ERR_HIP; } while (0)
    if (grow_scratch(S, n)) return RSQC_ERR_HIP;
    if (!S.count) FS_TRY(hipMalloc(&S.count, 16));
    uint64_t *k0 = (uint64_t *)S.k0, *k1 = (uint64_t *)S.k1;
    uint32_t *v0 = (uint32_t *)S.v0, *v1 = (uint32_t *)S.v1, *v2 = (uint32_t *)S.v2, *d_ns = (uint32_t *)S.count;
    size_t tmp_bytes = 0;
    FS_TRY(sort_pairs_u64(nullptr, tmp_bytes, c.file_index, k0, v0, v1, n, stream));
    if (grow_tmp(S, tmp_bytes)) return RSQC_ERR_HIP;
    const int T = 256, B = (int)((n + T - 1) / T);
    hipLaunchKernelGGL(frag_iota_kernel, dim3(B), dim3(T), 0, stream, v0, n);
    // (1) file order
    FS_TRY(sort_pairs_u64(S.tmp, tmp_bytes, c.file_index, k0, v0, v1, n, stream));
    // (2) stable sort by QNAME hash: groups, file order inside
    hipLaunchKernelGGL(frag_gather_u64_kernel, dim3(B), dim3(T), 0, stream, c.qhash, v1, k0, n);
    FS_TRY(sort_pairs_u64(S.tmp, tmp_bytes, k0, k1, v1, v2, n, stream));
    // (3) replay every group: samples (file index of the completing record, |isize|) -> k0 / v0 (free again by now)
    FS_TRY(hipMemsetAsync(d_ns, 0, 4, stream));
    hipLaunchKernelGGL(frag_groups_kernel, dim3(B), dim3(T), 0, stream, k1, v2, c, n, k0, v0, d_ns);
    uint32_t ns = 0;
    FS_TRY(hipMemcpyAsync(&ns, d_ns, 4, hipMemcpyDeviceToHost, stream));
    FS_TRY(hipStreamSynchronize(stream));
    if (!ns) return 0;
    // (4) the first max_samples samples in file order: only when there are more than that
    const uint32_t keep = std::min(ns, max_samples);
    if (keep < ns) FS_TRY(sort_pairs_u64(S.tmp, tmp_bytes, k0, k1, v0, v1, ns, stream));
    else { FS_TRY(hipMemcpyAsync(k1, k0, (size_t)ns * 8, hipMemcpyDeviceToDevice, stream)); FS_TRY(hipMemcpyAsync(v1, v0, (size_t)ns * 4, hipMemcpyDeviceToDevice, stream)); }
    n_kept = keep; remaining = max_samples - keep;
    // (5) histogram: the kept sizes sorted on the device, run lengths on the host (map<long long, unsigned long>, src/RNASeQC.cpp:171)
    size_t tb2 = 0;
    FS_TRY(rocprim::radix_sort_keys(nullptr, tb2, v1, v2, keep, 0, 32, stream));
    if (grow_tmp(S, tb2)) return RSQC_ERR_HIP;
    FS_TRY(rocprim::radix_sort_keys(S.tmp, tb2, v1, v2, keep, 0, 32, stream));
    if (keep > S.h_cap) {
        if (S.h_sizes) (void)hipHostFree(S.h_sizes);
        S.h_cap = keep + keep / 4 + 1024;
        FS_TRY(hipHostMalloc((void **)&S.h_sizes, (size_t)S.h_cap * 4, hipHostMallocDefault));
    }
    FS_TRY(hipMemcpyAsync(S.h_sizes, v2, (size_t)keep * 4, hipMemcpyDeviceToHost, stream));
    FS_TRY(hipStreamSynchronize(stream));
    for (uint32_t i = 0; i < keep;) {
        uint32_t j = i + 1;
        while (j < keep && S.h_sizes[j] == S.h_sizes[i]) ++j;
        sizes.push_back((int64_t)S.h_sizes[i]); counts.push_back((uint64_t)(j - i));
        i = j;
    }
#undef FS_TRY
    return 0;
}


// One thread per candidate in (qhash, file index) order; group starts replay src/Expression.cpp:461-476.
// Real fragments pile up in a dozen neighbouring bins, i.e. in two cache lines: memory-side atomics on them serialise
// (~3 ns each, 3.4 ms per million fragments when every fragment went to memory).  The histogram is therefore kept per
// workgroup in LDS over a grid-stride loop and flushed once: a few thousand global atomics per launch.
#define RSQC_GC_GROUP_BLOCKS 512
__global__ void __launch_bounds__(256)
gc_groups_kernel(const uint64_t *sorted_q, const uint32_t *order, const GcCandidates c, uint32_t n,
                 const DevReference R, unsigned long long *bins) {
    __shared__ uint32_t hist[RSQC_GC_BINS + 1];
    for (int i = threadIdx.x; i <= RSQC_GC_BINS; i += blockDim.x) hist[i] = 0u;
    __syncthreads();
    for (uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; j < n; j += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t q = sorted_q[j];
        if (j > 0 && sorted_q[j - 1] == q) continue;         // not a group start
        bool pending = false; uint32_t p_row = 0; int32_t p_end = 0;
        for (uint64_t k = j; k < n && sorted_q[k] == q; ++k) {
            const uint32_t e = order[k];
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
    for (int i = threadIdx.x; i <= RSQC_GC_BINS; i += blockDim.x) if (hist[i]) atomicAdd(&bins[i], (unsigned long long)hist[i]);
}

void free_sort_scratch(SortScratch &s) {
    for (void *p : {s.k0, s.k1, s.v0, s.v1, s.v2, s.tmp, s.count}) if (p) (void)hipFree(p);
    if (s.h_sizes) (void)hipHostFree(s.h_sizes);
    s = SortScratch{};
}

int run_gc_content(hipStream_t stream, const GcCandidates &c, uint32_t n, const DevReference &R, unsigned long long *bins, SortScratch &S) {
    if (n == 0) return 0;
#define GC_TRY(e) do { if ((e) != hipSuccess) return RSQC_ERR_HIP; } while (0)
    if (grow_scratch(S, n)) return RSQC_ERR_HIP;
    uint64_t *k0 = (uint64_t *)S.k0, *k1 = (uint64_t *)S.k1;
    uint32_t *v0 = (uint32_t *)S.v0, *v1 = (uint32_t *)S.v1, *v2 = (uint32_t *)S.v2;
    size_t tmp_bytes = 0;
    GC_TRY(sort_pairs_u64(nullptr, tmp_bytes, c.file_index, k0, v0, v1, n, stream));
    if (grow_tmp(S, tmp_bytes)) return RSQC_ERR_HIP;
    const int T = 256, B = (int)((n + T - 1) / T);
    hipLaunchKernelGGL(frag_iota_kernel, dim3(B), dim3(T), 0, stream, v0, n);
    GC_TRY(sort_pairs_u64(S.tmp, tmp_bytes, c.file_index, k0, v0, v1, n, stream));               // file order
    hipLaunchKernelGGL(frag_gather_u64_kernel, dim3(B), dim3(T), 0, stream, c.qhash, v1, k0, n);
    GC_TRY(sort_pairs_u64(S.tmp, tmp_bytes, k0, k1, v1, v2, n, stream));                         // stable by QNAME hash
    hipLaunchKernelGGL(gc_groups_kernel, dim3(B < RSQC_GC_GROUP_BLOCKS ? B : RSQC_GC_GROUP_BLOCKS), dim3(T), 0, stream, k1, v2, c, n, R, bins);
    GC_TRY(hipGetLastError());
#undef GC_TRY
    return 0;
}

}  // namespace rsqc
