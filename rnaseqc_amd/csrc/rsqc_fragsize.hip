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

// Runs K5 over `n` candidates (device arrays in `c`).  Returns 0 or an RSQC_ERR_* code; fills the
// histogram (ascending size) and the number of samples left.
int run_fragment_sizes(hipStream_t stream, const FragCandidates &c, uint32_t n, uint32_t max_samples,
                       std::vector<int64_t> &sizes, std::vector<uint64_t> &counts, uint32_t &remaining,
                       std::vector<uint64_t> *keep_file, std::vector<uint32_t> *keep_size) {
    sizes.clear(); counts.clear(); remaining = max_samples;
    if (keep_file) keep_file->clear();
    if (keep_size) keep_size->clear();
    if (n == 0) return 0;
    uint64_t *k0 = nullptr, *k1 = nullptr, *sf = nullptr, *sf2 = nullptr;
    uint32_t *v0 = nullptr, *v1 = nullptr, *v2 = nullptr, *ss = nullptr, *d_ns = nullptr, *sidx = nullptr, *sidx2 = nullptr;
    void *tmp = nullptr;
    int rc = 0;
    auto cleanup = [&]() {
        for (void *p : {(void *)k0, (void *)k1, (void *)sf, (void *)sf2, (void *)v0, (void *)v1, (void *)v2, (void *)ss,
                        (void *)d_ns, (void *)sidx, (void *)sidx2, tmp}) if (p) (void)hipFree(p);
    };
#define FS_TRY(e) do { if ((e) != hipSuccess) { cleanup(); return RSQC_ERR_HIP; } } while (0)
    FS_TRY(hipMalloc(&k0, (size_t)n * 8)); FS_TRY(hipMalloc(&k1, (size_t)n * 8));
    FS_TRY(hipMalloc(&v0, (size_t)n * 4)); FS_TRY(hipMalloc(&v1, (size_t)n * 4)); FS_TRY(hipMalloc(&v2, (size_t)n * 4));
    FS_TRY(hipMalloc(&sf, (size_t)n * 8)); FS_TRY(hipMalloc(&ss, (size_t)n * 4)); FS_TRY(hipMalloc(&d_ns, 4));
    size_t tmp_bytes = 0;
    FS_TRY(sort_pairs_u64(nullptr, tmp_bytes, c.file_index, k0, v0, v1, n, stream));
    FS_TRY(hipMalloc(&tmp, tmp_bytes + 256));
    const int T = 256, B = (int)((n + T - 1) / T);
    hipLaunchKernelGGL(frag_iota_kernel, dim3(B), dim3(T), 0, stream, v0, n);
    // (1) file order
    FS_TRY(sort_pairs_u64(tmp, tmp_bytes, c.file_index, k0, v0, v1, n, stream));
    // (2) stable sort by QNAME hash: groups, file order inside
    hipLaunchKernelGGL(frag_gather_u64_kernel, dim3(B), dim3(T), 0, stream, c.qhash, v1, k0, n);
    FS_TRY(sort_pairs_u64(tmp, tmp_bytes, k0, k1, v1, v2, n, stream));
    // (3) replay every group
    FS_TRY(hipMemsetAsync(d_ns, 0, 4, stream));
    hipLaunchKernelGGL(frag_groups_kernel, dim3(B), dim3(T), 0, stream, k1, v2, c, n, sf, ss, d_ns);
    uint32_t ns = 0;
    FS_TRY(hipMemcpyAsync(&ns, d_ns, 4, hipMemcpyDeviceToHost, stream));
    FS_TRY(hipStreamSynchronize(stream));
    if (ns) {
        // (4) the first max_samples samples in file order
        std::vector<uint64_t> h_file(ns); std::vector<uint32_t> h_size(ns);
        FS_TRY(hipMemcpy(h_file.data(), sf, (size_t)ns * 8, hipMemcpyDeviceToHost));
        FS_TRY(hipMemcpy(h_size.data(), ss, (size_t)ns * 4, hipMemcpyDeviceToHost));
        std::vector<uint32_t> ord(ns);
        for (uint32_t i = 0; i < ns; ++i) ord[i] = i;
        const uint32_t keep = std::min(ns, max_samples);
        if (keep < ns) std::nth_element(ord.begin(), ord.begin() + keep, ord.end(), [&](uint32_t x, uint32_t y) { return h_file[x] < h_file[y]; });
        std::map<int64_t, uint64_t> hist;                 // map<long long, unsigned long>, src/RNASeQC.cpp:171
        for (uint32_t i = 0; i < keep; ++i) hist[(int64_t)h_size[ord[i]]]++;
        for (auto &kv : hist) { sizes.push_back(kv.first); counts.push_back(kv.second); }
        remaining = max_samples - keep;
        if (keep_file && keep_size) {                     // the kept samples in file order: what a sharded run merges
            std::sort(ord.begin(), ord.begin() + keep, [&](uint32_t x, uint32_t y) { return h_file[x] < h_file[y]; });
            keep_file->reserve(keep); keep_size->reserve(keep);
            for (uint32_t i = 0; i < keep; ++i) { keep_file->push_back(h_file[ord[i]]); keep_size->push_back(h_size[ord[i]]); }
        }
    }
    cleanup();
#undef FS_TRY
    return rc;
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
    for (void *p : {s.k0, s.k1, s.v0, s.v1, s.v2, s.tmp}) if (p) (void)hipFree(p);
    s = SortScratch{};
}

int run_gc_content(hipStream_t stream, const GcCandidates &c, uint32_t n, const DevReference &R, unsigned long long *bins, SortScratch &S) {
    if (n == 0) return 0;
#define GC_TRY(e) do { if ((e) != hipSuccess) return RSQC_ERR_HIP; } while (0)
    if (n > S.cap_n) {
        void *tmp_keep = S.tmp; size_t tmp_bytes_keep = S.tmp_bytes;
        S.tmp = nullptr;
        free_sort_scratch(S);
        S.tmp = tmp_keep; S.tmp_bytes = tmp_bytes_keep;
        const size_t cap = (size_t)n + n / 4 + 1024;
        GC_TRY(hipMalloc(&S.k0, cap * 8)); GC_TRY(hipMalloc(&S.k1, cap * 8));
        GC_TRY(hipMalloc(&S.v0, cap * 4)); GC_TRY(hipMalloc(&S.v1, cap * 4)); GC_TRY(hipMalloc(&S.v2, cap * 4));
        S.cap_n = cap;
    }
    uint64_t *k0 = (uint64_t *)S.k0, *k1 = (uint64_t *)S.k1;
    uint32_t *v0 = (uint32_t *)S.v0, *v1 = (uint32_t *)S.v1, *v2 = (uint32_t *)S.v2;
    size_t tmp_bytes = 0;
    GC_TRY(sort_pairs_u64(nullptr, tmp_bytes, c.file_index, k0, v0, v1, n, stream));
    if (tmp_bytes + 256 > S.tmp_bytes) {
        if (S.tmp) (void)hipFree(S.tmp);
        S.tmp = nullptr; S.tmp_bytes = tmp_bytes + tmp_bytes / 4 + 4096;
        GC_TRY(hipMalloc(&S.tmp, S.tmp_bytes));
    }
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
