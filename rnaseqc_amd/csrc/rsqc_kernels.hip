// rsqc_kernels.hip -- hand-written HIP kernels for gfx950 (MI355X, wave64).
//
//   K1  classify_count_kernel   per-record path: gate cascade, CIGAR blocks, overlap query (staged, batched
//                               loads), exon / gene counters in workgroup LDS tables, coverage as a
//                               difference array, (gene, name-hash) pairs, vertical scalar counters
//   K1s classify_slow_kernel    general code for the records the fast path hands over
//   KR  read_length_kernel      order-dependent "Read Length" state machine over tile summaries
//   K4  frag_layout / frag_local / frag_count   per-gene distinct QNAME count (geneFragmentCounts) by
//                               partitioned key lists + LDS sets; dedup_* = the earlier open-addressing form
//   K3  gene_coverage_kernel    per gene: diff->coverage scan into LDS, per-exon CV, bias windows, masked
//                               gene mean/std/CV; workgroup sized to the gene (1 wave / 256 / 1024 threads)
//   reset_kernel, pack_results_kernel   accumulator reset and result packing around a pass
//   gc_pack / exon_gc / gc_candidates   --fasta: G/C bit mask of the reference, per-exon GC, fragment GC candidates
//
// This is integer / byte indexing work bound by instruction issue, load latency and atomics, not a
// contraction: no MFMA.  All wave-level idioms are written for 64-lane wavefronts.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <algorithm>
#include <cstdlib>

// Profiling build (make prof -> lib/librnaseqc_amd_prof.so, loaded with RSQC_LIB): K1 reads the shader clock at section
// marks and accumulates, per section, the cycles its waves spent since their previous mark (stalls included).
// Diagnostic only; the product library is compiled without it.
#ifdef RSQC_K1_PROF
namespace rsqc { __device__ __forceinline__ void k1_mark(int sec); __device__ __forceinline__ void k1_event(int id, bool cond); }
#define RSQC_MARK(sec) ::rsqc::k1_mark(sec)
#define RSQC_EVENT(id, cond) ::rsqc::k1_event(id, cond)
#endif

#include "rsqc_device.h"
#ifndef RSQC_K1_PROF
#define RSQC_FIN_STAMP(sec)
#define RSQC_FIN_SECT(base, sec)
#define RSQC_FIN_BEGIN
#endif

namespace rsqc {

// ------------------------------------------------------------------ wave helpers: rsqc_wave.h
}  // namespace rsqc
#include "rsqc_wave.h"
namespace rsqc {

#ifdef RSQC_K1_PROF
__device__ unsigned long long g_k1_prof[48];          // [sec] cycles, [16 + sec] marks, [32 + id] / [36 + id] slow-branch events
__shared__ unsigned long long s_prof_acc[48];
__shared__ unsigned long long s_prof_last[RSQC_K1_THREADS / 64];
__device__ __forceinline__ void k1_mark(int sec) {
    const unsigned long long t = __builtin_amdgcn_s_memtime();
    if (lane_id() == 0) {
        const int w = (int)(threadIdx.x >> 6);
        atomicAdd(&s_prof_acc[sec], t - s_prof_last[w]);
        atomicAdd(&s_prof_acc[16 + sec], 1ull);
        s_prof_last[w] = t;
    }
}
// [32 + id]: tiles (block rounds) in which at least one lane takes slow branch `id`; [36 + id]: lanes that take it
__device__ __forceinline__ void k1_event(int id, bool cond) {
    const unsigned long long m = __ballot(cond);
    if (m && lane_id() == 0) { atomicAdd(&s_prof_acc[32 + id], 1ull); atomicAdd(&s_prof_acc[36 + id], (unsigned long long)__popcll(m)); }
}
// end-of-file stage: [0..31] absolute shader-clock stamps of thread 0 of the first workgroup of the longest-gene launch of K3,
// [32..47] cycles per section of frag_local_kernel summed over its workgroups (thread 0), [48..63] the same for frag_count_kernel
__device__ unsigned long long g_fin_prof[64];
__shared__ unsigned long long s_fin_stamp[12];      // the stamps of this workgroup; the slowest workgroup of the launch publishes its set
#define RSQC_FIN_STAMP(sec) do { if (first == 0 && threadIdx.x == 0) s_fin_stamp[sec] = __builtin_amdgcn_s_memtime(); } while (0)
#define RSQC_FIN_SECT(base, sec) do { if (threadIdx.x == 0) { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); atomicAdd(&g_fin_prof[(base) + (sec)], t_ - fin_last); fin_last = t_; } } while (0)
#define RSQC_FIN_BEGIN unsigned long long fin_last = __builtin_amdgcn_s_memtime();
__device__ unsigned long long g_dbg_pair_hash[32768]; __device__ uint32_t g_dbg_pair_gene[32768]; __device__ uint32_t g_dbg_pair_count;
extern "C" __attribute__((visibility("default"))) int rsqc_debug_pairs(unsigned long long *hash, uint32_t *gene, uint32_t *count) {
    if (hipMemcpyFromSymbol(hash, HIP_SYMBOL(g_dbg_pair_hash), sizeof(g_dbg_pair_hash)) != hipSuccess) return -1;
    if (hipMemcpyFromSymbol(gene, HIP_SYMBOL(g_dbg_pair_gene), sizeof(g_dbg_pair_gene)) != hipSuccess) return -1;
    if (hipMemcpyFromSymbol(count, HIP_SYMBOL(g_dbg_pair_count), 4) != hipSuccess) return -1;
    return 0;
}
extern "C" __attribute__((visibility("default"))) int rsqc_debug_fin_prof(unsigned long long *out, int reset) {
    if (out && hipMemcpyFromSymbol(out, HIP_SYMBOL(g_fin_prof), sizeof(g_fin_prof)) != hipSuccess) return -1;
    if (reset) { unsigned long long z[64] = {0}; if (hipMemcpyToSymbol(HIP_SYMBOL(g_fin_prof), z, sizeof(z)) != hipSuccess) return -1; }
    return 0;
}
extern "C" __attribute__((visibility("default"))) int rsqc_debug_k1_prof(unsigned long long *out, int reset) {
    if (out && hipMemcpyFromSymbol(out, HIP_SYMBOL(g_k1_prof), sizeof(g_k1_prof)) != hipSuccess) return -1;
    if (reset) { unsigned long long z[48] = {0}; if (hipMemcpyToSymbol(HIP_SYMBOL(g_k1_prof), z, sizeof(z)) != hipSuccess) return -1; }
    return 0;
}
#endif

// ------------------------------------------------------------------ K1s (and the record loader): rsqc_k1s.h
}  // namespace rsqc
#include "rsqc_k1s.h"
namespace rsqc {

// ------------------------------------------------------------------ the gate pass of --legacy runs
// Under --legacy the per-record work of the feature stage belongs to classify_slow_kernel<true> (general code over EVERY record,
// rsqc_k1s.h); this kernel takes what comes before it: the gate cascade with the LegacyMode tests (src/RNASeQC.cpp:254-342), its
// counters, the fragment-size candidates and the Read-Length inputs.  (Rounds 1-2 ran the default rules here as well -- bin tables
// and a downward walk of the start-sorted exon rows; since round 3 they run in classify_ei_kernel, rsqc_k1.h, and round 5 removed
// the feature stage, scatter and pair emission this body still carried for them.)
struct LegacyGateShared {
    unsigned long long cnt[RSQC_N_COUNTERS];     // sum-type counters
    uint32_t cnt32[64];                          // one-per-record counters of the workgroup (a workgroup sees < 2^32 records)
    uint32_t rl[3];
};
__global__ void __launch_bounds__(RSQC_K1_THREADS, 4)
classify_count_kernel_legacy(DevAnnotation a, DevParams p, DevBatch b, DevAccum acc) {
    __shared__ LegacyGateShared S;
    const int l = lane_id();
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    for (int c = threadIdx.x; c < RSQC_N_COUNTERS; c += blockDim.x) S.cnt[c] = 0ull;
    if (threadIdx.x < 64) S.cnt32[threadIdx.x] = 0u;
    if (threadIdx.x == 0) { S.rl[0] = 0u; S.rl[1] = 0xFFFFFFFFu; S.rl[2] = 0u; }
    if (blockIdx.x == 0 && threadIdx.x == 0) *acc.pair_slow_count = 0u;     // written only by the slow kernel, which runs after this one
    __syncthreads();
    // One-per-record counters: a WaveSink (rsqc_read.h) -- per tile, lane c of one register receives the number of records
    // that increment counter c, and ONE LDS instruction adds the register to the workgroup's table.  The seven sum-type counters
    // stay per-lane sums, reduced every 31 tiles.
    uint32_t sum_e1mm = 0, sum_e1b = 0, sum_e2mm = 0, sum_e2b = 0, sum_mm = 0, sum_b = 0, sum_blk = 0;
    int pending = 0;
    uint32_t l_span = 0u, l_lmin = 0xFFFFFFFFu, l_lmax = 0u;
    auto flush_counts = [&]() {
        const uint32_t s0 = wave_sum(sum_e1mm), s1 = wave_sum(sum_e1b), s2 = wave_sum(sum_e2mm), s3 = wave_sum(sum_e2b),
                       s4 = wave_sum(sum_mm), s5 = wave_sum(sum_b), s6 = wave_sum(sum_blk);
        if (l == 0 && s0) atomicAdd(&S.cnt[RSQC_C_END1_MISMATCHES], (unsigned long long)s0);
        if (l == 0 && s1) atomicAdd(&S.cnt[RSQC_C_END1_BASES], (unsigned long long)s1);
        if (l == 0 && s2) atomicAdd(&S.cnt[RSQC_C_END2_MISMATCHES], (unsigned long long)s2);
        if (l == 0 && s3) atomicAdd(&S.cnt[RSQC_C_END2_BASES], (unsigned long long)s3);
        if (l == 0 && s4) atomicAdd(&S.cnt[RSQC_C_MISMATCHED_BASES], (unsigned long long)s4);
        if (l == 0 && s5) atomicAdd(&S.cnt[RSQC_C_TOTAL_BASES], (unsigned long long)s5);
        if (l == 0 && s6) atomicAdd(&S.cnt[RSQC_C_ALIGNMENT_BLOCKS], (unsigned long long)s6);
        sum_e1mm = sum_e1b = sum_e2mm = sum_e2b = sum_mm = sum_b = sum_blk = 0;
        pending = 0;
    };
    // every WAVE streams its own contiguous range of records
    constexpr uint32_t WPB = RSQC_K1_THREADS / 64;
    const uint64_t total_waves = (uint64_t)gridDim.x * WPB;
    const uint64_t per_wave = (((b.n + total_waves - 1) / total_waves) + 63ull) & ~63ull;
    const uint64_t wbeg = ((uint64_t)blockIdx.x * WPB + (uint64_t)wave) * per_wave;
    const uint64_t wend = wbeg + per_wave < b.n ? wbeg + per_wave : b.n;
    uint32_t seg = wbeg < b.n ? find_segment(b, wbeg) : 0u;
    int32_t u_tid = b.n_seg ? b.seg_tid[seg] : -1;
    // record words are staged one tile ahead (core words two tiles ahead: the CIGAR address comes from them)
    const int4 zero4 = {0, 0, 0, 0};
    int4 cur_cv = zero4, cur_av = zero4, nx_cv = zero4;
    uint32_t cur_cg[4] = {0, 0, 0, 0};
    const int4 *const core4 = reinterpret_cast<const int4 *>(b.core), *const aux4 = reinterpret_cast<const int4 *>(b.aux);
    if (wbeg + (uint64_t)l < wend) { cur_cv = ld32(core4 + wbeg, (uint32_t)l); cur_av = ld32(aux4 + wbeg, (uint32_t)l); }
    if (wbeg + 64ull + (uint64_t)l < wend) nx_cv = ld32(core4 + wbeg + 64, (uint32_t)l);
    {
        const uint32_t co = (uint32_t)cur_cv.w;                          // buffers carry 32 bytes of slack
        cur_cg[0] = ld32(b.cigar, co); cur_cg[1] = ld32(b.cigar, co + 1); cur_cg[2] = ld32(b.cigar, co + 2); cur_cg[3] = ld32(b.cigar, co + 3);
    }
    for (uint64_t w0 = wbeg; w0 < wend; w0 += 64) {
        const uint64_t i = w0 + (uint64_t)l;
        const bool valid = i < wend;
        while (seg + 1 < b.n_seg && b.seg_start[seg + 1] <= w0) { ++seg; u_tid = b.seg_tid[seg]; }
        const bool mixed = seg + 1 < b.n_seg && b.seg_start[seg + 1] < w0 + 64ull;   // a contig boundary inside the tile
        WaveSink cnt;
        bool big_any;
        {
            RecordCounters rc; Record r; Blocks B; bool hq = false;
            const int4 cv = cur_cv, av = cur_av;                                  // (zero for lanes past the range)
            r.pos = cv.x; r.mpos = cv.y; r.isize = cv.z;
            r.cigar = b.cigar + (uint32_t)cv.w;
            r.qhash = (uint64_t)(uint32_t)av.x | ((uint64_t)(uint32_t)av.y << 32);
            r.flag = (uint32_t)av.z & 0xFFFFu; r.l_qseq = (int32_t)((uint32_t)av.z >> 16);
            r.mapq = (uint32_t)av.w & 0xFFu; r.nm = (int32_t)(((uint32_t)av.w >> 8) & 0xFFu);
            r.tagbits = ((uint32_t)av.w >> 16) & 0xFFu; r.n_cigar = (uint32_t)av.w >> 24;
            bool ok = true;
            if (valid && (r.l_qseq == RSQC_LQSEQ_ESCAPE || r.nm == RSQC_NM_ESCAPE || r.n_cigar == RSQC_NCIGAR_ESCAPE)) {
                uint32_t lo = 0, hi = b.n_wide;                     // wide table is sorted by record index
                while (lo < hi) { uint32_t m = (lo + hi) >> 1; if (b.wide_index[m] < i) lo = m + 1; else hi = m; }
                if (lo >= b.n_wide || b.wide_index[lo] != i) ok = false;
                else { r.l_qseq = b.wide_l_qseq[lo]; r.nm = b.wide_nm[lo]; r.n_cigar = b.wide_n_cigar[lo]; }
            }
            r.tid = u_tid;
            if (mixed && valid) { uint32_t s2 = seg; while (s2 + 1 < b.n_seg && b.seg_start[s2 + 1] <= i) ++s2; r.tid = b.seg_tid[s2]; }
            if (valid && !ok) atomicExch(acc.error, RSQC_ERR_ARG);
            const bool lane_on = valid && ok;
            if (!lane_on) r.n_cigar = 0;
            CigarWalk cw;
            walk_cigar(r, cur_cg, cw, B);
            const bool go = gate_cascade<true, WaveSink>(a, p, r, cw, rc, hq, cnt, lane_on);
            if (!lane_on) { rc.e1_mm = rc.e1_bases = rc.e2_mm = rc.e2_bases = rc.mm = rc.bases = rc.blocks = 0; rc.rl_eligible = 0; rc.error = 0; rc.frag_candidate = 0; }
            if (go && a.have_bed && rc.frag_candidate) {          // src/RNASeQC.cpp:372
                const int32_t name = bed_interval_of(a, r);
                if (name >= 0) {
                    const uint32_t slot = atomicAdd(acc.frag.count, 1u);
                    if (slot < acc.frag.cap) {
                        acc.frag.file_index[slot] = b.record_base + i; acc.frag.qhash[slot] = r.qhash;
                        acc.frag.h2[slot] = b.qhash2 ? b.qhash2[i] : 0u;
                        acc.frag.name[slot] = name; acc.frag.endpos[slot] = rc.endpos;
                        const bool fok = !(r.flag & RSQC_FMREVERSE) && (r.flag & RSQC_FREVERSE) && r.pos != r.mpos;
                        const uint32_t sz = (uint32_t)(r.isize < 0 ? -(int64_t)r.isize : (int64_t)r.isize);
                        acc.frag.flag_size[slot] = (sz & 0x7FFFFFFFu) | (fok ? 0x80000000u : 0u);
                    } else atomicExch(acc.error, RSQC_ERR_CAPACITY);
                }
            }
            if (rc.error) atomicExch(acc.error, rc.error);
            sum_e1mm += rc.e1_mm; sum_e1b += rc.e1_bases; sum_e2mm += rc.e2_mm; sum_e2b += rc.e2_bases;
            sum_mm += rc.mm; sum_b += rc.bases; sum_blk += rc.blocks;
            big_any = (rc.bases | rc.mm | rc.blocks) >= (1u << 26);
            // Read-Length inputs: per-tile max span + batch-level extremes
            const uint32_t sp = rc.rl_eligible ? rc.rl_span : 0u;
            const uint32_t wsp = wave_max_u32(sp);
            if (l == 0) acc.tile_span[w0 >> 6] = wsp;
            l_span = sp > l_span ? sp : l_span;
            if (rc.rl_eligible) {
                const uint32_t lq = (uint32_t)rc.rl_lqseq;
                l_lmin = lq < l_lmin ? lq : l_lmin; l_lmax = lq > l_lmax ? lq : l_lmax;
            }
        }
        {   // stage the next tile
            const uint64_t i1 = i + 64ull, i2 = i + 128ull;
            const int4 t_cv = nx_cv;
            int4 t_av = zero4;
            if (i1 < wend) t_av = ld32(aux4 + w0 + 64, (uint32_t)l);
            const uint32_t co = (uint32_t)t_cv.w;
            cur_cg[0] = ld32(b.cigar, co); cur_cg[1] = ld32(b.cigar, co + 1); cur_cg[2] = ld32(b.cigar, co + 2); cur_cg[3] = ld32(b.cigar, co + 3);
            nx_cv = zero4;
            if (i2 < wend) nx_cv = ld32(core4 + w0 + 128, (uint32_t)l);
            cur_cv = t_cv; cur_av = t_av;
        }
        // the tile's one-per-record counters: lane c holds counter c's increment (one LDS instruction)
        if (l < RSQC_N_COUNTERS && cnt.vec) atomicAdd(&S.cnt32[l], cnt.vec);
        // the u32 sums cannot overflow within 31 tiles unless a record carries an absurd value: flush right away then
        if (++pending == 31 || __ballot(big_any) != 0ull) flush_counts();
    }
    flush_counts();
    {
        const uint32_t ws = wave_max_u32(l_span), wmn = wave_min_u32(l_lmin), wmx = wave_max_u32(l_lmax);
        if (l == 0) { atomicMax(&S.rl[0], ws); atomicMin(&S.rl[1], wmn); atomicMax(&S.rl[2], wmx); }
    }
    __syncthreads();
    for (int c = threadIdx.x; c < RSQC_N_COUNTERS; c += blockDim.x) {
        const unsigned long long v = S.cnt[c] + (unsigned long long)S.cnt32[c];
        if (v) atomicAdd(&acc.counters[c], v);
    }
    if (threadIdx.x == 0) {
        atomicMax(&acc.rl_stats[0], S.rl[0]); atomicMin(&acc.rl_stats[1], S.rl[1]); atomicMax(&acc.rl_stats[2], S.rl[2]);
        acc.pair_chunk_count[blockIdx.x] = 0u;       // (this kernel emits no (gene, name) pairs: the general kernel does, into the slow-path region)
    }
}

}  // namespace rsqc
#include "rsqc_k1.h"
namespace rsqc {

// rank table of the elementary-interval index (rsqc_read.h: EiRank): one thread per word of 64 positions of one contig
__global__ void __launch_bounds__(256)
ei_rank_kernel(const EiEntry *ei, uint32_t ei_lo, uint32_t ei_hi, EiRank *rank, uint32_t n_words) {
    const uint32_t w = blockIdx.x * blockDim.x + threadIdx.x;
    if (w >= n_words) return;
    uint32_t lo = ei_lo, hi = ei_hi;                      // first entry of the contig with pos >= w * 64
    while (lo < hi) { const uint32_t m = lo + ((hi - lo) >> 1); if (((uint32_t)ei[m].pos >> 6) < w) lo = m + 1; else hi = m; }
    EiRank r = {0u, 0u, lo, 0u};
    for (uint32_t j = lo; j < ei_hi && ((uint32_t)ei[j].pos >> 6) == w; ++j) {
        const uint32_t bit = (uint32_t)ei[j].pos & 63u;
        if (bit < 32) r.lo |= 1u << bit; else r.hi |= 1u << (bit - 32);
    }
    rank[w] = r;
}
void launch_ei_rank(hipStream_t s, const EiEntry *ei, uint32_t ei_lo, uint32_t ei_hi, EiRank *rank, uint32_t n_words) {
    if (n_words) hipLaunchKernelGGL(ei_rank_kernel, dim3((n_words + 255) / 256), dim3(256), 0, s, ei, ei_lo, ei_hi, rank, n_words);
}

// ------------------------------------------------------------------ KR: rsqc_kr.h
}  // namespace rsqc
#include "rsqc_kr.h"
namespace rsqc {

// ------------------------------------------------------------------ K4 (rsqc_k4.h) and the retirement of a batch's pairs
// The pairs of a batch live in per-K1-block chunks (+ one slow-path region), each chunk in file order; the pairs of
// batches that have been retired (rsqc_api.cpp) sit in one dense arena in file order.
#define RSQC_K4_SLOW_BLOCKS 32

// Retirement of a batch: its chunks, one after the other, appended to the arena.  Workgroup k < n_chunks copies chunk k
// (its destination = the sum of the counts before it); the last RSQC_K4_SLOW_BLOCKS workgroups share the slow-path region.
__global__ void __launch_bounds__(256)
pairs_append_kernel(const PairRec *src, uint32_t chunk_cap, const uint32_t *counts, uint32_t n_chunks,
                    uint32_t slow_base, uint32_t slow_cap, PairRec *dst_pairs) {
    __shared__ unsigned long long s_part[256];
    const uint32_t k = blockIdx.x < n_chunks ? blockIdx.x : n_chunks;
    unsigned long long before = 0;
    for (uint32_t j = threadIdx.x; j < k; j += blockDim.x) before += counts[j] < chunk_cap ? counts[j] : chunk_cap;
    s_part[threadIdx.x] = before;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) { if ((int)threadIdx.x < o) s_part[threadIdx.x] += s_part[threadIdx.x + o]; __syncthreads(); }
    const unsigned long long dst = s_part[0];
    uint32_t base, count, lo = 0, step = 1;
    if (blockIdx.x < n_chunks) { base = blockIdx.x * chunk_cap; count = counts[k] < chunk_cap ? counts[k] : chunk_cap; }
    else { base = slow_base; count = counts[n_chunks] < slow_cap ? counts[n_chunks] : slow_cap; lo = blockIdx.x - n_chunks; step = gridDim.x - n_chunks; }
    for (uint32_t i = lo * blockDim.x + threadIdx.x; i < count; i += step * blockDim.x) {
        dst_pairs[dst + i] = src[base + i];
    }
}
void launch_pairs_append(hipStream_t s, const PairRec *src, uint32_t chunk_cap, const uint32_t *counts,
                         uint32_t n_chunks, uint32_t slow_base, uint32_t slow_cap, PairRec *dst) {
    hipLaunchKernelGGL(pairs_append_kernel, dim3(n_chunks + RSQC_K4_SLOW_BLOCKS), dim3(256), 0, s, src, chunk_cap, counts,
                       n_chunks, slow_base, slow_cap, dst);
}

}  // namespace rsqc
#include "rsqc_k4.h"
namespace rsqc {

// ------------------------------------------------------------------ K3: rsqc_k3.h
}  // namespace rsqc
#include "rsqc_k3.h"
namespace rsqc {

// ------------------------------------------------------------------ --fasta
// One G/C bit per base from the FASTA text of a contig (64 bases per thread).
__global__ void __launch_bounds__(256)
gc_pack_kernel(const uint8_t *ascii, uint64_t len, unsigned long long *words) {
    const uint64_t n_words = (len + 63) / 64;
    for (uint64_t w = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; w < n_words; w += (uint64_t)gridDim.x * blockDim.x) {
        unsigned long long x = 0ull;
        const uint64_t b0 = w * 64, b1 = b0 + 64 < len ? b0 + 64 : len;
        for (uint64_t i = b0; i < b1; ++i) {
            const uint8_t ch = ascii[i] | 0x20u;                       // G g C c, src/Fasta.cpp:72
            if (ch == 'g' || ch == 'c') x |= 1ull << (i - b0);
        }
        words[w] = x;
    }
}
void launch_gc_pack(hipStream_t s, const uint8_t *ascii, uint64_t len, unsigned long long *words) {
    const uint64_t n_words = (len + 63) / 64;
    if (!n_words) return;
    hipLaunchKernelGGL(gc_pack_kernel, dim3((unsigned)std::min<uint64_t>((n_words + 255) / 256, 65536)), dim3(256), 0, s, ascii, len, words);
}

// Per-exon GC as fetched by computeCoverage (src/Metrics.cpp:299-303): getSeq(chr, start, start + length) -- the
// 1-based start used as a 0-based offset -- clipped at the contig end (bioio.hpp:306); -1 when the FASTA lacks the contig.
__global__ void __launch_bounds__(256)
exon_gc_kernel(DevAnnotation a, DevReference R, double *exon_gc) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (uint32_t)a.n_exons) return;
    int lo = 0, hi = a.n_contigs;                                      // contig of the row: last one with ex_lo <= i
    while (hi - lo > 1) { const int m = (lo + hi) >> 1; if (a.contig[m].ex_lo <= i) lo = m; else hi = m; }
    while (lo + 1 < a.n_contigs && a.contig[lo].ex_hi <= i) ++lo;      // (contigs without exons share ex_lo)
    const ExonRow row = a.ex[i];
    double v = -1.0;
    if (R.word_off[lo] != ~0ull) {
        const int64_t L = (int64_t)R.length[lo];
        int64_t s = row.start, e = (int64_t)row.start + ((int64_t)row.end - row.start + 1);
        if (s >= 0 && s < L) {
            if (e > L) e = L;
            v = gc_value(gc_count(R, lo, s, e), (uint64_t)(e - s));
        }
    }
    exon_gc[a.ex_id[i]] = v;
}
void launch_exon_gc(hipStream_t s, const DevAnnotation &a, const DevReference &R, double *exon_gc) {
    if (a.n_exons <= 0) return;
    hipLaunchKernelGGL(exon_gc_kernel, dim3((a.n_exons + 255) / 256), dim3(256), 0, s, a, R, exon_gc);
}

// Candidates of the fragment GC branch (src/Expression.cpp:459): contig in the FASTA, high quality, ONE aligned block
// that lies inside exactly ONE exon row (then exonic, alignedExons.size() == 1 and doExonMetrics hold), and
// 100 < |InsertSize| < 1000.  A separate pass over the batch, launched only when a reference is set: the per-read
// kernel of runs without --fasta is untouched.
// 1 024 threads per workgroup: a pass reserves its candidates' slots with ONE returning atomic on the list's counter, and one address
// completes about 88 of those per microsecond chip-wide -- with 256-record passes the 400 k reservations of 100 M records were the
// kernel's 3.1 ms (profiles/r6_kernel_stats_fasta.txt).
constexpr int GC_CAND_THREADS = 1024;
__global__ void __launch_bounds__(GC_CAND_THREADS)
gc_candidates_kernel(DevAnnotation a, DevParams p, DevBatch b, DevReference R, GcCandidates out, int *error) {
    __shared__ uint32_t s_base, s_count;
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i0 = (uint64_t)blockIdx.x * blockDim.x; i0 < b.n; i0 += stride) {
        const uint64_t i = i0 + threadIdx.x;
        if (threadIdx.x == 0) s_count = 0u;
        __syncthreads();
        bool emit = false; uint32_t row_hit = 0; int32_t endpos = 0, tid = 0; uint32_t flag_lq = 0; uint64_t qhash = 0;
        if (i < b.n) {
            Record r;
            if (load_record(b, i, find_segment(b, i0), r)) {               // (the pass's first record: a wave-uniform hint, scalar loads; load_record advances from it)
                RecordCounters rc; bool hq; uint32_t aligned; Blocks B;
                const bool go = gate_cascade(a, p, r, rc, hq, aligned, B);
                const int64_t isz = r.isize < 0 ? -(int64_t)r.isize : (int64_t)r.isize;
                if (go && hq && !p.legacy && B.nb == 1 && isz > 100 && isz < 1000 && R.word_off[r.tid] != ~0ull) {   // (--legacy never reaches the GC branch, src/RNASeQC.cpp:364)
                    const int32_t bs = B.bs[0], be = B.bs[0] + (int32_t)B.len[0];
                    uint32_t n_in = 0;
                    query_block(a, a.contig[r.tid], bs, be, read_strand_of(p, r.flag), (ClassFlags *)nullptr,
                                [&](uint32_t row_i, const ExonRow &, bool contained) { if (contained) { ++n_in; row_hit = row_i; } });
                    if (n_in == 1) {
                        emit = true; endpos = rc.endpos; tid = r.tid; qhash = r.qhash;
                        flag_lq = ((uint32_t)r.l_qseq & 0x7FFFFFFFu) | (r.pos != r.mpos ? 0x80000000u : 0u);
                    }
                }
            }
        }
        uint32_t my = 0;
        if (emit) my = atomicAdd(&s_count, 1u);                        // LDS: one global reservation per workgroup
        __syncthreads();
        if (threadIdx.x == 0 && s_count) s_base = atomicAdd(out.count, s_count);
        __syncthreads();
        if (emit) {
            const uint32_t slot = s_base + my;
            if (slot < out.cap) {
                uint32_t seg = find_segment(b, i0);
                while (seg + 1 < b.n_seg && b.seg_start[seg + 1] <= i) ++seg;
                out.file_index[slot] = batch_file_index(b, seg, i); out.qhash[slot] = qhash; out.row[slot] = row_hit;
                out.h2[slot] = b.qhash2 ? b.qhash2[i] : 0u;
                out.endpos[slot] = endpos; out.flag_lq[slot] = flag_lq; out.tid[slot] = tid;
            } else atomicExch(error, RSQC_ERR_CAPACITY);
        }
        __syncthreads();
    }
}
void launch_gc_candidates(hipStream_t s, const DevAnnotation &a, const DevParams &p, const DevBatch &b, const DevReference &R,
                          const GcCandidates &out, int *error) {
    if (!b.n) return;
    const uint64_t blocks = (b.n + GC_CAND_THREADS - 1) / GC_CAND_THREADS;
    hipLaunchKernelGGL(gc_candidates_kernel, dim3((unsigned)std::min<uint64_t>(blocks, 4096)), dim3(GC_CAND_THREADS), 0, s, a, p, b, R, out, error);
}



// ------------------------------------------------------------------ result packing
// exon_hit[id] = the exon has a map entry in the reference's exonCounts (a non-zero sum; src/RNASeQC.cpp:513).
// Runs right before the read-back, i.e. after a multi-GPU reduction of the sums.
__global__ void __launch_bounds__(256)
pack_results_kernel(const double *exon_acc, uint8_t *exon_hit, uint32_t n_exons) {
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n_exons; i += gridDim.x * blockDim.x) exon_hit[i] = exon_acc[i] > 0.0 ? 1 : 0;
}

// ------------------------------------------------------------------ reset
// one launch instead of a handful of memsets: zero the accumulator arena and the coverage array; the vector that
// holds rl_stats is written armed ({max span 0, min l_qseq UINT_MAX, max l_qseq 0, -})
__global__ void __launch_bounds__(256)
reset_kernel(uint4 *arena, size_t arena_vec, uint4 *cov, size_t cov_vec, size_t rl_vec) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    const uint4 z = {0u, 0u, 0u, 0u}, armed = {0u, 0xFFFFFFFFu, 0u, 0u};
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < cov_vec; i += stride) cov[i] = z;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < arena_vec; i += stride) arena[i] = i == rl_vec ? armed : z;
}

// ------------------------------------------------------------------ shard reduction (single-process multi-GPU)
// dst += src over the three reducible ranges of a context's result arena (rsqc_device_vectors): what the RCCL all_reduce of
// a one-process-per-GPU run does, for a process that drives several GPUs itself: the peer's ranges arrive by
// hipMemcpyPeerAsync (xGMI) and are added here.
__global__ void __launch_bounds__(256)
reduce_add_kernel(unsigned long long *du, const unsigned long long *su, size_t nu, double *df, const double *sf, size_t nf,
                  uint8_t *db, const uint8_t *sb, size_t nb) {
    const size_t stride = (size_t)gridDim.x * blockDim.x, t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (size_t i = t; i < nu; i += stride) du[i] += su[i];
    for (size_t i = t; i < nf; i += stride) df[i] += sf[i];
    for (size_t i = t; i < nb; i += stride) db[i] = (uint8_t)(db[i] + sb[i]);
}
void launch_reduce_add(hipStream_t s, unsigned long long *du, const unsigned long long *su, size_t nu, double *df, const double *sf, size_t nf,
                       uint8_t *db, const uint8_t *sb, size_t nb) {
    hipLaunchKernelGGL(reduce_add_kernel, dim3(512), dim3(256), 0, s, du, su, nu, df, sf, nf, db, sb, nb);
}

// ------------------------------------------------------------------ launch wrappers
void launch_pack_results(hipStream_t s, const double *exon_acc, uint8_t *exon_hit, uint32_t n_exons) {
    if (n_exons) hipLaunchKernelGGL(pack_results_kernel, dim3((n_exons + 255) / 256 < 1024 ? (n_exons + 255) / 256 : 1024), dim3(256), 0, s, exon_acc, exon_hit, n_exons);
}
void launch_reset(hipStream_t s, void *arena, size_t arena_bytes, void *cov, size_t cov_bytes, uint32_t *rl_min) {
    // both allocations are 16-byte multiples with slack (rsqc_api.cpp: dev_alloc)
    // rl_min = &rl_stats[1]; rl_stats starts a 16-byte vector of the arena (off_misc + 32)
    const size_t rl_vec = (size_t)((char *)rl_min - 4 - (char *)arena) / 16;
    hipLaunchKernelGGL(reset_kernel, dim3(1024), dim3(256), 0, s, (uint4 *)arena, (arena_bytes + 15) / 16, (uint4 *)cov, (cov_bytes + 15) / 16, rl_vec);
}
void launch_frag_compact(hipStream_t s, const FragCandidates &src, const FragCandidates &dst, uint64_t n_rec, int k1_grid) {
    hipLaunchKernelGGL(frag_compact_kernel, dim3(k1_grid), dim3(256), 0, s, src, dst, (uint32_t)n_rec, (uint32_t)k1_grid);
}
void launch_classify(hipStream_t s, int grid, int variant, const DevAnnotation &a, const DevParams &p, const DevBatch &b,
                     const DevAccum &acc) {
    if (variant < 0) hipLaunchKernelGGL(classify_count_kernel_legacy, dim3(grid), dim3(RSQC_K1_THREADS), 0, s, a, p, b, acc);
    else {
#if defined(RSQC_K1_PROF) || defined(RSQC_DIAG_KNOBS)
        // (diagnostic builds) RSQC_K1_LDS_PAD: extra dynamic LDS per workgroup, i.e. fewer resident waves -- tells latency-bound from issue-bound
        static const unsigned pad = getenv("RSQC_K1_LDS_PAD") ? (unsigned)atoi(getenv("RSQC_K1_LDS_PAD")) : 0u;
#else
        const unsigned pad = 0u;
#endif
        const K1Args A{a, p, b, acc};
        if (a.have_bed) hipLaunchKernelGGL(classify_ei_kernel<true>, dim3(grid), dim3(RSQC_K1_THREADS), pad, s, A);
        else hipLaunchKernelGGL(classify_ei_kernel<false>, dim3(grid), dim3(RSQC_K1_THREADS), pad, s, A);
    }
}
// the records classify_ei_kernel deferred (more than eight operations / three blocks): one wave per call of 64, grid-stride
void launch_classify_long(hipStream_t s, int k1_grid, const DevAnnotation &a, const DevParams &p, const DevBatch &b, const DevAccum &acc) {
    const K1Args A{a, p, b, acc};
    hipLaunchKernelGGL(classify_long_kernel, dim3((unsigned)rsqc_long_grid(k1_grid)), dim3(RSQC_K1_THREADS), 0, s, A, (uint32_t)k1_grid);
}
void launch_classify_slow(hipStream_t s, const DevAnnotation &a, const DevParams &p, const DevBatch &b,
                          const DevAccum &acc) {
    if (p.legacy) {
        const uint64_t blocks = (b.n + RSQC_SLOW_THREADS - 1) / RSQC_SLOW_THREADS;
        hipLaunchKernelGGL(classify_slow_kernel<true>, dim3((unsigned)std::min<uint64_t>(blocks ? blocks : 1, RSQC_SLOW_LEGACY_GRID)), dim3(RSQC_SLOW_THREADS), 0, s, a, p, b, acc);
    } else {
        // the number of listed records is only known on the device: enough workgroups for 1 record in 200 to take ONE record per
        // thread (the code is a chain of dependent loads: parallelism, not iterations); workgroups beyond the list leave at once
        const uint64_t blocks = std::max<uint64_t>(64, std::min<uint64_t>(2048, b.n / 200 / RSQC_SLOW_THREADS + 1));
        hipLaunchKernelGGL(classify_slow_kernel<false>, dim3((unsigned)blocks), dim3(RSQC_SLOW_THREADS), 0, s, a, p, b, acc);
    }
}
void launch_read_length(hipStream_t s, const DevAnnotation &a, const DevParams &p, const DevBatch &b,
                        const DevAccum &acc, uint32_t *summary) {
    hipLaunchKernelGGL(read_length_kernel, dim3(acc.rl_seg ? b.n_seg : 1u), dim3(64), 0, s, a, p, b, acc, summary);
}
void launch_frag_layout(hipStream_t s, const unsigned long long *gene_reads, uint32_t n_genes, const FragPlan &P, int *error) {
    const uint32_t blocks = (n_genes + 1023u) / 1024u;
    if (!blocks) return;
    hipLaunchKernelGGL(frag_layout_totals_kernel, dim3(blocks), dim3(1024), 0, s, gene_reads, n_genes, P.blk_space, P.blk_parts, error);
    hipLaunchKernelGGL(frag_layout_kernel, dim3(blocks), dim3(1024), 0, s, gene_reads, n_genes, P.blk_space, P.blk_parts, P.part_first,
                       P.ginfo, P.cursor, P.part_info, P.full_n);
}
// list_blocks: workgroups that share the dense region behind the chunks (0 = the default for a batch's slow-path region)
void launch_frag_local(hipStream_t s, const DevAccum &acc, uint32_t n_chunks, const FragPlan &P, uint32_t list_blocks) {
    hipLaunchKernelGGL(frag_local_kernel, dim3(frag_local_chunk_wgs(n_chunks) + (list_blocks ? list_blocks : RSQC_K4_SLOW_BLOCKS)), dim3(RSQC_K4L_THREADS), 0, s, acc.pairs, acc.pair_chunk_cap, acc.pair_chunk_count, n_chunks, acc.pair_slow_base, acc.pair_slow_cap,
                       P.ginfo, P.cursor, P.list, acc.error);
}
void launch_frag_count(hipStream_t s, uint32_t n_genes, const FragPlan &P, uint32_t parts_bound, unsigned long long *gene_frag, int *error) {
    if (n_genes == 0) return;                   // (no layout was written: launch_frag_layout returns early too)
    const uint32_t grid = parts_bound < 16384u ? (parts_bound ? parts_bound : 1u) : 16384u;
    hipLaunchKernelGGL((frag_count_kernel<RSQC_K4_PART_SLOTS / 2>), dim3(grid), dim3(RSQC_K4_COUNT_THREADS), 0, s, P.part_first + n_genes, P.cursor, P.part_info, P.list, gene_frag,
                       P.full_list, P.full_n, error);
    hipLaunchKernelGGL((frag_count_kernel<RSQC_K4_PART_SLOTS>), dim3(grid < 1024u ? grid : 1024u), dim3(RSQC_K4_COUNT_THREADS), 0, s, P.part_first + n_genes, P.cursor, P.part_info, P.list, gene_frag,
                       P.full_list, P.full_n, error);
}
#ifndef RSQC_K3_MEDIUM_T
#define RSQC_K3_MEDIUM_T uint32_t
#endif
#ifndef RSQC_K3_SMALL_T
#define RSQC_K3_SMALL_T uint32_t
#endif
void launch_gene_coverage(hipStream_t s, hipStream_t s2, hipStream_t s3, const GeneCovArgs &A, uint32_t n_large, uint32_t n_medium, uint32_t n_xlarge,
                          uint32_t n_le6144, uint32_t n_le3072, uint32_t n_le2048, uint32_t n_le1024) {
    if (A.n_listed <= 0) return;
    // gene_order is sorted by coding length, longest first: [0, n_large) x 1024 threads,
    // [n_large, n_large + n_medium) x 256 threads, the rest one wave each; the launches are independent
    // (disjoint genes) and go to their own streams so that they overlap
    const uint32_t n = (uint32_t)A.n_listed, n_small = n - n_large - n_medium;
    const bool wide = A.bias_window > 128;
#define RSQC_K3_LAUNCH(T, COVT, CAP, COUNT, FIRST, STREAM)                                                                   \
    if (COUNT) {                                                                                                        \
        if (wide) hipLaunchKernelGGL((gene_coverage_kernel<T, RSQC_MAX_BIAS_WINDOW, COVT, CAP>), dim3(COUNT), dim3(T), 0, STREAM, A, FIRST); \
        else hipLaunchKernelGGL((gene_coverage_kernel<T, 128, COVT, CAP>), dim3(COUNT), dim3(T), 0, STREAM, A, FIRST);  \
    }
    // the 1024-thread class in two LDS sizes: a workgroup that holds 146 KB keeps its CU to itself, one that holds 64 KB leaves
    // room for the fragment workgroups running beside it.  (The runtime maps streams onto four hardware queues: with K4 on the
    // context's stream there are three for K3; the 64 KB class goes in front of the one-wave classes.)
    // The other classes in several LDS sizes too (round 6): a workgroup's LDS is its gene's coverage vector, and sized for the LONGEST gene of
    // a class it limits the workgroups a CU holds -- 8 one-wave workgroups (16 KB each: a quarter of the CU's wave slots) for the 53 k genes
    // of up to 4 096 bases, of which 38 k have at most 1 024 -- and takes the LDS the fragment kernels beside them need.  gene_order is sorted
    // by coding length, longest first, so a class is a range of it; the counts of genes of up to 6 144 / 3 072 / 2 048 / 1 024 bases come
    // from the host (end-of-file kernels 1.57 -> 1.46 ms with the one-wave class in three sizes, call r6j).
#ifndef RSQC_K3_SMALL_SPLIT
#define RSQC_K3_SMALL_SPLIT 1
#endif
    static_assert(RSQC_K3_SMALL_MAX == 4096 && RSQC_K3_MEDIUM_MAX == 12288, "the split points below sit inside these classes");
    if (!RSQC_K3_SMALL_SPLIT || n_le3072 > n_small || n_le6144 < n_small || n_le6144 > n_small + n_medium) { n_le6144 = n_small; n_le3072 = 0; n_le2048 = 0; n_le1024 = 0; }
    if (n_le2048 > n_le3072) n_le2048 = n_le3072;
    if (n_le1024 > n_le2048) n_le1024 = n_le2048;
    const uint32_t n_med_short = n_le6144 - n_small;                       // genes of 4 097 .. 6 144 bases: the end of the medium class
    RSQC_K3_LAUNCH(1024, uint16_t, RSQC_K3_LARGE_LDS16, n_xlarge, 0u, s)
    RSQC_K3_LAUNCH(256, RSQC_K3_MEDIUM_T, RSQC_K3_MEDIUM_MAX, n_medium - n_med_short, n_large, s2)
    RSQC_K3_LAUNCH(256, RSQC_K3_MEDIUM_T, 6144, n_med_short, n_large + n_medium - n_med_short, s2)
    RSQC_K3_LAUNCH(1024, uint16_t, RSQC_K3_LARGE2_LDS16, n_large - n_xlarge, n_xlarge, s3)
    // the one-wave classes behind the 256-thread ones (0.3 + 0.2 ms), NOT behind the 64 KB class: beside the fragment kernels that one takes
    // 0.8 ms, and queued behind it the one-wave classes ended after the fragment count -- the stage's last kernel (timeline of call r6k)
    RSQC_K3_LAUNCH(64, RSQC_K3_SMALL_T, RSQC_K3_SMALL_MAX, n_small - n_le3072, n_large + n_medium, s2)
    RSQC_K3_LAUNCH(64, RSQC_K3_SMALL_T, 3072, n_le3072 - n_le2048, n - n_le3072, s2)
    RSQC_K3_LAUNCH(64, RSQC_K3_SMALL_T, 2048, n_le2048 - n_le1024, n - n_le2048, s2)
    RSQC_K3_LAUNCH(64, RSQC_K3_SMALL_T, 1024, n_le1024, n - n_le1024, s2)
#undef RSQC_K3_LAUNCH
}

}  // namespace rsqc
