// rsqc_api.cpp -- the C ABI of include/rnaseqc_amd.h on top of the HIP kernels.
//
// One context = one GPU = one shard of contigs.  Everything the per-record path reads
// (annotation index, uploaded batches) and writes (count vectors, per-base coverage,
// de-dup tables) stays resident in HBM; the host only builds the index once, enqueues
// work on the context's stream and reads the small result vectors back at end of file.
// There is no CPU fallback: without a HIP device rsqc_create() fails.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>   // types only: the library is bound at run time (rccl_api)
#include <dlfcn.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include "rsqc_device.h"
#include "rsqc_index.h"
#include "rsqc_decode.h"

// Knobs that make the library SKIP work (wrong or incomplete results) exist only in the diagnostic build (`make prof`,
// -DRSQC_K1_PROF): the product library does not read them.
#if defined(RSQC_K1_PROF) || defined(RSQC_DIAG_KNOBS)       /* (`make variant NAME=diag DEFS=-DRSQC_DIAG_KNOBS`: the knobs without the section timers) */
#define RSQC_DIAG(name) getenv(name)
#else
#define RSQC_DIAG(name) ((const char *)nullptr)
#endif
// (diagnostic build only, RSQC_HOST_TRACE=1: host clock at the steps of a pass to stderr -- where a pass spends what the kernels' events do not show)
#if defined(RSQC_K1_PROF) || defined(RSQC_DIAG_KNOBS)
static void host_trace(const char *what) {
    static const bool on = getenv("RSQC_HOST_TRACE") != nullptr;
    if (!on) return;
    static std::chrono::steady_clock::time_point last = std::chrono::steady_clock::now();
    const auto now = std::chrono::steady_clock::now();
    fprintf(stderr, "[host] %-28s +%8.1f us\n", what, std::chrono::duration<double, std::micro>(now - last).count());
    last = now;
}
#define RSQC_TRACE(what) host_trace(what)
#else
#define RSQC_TRACE(what) ((void)0)
#endif


using namespace rsqc;

namespace {

struct DevBuf {
    void *p = nullptr; size_t bytes = 0;
    void release() { if (p) (void)hipFree(p); p = nullptr; bytes = 0; }
};

struct UploadedBatch {
    DevBatch d{};
    std::vector<DevBuf> bufs;
    uint64_t n = 0, n_cigar_total = 0, file_index_base = 0;
    std::vector<uint64_t> seg_file_index, seg_records;   // a batch of several file ranges (rsqc_batch.seg_file_index): per segment
    DevBuf rl_seg;                                   // ... and its per-segment Read-Length inputs [3 * n_seg] (armed at every submit)
    bool in_use = false;
    bool pooled = false;           // transient upload: its buffers go back to the context's pool
};

// What a submitted batch leaves behind for the end-of-file stage is written into WORST-CASE sized buffers (the counts
// are only known on the device).  A batch that has completed is RETIRED at the next rsqc_submit / rsqc_wait: its
// counts are read from a page-locked mirror, what it actually emitted is appended to a growing arena, and the
// worst-case buffers go back to the pool -- so the memory held until rsqc_finalize is what was emitted (16 B per
// (gene, name) pair, 28 B per fragment-size candidate, 32 B per GC candidate) plus the buffers of the batches in flight.
struct PairBuf {                // (gene, qname-hash) pairs of one submitted batch
    DevBuf rec, counts;         // rec: PairRec[cap] {gene, second name hash (zero for a batch without rsqc_batch.qhash2), name hash}; counts: [n_chunks] per K1 block, then [1] slow-path counter
    uint64_t cap = 0;           // pair slots allocated
    uint32_t n_chunks = 0, chunk_cap = 0, slow_base = 0, slow_cap = 0;
    uint32_t counts_cap = 0;
    uint64_t pairs_bound = 0;   // most pairs the batch can have emitted
    uint32_t *h_counts = nullptr;   // page-locked mirror of `counts` (copied when the batch's kernels are done)
    hipEvent_t done = nullptr;      // the counts have arrived in h_counts
    hipEvent_t kernels = nullptr;   // the batch's per-record kernels are through (what the copy of the counts waits for, on a side stream)
    bool used = false;
};

struct FragBuf {                // fragment-size candidates of one submitted batch (BED runs only)
    DevBuf file, qhash, name, endpos, fs, h2, count;
    DevBuf r_file, r_qhash, r_name, r_endpos, r_fs, r_h2, r_counts;   // the per-record kernel's workgroup regions (packed into the columns above by frag_compact_kernel)
    uint32_t cap = 0, grid_cap = 0;
    uint32_t *h_count = nullptr;
    bool used = false;
};
struct GcBuf {                  // fragment GC candidates of one submitted batch (--fasta runs only)
    DevBuf file, qhash, row, endpos, flag_lq, tid, h2, count;
    uint32_t cap = 0;
    uint32_t *h_count = nullptr;
    bool used = false;
};
// growing device arrays of the retired batches: a few parallel columns with one fill level
struct Arena {
    DevBuf col[8]; size_t width[8] = {0, 0, 0, 0, 0, 0, 0, 0}; int n_col = 0;
    uint64_t used = 0, cap = 0;
};

// device-side BAM decode (rsqc_decode_*): the window buffers are sized once for the largest call and reused; the batch
// columns a window is parsed into are read by the per-read kernels of that window before the next window's kernels
// (same stream) overwrite them
struct DecodeState {
    bool active = false;
    BamTagSpec tags{};
    uint64_t next_file_index = 0, records = 0;
    uint32_t head = 1u << 22;          // room in front of the window for the carried-over part of a record
    uint32_t tail = 0;                 // bytes carried over, parked at [head - tail, head)
    size_t out_cap = 0, comp_cap = 0, blk_cap = 0;
    DevBuf comp, blocks, ubuf, seg, seg_rec0, seg_ops0, rec_off, ops_at, mark, core, aux, qh2, cigar, seg_tid, seg_start,
           wide_index, wide_nm, wide_lq, wide_nc, sum, carry, tailtmp, scratch;
    DecodeSummary *h_sum = nullptr;    // page-locked
    DevBgzfBlock *h_blocks = nullptr;  // page-locked, blk_cap entries
    bool unsorted = false;
    uint64_t n_bad = 0;
    std::vector<std::string> bad_names;
    std::vector<const char *> bad_ptrs;
    // RSQC_DECODE_PROFILE: stage times of the stream, printed at rsqc_decode_end
    bool profile = false; hipEvent_t pe[4] = {nullptr, nullptr, nullptr, nullptr};
    double ms_copy = 0, ms_inflate = 0, ms_parse = 0, ms_call = 0; uint64_t prof_in = 0, prof_out = 0, prof_calls = 0;
    std::chrono::steady_clock::time_point prof_t0;
    // a call is enqueued, then finished (its summary read, its records submitted) -- at once, or by the next call when the
    // stream is pipelined: the next call's file bytes then cross PCIe beside this call's kernels
    bool pipelined = false, pending = false, pend_limited = false;
    int slot = 0;                      // which half of comp / blocks / h_blocks the call in flight reads
    DecodeWindow pend_w{};
    std::chrono::steady_clock::time_point pend_wall0;
    hipStream_t copy_stream = nullptr; hipEvent_t ev_copy = nullptr;
    std::vector<int32_t> run_tid;      // contig segments of the last window
    rsqc_batch last{};                 // the last window's batch (device pointers): rsqc_decode_window::device_batch
};

}  // namespace

struct rsqc_ctx {
    rsqc_params params{};
    int device = 0;
    hipStream_t stream = nullptr;
    std::string last_error;
    int sticky = 0;
    int k1_variant = 41, k1_grid = 256 * 20;  // workgroups of the per-read kernel (RSQC_K1_GRID overrides), set once at create: four rounds of five workgroups per CU

    // annotation (host copies needed at finalize)
    bool have_ann = false;
    int32_t n_ref = 0, n_contigs = 0, n_genes = 0, n_listed = 0, n_exons = 0;
    std::vector<uint32_t> exon_row_id;          // row -> exon id
    std::vector<DevBuf> ann_bufs;
    DevAnnotation dann{};
    DevParams dparams{};
    // K3 inputs
    const uint32_t *d_ge_off = nullptr, *d_ge_row = nullptr, *d_gene_cov_off = nullptr, *d_gene_coding = nullptr;
    const uint8_t *d_gene_flags = nullptr, *d_gene_owned = nullptr;   // owned by ann_bufs
    const uint32_t *d_gene_order = nullptr;
    uint32_t k3_large = 0, k3_medium = 0, k3_xlarge = 0, k3_le6144 = 0, k3_le3072 = 0, k3_le2048 = 0, k3_le1024 = 0;
    int stream_prio = 0, prio_side = 0;         // RSQC_STREAM_PRIO (rsqc_create)
    uint32_t n_exons_outside_gene = 0;          // of the annotation in use (rsqc_results.exons_outside_gene_row)
    hipEvent_t fin_e0 = nullptr, fin_e1 = nullptr;
    uint64_t cov_entries = 0;
    bool have_bed = false;

    // accumulators
    // one device arena holds every small result vector (single memset at reset, single D2H at finalize):
    // u64[3G+K] | u64 bias3,bias5[L] | f64 exon_acc[E] | f64 gmean,gstd,gcv[L] | f64 ecv[E] | u8 gvalid[L] | u8 ecv_valid[E] | u8 exon_hit[E] | misc[64]
    DevBuf d_arena, d_cov, d_ovf_index, d_tiles;
    DevBuf d_defer;                             // classify_ei_kernel's deferred list (per-workgroup regions, then the dense list): reused batch after batch (stream order)
    DevBuf d_ei_rank;                                   // rank table of the interval index
    char *h_arena = nullptr;                      // pinned host mirror
    size_t arena_bytes = 0, off_u64 = 0, off_exon = 0, off_gmean = 0, off_gstd = 0, off_gcv = 0, off_bias3 = 0,
           off_bias5 = 0, off_ecv = 0, off_gvalid = 0, off_ecvv = 0, off_ehit = 0, off_misc = 0;
    hipStream_t stream2 = nullptr, stream3 = nullptr, stream4 = nullptr;   // K3 (three size classes) runs beside K4
    hipEvent_t ev_fork = nullptr, ev_join = nullptr, ev_join3 = nullptr, ev_join4 = nullptr;
    DevAccum acc{};
    uint64_t tile_cap = 0;
    std::vector<PairBuf> pair_pool;
    std::vector<size_t> pairs_in_flight;        // indices into pair_pool, submission order (batches not retired yet)
    Arena pair_arena, frag_arena, gc_arena;     // what the retired batches emitted
    DevBuf d_arena_count;                       // u32: pair_arena.used for the K4 launch over the arena
    std::vector<DevBuf> parked;                 // outgrown arena columns, freed at the next synchronisation point
    DevBuf d_table, d_tab_off, d_tab_cap;
    std::vector<FragBuf> frag_pool;
    std::vector<size_t> frags_in_flight;
    uint32_t frag_remaining = 0;
    // --fasta
    bool have_ref = false;
    DevBuf d_ref_bits, d_ref_off, d_ref_len, d_gc_bins, d_exon_gc;
    DevReference dref{};
    std::vector<GcBuf> gc_pool;
    std::vector<size_t> gcs_in_flight;
    std::vector<uint64_t> h_gc;                 // [RSQC_GC_BINS + 1]
    std::vector<double> h_exon_gc;              // by exon id
    SortScratch gc_scratch, frag_scratch;
    uint32_t frag_kept = 0;                     // samples run_fragment_sizes left in frag_scratch (k1 / v1)
    // K3 outputs
    bool finalized = false;

    // batches
    std::vector<UploadedBatch *> resident;
    std::vector<UploadedBatch *> transient;     // owned by submit(), freed at wait()
    uint64_t next_record_base = 0;
    bool have_ranges = false;                   // a batch of several file ranges was submitted in this pass: Read Length is composed on the host
    bool have_composed_rl = false; int32_t composed_rl = 0;
    bool early_copied = false;                  // run_finalize_kernels copied everything but geneFragmentCounts and the status words beside the fragment kernels
    std::vector<uint32_t> h_rl_arm;
    int name_mode = -1;                         // -1 no batch yet in this pass; 0 batches without qhash2 (64-bit names); 1 with (96-bit names)
    // per submitted batch: file index of its first record and the Read-Length transfer function the KR kernel leaves
    // on the device (rsqc_shard_info)
    std::vector<uint64_t> batch_file_index, batch_records;
    DevBuf d_rl_summary;
    uint32_t *h_rl_raw = nullptr;               // page-locked (a copy into pageable memory would hold the host until the coverage kernel ahead of it ended)
    size_t h_rl_raw_cap = 0;                    // ... words
    std::vector<uint32_t> h_rl_offset, h_rl_span;
    std::vector<int32_t> h_rl_state;
    std::vector<uint64_t> h_sample_file;        // fragment-size samples kept by this shard (first N by file index), ascending
    std::vector<uint32_t> h_sample_size;

    // timing
    std::vector<std::pair<hipEvent_t, hipEvent_t>> k1_events, h2d_events, long_events;
    std::vector<DevBuf> upload_pool;            // device buffers of retired transient uploads
    std::vector<hipEvent_t> event_pool;
    rsqc_timing timing{};

    DecodeState dec;

    // host results
    std::vector<uint64_t> h_fcount;
    std::vector<int64_t> h_fsize;
    rsqc_results results{};
};

namespace {

int fail(rsqc_ctx *c, int code, const std::string &msg) {
    if (c) { c->last_error = msg; }
    return code;
}

#define HIP_TRY(c, expr)                                                                         \
    do {                                                                                         \
        hipError_t e_ = (expr);                                                                  \
        if (e_ != hipSuccess)                                                                    \
            return fail((c), RSQC_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_));   \
    } while (0)

int dev_alloc(rsqc_ctx *c, DevBuf &b, size_t bytes, bool zero) {
    if (bytes == 0) bytes = 16;
    bytes = (bytes + 15) & ~(size_t)15;          // whole 16-byte vectors (the reset kernel clears uint4s)
    if (b.bytes < bytes) {
        b.release();
        HIP_TRY(c, hipMalloc(&b.p, bytes));
        b.bytes = bytes;
    }
    if (zero) HIP_TRY(c, hipMemsetAsync(b.p, 0, b.bytes, c->stream));
    return 0;
}

// device buffer of at least `bytes` from the context's pool of retired upload buffers (smallest that fits), or empty
DevBuf take_pooled(rsqc_ctx *c, size_t bytes);

template <class T>
int upload(rsqc_ctx *c, std::vector<DevBuf> &owner, const T *host, size_t n, const T **out, bool from_pool = false) {
    DevBuf b;
    size_t bytes = n * sizeof(T);
    if (from_pool) {
        b = take_pooled(c, bytes + 32);
        if (b.p) {
            if (bytes) HIP_TRY(c, hipMemcpyAsync(b.p, host, bytes, hipMemcpyHostToDevice, c->stream));
            owner.push_back(b);
            *out = (const T *)b.p;
            return 0;
        }
    }
    // 32 bytes of slack: kernels load a few entries past the end with unconditional, ignored loads
    HIP_TRY(c, hipMalloc(&b.p, bytes + 32));
    b.bytes = bytes + 32;
    if (bytes) HIP_TRY(c, hipMemcpyAsync(b.p, host, bytes, hipMemcpyHostToDevice, c->stream));
    owner.push_back(b);
    *out = (const T *)b.p;
    return 0;
}

hipEvent_t get_event(rsqc_ctx *c) {
    if (!c->event_pool.empty()) { hipEvent_t e = c->event_pool.back(); c->event_pool.pop_back(); return e; }
    hipEvent_t e = nullptr;
    (void)hipEventCreate(&e);
    return e;
}

int check_device_error(rsqc_ctx *c) {
    int err = 0;
    HIP_TRY(c, hipMemcpyAsync(&err, c->acc.error, sizeof(int), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    if (err) {
        c->sticky = err;
        return fail(c, err, err == RSQC_ERR_BAD_CIGAR ? "Unrecognized Cigar Op" :
                            err == RSQC_ERR_CAPACITY ? "a device-side capacity was exceeded" :
                            err == RSQC_ERR_EMPTY_MEDIAN ? "Cannot compute median of an empty list" : "device error");
    }
    return 0;
}

int resolve_events(rsqc_ctx *c) {
    for (auto &pr : c->k1_events) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, pr.first, pr.second) == hipSuccess) c->timing.classify_ms += ms;
        c->event_pool.push_back(pr.first); c->event_pool.push_back(pr.second);
    }
    c->k1_events.clear();
    for (auto &pr : c->long_events) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, pr.first, pr.second) == hipSuccess) c->timing.classify_long_ms += ms;
        c->event_pool.push_back(pr.second);          // (.first is the K1 pair's second event, recycled above)
    }
    c->long_events.clear();
    for (auto &pr : c->h2d_events) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, pr.first, pr.second) == hipSuccess) c->timing.h2d_ms += ms;
        c->event_pool.push_back(pr.first); c->event_pool.push_back(pr.second);
    }
    c->h2d_events.clear();
    return 0;
}

void free_batch(UploadedBatch *u);

DevBuf take_pooled(rsqc_ctx *c, size_t bytes) {
    size_t best = (size_t)-1;
    for (size_t i = 0; i < c->upload_pool.size(); ++i)
        if (c->upload_pool[i].bytes >= bytes && (best == (size_t)-1 || c->upload_pool[i].bytes < c->upload_pool[best].bytes)) best = i;
    DevBuf b;
    if (best != (size_t)-1) { b = c->upload_pool[best]; c->upload_pool.erase(c->upload_pool.begin() + (long)best); }
    return b;
}
void retire_batch(rsqc_ctx *c, UploadedBatch *u) {
    if (!u->pooled) { free_batch(u); return; }
    for (auto &b : u->bufs) {
        if (c->upload_pool.size() < 64) c->upload_pool.push_back(b); else b.release();
    }
    u->bufs.clear();
    delete u;
}

int zero_accumulators(rsqc_ctx *c) {
    // one zeroing kernel + one store (rl_stats[1] = min l_qseq starts at UINT_MAX) instead of several memsets
    launch_reset(c->stream, c->d_arena.p, c->arena_bytes, c->d_cov.p, c->d_cov.bytes, (uint32_t *)((char *)c->d_arena.p + c->off_misc + 36));
    HIP_TRY(c, hipGetLastError());
    for (auto &pb : c->pair_pool) pb.used = false;
    c->pairs_in_flight.clear();
    c->pair_arena.used = c->frag_arena.used = c->gc_arena.used = 0;
    for (auto &fb : c->frag_pool) fb.used = false;
    c->frags_in_flight.clear();
    c->h_fsize.clear(); c->h_fcount.clear();
    c->frag_remaining = c->have_bed ? c->params.fragment_samples : 0;
    for (auto &gb : c->gc_pool) gb.used = false;
    c->gcs_in_flight.clear();
    if (c->have_ref) {
        HIP_TRY(c, hipMemsetAsync(c->d_gc_bins.p, 0, (RSQC_GC_BINS + 1) * 8, c->stream));
        c->h_gc.assign(RSQC_GC_BINS + 1, 0);
    }
    c->finalized = false;
    c->next_record_base = 0; c->name_mode = -1; c->have_ranges = false; c->have_composed_rl = false;
    c->batch_file_index.clear(); c->batch_records.clear();
    c->h_rl_offset.clear(); c->h_rl_span.clear(); c->h_rl_state.clear();
    c->h_sample_file.clear(); c->h_sample_size.clear(); c->frag_kept = 0;
    c->sticky = 0;
    return 0;
}

int upload_batch(rsqc_ctx *c, const rsqc_batch *b, UploadedBatch *u, bool pooled = false) {
    u->pooled = pooled;
    if (b->n > 0xFFFFFFF0ull || b->n_cigar_total >= (1ull << 30)) return fail(c, RSQC_ERR_ARG, "batch too large");
    DevBatch &d = u->d;
    d.n = b->n; d.n_seg = b->n_seg; d.n_wide = b->n_wide;
    u->n = b->n; u->n_cigar_total = b->n_cigar_total; u->file_index_base = b->file_index_base;
    int rc;
#define UP(field, count) if ((rc = upload(c, u->bufs, b->field, (size_t)(count), &d.field, pooled))) return rc
    UP(core, b->n); UP(aux, b->n);
    UP(cigar, b->n_cigar_total);
    UP(seg_tid, b->n_seg); UP(seg_start, (size_t)b->n_seg + 1);
    UP(wide_index, b->n_wide); UP(wide_nm, b->n_wide); UP(wide_l_qseq, b->n_wide); UP(wide_n_cigar, b->n_wide);
    d.qhash2 = nullptr;
    if (b->qhash2) UP(qhash2, b->n);
    d.seg_file_index = nullptr;
    u->seg_file_index.clear(); u->seg_records.clear();
    if (b->seg_file_index && b->n_seg) {
        UP(seg_file_index, b->n_seg);
        u->seg_file_index.assign(b->seg_file_index, b->seg_file_index + b->n_seg);
        for (uint32_t k = 0; k < b->n_seg; ++k) u->seg_records.push_back(b->seg_start[k + 1] - b->seg_start[k]);
        for (uint32_t k = 0; k + 1 < b->n_seg; ++k)
            if (u->seg_file_index[k + 1] < u->seg_file_index[k] + u->seg_records[k]) return fail(c, RSQC_ERR_ARG, "rsqc_batch.seg_file_index: the ranges must ascend and not overlap");
        DevBuf rs = pooled ? take_pooled(c, (size_t)b->n_seg * 12 + 32) : DevBuf{};
        if (!rs.p) { HIP_TRY(c, hipMalloc(&rs.p, (size_t)b->n_seg * 12 + 32)); rs.bytes = (size_t)b->n_seg * 12 + 32; }
        u->bufs.push_back(rs); u->rl_seg = rs;           // (owned through bufs)
        u->file_index_base = u->seg_file_index[0];
    }
#undef UP
    return 0;
}

void free_batch(UploadedBatch *u) {
    for (auto &b : u->bufs) b.release();
    delete u;
}
// a transient batch whose kernels have completed: keep its device buffers for the next rsqc_submit
void retire_batch(rsqc_ctx *c, UploadedBatch *u);

int arena_reserve(rsqc_ctx *c, Arena &a, uint64_t extra) {
    if (a.used + extra <= a.cap) return 0;
    // doubling, from a floor of 16 M entries: a handful of growth steps for any input.  The old columns are not freed here
    // (hipFree synchronises the device, in the middle of the decode / kernel pipeline): they are parked until the next
    // point where the stream has been synchronised anyway.
    const uint64_t ncap = std::max<uint64_t>(std::max<uint64_t>(a.used + extra, 2 * a.cap), 1ull << 24);
    for (int k = 0; k < a.n_col; ++k) {
        DevBuf nb;
        HIP_TRY(c, hipMalloc(&nb.p, ncap * a.width[k] + 64));
        nb.bytes = ncap * a.width[k] + 64;
        if (a.used) HIP_TRY(c, hipMemcpyAsync(nb.p, a.col[k].p, a.used * a.width[k], hipMemcpyDeviceToDevice, c->stream));
        if (a.col[k].p) c->parked.push_back(a.col[k]);
        a.col[k] = nb;
    }
    a.cap = ncap;
    return 0;
}
void free_parked(rsqc_ctx *c) {                 // caller: the stream has been synchronised
    for (auto &b : c->parked) b.release();
    c->parked.clear();
}

// Retires every in-flight batch whose kernels have completed -- all of them when `all`, waiting for each batch's `done` event
// (the caller need NOT have synchronised: `done` is recorded on the side stream behind the copy of the batch's pair counts, which
// waits for `kernels`, recorded on the main stream behind the batch's last kernel AND behind the copies of its fragment-size / GC
// candidate counts -- so those mirrors are valid too once `done` has fired).
int retire_completed(rsqc_ctx *c, bool all) {
    size_t keep = 0;
    for (size_t k = 0; k < c->pairs_in_flight.size(); ++k) {
        const size_t idx = c->pairs_in_flight[k];
        PairBuf &pb = c->pair_pool[idx];
        // in submission order only: the arena keeps file order, which the fragment de-dup's LDS pass relies on for locality
        const bool done = keep == k && (all || hipEventQuery(pb.done) == hipSuccess);
        if (!done) { c->pairs_in_flight[keep++] = idx; continue; }
        if (all) HIP_TRY(c, hipEventSynchronize(pb.done));          // (the counts travel on a side stream)
        uint64_t total = 0;
        for (uint32_t j = 0; j < pb.n_chunks; ++j) total += std::min(pb.h_counts[j], pb.chunk_cap);
        total += std::min(pb.h_counts[pb.n_chunks], pb.slow_cap);
        if (c->pair_arena.used + total > 0xFFFFFFF0ull) return fail(c, RSQC_ERR_CAPACITY, "more than 2^32 (gene, name) pairs in one pass");
        int rc = arena_reserve(c, c->pair_arena, total);
        if (rc) return rc;
        if (total) launch_pairs_append(c->stream, (const PairRec *)pb.rec.p, pb.chunk_cap, (const uint32_t *)pb.counts.p,
                                       pb.n_chunks, pb.slow_base, pb.slow_cap, (PairRec *)c->pair_arena.col[0].p + c->pair_arena.used);
        c->pair_arena.used += total;
        pb.used = false;                       // (stream order: the append reads the buffer before a later batch writes it)
    }
    c->pairs_in_flight.resize(keep);
    // candidate lists are dense [0, count): plain device-to-device copies.  They are retired together with their batch's
    // pairs: a candidate buffer is in flight exactly as long as the pair buffer submitted with it.
    auto retire_list = [&](std::vector<size_t> &in_flight, size_t n_keep, Arena &arena, auto &&buf_of) -> int {
        const size_t n_retire = in_flight.size() > n_keep ? in_flight.size() - n_keep : 0;
        for (size_t k = 0; k < n_retire; ++k) {
            uint32_t count = 0; uint32_t cap = 0; const void *src[8]; bool *used = nullptr;
            buf_of(in_flight[k], count, cap, src, used);
            if (count > cap) return fail(c, RSQC_ERR_CAPACITY, "candidate overflow");
            int rc = arena_reserve(c, arena, count);
            if (rc) return rc;
            for (int f = 0; f < arena.n_col && count; ++f)
                HIP_TRY(c, hipMemcpyAsync((char *)arena.col[f].p + arena.used * arena.width[f], src[f], (size_t)count * arena.width[f], hipMemcpyDeviceToDevice, c->stream));
            arena.used += count;
            *used = false;
        }
        in_flight.erase(in_flight.begin(), in_flight.begin() + (long)n_retire);
        return 0;
    };
    int rc = retire_list(c->frags_in_flight, c->have_bed ? keep : c->frags_in_flight.size(), c->frag_arena,
                         [&](size_t i, uint32_t &count, uint32_t &cap, const void **src, bool *&used) {
                             FragBuf &fb = c->frag_pool[i]; count = *fb.h_count; cap = fb.cap; used = &fb.used;
                             src[0] = fb.file.p; src[1] = fb.qhash.p; src[2] = fb.name.p; src[3] = fb.endpos.p; src[4] = fb.fs.p; src[5] = fb.h2.p;
                         });
    if (rc) return rc;
    return retire_list(c->gcs_in_flight, (c->have_ref && !c->dparams.legacy) ? keep : c->gcs_in_flight.size(), c->gc_arena,
                       [&](size_t i, uint32_t &count, uint32_t &cap, const void **src, bool *&used) {
                           GcBuf &gb = c->gc_pool[i]; count = *gb.h_count; cap = gb.cap; used = &gb.used;
                           src[0] = gb.file.p; src[1] = gb.qhash.p; src[2] = gb.row.p; src[3] = gb.endpos.p; src[4] = gb.flag_lq.p; src[5] = gb.tid.p; src[6] = gb.h2.p;
                       });
}

PairBuf *acquire_pairs(rsqc_ctx *c, uint64_t cap, uint32_t n_counts, size_t *index) {
    for (size_t i = 0; i < c->pair_pool.size(); ++i)
        if (!c->pair_pool[i].used && c->pair_pool[i].cap >= cap && c->pair_pool[i].counts_cap >= n_counts) {
            c->pair_pool[i].used = true; *index = i; return &c->pair_pool[i];
        }
    PairBuf pb;
    if (hipMalloc(&pb.rec.p, (size_t)cap * sizeof(PairRec)) != hipSuccess) return nullptr;
    if (hipMalloc(&pb.counts.p, (size_t)n_counts * 4) != hipSuccess) return nullptr;
    pb.rec.bytes = (size_t)cap * sizeof(PairRec); pb.counts.bytes = (size_t)n_counts * 4;
    if (hipHostMalloc((void **)&pb.h_counts, (size_t)n_counts * 4, hipHostMallocDefault) != hipSuccess) return nullptr;
    if (hipEventCreateWithFlags(&pb.done, hipEventDisableTiming) != hipSuccess) return nullptr;
    if (hipEventCreateWithFlags(&pb.kernels, hipEventDisableTiming) != hipSuccess) return nullptr;
    pb.cap = cap; pb.counts_cap = n_counts; pb.used = true;
    c->pair_pool.push_back(pb);
    *index = c->pair_pool.size() - 1;
    return &c->pair_pool.back();
}

int run_batch(rsqc_ctx *c, UploadedBatch *u) {
    // record indices ride in 32 bits inside the per-record kernel (31 in its overflow list): checked before anything is recorded for the batch
    if (u->n >= (1ull << 31)) return fail(c, RSQC_ERR_ARG, "batch too large (split it)");
    if (!c->have_ann) return fail(c, RSQC_ERR_ARG, "rsqc_set_annotation must precede rsqc_submit");
    if (c->finalized) return fail(c, RSQC_ERR_ARG, "rsqc_reset required after rsqc_finalize");
    if (u->n == 0) return 0;
    // one name identity per pass: a fragment whose mates carry (qhash, h2) and (qhash, 0) would count as two names
    {
        const int mode = u->d.qhash2 ? 1 : 0;
        if (c->name_mode < 0) c->name_mode = mode;
        else if (c->name_mode != mode)
            return fail(c, RSQC_ERR_ARG, "rsqc_batch.qhash2 must be given for every batch of a pass or for none (the name identity is 96 or 64 bits for the whole pass)");
    }
    const uint64_t tiles = (u->n + RSQC_K1_THREADS - 1) / RSQC_K1_THREADS;
    const uint64_t wave_tiles = (u->n + 63) / 64 + 64;
    if (wave_tiles > c->tile_cap) {
        c->tile_cap = wave_tiles + wave_tiles / 4 + 64;
        int rc = dev_alloc(c, c->d_tiles, c->tile_cap * sizeof(uint32_t), false);
        if (rc) return rc;
        c->acc.tile_span = (uint32_t *)c->d_tiles.p;
    }
    // (gene, qname-hash) pairs of this batch: every K1 block owns a private chunk sized for the
    // worst case of its tiles (FAST_SET pairs per record); 1 M extra slots serve the slow path
    // Workgroups: the context's grid for a batch that fills it; a smaller batch gets one round of the chip (256 CUs x 5) or 2048
    // records per workgroup, whichever is more -- with a workgroup per 256 records (round 4) a 0.8 M-record batch ran three rounds of
    // workgroups that each initialised and flushed their LDS tables for ONE tile per wave (profiles/r5_kernel_stats_dist_selftest_before.txt)
    const int grid = (int)std::min<uint64_t>(tiles, std::min<uint64_t>((uint64_t)c->k1_grid, std::max<uint64_t>(1280, u->n / 2048)));
    const uint64_t total_waves = (uint64_t)grid * (RSQC_K1_THREADS / 64);
    const uint64_t per_wave = (((u->n + total_waves - 1) / total_waves) + 63ull) & ~63ull;   // as in the kernel
    if (!c->dparams.legacy) {                  // the deferred list: a slot per record (a workgroup's region is its record range) + the dense list (whole calls of 64 per workgroup)
        const size_t region_words = u->n + 64, list_words = u->n + 64 * (size_t)grid + 64;
        if (c->d_defer.bytes < (region_words + list_words) * 4) { int rc = dev_alloc(c, c->d_defer, (region_words + list_words + u->n / 2) * 4, false); if (rc) return rc; }
        c->acc.defer_index = (uint32_t *)c->d_defer.p; c->acc.defer_list = c->acc.defer_index + region_words;
    }
    const uint64_t chunk_cap = per_wave * (RSQC_K1_THREADS / 64) * FAST_SET;
    // --legacy: every pair comes from the general kernel (one per gene a record is counted to; 4 per record is far
    // above what annotations produce -- beyond it the run fails with RSQC_ERR_CAPACITY)
    const uint64_t slow_cap = c->dparams.legacy ? std::max<uint64_t>(1ull << 20, 4ull * u->n) + (uint64_t)RSQC_SLOW_LEGACY_GRID * RSQC_SLOW_RES : 1ull << 20;
    // chunks: one per K1 workgroup, then one per workgroup of classify_long_kernel (the records K1 defers; none under --legacy)
    const int n_chunks = grid + (c->dparams.legacy ? 0 : rsqc_long_grid(grid));
    const uint64_t want = chunk_cap * (uint64_t)n_chunks + slow_cap;
    if (want > 0xFFFFFFF0ull) return fail(c, RSQC_ERR_ARG, "batch too large (split it)");
    { int rcr = retire_completed(c, false); if (rcr) return rcr; }     // completed batches hand their buffers back first
    size_t pidx = 0;
    PairBuf *pb = acquire_pairs(c, want, (uint32_t)n_chunks + 1, &pidx);
    if (!pb) return fail(c, RSQC_ERR_HIP, "hipMalloc(pair buffer) failed");
    pb->n_chunks = (uint32_t)n_chunks; pb->chunk_cap = (uint32_t)chunk_cap;
    pb->pairs_bound = std::min<uint64_t>(want, u->n * (uint64_t)FAST_SET + slow_cap);
    pb->slow_base = (uint32_t)(chunk_cap * (uint64_t)n_chunks); pb->slow_cap = (uint32_t)slow_cap;
    c->pairs_in_flight.push_back(pidx);
    // (no per-batch memsets: every K1 workgroup writes its own chunk count, workgroup 0 zeroes the slow-path pair
    //  counter, and the overflow counter is re-armed by the last kernel of the previous batch / the reset kernel)
    DevAccum acc = c->acc;
    acc.pairs = (PairRec *)pb->rec.p;
    acc.pair_chunk_cap = pb->chunk_cap; acc.pair_chunk_count = (uint32_t *)pb->counts.p;
    acc.pair_slow_base = pb->slow_base; acc.pair_slow_cap = pb->slow_cap;
    acc.pair_slow_count = (uint32_t *)pb->counts.p + n_chunks;
    FragCandidates frag_dense{};
    if (c->have_bed) {
        size_t fidx = c->frag_pool.size();
        for (size_t i = 0; i < c->frag_pool.size(); ++i) if (!c->frag_pool[i].used && c->frag_pool[i].cap >= u->n && (c->dparams.legacy || c->frag_pool[i].grid_cap >= (uint32_t)grid)) { fidx = i; break; }
        if (fidx == c->frag_pool.size()) {
            FragBuf fb; fb.cap = (uint32_t)u->n;
            int rc2;
            if ((rc2 = dev_alloc(c, fb.file, u->n * 8, false)) || (rc2 = dev_alloc(c, fb.qhash, u->n * 8, false)) ||
                (rc2 = dev_alloc(c, fb.name, u->n * 4, false)) || (rc2 = dev_alloc(c, fb.endpos, u->n * 4, false)) ||
                (rc2 = dev_alloc(c, fb.fs, u->n * 4, false)) || (rc2 = dev_alloc(c, fb.h2, u->n * 4, false)) || (rc2 = dev_alloc(c, fb.count, 16, false))) return rc2;
            if (!c->dparams.legacy) {
                fb.grid_cap = (uint32_t)grid;
                if ((rc2 = dev_alloc(c, fb.r_file, u->n * 8, false)) || (rc2 = dev_alloc(c, fb.r_qhash, u->n * 8, false)) ||
                    (rc2 = dev_alloc(c, fb.r_name, u->n * 4, false)) || (rc2 = dev_alloc(c, fb.r_endpos, u->n * 4, false)) ||
                    (rc2 = dev_alloc(c, fb.r_fs, u->n * 4, false)) || (rc2 = dev_alloc(c, fb.r_h2, u->n * 4, false)) || (rc2 = dev_alloc(c, fb.r_counts, (size_t)grid * 4, false))) return rc2;
            }
            HIP_TRY(c, hipHostMalloc((void **)&fb.h_count, 16, hipHostMallocDefault));
            c->frag_pool.push_back(fb);
        }
        FragBuf &fb = c->frag_pool[fidx];
        fb.used = true;
        c->frags_in_flight.push_back(fidx);
        HIP_TRY(c, hipMemsetAsync(fb.count.p, 0, 16, c->stream));
        acc.frag.file_index = (uint64_t *)fb.file.p; acc.frag.qhash = (uint64_t *)fb.qhash.p;
        acc.frag.name = (int32_t *)fb.name.p; acc.frag.endpos = (int32_t *)fb.endpos.p;
        acc.frag.flag_size = (uint32_t *)fb.fs.p; acc.frag.count = (uint32_t *)fb.count.p; acc.frag.cap = fb.cap; acc.frag.chunk_count = nullptr;
        acc.frag.h2 = (uint32_t *)fb.h2.p;
        frag_dense = acc.frag;
        if (!c->dparams.legacy) {                 // the per-record kernel writes workgroup regions; frag_compact_kernel packs them (below)
            acc.frag.file_index = (uint64_t *)fb.r_file.p; acc.frag.qhash = (uint64_t *)fb.r_qhash.p; acc.frag.name = (int32_t *)fb.r_name.p;
            acc.frag.endpos = (int32_t *)fb.r_endpos.p; acc.frag.flag_size = (uint32_t *)fb.r_fs.p; acc.frag.chunk_count = (uint32_t *)fb.r_counts.p;
            acc.frag.h2 = (uint32_t *)fb.r_h2.p;
        }
    }
    // file order is part of the boundary: the index of record 0 in the whole file comes from the caller (it decides the
    // first-N cut-off of the fragment-size sampler and the order in which shards are composed); batches arrive in file order
    if (u->file_index_base < c->next_record_base)
        return fail(c, RSQC_ERR_ARG, "batches must be submitted in file order (file_index_base below the end of the previous batch)");
    constexpr size_t kMaxBatches = 1u << 14;
    const bool ranges = !u->seg_file_index.empty();
    if (ranges && c->dparams.legacy) return fail(c, RSQC_ERR_ARG, "rsqc_batch.seg_file_index is not supported with --legacy");
    const size_t n_entries = ranges ? u->seg_file_index.size() : 1;          // order-dependent summaries: one per batch, or one per range
    if (c->batch_file_index.size() + n_entries > kMaxBatches) return fail(c, RSQC_ERR_CAPACITY, "more than 16384 batches / file ranges in one pass");
    if (!c->d_rl_summary.p) { int rc3 = dev_alloc(c, c->d_rl_summary, kMaxBatches * RSQC_RL_SUMMARY_WORDS * 4, false); if (rc3) return rc3; }
    uint32_t *rl_slot = (uint32_t *)c->d_rl_summary.p + c->batch_file_index.size() * RSQC_RL_SUMMARY_WORDS;
    DevBatch d = u->d;
    d.record_base = u->file_index_base;
    acc.rl_seg = nullptr;
    if (ranges) {
        for (size_t k = 0; k < n_entries; ++k) { c->batch_file_index.push_back(u->seg_file_index[k]); c->batch_records.push_back(u->seg_records[k]); }
        c->next_record_base = u->seg_file_index.back() + u->seg_records.back();
        c->have_ranges = true;
        // the per-segment Read-Length inputs, armed {max span 0, min l_qseq UINT_MAX, max l_qseq 0}
        c->h_rl_arm.resize(3 * n_entries);
        for (size_t k = 0; k < n_entries; ++k) { c->h_rl_arm[3 * k] = 0u; c->h_rl_arm[3 * k + 1] = 0xFFFFFFFFu; c->h_rl_arm[3 * k + 2] = 0u; }
        HIP_TRY(c, hipMemcpyAsync(u->rl_seg.p, c->h_rl_arm.data(), n_entries * 12, hipMemcpyHostToDevice, c->stream));
        acc.rl_seg = (uint32_t *)u->rl_seg.p;
    } else {
        c->batch_file_index.push_back(u->file_index_base); c->batch_records.push_back(u->n);
        c->next_record_base = u->file_index_base + u->n;
    }
    hipEvent_t e0 = get_event(c), e1 = get_event(c);
    HIP_TRY(c, hipEventRecord(e0, c->stream));
    launch_classify(c->stream, grid, c->dparams.legacy ? -1 : c->k1_variant, c->dann, c->dparams, d, acc);
    HIP_TRY(c, hipEventRecord(e1, c->stream));
    c->k1_events.emplace_back(e0, e1);
    if (!c->dparams.legacy) {                       // the records it deferred: timed on its own (rsqc_timing.classify_long_ms)
        hipEvent_t e2 = get_event(c);
        launch_classify_long(c->stream, grid, c->dann, c->dparams, d, acc);
        HIP_TRY(c, hipEventRecord(e2, c->stream));
        c->long_events.emplace_back(e1, e2);
    }
    if (c->have_bed && !c->dparams.legacy) launch_frag_compact(c->stream, acc.frag, frag_dense, u->n, grid);
    launch_classify_slow(c->stream, c->dann, c->dparams, d, acc);
    launch_read_length(c->stream, c->dann, c->dparams, d, acc, rl_slot);
    if (c->have_ref && !c->dparams.legacy) {          // --fasta: fragment GC candidates, a separate pass over the batch
        size_t gidx = c->gc_pool.size();
        for (size_t i = 0; i < c->gc_pool.size(); ++i) if (!c->gc_pool[i].used && c->gc_pool[i].cap >= u->n) { gidx = i; break; }
        if (gidx == c->gc_pool.size()) {
            GcBuf gb; gb.cap = (uint32_t)u->n;
            int rc2;
            if ((rc2 = dev_alloc(c, gb.file, u->n * 8, false)) || (rc2 = dev_alloc(c, gb.qhash, u->n * 8, false)) ||
                (rc2 = dev_alloc(c, gb.row, u->n * 4, false)) || (rc2 = dev_alloc(c, gb.endpos, u->n * 4, false)) ||
                (rc2 = dev_alloc(c, gb.flag_lq, u->n * 4, false)) || (rc2 = dev_alloc(c, gb.tid, u->n * 4, false)) || (rc2 = dev_alloc(c, gb.h2, u->n * 4, false)) ||
                (rc2 = dev_alloc(c, gb.count, 16, false))) return rc2;
            HIP_TRY(c, hipHostMalloc((void **)&gb.h_count, 16, hipHostMallocDefault));
            c->gc_pool.push_back(gb);
        }
        GcBuf &gb = c->gc_pool[gidx];
        gb.used = true;
        c->gcs_in_flight.push_back(gidx);
        HIP_TRY(c, hipMemsetAsync(gb.count.p, 0, 16, c->stream));
        GcCandidates gc{(uint64_t *)gb.file.p, (uint64_t *)gb.qhash.p, (uint32_t *)gb.row.p, (int32_t *)gb.endpos.p,
                        (uint32_t *)gb.flag_lq.p, (int32_t *)gb.tid.p, (uint32_t *)gb.count.p, gb.cap, (uint32_t *)gb.h2.p};
        launch_gc_candidates(c->stream, c->dann, c->dparams, d, c->dref, gc, acc.error);
    }
    // the batch's counts, for its retirement: page-locked mirrors + an event that tells when they are valid.  The copy of the
    // pair counts rides on a side stream: a copy between two kernels of the main stream costs ~40 us of queue hand-over there
    // (profiles/r4_step_timeline.txt), and nothing on the main stream reads what it brings
    if (c->have_bed) { FragBuf &fb = c->frag_pool[c->frags_in_flight.back()]; HIP_TRY(c, hipMemcpyAsync(fb.h_count, fb.count.p, 4, hipMemcpyDeviceToHost, c->stream)); }
    if (c->have_ref && !c->dparams.legacy) { GcBuf &gb = c->gc_pool[c->gcs_in_flight.back()]; HIP_TRY(c, hipMemcpyAsync(gb.h_count, gb.count.p, 4, hipMemcpyDeviceToHost, c->stream)); }
    HIP_TRY(c, hipEventRecord(pb->kernels, c->stream));
    HIP_TRY(c, hipStreamWaitEvent(c->stream2, pb->kernels, 0));
    HIP_TRY(c, hipMemcpyAsync(pb->h_counts, pb->counts.p, ((size_t)n_chunks + 1) * 4, hipMemcpyDeviceToHost, c->stream2));
    HIP_TRY(c, hipEventRecord(pb->done, c->stream2));
    HIP_TRY(c, hipGetLastError());
    c->timing.classify_launches += 1;
    c->timing.classify_records += u->n;
    c->timing.classify_bytes += 32ull * u->n + 4ull * u->n_cigar_total;
    return 0;
}

}  // namespace

// ------------------------------------------------------------------------------------ API

extern "C" {

int rsqc_create(const rsqc_params *params, rsqc_ctx **out) {
    if (!params || !out) return RSQC_ERR_ARG;
    if (params->abi_version != RSQC_ABI_VERSION) return RSQC_ERR_ARG;
    if (params->n_filter_tags < 0 || params->n_filter_tags > RSQC_MAX_FILTER_TAGS) return RSQC_ERR_ARG;
    if (params->bias_offset < 0 || params->bias_window < 1 || params->bias_window > RSQC_MAX_BIAS_WINDOW) return RSQC_ERR_ARG;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return RSQC_ERR_NO_DEVICE;
    if (params->device < 0 || params->device >= ndev) return RSQC_ERR_NO_DEVICE;
    rsqc_ctx *c = new rsqc_ctx();
    c->params = *params;
    c->device = params->device;
    // Stream priorities (RSQC_STREAM_PRIO = 1 or 2; default 0 = none): the context's stream -- the per-record kernels and the fragment-counting
    // chain, the end-of-file stage's critical path -- above the side streams of the coverage kernels.  Measured (calls r6i, r6l): the fragment
    // scatter gets faster (1.24 -> 1.00 ms), the coverage kernels and the fragment count behind them slower, the stage's end does not move:
    // it is the SUM of the kernels' work on the chip that sets it.  Left as a switch.
    if (const char *e = getenv("RSQC_STREAM_PRIO")) c->stream_prio = atoi(e);
    int prio_least = 0, prio_greatest = 0;
    if (hipSetDevice(c->device) == hipSuccess) (void)hipDeviceGetStreamPriorityRange(&prio_least, &prio_greatest);
    c->prio_side = c->stream_prio ? prio_least : 0;
    if (hipSetDevice(c->device) != hipSuccess ||
        (c->stream_prio == 1 ? hipStreamCreateWithPriority(&c->stream, hipStreamNonBlocking, prio_greatest) : hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking)) != hipSuccess) {
        delete c;
        return RSQC_ERR_HIP;
    }
    c->dparams.mapq_threshold = params->mapq_threshold;
    c->dparams.base_mismatch = params->base_mismatch;
    c->dparams.chimeric_distance = params->chimeric_distance;
    c->dparams.stranded = params->stranded;
    c->dparams.unpaired = params->unpaired;
    c->dparams.exclude_chimeric = params->exclude_chimeric;
    c->dparams.n_filter_tags = params->n_filter_tags;
    c->dparams.legacy = params->legacy ? 1 : 0;
    c->pair_arena.n_col = 1; c->pair_arena.width[0] = sizeof(PairRec);   // {gene, second name hash, name hash}
    c->frag_arena.n_col = 6; { const size_t w[6] = {8, 8, 4, 4, 4, 4}; for (int k = 0; k < 6; ++k) c->frag_arena.width[k] = w[k]; }   // ..., second name hash
    c->gc_arena.n_col = 7; { const size_t w[7] = {8, 8, 4, 4, 4, 4, 4}; for (int k = 0; k < 7; ++k) c->gc_arena.width[k] = w[k]; }   // ..., second name hash
    if (const char *e = getenv("RSQC_K1_GRID")) c->k1_grid = std::min(16384, std::max(1, atoi(e)));
    *out = c;
    return RSQC_OK;
}

void rsqc_destroy(rsqc_ctx *c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    (void)hipStreamSynchronize(c->stream);
    for (auto *u : c->resident) if (u) free_batch(u);
    for (auto *u : c->transient) free_batch(u);
    for (auto &b : c->upload_pool) b.release();
    for (auto &b : c->ann_bufs) b.release();
    for (auto &pb : c->pair_pool) { pb.rec.release(); pb.counts.release(); if (pb.h_counts) (void)hipHostFree(pb.h_counts); if (pb.done) (void)hipEventDestroy(pb.done); if (pb.kernels) (void)hipEventDestroy(pb.kernels); }
    for (auto &fb : c->frag_pool) { fb.file.release(); fb.qhash.release(); fb.name.release(); fb.endpos.release(); fb.fs.release(); fb.h2.release(); fb.count.release(); fb.r_file.release(); fb.r_qhash.release(); fb.r_name.release(); fb.r_endpos.release(); fb.r_fs.release(); fb.r_h2.release(); fb.r_counts.release(); if (fb.h_count) (void)hipHostFree(fb.h_count); }
    for (auto &gb : c->gc_pool) { gb.file.release(); gb.qhash.release(); gb.row.release(); gb.endpos.release(); gb.flag_lq.release(); gb.tid.release(); gb.h2.release(); gb.count.release(); if (gb.h_count) (void)hipHostFree(gb.h_count); }
    for (Arena *a : {&c->pair_arena, &c->frag_arena, &c->gc_arena}) for (int k = 0; k < a->n_col; ++k) a->col[k].release();
    c->d_arena_count.release(); c->d_rl_summary.release();
    {
        DecodeState &D = c->dec;
        for (DevBuf *b : {&D.comp, &D.blocks, &D.ubuf, &D.seg, &D.seg_rec0, &D.seg_ops0, &D.rec_off, &D.ops_at, &D.mark, &D.core, &D.aux, &D.qh2, &D.cigar,
                          &D.seg_tid, &D.seg_start, &D.wide_index, &D.wide_nm, &D.wide_lq, &D.wide_nc, &D.sum, &D.carry, &D.tailtmp, &D.scratch}) b->release();
        if (D.h_sum) (void)hipHostFree(D.h_sum);
        if (D.h_blocks) (void)hipHostFree(D.h_blocks);
        if (D.ev_copy) (void)hipEventDestroy(D.ev_copy);
        if (D.copy_stream) (void)hipStreamDestroy(D.copy_stream);
        for (auto &e : D.pe) if (e) (void)hipEventDestroy(e);
    }
    for (auto &b : c->parked) b.release();
    free_sort_scratch(c->gc_scratch); free_sort_scratch(c->frag_scratch);
    c->d_ref_bits.release(); c->d_ref_off.release(); c->d_ref_len.release(); c->d_gc_bins.release(); c->d_exon_gc.release();
    DevBuf *all[] = {&c->d_arena, &c->d_cov, &c->d_ovf_index, &c->d_tiles, &c->d_defer, &c->d_ei_rank, &c->d_table, &c->d_tab_off, &c->d_tab_cap};
    if (c->h_arena) (void)hipHostFree(c->h_arena);
    if (c->h_rl_raw) (void)hipHostFree(c->h_rl_raw);
    if (c->stream2) (void)hipStreamDestroy(c->stream2);
    if (c->stream3) (void)hipStreamDestroy(c->stream3);
    if (c->stream4) (void)hipStreamDestroy(c->stream4);
    if (c->ev_join3) (void)hipEventDestroy(c->ev_join3);
    if (c->ev_join4) (void)hipEventDestroy(c->ev_join4);
    if (c->ev_fork) (void)hipEventDestroy(c->ev_fork);
    if (c->ev_join) (void)hipEventDestroy(c->ev_join);
    for (auto *b : all) b->release();
    for (auto &pr : c->k1_events) { (void)hipEventDestroy(pr.first); (void)hipEventDestroy(pr.second); }
    for (auto &pr : c->long_events) (void)hipEventDestroy(pr.second);      // (.first is the K1 pair's second event)
    for (auto e : c->event_pool) (void)hipEventDestroy(e);
    (void)hipStreamDestroy(c->stream);
    delete c;
}

int rsqc_set_annotation(rsqc_ctx *c, const rsqc_annotation *a, const uint8_t *owned_contig) {
    if (!c || !a) return RSQC_ERR_ARG;
    if (c->have_ann) return fail(c, RSQC_ERR_ARG, "annotation already set");
    HIP_TRY(c, hipSetDevice(c->device));
    const int nc = a->n_contigs, G = a->n_genes, L = a->n_genes_listed, E = a->n_exons;
    c->n_ref = a->n_ref; c->n_contigs = nc; c->n_genes = G; c->n_listed = L; c->n_exons = E;
    HostIndex hx;
    {
        std::string err;
        int brc = hx.build(a, owned_contig, err);
        if (brc) return fail(c, brc, err);
    }
    const uint64_t run = hx.cov_entries;
    c->cov_entries = run;
    c->exon_row_id.assign(a->exon_row_id, a->exon_row_id + E);

    // ---- upload ---------------------------------------------------------------------------------
    DevAnnotation &d = c->dann;
    d.n_ref = a->n_ref; d.n_contigs = nc; d.n_genes = G; d.n_listed = L; d.n_exons = E;
    d.bin_shift = HostIndex::kBinShift;
    int rc;
#define UPV(dst, vec) if ((rc = upload(c, c->ann_bufs, (vec).data(), (vec).size(), &(dst)))) return rc
#define UPA(dst, ptr, n) if ((rc = upload(c, c->ann_bufs, (ptr), (size_t)(n), &(dst)))) return rc
    UPV(d.ex, hx.ex_rows); UPV(d.gb, hx.gb); UPV(d.contig, hx.contig);
    UPV(d.ex_binhi, hx.ex_binhi); UPV(d.gb_bin, hx.gb_bin); UPV(d.ex_cov, hx.ex_cov); UPV(d.ex_pmax, hx.ex_pmax);
    UPV(d.ex_id, c->exon_row_id);
    UPV(d.ei, hx.ei); UPV(d.ei_coarse, hx.ei_coarse);
    if ((rc = dev_alloc(c, c->d_ei_rank, ((size_t)hx.rank_words + 1) * sizeof(EiRank), true)))
        return fail(c, rc, "no device memory for the interval index's rank table: " + std::to_string((((size_t)hx.rank_words + 1) * sizeof(EiRank)) >> 20) +
                           " MiB (16 bytes per 64 positions up to every contig's last feature; 775 MB for the human contig lengths)");
    d.ei_rank = (const EiRank *)c->d_ei_rank.p;
    for (int k = 0; k < nc; ++k)                  // (stream order: after the upload of the entries)
        launch_ei_rank(c->stream, d.ei, hx.ei_range[(size_t)k], hx.ei_range[(size_t)k + 1],
                       (EiRank *)c->d_ei_rank.p + hx.contig[(size_t)k].rk_base, hx.contig[(size_t)k].rk_words);
    d.legacy = nullptr;
    if (c->params.legacy) {                       // tables of the --legacy rules (rsqc_read.h: LegacyTables)
        LegacyTables lt{};
        UPV(lt.gr, hx.gr_rows); UPV(lt.gr_pmax, hx.g_pmax); UPV(lt.gr_range, hx.g_range); UPV(lt.ex_ord, hx.ex_ord); UPV(lt.gr_binhi, hx.gr_binhi);
        std::vector<LegacyTables> one(1, lt);
        UPV(d.legacy, one);
    }
    auto &gene_cov_off = hx.gene_cov_off; auto &gene_coding = hx.gene_coding;
    auto &gene_flags = hx.gene_flags; auto &gene_owned = hx.gene_owned;
    // empty BED until rsqc_set_bed
    std::vector<uint32_t> zero_range((size_t)nc + 1, 0);
    UPV(d.bed_range, zero_range);
    d.bed_start = d.bed_end = d.bed_pmax = nullptr; d.bed_binhi = d.bed_bin_base = nullptr; d.have_bed = 0;
    UPA(c->d_ge_off, a->gene_exon_off, (size_t)G + 1);
    UPA(c->d_ge_row, a->gene_exon_row, E);
    UPV(c->d_gene_cov_off, gene_cov_off);
    UPV(c->d_gene_coding, gene_coding);
    UPV(c->d_gene_flags, gene_flags);
    UPV(c->d_gene_owned, gene_owned);
    std::vector<uint32_t> gene_order((size_t)std::max(L, 1), 0);
    for (int g = 0; g < L; ++g) gene_order[(size_t)g] = (uint32_t)g;
    std::stable_sort(gene_order.begin(), gene_order.begin() + L, [&](uint32_t x, uint32_t y) { return gene_coding[x] > gene_coding[y]; });
    UPV(c->d_gene_order, gene_order);
    // workgroup size classes of the end-of-file coverage stage (rsqc_kernels.hip, K3)
    c->k3_large = c->k3_medium = c->k3_xlarge = c->k3_le6144 = c->k3_le3072 = c->k3_le2048 = c->k3_le1024 = 0;
    for (int k = 0; k < L; ++k) {
        const uint32_t len = gene_coding[gene_order[(size_t)k]];
        if (len > (uint32_t)RSQC_K3_MEDIUM_MAX) c->k3_large++; else if (len > (uint32_t)RSQC_K3_SMALL_MAX) c->k3_medium++;
        if (len > (uint32_t)RSQC_K3_LARGE2_LDS16) c->k3_xlarge++;
        if (len <= 6144u) c->k3_le6144++;
        if (len <= 3072u) c->k3_le3072++;
        if (len <= 2048u) c->k3_le2048++;
        if (len <= 1024u) c->k3_le1024++;
    }
#undef UPV
#undef UPA
    // ---- accumulators -----------------------------------------------------------------------------
    const size_t n_u64 = (size_t)G * 3 + RSQC_N_COUNTERS;
    const size_t Lz = (size_t)std::max(L, 1), Ez = (size_t)std::max(E, 1);
    auto pad8 = [](size_t x) { return (x + 7) & ~(size_t)7; };
    size_t at = 0;
    // three runs of one element type each, so that a sharded run sum-reduces everything with three collectives
    // (rsqc_device_vectors): u64 counts | f64 sums and owner-only statistics | u8 validity flags
    c->off_u64 = at; at += n_u64 * 8;
    c->off_bias3 = at; at += Lz * 8;
    c->off_bias5 = at; at += Lz * 8;
    c->off_exon = at; at += Ez * 8;
    c->off_gmean = at; at += Lz * 8;
    c->off_gstd = at; at += Lz * 8;
    c->off_gcv = at; at += Lz * 8;
    c->off_ecv = at; at += Ez * 8;
    c->off_gvalid = at; at += pad8(Lz);
    c->off_ecvv = at; at += pad8(Ez);
    c->off_ehit = at; at += pad8(Ez);
    at = (at + 15) & ~(size_t)15;                 // rl_stats (off_misc + 32) starts a 16-byte vector: the reset kernel arms it
    c->off_misc = at; at += 64;
    c->arena_bytes = at;
    if ((rc = dev_alloc(c, c->d_arena, at, false))) return rc;
    HIP_TRY(c, hipHostMalloc((void **)&c->h_arena, at, hipHostMallocDefault));
    if ((rc = dev_alloc(c, c->d_cov, (size_t)(run + 64) * 4, false))) return rc;
    const uint32_t ovf_cap = 1u << 20;
    if ((rc = dev_alloc(c, c->d_ovf_index, (size_t)ovf_cap * 8, false))) return rc;
    HIP_TRY(c, hipStreamCreateWithPriority(&c->stream2, hipStreamNonBlocking, c->prio_side));
    HIP_TRY(c, hipStreamCreateWithPriority(&c->stream3, hipStreamNonBlocking, c->prio_side));
    HIP_TRY(c, hipStreamCreateWithPriority(&c->stream4, hipStreamNonBlocking, c->prio_side));
    HIP_TRY(c, hipEventCreateWithFlags(&c->ev_join3, hipEventDisableTiming));
    HIP_TRY(c, hipEventCreateWithFlags(&c->ev_join4, hipEventDisableTiming));
    HIP_TRY(c, hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming));
    HIP_TRY(c, hipEventCreateWithFlags(&c->ev_join, hipEventDisableTiming));
    char *A = (char *)c->d_arena.p;
    DevAccum &acc = c->acc;
    acc.gene_reads = (unsigned long long *)(A + c->off_u64);
    acc.gene_unique = acc.gene_reads + G;
    acc.gene_frag = acc.gene_unique + G;
    acc.counters = acc.gene_frag + G;
    acc.exon_acc = (double *)(A + c->off_exon);
    acc.cov_diff = (uint32_t *)c->d_cov.p;
    acc.ovf_count = (uint32_t *)(A + c->off_misc);
    acc.read_length = (int32_t *)(A + c->off_misc + 8);
    acc.error = (int *)(A + c->off_misc + 16);
    acc.rl_stats = (uint32_t *)(A + c->off_misc + 32);
    acc.defer_total = (uint32_t *)(A + c->off_misc + 48);    // (zeroed with the arena, re-armed by the last kernel of every batch)
    acc.ovf_index = (uint64_t *)c->d_ovf_index.p; acc.ovf_cap = ovf_cap;
    c->have_ann = true;
    if ((rc = zero_accumulators(c))) return rc;
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    c->last_error = hx.warning;                  // (RSQC_OK with a warning: an exon outside its gene's row, see rsqc_index.h)
    c->n_exons_outside_gene = hx.n_exons_outside_gene;
    return RSQC_OK;
}

int rsqc_set_reference(rsqc_ctx *c, const rsqc_reference *ref) {
    if (!c || !ref || !c->have_ann) return RSQC_ERR_ARG;
    if (c->have_ref) return fail(c, RSQC_ERR_ARG, "reference already set");
    HIP_TRY(c, hipSetDevice(c->device));
    const int nc = c->n_contigs;
    std::vector<unsigned long long> off((size_t)nc, ~0ull), len((size_t)nc, 0ull);
    unsigned long long words = 0, longest = 0;
    for (int i = 0; i < ref->n; ++i) {
        const int k = ref->contig[i];
        if (k < 0 || k >= nc || off[(size_t)k] != ~0ull) return fail(c, RSQC_ERR_ARG, "reference contig out of range or repeated");
        if (ref->length[i] && !ref->sequence[i]) return fail(c, RSQC_ERR_ARG, "reference contig without bases");
        off[(size_t)k] = words; len[(size_t)k] = ref->length[i];
        words += (ref->length[i] + 63) / 64;
        longest = std::max<unsigned long long>(longest, ref->length[i]);
    }
    int rc;
    if ((rc = dev_alloc(c, c->d_ref_bits, (size_t)(words + 2) * 8, false)) || (rc = dev_alloc(c, c->d_ref_off, (size_t)std::max(nc, 1) * 8, false)) ||
        (rc = dev_alloc(c, c->d_ref_len, (size_t)std::max(nc, 1) * 8, false)) || (rc = dev_alloc(c, c->d_gc_bins, (RSQC_GC_BINS + 1) * 8, false)) ||
        (rc = dev_alloc(c, c->d_exon_gc, (size_t)std::max(c->n_exons, 1) * 8, false))) return rc;
    HIP_TRY(c, hipMemcpyAsync(c->d_ref_off.p, off.data(), (size_t)nc * 8, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipMemcpyAsync(c->d_ref_len.p, len.data(), (size_t)nc * 8, hipMemcpyHostToDevice, c->stream));
    // the bases pass through one staging buffer sized for the longest contig and are packed to one bit each
    DevBuf stage;
    if ((rc = dev_alloc(c, stage, (size_t)longest + 64, false))) return rc;
    for (int i = 0; i < ref->n; ++i) {
        if (!ref->length[i]) continue;
        HIP_TRY(c, hipMemcpyAsync(stage.p, ref->sequence[i], (size_t)ref->length[i], hipMemcpyHostToDevice, c->stream));
        launch_gc_pack(c->stream, (const uint8_t *)stage.p, ref->length[i], (unsigned long long *)c->d_ref_bits.p + off[(size_t)ref->contig[i]]);
        HIP_TRY(c, hipStreamSynchronize(c->stream));          // the caller's string and the staging buffer are reused
    }
    stage.release();
    c->dref = DevReference{(const unsigned long long *)c->d_ref_bits.p, (const unsigned long long *)c->d_ref_off.p,
                           (const unsigned long long *)c->d_ref_len.p};
    launch_exon_gc(c->stream, c->dann, c->dref, (double *)c->d_exon_gc.p);
    c->h_exon_gc.assign((size_t)std::max(c->n_exons, 1), -1.0);
    HIP_TRY(c, hipMemcpyAsync(c->h_exon_gc.data(), c->d_exon_gc.p, (size_t)c->n_exons * 8, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipMemsetAsync(c->d_gc_bins.p, 0, (RSQC_GC_BINS + 1) * 8, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    HIP_TRY(c, hipGetLastError());
    c->h_gc.assign(RSQC_GC_BINS + 1, 0);
    c->have_ref = true;
    return RSQC_OK;
}

int rsqc_set_bed(rsqc_ctx *c, const rsqc_bed *bed) {
    if (!c || !bed || !c->have_ann) return RSQC_ERR_ARG;
    HIP_TRY(c, hipSetDevice(c->device));
    const int nc = c->n_contigs, n = bed->n_intervals;
    std::vector<uint32_t> range((size_t)nc + 1, 0);
    std::vector<int32_t> pmax((size_t)n);
    for (int i = 0; i < n; ++i) {
        if (bed->contig[i] < 0 || bed->contig[i] >= nc) return fail(c, RSQC_ERR_ARG, "BED contig out of range");
        if (i && (bed->contig[i] < bed->contig[i - 1] ||
                  (bed->contig[i] == bed->contig[i - 1] && bed->start[i] < bed->start[i - 1])))
            return fail(c, RSQC_ERR_ARG, "BED intervals must be grouped by contig id and ascending by start");
        range[(size_t)bed->contig[i] + 1]++;
    }
    for (int k = 0; k < nc; ++k) range[(size_t)k + 1] += range[(size_t)k];
    for (int k = 0; k < nc; ++k) {
        int32_t m = INT32_MIN;
        for (uint32_t i = range[(size_t)k]; i < range[(size_t)k + 1]; ++i) { m = std::max(m, bed->end[i]); pmax[i] = m; }
    }
    int rc;
    if ((rc = upload(c, c->ann_bufs, bed->start, (size_t)n, &c->dann.bed_start))) return rc;
    if ((rc = upload(c, c->ann_bufs, bed->end, (size_t)n, &c->dann.bed_end))) return rc;
    if ((rc = upload(c, c->ann_bufs, pmax.data(), pmax.size(), &c->dann.bed_pmax))) return rc;
    if ((rc = upload(c, c->ann_bufs, range.data(), range.size(), &c->dann.bed_range))) return rc;
    {   // bin table (DevAnnotation::bed_binhi): a block's upper bound is one load and a short step down
        std::vector<uint32_t> bin_base((size_t)nc + 1, 0), binhi;
        for (int k = 0; k < nc; ++k) {
            const uint32_t lo = range[(size_t)k], hi = range[(size_t)k + 1];
            const int32_t top = hi > lo ? std::max(bed->start[hi - 1], 0) : 0;
            const uint32_t nb = hi > lo ? ((uint32_t)top >> RSQC_BED_BIN_SHIFT) + 1u : 1u;
            bin_base[(size_t)k + 1] = bin_base[(size_t)k] + nb;
            uint32_t row = lo;
            for (uint32_t b = 0; b < nb; ++b) {
                const int64_t limit = ((int64_t)b + 1) << RSQC_BED_BIN_SHIFT;
                while (row < hi && (int64_t)bed->start[row] < limit) ++row;
                binhi.push_back(row);
            }
        }
        if ((rc = upload(c, c->ann_bufs, binhi.data(), binhi.size(), &c->dann.bed_binhi))) return rc;
        if ((rc = upload(c, c->ann_bufs, bin_base.data(), bin_base.size(), &c->dann.bed_bin_base))) return rc;
    }
    c->dann.have_bed = 1;
    c->have_bed = true;
    c->frag_remaining = c->params.fragment_samples;
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return RSQC_OK;
}

int rsqc_submit(rsqc_ctx *c, const rsqc_batch *b) {
    if (!c || !b) return RSQC_ERR_ARG;
    if (c->sticky) return c->sticky;
    HIP_TRY(c, hipSetDevice(c->device));
    UploadedBatch *u = new UploadedBatch();
    hipEvent_t e0 = get_event(c), e1 = get_event(c);
    (void)hipEventRecord(e0, c->stream);
    int rc = upload_batch(c, b, u, /*pooled=*/true);
    (void)hipEventRecord(e1, c->stream);
    c->h2d_events.emplace_back(e0, e1);           // resolved at rsqc_wait / rsqc_finalize
    if (rc) { free_batch(u); return rc; }
    c->transient.push_back(u);
    // asynchronous from here on: with pinned source arrays (rsqc_host_alloc) the copies are DMA transfers the
    // call does not wait for; the caller keeps the arrays alive and unmodified until rsqc_wait
    return run_batch(c, u);
}

int rsqc_wait(rsqc_ctx *c) {
    if (!c) return RSQC_ERR_ARG;
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    resolve_events(c);
    for (auto *u : c->transient) retire_batch(c, u);
    c->transient.clear();
    free_parked(c);
    if (c->have_ann) {
        int rc = retire_completed(c, true);      // everything submitted so far has completed
        if (rc) return rc;
        return check_device_error(c);
    }
    return RSQC_OK;
}

int rsqc_upload(rsqc_ctx *c, const rsqc_batch *b, int *handle_out) {
    if (!c || !b || !handle_out) return RSQC_ERR_ARG;
    HIP_TRY(c, hipSetDevice(c->device));
    UploadedBatch *u = new UploadedBatch();
    int rc = upload_batch(c, b, u);
    if (rc) { free_batch(u); return rc; }
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    c->resident.push_back(u);
    *handle_out = (int)c->resident.size() - 1;
    return RSQC_OK;
}

int rsqc_submit_resident(rsqc_ctx *c, int handle) {
    if (!c || handle < 0 || handle >= (int)c->resident.size() || !c->resident[(size_t)handle]) return RSQC_ERR_ARG;
    if (c->sticky) return c->sticky;
    HIP_TRY(c, hipSetDevice(c->device));
    RSQC_TRACE("submit_resident: enter");
    const int rc = run_batch(c, c->resident[(size_t)handle]);
    RSQC_TRACE("submit_resident: enqueued");
    return rc;
}

int rsqc_release(rsqc_ctx *c, int handle) {
    if (!c || handle < 0 || handle >= (int)c->resident.size() || !c->resident[(size_t)handle]) return RSQC_ERR_ARG;
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    free_batch(c->resident[(size_t)handle]);
    c->resident[(size_t)handle] = nullptr;
    return RSQC_OK;
}

int rsqc_reset(rsqc_ctx *c) {
    if (!c || !c->have_ann) return RSQC_ERR_ARG;
    HIP_TRY(c, hipSetDevice(c->device));
    RSQC_TRACE("reset: enter");
    const int rc = zero_accumulators(c);
    RSQC_TRACE("reset: enqueued");
    return rc;
}

// one D2H of the whole arena into the pinned mirror, then unpack into the results struct
// The device holds every result vector in its final order (genes by listed id, exons by exon id), so the read-back
// is ONE D2H of the arena into the page-locked mirror; the results struct points straight into the mirror.
static int read_back(rsqc_ctx *c) {
    const int G = c->n_genes, L = c->n_listed, E = c->n_exons;
    char *A = (char *)c->d_arena.p;
    if (c->early_copied) {                           // (run_finalize_kernels sent the rest ahead, beside the fragment kernels)
        const size_t frag_lo = c->off_u64 + (size_t)2 * (size_t)G * 8;
        HIP_TRY(c, hipMemcpyAsync(c->h_arena + frag_lo, (char *)c->d_arena.p + frag_lo, (size_t)G * 8, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(c, hipMemcpyAsync(c->h_arena + c->off_misc, (char *)c->d_arena.p + c->off_misc, c->arena_bytes - c->off_misc, hipMemcpyDeviceToHost, c->stream));
        c->early_copied = false;
        RSQC_TRACE("read_back: copies enqueued");
    } else {
        launch_pack_results(c->stream, (const double *)(A + c->off_exon), (uint8_t *)(A + c->off_ehit), (uint32_t)E);
        HIP_TRY(c, hipMemcpyAsync(c->h_arena, c->d_arena.p, c->arena_bytes, hipMemcpyDeviceToHost, c->stream));
    }
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    RSQC_TRACE("read_back: stream idle");
    char *H = c->h_arena;
    uint64_t *u = (uint64_t *)(H + c->off_u64);
    rsqc_results &R = c->results;
    for (int k = 0; k < RSQC_N_COUNTERS; ++k) R.counters[k] = u[3 * (size_t)G + (size_t)k];
    R.n_genes_listed = L; R.n_exons = E;
    R.gene_reads = u; R.gene_unique = u + G; R.gene_fragments = u + 2 * (size_t)G;
    R.exon_reads = (double *)(H + c->off_exon); R.exon_hit = (uint8_t *)(H + c->off_ehit);
    R.read_length = c->have_composed_rl ? c->composed_rl : *(const int32_t *)(H + c->off_misc + 8);
    R.gene_cov_mean = (double *)(H + c->off_gmean); R.gene_cov_std = (double *)(H + c->off_gstd); R.gene_cov_cv = (double *)(H + c->off_gcv);
    R.gene_cov_valid = (uint8_t *)(H + c->off_gvalid); R.exon_cv = (double *)(H + c->off_ecv); R.exon_cv_valid = (uint8_t *)(H + c->off_ecvv);
    R.bias_three = (uint64_t *)(H + c->off_bias3); R.bias_five = (uint64_t *)(H + c->off_bias5);
    R.n_fragment_sizes = (uint32_t)c->h_fsize.size();
    R.fragment_size = c->h_fsize.data(); R.fragment_count = c->h_fcount.data();
    R.fragment_samples_remaining = c->frag_remaining;
    R.have_reference = c->have_ref ? 1 : 0;
    R.gc_bins = c->have_ref ? c->h_gc.data() : nullptr;
    R.gc_out_of_range = c->have_ref ? c->h_gc[RSQC_GC_BINS] : 0;
    R.exon_gc = c->have_ref ? c->h_exon_gc.data() : nullptr;
    R.exons_outside_gene_row = c->n_exons_outside_gene;
    c->timing.slow_records = *(const uint32_t *)(H + c->off_misc + 4);
    const int err = *(const int *)(H + c->off_misc + 16);
    if (err) {
        c->sticky = err;
        return fail(c, err, err == RSQC_ERR_BAD_CIGAR ? "Unrecognized Cigar Op" :
                            err == RSQC_ERR_CAPACITY ? "a device-side capacity was exceeded" :
                            err == RSQC_ERR_EMPTY_MEDIAN ? "Cannot compute median of an empty list" : "device error");
    }
    return 0;
}

// end-of-file kernels (K3 beside K4, K5 for BED runs); results stay on the device
static int run_finalize_kernels(rsqc_ctx *c, bool early_readback = false) {
    int rc;
    c->early_copied = false;
    const int G = c->n_genes, L = c->n_listed;
    char *A = (char *)c->d_arena.p;
    hipEvent_t e0 = get_event(c), e1 = get_event(c);
        HIP_TRY(c, hipEventRecord(e0, c->stream));
        // the side streams start from here (recorded BEFORE the K4 kernels are enqueued on the main stream)
        HIP_TRY(c, hipEventRecord(c->ev_fork, c->stream));
        // ---- K3 on the side streams: coverage scan + per-gene statistics + bias.  Enqueued BEFORE the fragment stage: with several batches
        //      in flight that stage begins with a host-side wait (retire_completed below), and the coverage kernels -- which depend on
        //      the fork event only -- would otherwise not even be queued while the host sleeps -----------------------
        HIP_TRY(c, hipStreamWaitEvent(c->stream2, c->ev_fork, 0));
        HIP_TRY(c, hipStreamWaitEvent(c->stream3, c->ev_fork, 0));
        HIP_TRY(c, hipStreamWaitEvent(c->stream4, c->ev_fork, 0));
        GeneCovArgs Ga{};
        Ga.ge_off = c->d_ge_off; Ga.ge_row = c->d_ge_row;
        Ga.ex = c->dann.ex; Ga.ex_cov = c->dann.ex_cov; Ga.ex_id = c->dann.ex_id;
        Ga.gene_cov_off = c->d_gene_cov_off; Ga.gene_coding = c->d_gene_coding;
        Ga.gene_flags = c->d_gene_flags; Ga.gene_owned = c->d_gene_owned;
        Ga.gene_order = c->d_gene_order;
        Ga.gene_reads = c->acc.gene_reads; Ga.cov = c->acc.cov_diff; Ga.n_listed = L;
        Ga.mask = c->params.coverage_mask; Ga.bias_offset = c->params.bias_offset; Ga.bias_window = c->params.bias_window;
        Ga.bias_gene_length = c->params.bias_gene_length;
        Ga.g_mean = (double *)(A + c->off_gmean); Ga.g_std = (double *)(A + c->off_gstd); Ga.g_cv = (double *)(A + c->off_gcv);
        Ga.g_valid = (uint8_t *)(A + c->off_gvalid); Ga.e_cv = (double *)(A + c->off_ecv); Ga.e_cv_valid = (uint8_t *)(A + c->off_ecvv);
        Ga.bias3 = (unsigned long long *)(A + c->off_bias3); Ga.bias5 = (unsigned long long *)(A + c->off_bias5);
        Ga.error = c->acc.error;
        {
            uint32_t nl = c->k3_large, nm = c->k3_medium, nx = c->k3_xlarge;
            if (const char *e = RSQC_DIAG("RSQC_K3_FORCE")) {        // diagnostic build only: 1 = all 1024-thread, 2 = all 256, 3 = all one-wave
                const int f = atoi(e);
                if (f == 1) { nl = (uint32_t)L; nm = 0; } else if (f == 2) { nl = 0; nm = (uint32_t)L; nx = 0; } else if (f == 3) { nl = 0; nm = 0; nx = 0; } else if (f == 4) { nl = (uint32_t)L; nm = 0; nx = (uint32_t)L; }
            }
            if (!RSQC_DIAG("RSQC_DIAG_SKIP_K3")) launch_gene_coverage(c->stream2, c->stream3, c->stream4, Ga, nl, nm, nx, c->k3_le6144, c->k3_le3072, c->k3_le2048, c->k3_le1024);   // (diagnostic build only: results incomplete)
        }
        // ---- K4 on the main stream: per-gene distinct QNAMEs -------------------------------------------------
        // Several batches still in flight (a host that enqueued its batches faster than the device ran them -- bench.py's resident
        // batches, one per contig of a sharded run): they are retired into the arena first, so that the fragment stage runs ONE pass
        // over one dense list instead of one launch per batch over worst-case chunk tables (26 batches: 2.0 ms of frag_local
        // instead of 1.1).  Costs one wait for the batches' kernels here; a single batch in flight is used in place, uncopied.
        if (c->pairs_in_flight.size() > 1) { if ((rc = retire_completed(c, true))) return rc; }
        uint64_t pair_bound = c->pair_arena.used;
        for (size_t idx : c->pairs_in_flight) pair_bound += c->pair_pool[idx].pairs_bound;
        {
            // streaming form: survivors appended to per-partition key lists, then counted per partition in LDS.
            // bounds from the host's pair bound: partitions <= pairs / PART_READS + G, keys <= 2 x pairs + SUB_CAP x parts
            const uint64_t Gz = (uint64_t)std::max(G, 1);
            const uint64_t parts_bound = pair_bound / RSQC_K4_PART_READS + Gz + 1;
            const uint64_t keys_bound = 2 * pair_bound + RSQC_K4_SUB_CAP * std::min<uint64_t>(parts_bound, pair_bound / RSQC_K4_PART_READS + 1) + 16 * Gz + 16;   // (+ the round-up of every gene's space to 16 entries)
            if (parts_bound > 0xFFFFFFF0ull) return fail(c, RSQC_ERR_CAPACITY, "too many fragment partitions");
            const uint64_t lay_blocks = (Gz + 1023) / 1024;
            if ((rc = dev_alloc(c, c->d_tab_off, (Gz + 2) * 28 + 64 + lay_blocks * 12 + 64, false))) return rc;  // per-gene rows | part_first | layout totals
            if ((rc = dev_alloc(c, c->d_tab_cap, parts_bound * 24 + 128, false))) return rc;                  // per-partition rows | cursor | list of the fuller ones + its counter
            if (c->d_table.bytes < (size_t)keys_bound * sizeof(FragKey)) { if ((rc = dev_alloc(c, c->d_table, (size_t)keys_bound * sizeof(FragKey) + (1u << 20), false))) return rc; }
            FragPlan P;
            P.ginfo = (uint4 *)c->d_tab_off.p;
            P.part_first = (uint32_t *)(P.ginfo + Gz + 1);
            P.blk_space = (unsigned long long *)(((uintptr_t)(P.part_first + Gz + 2) + 15) & ~(uintptr_t)15);
            P.blk_parts = (uint32_t *)(P.blk_space + lay_blocks);
            P.part_info = (uint4 *)c->d_tab_cap.p; P.cursor = (uint32_t *)(P.part_info + parts_bound);
            P.full_list = P.cursor + parts_bound; P.full_n = P.full_list + parts_bound;
            P.list = (FragKey *)c->d_table.p;
            launch_frag_layout(c->stream, c->acc.gene_reads, (uint32_t)G, P, c->acc.error);
            if (c->pair_arena.used && !RSQC_DIAG("RSQC_DIAG_SKIP_K4")) {     // the retired batches: one dense list, cut into pieces
                if ((rc = dev_alloc(c, c->d_arena_count, 16, false))) return rc;
                const uint32_t used32 = (uint32_t)c->pair_arena.used;
                HIP_TRY(c, hipMemcpyAsync(c->d_arena_count.p, &used32, 4, hipMemcpyHostToDevice, c->stream));
                DevAccum acc = c->acc;
                acc.pairs = (PairRec *)c->pair_arena.col[0].p;
                acc.pair_chunk_cap = 0; acc.pair_chunk_count = (uint32_t *)c->d_arena_count.p;
                acc.pair_slow_base = 0; acc.pair_slow_cap = used32;
                launch_frag_local(c->stream, acc, 0, P, (uint32_t)std::min<uint64_t>(4096, c->pair_arena.used / 1024 + 1));
            }
            for (size_t idx : c->pairs_in_flight) {
                if (RSQC_DIAG("RSQC_DIAG_SKIP_K4")) break;                // (diagnostic build only: results incomplete)
                PairBuf &pb = c->pair_pool[idx];
                DevAccum acc = c->acc;
                acc.pairs = (PairRec *)pb.rec.p;
                acc.pair_chunk_cap = pb.chunk_cap; acc.pair_chunk_count = (uint32_t *)pb.counts.p;
                acc.pair_slow_base = pb.slow_base; acc.pair_slow_cap = pb.slow_cap;
                acc.pair_slow_count = (uint32_t *)pb.counts.p + pb.n_chunks;
                // (--legacy: all the batch's pairs sit in its dense region -- shared by as many workgroups as a list of that size gets below,
                //  not by the 32 that serve the default rules' few thousand slow-path pairs: 9.3 -> ~1 ms per 100 M records)
                launch_frag_local(c->stream, acc, pb.n_chunks, P, c->dparams.legacy ? (uint32_t)std::min<uint64_t>(4096, pb.slow_cap / 8192 + 32) : 0u);
            }
            if (!RSQC_DIAG("RSQC_DIAG_SKIP_K4")) launch_frag_count(c->stream, (uint32_t)G, P, (uint32_t)parts_bound, c->acc.gene_frag, c->acc.error);
        }
        RSQC_TRACE("finalize: K3 + K4 enqueued");
        {   // per-batch Read-Length transfer functions (rsqc_shard_info): final since the last batch's read_length_kernel; the copy
            // goes out on a side stream behind its coverage kernel (behind frag_count on the main stream it cost a queue hand-over:
            // ~40 us) and is covered by that stream's join below; the destination is page-locked, the call returns at once
            const size_t nb = c->batch_file_index.size();
            if (nb * RSQC_RL_SUMMARY_WORDS > c->h_rl_raw_cap) {
                if (c->h_rl_raw) { (void)hipHostFree(c->h_rl_raw); c->h_rl_raw = nullptr; c->h_rl_raw_cap = 0; }
                const size_t want = std::max<size_t>(2 * nb, 64) * RSQC_RL_SUMMARY_WORDS;
                HIP_TRY(c, hipHostMalloc((void **)&c->h_rl_raw, want * 4, hipHostMallocDefault));
                c->h_rl_raw_cap = want;
            }
            RSQC_TRACE("finalize: K3 enqueued");
            if (nb) HIP_TRY(c, hipMemcpyAsync(c->h_rl_raw, c->d_rl_summary.p, nb * RSQC_RL_SUMMARY_WORDS * 4, hipMemcpyDeviceToHost, c->stream2));
            RSQC_TRACE("finalize: rl copy call returned");
        }
        HIP_TRY(c, hipEventRecord(c->ev_join3, c->stream3));
        HIP_TRY(c, hipEventRecord(c->ev_join4, c->stream4));
        if (early_readback) {
            // rsqc_finalize of ONE context: everything the fragment kernels (K4, still running on the main stream; K5 behind it) do not write --
            // all of the arena but geneFragmentCounts and the status words -- crosses PCIe NOW, behind the coverage kernels on their
            // stream, instead of behind K4 (the 9 MB copy was 0.17 ms at the end of every pass); read_back then fetches the rest
            const size_t frag_lo = c->off_u64 + (size_t)2 * (size_t)c->n_genes * 8, frag_hi = frag_lo + (size_t)c->n_genes * 8;
            HIP_TRY(c, hipStreamWaitEvent(c->stream2, c->ev_join3, 0));
            HIP_TRY(c, hipStreamWaitEvent(c->stream2, c->ev_join4, 0));
            launch_pack_results(c->stream2, (const double *)(A + c->off_exon), (uint8_t *)(A + c->off_ehit), (uint32_t)c->n_exons);
            if (frag_lo) HIP_TRY(c, hipMemcpyAsync(c->h_arena, c->d_arena.p, frag_lo, hipMemcpyDeviceToHost, c->stream2));
            HIP_TRY(c, hipMemcpyAsync(c->h_arena + frag_hi, (char *)c->d_arena.p + frag_hi, c->off_misc - frag_hi, hipMemcpyDeviceToHost, c->stream2));
            c->early_copied = true;
        }
        HIP_TRY(c, hipEventRecord(c->ev_join, c->stream2));
        HIP_TRY(c, hipStreamWaitEvent(c->stream, c->ev_join, 0));
        HIP_TRY(c, hipStreamWaitEvent(c->stream, c->ev_join3, 0));
        HIP_TRY(c, hipStreamWaitEvent(c->stream, c->ev_join4, 0));
        HIP_TRY(c, hipGetLastError());
        HIP_TRY(c, hipEventRecord(e1, c->stream));
        // ---- K5: fragment-size sampler (BED runs) ---------------------------------------------------
        if (c->have_bed) {
            // the batches still in flight join the retired ones in the candidate arena (file order), then one pairing pass.
            // K5 depends on the per-record kernels only (their candidates, and the counts copied behind them), not on K3 / K4: it runs BESIDE the
            // end-of-file kernels, on the side stream whose coverage class ends first (the 64 KB class: ~0.5 ms), from the moment the fork event
            // has fired -- rounds 4-6 waited for the whole stage here and ran it behind (1.3 ms of every --bed pass).  Only when the arena must
            // grow while it holds retired batches (its old columns are copied on the main stream) does the old order apply.
            uint64_t incoming = 0;
            bool beside = true;
#ifdef RSQC_K5_SERIAL
            beside = false;                                   // (A/B build: the order of rounds 4-6)
#endif
            HIP_TRY(c, hipEventSynchronize(c->ev_fork));      // every batch's kernels are done, the mirrors of the candidate counts are valid
            for (size_t k = 0; k < c->frags_in_flight.size(); ++k) incoming += *c->frag_pool[c->frags_in_flight[k]].h_count;
            if (c->frag_arena.used && c->frag_arena.used + incoming > c->frag_arena.cap) beside = false;
            hipStream_t ks = beside ? c->stream4 : c->stream;
            if (!beside) HIP_TRY(c, hipStreamSynchronize(c->stream));
            for (size_t k = 0; k < c->frags_in_flight.size(); ++k) {
                FragBuf &fb = c->frag_pool[c->frags_in_flight[k]];
                const uint32_t n = *fb.h_count;               // (copied when the batch was submitted; the stream has been synchronised)
                if (n > fb.cap) return fail(c, RSQC_ERR_CAPACITY, "fragment candidate overflow");
                if ((rc = arena_reserve(c, c->frag_arena, n))) return rc;
                const void *src[6] = {fb.file.p, fb.qhash.p, fb.name.p, fb.endpos.p, fb.fs.p, fb.h2.p};
                for (int f = 0; f < 6 && n; ++f)
                    HIP_TRY(c, hipMemcpyAsync((char *)c->frag_arena.col[f].p + c->frag_arena.used * c->frag_arena.width[f], src[f],
                                              (size_t)n * c->frag_arena.width[f], hipMemcpyDeviceToDevice, ks));
                c->frag_arena.used += n;
                fb.used = false;
            }
            c->frags_in_flight.clear();
            const uint64_t total = c->frag_arena.used;
            if (total > 0xFFFFFFF0ull) return fail(c, RSQC_ERR_CAPACITY, "too many fragment-size candidates");
            FragCandidates fc{(uint64_t *)c->frag_arena.col[0].p, (uint64_t *)c->frag_arena.col[1].p, (int32_t *)c->frag_arena.col[2].p,
                              (int32_t *)c->frag_arena.col[3].p, (uint32_t *)c->frag_arena.col[4].p, nullptr, (uint32_t)total, nullptr, (uint32_t *)c->frag_arena.col[5].p};
            const auto tf0 = std::chrono::steady_clock::now();
            rc = run_fragment_sizes(ks, fc, (uint32_t)total, c->params.fragment_samples, c->h_fsize, c->h_fcount,
                                    c->frag_remaining, c->frag_scratch, c->frag_kept, c->acc.error);
            if (beside && !rc) HIP_TRY(c, hipStreamSynchronize(ks));      // (run_fragment_sizes returns without a wait when it keeps no samples)
            c->timing.fragment_sizes_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tf0).count();
            if (rc) return fail(c, rc, "fragment-size stage failed");
        }
        // ---- fragment GC content (--fasta runs): the same mate pairing, no cut-off --------------------------
        if (c->have_ref && (!c->gcs_in_flight.empty() || c->gc_arena.used)) {
            HIP_TRY(c, hipStreamSynchronize(c->stream));
            GcCandidates gc{};
            uint64_t total = 0;
            if (c->gc_arena.used == 0 && c->gcs_in_flight.size() == 1) {          // one batch: its candidate arrays are used in place
                GcBuf &gb = c->gc_pool[c->gcs_in_flight[0]];
                total = *gb.h_count;
                if (total > gb.cap) return fail(c, RSQC_ERR_CAPACITY, "GC candidate overflow");
                gc = GcCandidates{(uint64_t *)gb.file.p, (uint64_t *)gb.qhash.p, (uint32_t *)gb.row.p, (int32_t *)gb.endpos.p,
                                  (uint32_t *)gb.flag_lq.p, (int32_t *)gb.tid.p, nullptr, (uint32_t)total, (uint32_t *)gb.h2.p};
            } else {
                for (size_t k = 0; k < c->gcs_in_flight.size(); ++k) {
                    GcBuf &gb = c->gc_pool[c->gcs_in_flight[k]];
                    const uint32_t n = *gb.h_count;
                    if (n > gb.cap) return fail(c, RSQC_ERR_CAPACITY, "GC candidate overflow");
                    if ((rc = arena_reserve(c, c->gc_arena, n))) return rc;
                    const void *src[7] = {gb.file.p, gb.qhash.p, gb.row.p, gb.endpos.p, gb.flag_lq.p, gb.tid.p, gb.h2.p};
                    for (int f = 0; f < 7 && n; ++f)
                        HIP_TRY(c, hipMemcpyAsync((char *)c->gc_arena.col[f].p + c->gc_arena.used * c->gc_arena.width[f], src[f],
                                                  (size_t)n * c->gc_arena.width[f], hipMemcpyDeviceToDevice, c->stream));
                    c->gc_arena.used += n;
                    gb.used = false;
                }
                c->gcs_in_flight.clear();
                total = c->gc_arena.used;
                gc = GcCandidates{(uint64_t *)c->gc_arena.col[0].p, (uint64_t *)c->gc_arena.col[1].p, (uint32_t *)c->gc_arena.col[2].p,
                                  (int32_t *)c->gc_arena.col[3].p, (uint32_t *)c->gc_arena.col[4].p, (int32_t *)c->gc_arena.col[5].p, nullptr, (uint32_t)total, (uint32_t *)c->gc_arena.col[6].p};
            }
            if (total > 0xFFFFFFF0ull) return fail(c, RSQC_ERR_CAPACITY, "too many GC candidates");
            if (total) {
                rc = run_gc_content(c->stream, gc, (uint32_t)total, c->dref, (unsigned long long *)c->d_gc_bins.p, c->gc_scratch, c->acc.error);
                if (rc) return fail(c, rc, "GC content stage failed");
            }
            HIP_TRY(c, hipMemcpyAsync(c->h_gc.data(), c->d_gc_bins.p, (RSQC_GC_BINS + 1) * 8, hipMemcpyDeviceToHost, c->stream));
            HIP_TRY(c, hipStreamSynchronize(c->stream));
        }
    c->fin_e0 = e0; c->fin_e1 = e1;
    return 0;
}
static void unpack_rl_summaries(rsqc_ctx *c) {           // after the stream has been synchronised
    const size_t nb = c->batch_file_index.size();
    c->h_rl_offset.assign(nb + 1, 0); c->h_rl_span.clear(); c->h_rl_state.clear();
    for (size_t k = 0; k < nb; ++k) {
        const uint32_t *w = c->h_rl_raw + k * RSQC_RL_SUMMARY_WORDS;
        const uint32_t P = std::min<uint32_t>(w[0], 128u);
        for (uint32_t j = 0; j < P; ++j) { c->h_rl_span.push_back(w[2 + 2 * j]); c->h_rl_state.push_back((int32_t)w[3 + 2 * j]); }
        c->h_rl_offset[k + 1] = (uint32_t)c->h_rl_span.size();
    }
}
// "Read Length" of a pass that held batches of several file ranges: the summaries (one transfer function per batch or range) applied in
// ascending file index from state 0 -- what read_length_kernel does on the device for a pass of plain batches (src/RNASeQC.cpp:275-278)
static int32_t compose_read_length(const rsqc_ctx *c) {
    std::vector<size_t> order(c->batch_file_index.size());
    for (size_t k = 0; k < order.size(); ++k) order[k] = k;
    std::stable_sort(order.begin(), order.end(), [&](size_t x, size_t y) { return c->batch_file_index[x] < c->batch_file_index[y]; });
    uint32_t r = 0;
    for (size_t k : order)
        for (uint32_t j = c->h_rl_offset[k]; j < c->h_rl_offset[k + 1]; ++j)
            if (c->h_rl_span[j] > r) { r = (uint32_t)c->h_rl_state[j]; break; }     // the first key above the state decides (rsqc_kr.h)
    return (int32_t)r;
}
static void finish_finalize_bookkeeping(rsqc_ctx *c) {
    float ms = 0.f;
    if (c->fin_e0 && hipEventElapsedTime(&ms, c->fin_e0, c->fin_e1) == hipSuccess) c->timing.finalize_ms += ms;
    if (c->fin_e0) { c->event_pool.push_back(c->fin_e0); c->event_pool.push_back(c->fin_e1); c->fin_e0 = c->fin_e1 = nullptr; }
    resolve_events(c);
    for (auto *u : c->transient) retire_batch(c, u);
    c->transient.clear();
    unpack_rl_summaries(c);
    if (c->have_ranges) { c->composed_rl = compose_read_length(c); c->have_composed_rl = true; c->results.read_length = c->composed_rl; }
    free_parked(c);
    c->finalized = true;
}

int rsqc_finalize(rsqc_ctx *c, rsqc_results *out) {
    if (!c || !out || !c->have_ann) return RSQC_ERR_ARG;
    if (c->sticky) return c->sticky;
    HIP_TRY(c, hipSetDevice(c->device));
    int rc;
    if (!c->finalized) {
        RSQC_TRACE("finalize: enter");
        if ((rc = run_finalize_kernels(c, /*early_readback=*/true))) return rc;
        RSQC_TRACE("finalize: kernels enqueued");
        // ---- one read-back of every result vector (also carries the device error flag) ----------------
        if ((rc = read_back(c))) return rc;
        finish_finalize_bookkeeping(c);
        RSQC_TRACE("finalize: bookkeeping done");
    } else if ((rc = read_back(c))) return rc;
    *out = c->results;
    return RSQC_OK;
}

int rsqc_finalize_device(rsqc_ctx *c) {
    if (!c || !c->have_ann) return RSQC_ERR_ARG;
    if (c->sticky) return c->sticky;
    HIP_TRY(c, hipSetDevice(c->device));
    if (c->finalized) return RSQC_OK;
    int rc;
    if ((rc = run_finalize_kernels(c))) return rc;
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    finish_finalize_bookkeeping(c);
    return RSQC_OK;
}

void *rsqc_host_alloc(size_t bytes) {
    void *p = nullptr;
    if (hipHostMalloc(&p, bytes ? bytes : 16, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    return p;
}
void rsqc_host_free(void *p) { if (p) (void)hipHostFree(p); }

int rsqc_device_accumulators(rsqc_ctx *c, void **u64_base, uint64_t *u64_count, void **f64_base, uint64_t *f64_count) {
    if (!c || !c->have_ann || !u64_base || !u64_count || !f64_base || !f64_count) return RSQC_ERR_ARG;
    *u64_base = (char *)c->d_arena.p + c->off_u64; *u64_count = (uint64_t)c->n_genes * 3 + RSQC_N_COUNTERS;
    *f64_base = (char *)c->d_arena.p + c->off_exon; *f64_count = (uint64_t)c->n_exons;
    return RSQC_OK;
}

int rsqc_shard_summary(rsqc_ctx *c, rsqc_shard_info *out) {
    if (!c || !out || !c->have_ann || !c->finalized) return RSQC_ERR_ARG;
    out->n_batches = (uint32_t)c->batch_file_index.size();
    out->batch_file_index = c->batch_file_index.data(); out->batch_records = c->batch_records.data();
    out->rl_offset = c->h_rl_offset.data(); out->rl_span = c->h_rl_span.data(); out->rl_state = c->h_rl_state.data();
    if (c->frag_kept && c->h_sample_file.size() != c->frag_kept) {      // fetched on demand: only sharded runs look at the samples
        HIP_TRY(c, hipSetDevice(c->device));
        c->h_sample_file.resize(c->frag_kept); c->h_sample_size.resize(c->frag_kept);
        HIP_TRY(c, hipMemcpy(c->h_sample_file.data(), c->frag_scratch.k1, (size_t)c->frag_kept * 8, hipMemcpyDeviceToHost));
        HIP_TRY(c, hipMemcpy(c->h_sample_size.data(), c->frag_scratch.v1, (size_t)c->frag_kept * 4, hipMemcpyDeviceToHost));
    }
    out->n_samples = (uint32_t)c->h_sample_file.size();
    out->sample_file_index = c->h_sample_file.data(); out->sample_size = c->h_sample_size.data();
    return RSQC_OK;
}

int rsqc_device_vectors(rsqc_ctx *c, rsqc_device_range out[3]) {
    if (!c || !c->have_ann || !out) return RSQC_ERR_ARG;
    char *A = (char *)c->d_arena.p;
    out[0].base = A + c->off_u64;    out[0].count = (c->off_exon - c->off_u64) / 8;
    out[1].base = A + c->off_exon;   out[1].count = (c->off_gvalid - c->off_exon) / 8;
    out[2].base = A + c->off_gvalid; out[2].count = c->off_ehit - c->off_gvalid;
    return RSQC_OK;
}

int rsqc_reduce_peer(rsqc_ctx *dst, rsqc_ctx *src) {
    if (!dst || !src || dst == src || !dst->have_ann || !src->have_ann || !dst->finalized || !src->finalized) return RSQC_ERR_ARG;
    if (dst->arena_bytes != src->arena_bytes || dst->n_genes != src->n_genes || dst->n_exons != src->n_exons)
        return fail(dst, RSQC_ERR_ARG, "rsqc_reduce_peer: the two contexts hold different annotations");
    HIP_TRY(src, hipSetDevice(src->device));
    HIP_TRY(src, hipStreamSynchronize(src->stream));
    HIP_TRY(dst, hipSetDevice(dst->device));
    // the peer's three ranges are contiguous in its arena: [off_u64, off_ehit)
    const size_t lo = dst->off_u64, hi = dst->off_ehit, bytes = hi - lo;
    DevBuf tmp;
    int rc = dev_alloc(dst, tmp, bytes, false);
    if (rc) return rc;
    HIP_TRY(dst, hipMemcpyPeerAsync(tmp.p, dst->device, (const char *)src->d_arena.p + lo, src->device, bytes, dst->stream));
    char *D = (char *)dst->d_arena.p, *T = (char *)tmp.p - lo;
    launch_reduce_add(dst->stream, (unsigned long long *)(D + dst->off_u64), (const unsigned long long *)(T + dst->off_u64), (dst->off_exon - dst->off_u64) / 8,
                      (double *)(D + dst->off_exon), (const double *)(T + dst->off_exon), (dst->off_gvalid - dst->off_exon) / 8,
                      (uint8_t *)(D + dst->off_gvalid), (const uint8_t *)(T + dst->off_gvalid), dst->off_ehit - dst->off_gvalid);
    HIP_TRY(dst, hipGetLastError());
    HIP_TRY(dst, hipStreamSynchronize(dst->stream));
    tmp.release();
    // the device error flags travel too: a shard's failure is the run's failure
    int err = 0;
    HIP_TRY(src, hipSetDevice(src->device));
    HIP_TRY(src, hipMemcpy(&err, src->acc.error, sizeof(int), hipMemcpyDeviceToHost));
    if (err) { dst->sticky = err; return fail(dst, err, "a shard reported a device-side error"); }
    return RSQC_OK;
}

// ---- the exchange step of a sharded run as ONE RCCL reduction per result range (north_star: "an RCCL
// reduce of the per-gene count vectors and scalar metrics over xGMI at end-of-file") ---------------------------------------
// One process drives the node's GPUs (the command line with --gpus), so the communicators come from ncclCommInitAll over
// the contexts' devices and the three reductions of every GPU are issued inside one group call, each on its context's
// stream.  librccl is bound at run time (like libdeflate in the host reader): a machine without it, or two contexts on
// one device (a communicator cannot hold a device twice: the single-GPU test configuration RSQC_GPU_LIST=0,0), takes the
// peer-copy path below instead.
namespace {
struct RcclApi {
    void *lib = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t *, int, const int *) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    ncclResult_t (*Reduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, int, ncclComm_t, hipStream_t) = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
    bool ok = false;
};
static RcclApi *rccl_api_ptr() {
    static RcclApi A;
    static bool tried = false;
    if (tried) return &A;
    tried = true;
    if (getenv("RSQC_NO_RCCL")) return &A;
    // the librccl that sits beside the HIP runtime THIS library runs on: a process may hold a second ROCm stack (PyTorch
    // bundles its own runtime and RCCL), and a communicator of that one cannot touch this runtime's allocations
    std::vector<std::string> names;
    Dl_info di{};
    if (dladdr(reinterpret_cast<const void *>(static_cast<hipError_t (*)(hipStream_t)>(&hipStreamSynchronize)), &di) && di.dli_fname) {
        std::string dir(di.dli_fname);
        const size_t slash = dir.find_last_of('/');
        if (slash != std::string::npos) { dir.resize(slash + 1); names.push_back(dir + "librccl.so.1"); names.push_back(dir + "librccl.so"); }
    }
    names.push_back("librccl.so.1"); names.push_back("librccl.so");
    for (const std::string &name : names) { A.lib = dlopen(name.c_str(), RTLD_NOW | RTLD_LOCAL); if (A.lib) break; }
    if (!A.lib) return &A;
#define RSQC_RCCL_SYM(field, sym) A.field = reinterpret_cast<decltype(A.field)>(dlsym(A.lib, sym))
    RSQC_RCCL_SYM(CommInitAll, "ncclCommInitAll"); RSQC_RCCL_SYM(CommDestroy, "ncclCommDestroy"); RSQC_RCCL_SYM(GroupStart, "ncclGroupStart");
    RSQC_RCCL_SYM(GroupEnd, "ncclGroupEnd"); RSQC_RCCL_SYM(Reduce, "ncclReduce"); RSQC_RCCL_SYM(GetErrorString, "ncclGetErrorString");
#undef RSQC_RCCL_SYM
    A.ok = A.CommInitAll && A.CommDestroy && A.GroupStart && A.GroupEnd && A.Reduce && A.GetErrorString;
    return &A;
}
}  // namespace

// A group = the contexts of one sharded run + (when RCCL is usable on their devices) one communicator per context, made
// ONCE: ncclCommInitAll over eight GPUs takes longer than the whole BAM loop of a 100 M-record file, so the command line
// brings the group up beside the GTF parse, outside the reference's `Average Reads/Sec` window (src/RNASeQC.cpp:385-394),
// and the end-of-file exchange only issues the reductions.
struct rsqc_group {
    std::vector<rsqc_ctx *> ctxs;
    std::vector<ncclComm_t> comms;       // empty: the peer-copy path
    std::string note;                    // why RCCL is not in use (for -vv)
    double init_ms = 0.0, last_reduce_ms = 0.0;
};

int rsqc_group_create(rsqc_ctx **ctxs, int n, rsqc_group **out) {
    if (!out) return RSQC_ERR_ARG;
    *out = nullptr;
    if (!ctxs || n < 1) return RSQC_ERR_ARG;
    for (int i = 0; i < n; ++i) if (!ctxs[i]) return RSQC_ERR_ARG;
    rsqc_group *g = new rsqc_group();
    g->ctxs.assign(ctxs, ctxs + n);
    const auto t0 = std::chrono::steady_clock::now();
    bool distinct = true;
    for (int i = 0; i < n; ++i) for (int j = 0; j < i; ++j) if (ctxs[i]->device == ctxs[j]->device) distinct = false;
    RcclApi &R = *rccl_api_ptr();
    if (!R.ok) g->note = getenv("RSQC_NO_RCCL") ? "RSQC_NO_RCCL is set" : "librccl not found";
    else if (!distinct) g->note = "two contexts share a device";
    else {
        std::vector<int> devs((size_t)n);
        for (int i = 0; i < n; ++i) devs[(size_t)i] = ctxs[i]->device;
        g->comms.assign((size_t)n, nullptr);
        const ncclResult_t r = R.CommInitAll(g->comms.data(), n, devs.data());
        if (r != ncclSuccess) {
            // no P2P / no shared memory / a mismatched RCCL: not an error of the run -- the peer-copy path sums the shards
            g->note = std::string("ncclCommInitAll: ") + R.GetErrorString(r);
            for (ncclComm_t c : g->comms) if (c) (void)R.CommDestroy(c);
            g->comms.clear();
        }
        (void)hipSetDevice(ctxs[0]->device);
    }
    g->init_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    *out = g;
    return RSQC_OK;
}

void rsqc_group_destroy(rsqc_group *g) {
    if (!g) return;
    if (!g->comms.empty()) { RcclApi &R = *rccl_api_ptr(); for (ncclComm_t c : g->comms) if (c) (void)R.CommDestroy(c); }
    delete g;
}

int rsqc_group_info(const rsqc_group *g, int *uses_rccl, double *init_ms, double *last_reduce_ms, const char **note) {
    if (!g) return RSQC_ERR_ARG;
    if (uses_rccl) *uses_rccl = g->comms.empty() ? 0 : 1;
    if (init_ms) *init_ms = g->init_ms;
    if (last_reduce_ms) *last_reduce_ms = g->last_reduce_ms;
    if (note) *note = g->note.c_str();
    return RSQC_OK;
}

static int shard_error_flags(rsqc_ctx *root, const std::vector<rsqc_ctx *> &ctxs) {
    for (size_t i = 1; i < ctxs.size(); ++i) {      // the device error flags travel too: a shard's failure is the run's failure
        int err = 0;
        HIP_TRY(ctxs[i], hipSetDevice(ctxs[i]->device));
        HIP_TRY(ctxs[i], hipMemcpy(&err, ctxs[i]->acc.error, sizeof(int), hipMemcpyDeviceToHost));
        if (err) { root->sticky = err; return fail(root, err, "a shard reported a device-side error"); }
    }
    HIP_TRY(root, hipSetDevice(root->device));
    return RSQC_OK;
}

int rsqc_group_reduce(rsqc_group *g, int *used_rccl) {
    if (used_rccl) *used_rccl = 0;
    if (!g || g->ctxs.empty()) return RSQC_ERR_ARG;
    const int n = (int)g->ctxs.size();
    rsqc_ctx *root = g->ctxs[0];
    for (int i = 0; i < n; ++i) {
        rsqc_ctx *c = g->ctxs[(size_t)i];
        if (!c->have_ann || !c->finalized) return RSQC_ERR_ARG;
        if (c->arena_bytes != root->arena_bytes || c->n_genes != root->n_genes || c->n_exons != root->n_exons)
            return fail(root, RSQC_ERR_ARG, "rsqc_group_reduce: the contexts hold different annotations");
    }
    const auto t0 = std::chrono::steady_clock::now();
    auto done = [&](int rc) { g->last_reduce_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(); return rc; };
    if (!g->comms.empty()) {
        RcclApi &R = *rccl_api_ptr();
        // the three reducible ranges of the arena (rsqc_device_vectors): u64 counts | f64 sums + owner-only statistics | u8 flags
        const size_t n_u64 = (root->off_exon - root->off_u64) / 8, n_f64 = (root->off_gvalid - root->off_exon) / 8, n_u8 = root->off_ehit - root->off_gvalid;
        ncclResult_t r = R.GroupStart();
        bool issued = false;
        if (r == ncclSuccess) {
            for (int i = 0; i < n && r == ncclSuccess; ++i) {
                rsqc_ctx *c = g->ctxs[(size_t)i];
                char *A = (char *)c->d_arena.p;
                (void)hipSetDevice(c->device);
                r = R.Reduce(A + c->off_u64, A + c->off_u64, n_u64, ncclUint64, ncclSum, 0, g->comms[(size_t)i], c->stream);
                if (r == ncclSuccess) r = R.Reduce(A + c->off_exon, A + c->off_exon, n_f64, ncclFloat64, ncclSum, 0, g->comms[(size_t)i], c->stream);
                if (r == ncclSuccess) r = R.Reduce(A + c->off_gvalid, A + c->off_gvalid, n_u8, ncclUint8, ncclSum, 0, g->comms[(size_t)i], c->stream);
                issued = true;
            }
            const ncclResult_t re = R.GroupEnd();
            if (r == ncclSuccess) r = re;
        }
        if (r != ncclSuccess) {
            // a reduction that was (partly) enqueued may have changed ctxs[0]'s ranges, and waiting on a half-issued group can
            // hang: nothing is synchronised, the run ends here.  A failure before anything was issued takes the peer path.
            if (issued) return done(fail(root, RSQC_ERR_HIP, std::string("RCCL reduction failed after it was issued: ") + R.GetErrorString(r)));
            g->note = std::string("ncclGroupStart: ") + R.GetErrorString(r);
        } else {
            for (int i = 0; i < n; ++i) {
                rsqc_ctx *c = g->ctxs[(size_t)i];
                (void)hipSetDevice(c->device);
                if (hipStreamSynchronize(c->stream) != hipSuccess) return done(fail(root, RSQC_ERR_HIP, "hipStreamSynchronize after the RCCL reduction failed"));
            }
            const int rc = shard_error_flags(root, g->ctxs);
            if (rc == RSQC_OK && used_rccl) *used_rccl = 1;
            return done(rc);
        }
    }
    for (int i = 1; i < n; ++i) { const int rc = rsqc_reduce_peer(root, g->ctxs[(size_t)i]); if (rc != RSQC_OK) return done(rc); }
    return done(RSQC_OK);
}

// create + reduce + destroy in one call (a caller that does not mind the bring-up inside its timed region)
int rsqc_reduce_group(rsqc_ctx **ctxs, int n, int *used_rccl) {
    if (used_rccl) *used_rccl = 0;
    rsqc_group *g = nullptr;
    int rc = rsqc_group_create(ctxs, n, &g);
    if (rc != RSQC_OK) return rc;
    rc = rsqc_group_reduce(g, used_rccl);
    rsqc_group_destroy(g);
    return rc;
}

int rsqc_refresh_results(rsqc_ctx *c, rsqc_results *out) {
    if (!c || !out || !c->have_ann || !c->finalized) return RSQC_ERR_ARG;
    HIP_TRY(c, hipSetDevice(c->device));
    int rc = read_back(c);
    if (rc) return rc;
    *out = c->results;
    return RSQC_OK;
}

int rsqc_get_timing(rsqc_ctx *c, rsqc_timing *out) {
    if (!c || !out) return RSQC_ERR_ARG;
    (void)hipSetDevice(c->device);
    (void)hipStreamSynchronize(c->stream);
    resolve_events(c);
    *out = c->timing;
    return RSQC_OK;
}

int rsqc_reset_timing(rsqc_ctx *c) {
    if (!c) return RSQC_ERR_ARG;
    (void)hipSetDevice(c->device);
    (void)hipStreamSynchronize(c->stream);
    resolve_events(c);
    c->timing = rsqc_timing{};
    return RSQC_OK;
}

// ---- device-side BAM decode ---------------------------------------------------------------------------------------
namespace { int decode_reserve(rsqc_ctx *c, size_t out_bytes, size_t comp_bytes, size_t n_blocks); }
int rsqc_decode_begin(rsqc_ctx *c, const rsqc_decode_params *p) {
    if (!c || !p || p->n_ref < 0) return RSQC_ERR_ARG;
    if (!c->have_ann) return fail(c, RSQC_ERR_ARG, "rsqc_set_annotation must precede rsqc_decode_begin");
    HIP_TRY(c, hipSetDevice(c->device));
    DecodeState &D = c->dec;
    D.tags = BamTagSpec{};
    D.tags.n_ref = p->n_ref;
    if (p->has_chimeric_tag) { D.tags.have_ch = 1; D.tags.ch0 = (uint8_t)p->chimeric_tag[0]; D.tags.ch1 = (uint8_t)p->chimeric_tag[1]; }
    D.tags.n_filter = (uint8_t)c->params.n_filter_tags;
    for (int k = 0; k < c->params.n_filter_tags; ++k) { D.tags.f0[k] = (uint8_t)p->filter_tag[k][0]; D.tags.f1[k] = (uint8_t)p->filter_tag[k][1]; }
    D.next_file_index = p->file_index_base; D.records = 0; D.tail = 0;
    D.unsorted = false; D.n_bad = 0; D.bad_names.clear();
    D.pipelined = p->pipelined != 0; D.pending = false; D.slot = 0;
    if (!D.copy_stream) { HIP_TRY(c, hipStreamCreateWithFlags(&D.copy_stream, hipStreamNonBlocking)); HIP_TRY(c, hipEventCreateWithFlags(&D.ev_copy, hipEventDisableTiming)); }
    D.profile = getenv("RSQC_DECODE_PROFILE") != nullptr;
    D.ms_copy = D.ms_inflate = D.ms_parse = D.ms_call = 0; D.prof_in = D.prof_out = D.prof_calls = 0;
    D.prof_t0 = std::chrono::steady_clock::now();
    if (D.profile && !D.pe[0]) for (auto &e : D.pe) HIP_TRY(c, hipEventCreate(&e));
    int rc;
    if ((rc = dev_alloc(c, D.sum, sizeof(DecodeSummary), false)) || (rc = dev_alloc(c, D.carry, sizeof(DecodeCarry), true)) ||
        (rc = dev_alloc(c, D.scratch, DEC_SCRATCH_WORDS * 4, false))) return rc;
    if (!D.h_sum) HIP_TRY(c, hipHostMalloc((void **)&D.h_sum, sizeof(DecodeSummary), hipHostMallocDefault));
    if (p->reserve_inflated_bytes) {
        const size_t want = (size_t)std::min<uint64_t>(p->reserve_inflated_bytes, (1ull << 31) - D.head);
        if ((rc = decode_reserve(c, want, want / 2, want / 32768 + 64))) return rc;
    }
    D.active = true;
    return RSQC_OK;
}

namespace {
// buffers for a window of `out_bytes` inflated bytes behind the head room
int decode_reserve(rsqc_ctx *c, size_t out_bytes, size_t comp_bytes, size_t n_blocks) {
    DecodeState &D = c->dec;
    int rc;
    if (comp_bytes + 64 > D.comp_cap) {                        // (two halves: the call in flight and the one being copied)
        D.comp_cap = (comp_bytes + comp_bytes / 4 + 64 + 255) & ~(size_t)255;
        if ((rc = dev_alloc(c, D.comp, 2 * D.comp_cap, false))) return rc;
    }
    if (n_blocks > D.blk_cap) {
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        D.blk_cap = n_blocks + n_blocks / 4 + 64;
        if ((rc = dev_alloc(c, D.blocks, 2 * D.blk_cap * sizeof(DevBgzfBlock), false))) return rc;
        if (D.h_blocks) (void)hipHostFree(D.h_blocks);
        HIP_TRY(c, hipHostMalloc((void **)&D.h_blocks, 2 * D.blk_cap * sizeof(DevBgzfBlock), hipHostMallocDefault));
    }
    if (out_bytes <= D.out_cap) return 0;
    const size_t cap = std::max<size_t>(out_bytes + out_bytes / 8, 64u << 20);
    const size_t W = (size_t)D.head + cap;
    // the window buffer keeps the carried-over bytes
    DevBuf nu;
    HIP_TRY(c, hipMalloc(&nu.p, W + 256)); nu.bytes = W + 256;
    if (D.tail) HIP_TRY(c, hipMemcpyAsync((char *)nu.p + D.head - D.tail, (char *)D.ubuf.p + D.head - D.tail, D.tail, hipMemcpyDeviceToDevice, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    D.ubuf.release(); D.ubuf = nu;
    const size_t n_seg = W / DEC_SEG_BYTES + 4, n_rec = W / 36 + 4;
    if ((rc = dev_alloc(c, D.seg, n_seg * sizeof(BamSegment), false)) || (rc = dev_alloc(c, D.seg_rec0, n_seg * 4, false)) ||
        (rc = dev_alloc(c, D.seg_ops0, n_seg * 4, false)) || (rc = dev_alloc(c, D.rec_off, n_rec * 4, false)) ||
        (rc = dev_alloc(c, D.ops_at, n_rec * 4, false)) || (rc = dev_alloc(c, D.mark, n_rec, false)) ||
        (rc = dev_alloc(c, D.core, n_rec * 16 + 64, false)) || (rc = dev_alloc(c, D.aux, n_rec * 16 + 64, false)) || (rc = dev_alloc(c, D.qh2, n_rec * 4 + 64, false)) ||
        (rc = dev_alloc(c, D.cigar, W + 256, false)) || (rc = dev_alloc(c, D.seg_tid, n_rec * 4 + 64, false)) ||
        (rc = dev_alloc(c, D.seg_start, (n_rec + 1) * 8 + 64, false)) || (rc = dev_alloc(c, D.wide_index, n_rec * 8 + 64, false)) ||
        (rc = dev_alloc(c, D.wide_nm, n_rec * 4 + 64, false)) || (rc = dev_alloc(c, D.wide_lq, n_rec * 4 + 64, false)) ||
        (rc = dev_alloc(c, D.wide_nc, n_rec * 4 + 64, false))) return rc;
    D.out_cap = cap;
    return 0;
}
}  // namespace

namespace {
// second half of a call: wait for the window's kernels, read its summary, submit its records as a batch, park what is left
int decode_finish(rsqc_ctx *c, rsqc_decode_window *out) {
    DecodeState &D = c->dec;
    if (!D.pending) return RSQC_OK;
    D.pending = false;
    const DecodeWindow &W = D.pend_w;
    int rc;
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    if (D.profile) {
        float a = 0, b = 0, d = 0;
        (void)hipEventElapsedTime(&a, D.pe[0], D.pe[1]); (void)hipEventElapsedTime(&b, D.pe[1], D.pe[2]); (void)hipEventElapsedTime(&d, D.pe[2], D.pe[3]);
        D.ms_copy += a; D.ms_inflate += b; D.ms_parse += d;
        D.ms_call += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - D.pend_wall0).count();
        D.prof_calls++;
    }
    HIP_TRY(c, hipGetLastError());
    const DecodeSummary &S = *D.h_sum;
    if (S.status & DEC_ST_INFLATE) {
        c->sticky = RSQC_ERR_INPUT;
        return fail(c, RSQC_ERR_INPUT, "BGZF inflate failed (corrupt block " + std::to_string((S.inflate_fail >> 4) - 1) + " of the call, code " + std::to_string(S.inflate_fail & 15u) + ")");
    }
    if (S.status & DEC_ST_BAD_RECORD) { c->sticky = RSQC_ERR_INPUT; return fail(c, RSQC_ERR_INPUT, "bad BAM record"); }
    // the reference's stderr diagnostics
    if (S.unsorted) D.unsorted = true;
    for (uint32_t k = 0; k < S.n_bad && k < DEC_MAX_BAD && D.bad_names.size() < DEC_MAX_BAD; ++k) {
        uint8_t raw[36 + 256] = {0};
        const size_t room = std::min<size_t>(sizeof raw, (size_t)W.end - S.bad_off[k]);
        HIP_TRY(c, hipMemcpy(raw, (const char *)D.ubuf.p + S.bad_off[k], room, hipMemcpyDeviceToHost));
        const size_t l_name = raw[12];
        D.bad_names.emplace_back((const char *)raw + 36, strnlen((const char *)raw + 36, std::min(l_name, room > 36 ? room - 36 : 0)));
    }
    D.n_bad += S.n_bad;
    // what is left of the window: an incomplete record stays in front of the next one
    const uint32_t left = D.pend_limited ? 0u : W.end - S.consumed_end;
    if (left) {
        if ((rc = dev_alloc(c, D.tailtmp, left, false))) return rc;
        HIP_TRY(c, hipMemcpyAsync(D.tailtmp.p, (const char *)D.ubuf.p + S.consumed_end, left, hipMemcpyDeviceToDevice, c->stream));
        if (left <= D.head) HIP_TRY(c, hipMemcpyAsync((char *)D.ubuf.p + D.head - left, D.tailtmp.p, left, hipMemcpyDeviceToDevice, c->stream));
    }
    D.run_tid.assign(S.n_seg, 0);
    if (S.n_seg) HIP_TRY(c, hipMemcpy(D.run_tid.data(), D.seg_tid.p, (size_t)S.n_seg * 4, hipMemcpyDeviceToHost));
    if (out) { out->n_records = S.n_rec; out->n_runs = S.n_seg; out->run_tid = D.run_tid.data(); out->device_batch = rsqc_batch{}; }
    if (S.n_rec) {
        D.last.n = S.n_rec; D.last.file_index_base = D.next_file_index; D.last.core = W.core; D.last.aux = W.aux; D.last.qhash2 = W.qh2; D.last.cigar = W.cigar;
        D.last.n_cigar_total = S.n_ops; D.last.n_seg = S.n_seg; D.last.seg_tid = W.seg_tid; D.last.seg_start = W.seg_start;
        D.last.n_wide = S.n_wide; D.last.wide_index = W.wide_index; D.last.wide_nm = W.wide_nm; D.last.wide_l_qseq = W.wide_lq; D.last.wide_n_cigar = W.wide_nc;
        if (out && !D.pipelined) out->device_batch = D.last;     // (pipelined: the next call's kernels are already queued into these buffers)
        UploadedBatch *u = new UploadedBatch();
        u->pooled = false;
        u->n = S.n_rec; u->n_cigar_total = S.n_ops; u->file_index_base = D.next_file_index;
        DevBatch &d = u->d;
        d.n = S.n_rec; d.core = W.core; d.aux = W.aux; d.qhash2 = W.qh2; d.cigar = W.cigar;
        d.n_seg = S.n_seg; d.seg_tid = W.seg_tid; d.seg_start = W.seg_start;
        d.n_wide = S.n_wide; d.wide_index = W.wide_index; d.wide_nm = W.wide_nm; d.wide_l_qseq = W.wide_lq; d.wide_n_cigar = W.wide_nc;
        c->transient.push_back(u);
        D.next_file_index += S.n_rec; D.records += S.n_rec;
        if ((rc = run_batch(c, u))) return rc;
    }
    if (left > D.head) {
        // a record larger than the head room: every window buffer is rebuilt around a larger one (the per-read kernels of
        // this window finish first: releasing device memory waits for them)
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        uint32_t nh = D.head; while (nh < left) nh <<= 1;
        const size_t keep = D.out_cap;
        D.head = nh; D.out_cap = 0; D.tail = 0;
        if ((rc = decode_reserve(c, keep, 0, 0))) return rc;
        HIP_TRY(c, hipMemcpyAsync((char *)D.ubuf.p + D.head - left, D.tailtmp.p, left, hipMemcpyDeviceToDevice, c->stream));
    }
    D.tail = left;
    return RSQC_OK;
}
}  // namespace

int rsqc_decode_submit(rsqc_ctx *c, const void *compressed, uint64_t compressed_bytes, const rsqc_bgzf_block *blocks, uint32_t n_blocks,
                       uint32_t skip_bytes, uint64_t limit_bytes, rsqc_decode_window *out) {
    if (!c || (!compressed && compressed_bytes) || (!blocks && n_blocks)) return RSQC_ERR_ARG;
    if (c->sticky) return c->sticky;
    DecodeState &D = c->dec;
    if (!D.active) return fail(c, RSQC_ERR_ARG, "rsqc_decode_begin must precede rsqc_decode_submit");
    if (out) { out->n_records = 0; out->n_runs = 0; out->run_tid = nullptr; out->device_batch = rsqc_batch{}; }
    D.last = rsqc_batch{};
    HIP_TRY(c, hipSetDevice(c->device));
    uint64_t total = 0, raw_total = 0;
    uint32_t n_gpu = n_blocks;                                          // blocks [n_gpu, n_blocks) arrive inflated (RSQC_BGZF_INFLATED)
    for (uint32_t k = 0; k < n_blocks; ++k) {
        const rsqc_bgzf_block &b = blocks[k];
        if (b.out_bytes > 65536u || b.in_offset > compressed_bytes || b.in_bytes > compressed_bytes - b.in_offset)
            return fail(c, RSQC_ERR_ARG, "BGZF block outside the compressed buffer or with ISIZE above 64 KiB");
        if (b.flags & RSQC_BGZF_INFLATED) {
            if (n_gpu == n_blocks) n_gpu = k;
            if (b.in_bytes != b.out_bytes || b.in_offset != blocks[n_gpu].in_offset + raw_total)
                return fail(c, RSQC_ERR_ARG, "inflated blocks must lie one after the other in the buffer, in_bytes == out_bytes");
            raw_total += b.out_bytes;
        } else if (n_gpu != n_blocks) return fail(c, RSQC_ERR_ARG, "inflated blocks must form one run at the end of the call");
        total += b.out_bytes;
    }
    if (total + D.head > (1ull << 31)) return fail(c, RSQC_ERR_ARG, "too much inflated data in one rsqc_decode_submit (2 GiB with the bytes carried over)");
    if (skip_bytes > total) return fail(c, RSQC_ERR_ARG, "skip_bytes beyond the inflated data");
    int rc;
    // buffers that have to grow are in use by the call in flight: it is finished first (rare: rsqc_decode_params.reserve_inflated_bytes)
    const int slot = D.slot ^ 1;
    if (D.pending && ((size_t)total > D.out_cap || (size_t)compressed_bytes + 64 > D.comp_cap || n_blocks > D.blk_cap)) {
        if ((rc = decode_finish(c, out))) return rc;
        // (finishing the call in flight may have enlarged the head room for a carried-over record: the limit is about THIS origin)
        if (total + D.head > (1ull << 31)) return fail(c, RSQC_ERR_ARG, "too much inflated data in one rsqc_decode_submit (2 GiB with the bytes carried over)");
    }
    if ((rc = decode_reserve(c, (size_t)total, (size_t)compressed_bytes, n_blocks))) return rc;
    DevBgzfBlock *hb = D.h_blocks + (size_t)slot * D.blk_cap;
    uint32_t raw_at = D.head;                                           // where the caller-inflated run goes in the window
    uint32_t head_used = D.head;                                        // the window origin the table below was laid out for
    auto lay_out_blocks = [&]() {
        head_used = D.head;
        uint32_t at = D.head;
        for (uint32_t k = 0; k < n_gpu; ++k) { hb[k] = DevBgzfBlock{blocks[k].in_offset, blocks[k].in_bytes, blocks[k].out_bytes, at, blocks[k].crc32}; at += blocks[k].out_bytes; }
        raw_at = at;
    };
    lay_out_blocks();
    // the file bytes go up on the copy stream, beside the kernels of the call before this one (pipelined streams)
    uint8_t *dcomp = (uint8_t *)D.comp.p + (size_t)slot * D.comp_cap;
    DevBgzfBlock *dblk = (DevBgzfBlock *)D.blocks.p + (size_t)slot * D.blk_cap;
    if (compressed_bytes) HIP_TRY(c, hipMemcpyAsync(dcomp, compressed, (size_t)compressed_bytes, hipMemcpyHostToDevice, D.copy_stream));
    if (n_gpu) HIP_TRY(c, hipMemcpyAsync(dblk, hb, (size_t)n_gpu * sizeof(DevBgzfBlock), hipMemcpyHostToDevice, D.copy_stream));
    HIP_TRY(c, hipEventRecord(D.ev_copy, D.copy_stream));
    // the call before this one: its kernels have had the time of this call's preparation
    // (an error from here on leaves with the copy drained: the caller's buffer is the caller's again when the call returns)
    if (D.pending) { if ((rc = decode_finish(c, out))) { (void)hipStreamSynchronize(D.copy_stream); return rc; } }
    if (D.head != head_used) {
        // the call just finished left a partial record larger than the head room, and decode_finish moved the window origin to
        // make room for it: the block table above was laid out for the old origin -- lay it out again and send it once more
        // (inflating to the old places would overwrite the carried bytes and shift the window)
        if (total + D.head > (1ull << 31)) { (void)hipStreamSynchronize(D.copy_stream); return fail(c, RSQC_ERR_ARG, "too much inflated data in one rsqc_decode_submit (2 GiB with the bytes carried over)"); }
        HIP_TRY(c, hipStreamSynchronize(D.copy_stream));               // (the first copy of the table reads hb)
        if ((rc = decode_reserve(c, (size_t)total, (size_t)compressed_bytes, n_blocks))) return rc;
        lay_out_blocks();
        if (n_gpu) HIP_TRY(c, hipMemcpyAsync(dblk, hb, (size_t)n_gpu * sizeof(DevBgzfBlock), hipMemcpyHostToDevice, D.copy_stream));
        HIP_TRY(c, hipEventRecord(D.ev_copy, D.copy_stream));
    }
    if (skip_bytes && D.tail) { (void)hipStreamSynchronize(D.copy_stream); return fail(c, RSQC_ERR_ARG, "skip_bytes in the middle of a record"); }
    D.slot = slot;
    D.pend_wall0 = std::chrono::steady_clock::now();
    if (D.profile) HIP_TRY(c, hipEventRecord(D.pe[0], c->stream));
    HIP_TRY(c, hipStreamWaitEvent(c->stream, D.ev_copy, 0));
    HIP_TRY(c, hipMemsetAsync(D.sum.p, 0, sizeof(DecodeSummary), c->stream));
    if (D.profile) HIP_TRY(c, hipEventRecord(D.pe[1], c->stream));
    // the inflate kernel's form follows the call's compression ratio: below 5x the rounds are short matches and literals and the
    // one-pass commit pays; above, long matches dominate and it only costs (RSQC_INFLATE_ONE_PASS=0/1 forces a form)
    static const int force_one_pass = getenv("RSQC_INFLATE_ONE_PASS") ? atoi(getenv("RSQC_INFLATE_ONE_PASS")) : -1;
    const bool one_pass = force_one_pass >= 0 ? force_one_pass != 0 : total < 5 * (uint64_t)std::max<uint64_t>(compressed_bytes - raw_total, 1);
    launch_bgzf_inflate(c->stream, dcomp, dblk, n_gpu, (uint8_t *)D.ubuf.p, (DecodeSummary *)D.sum.p, one_pass);
    if (raw_total)                                                      // the caller-inflated run: staged with the file bytes, now moved into the window
        HIP_TRY(c, hipMemcpyAsync((char *)D.ubuf.p + raw_at, dcomp + blocks[n_gpu].in_offset, (size_t)raw_total, hipMemcpyDeviceToDevice, c->stream));
    const bool limited = limit_bytes && limit_bytes < total;
    DecodeWindow &W = D.pend_w;
    W = DecodeWindow{};
    W.buf = (const uint8_t *)D.ubuf.p;
    W.start = D.head - D.tail + skip_bytes;
    W.end = D.head + (uint32_t)(limited ? limit_bytes : total);
    if (W.start > W.end) W.start = W.end;
    W.n_seg = (W.end - W.start + DEC_SEG_BYTES - 1) / DEC_SEG_BYTES;
    W.seg = (BamSegment *)D.seg.p; W.seg_rec0 = (uint32_t *)D.seg_rec0.p; W.seg_ops0 = (uint32_t *)D.seg_ops0.p;
    W.rec_off = (uint32_t *)D.rec_off.p; W.ops_at = (uint32_t *)D.ops_at.p; W.mark = (uint8_t *)D.mark.p;
    W.core = (rsqc_rec_core *)D.core.p; W.aux = (rsqc_rec_aux *)D.aux.p; W.qh2 = (uint32_t *)D.qh2.p; W.cigar = (uint32_t *)D.cigar.p;
    W.seg_tid = (int32_t *)D.seg_tid.p; W.seg_start = (uint64_t *)D.seg_start.p;
    W.wide_index = (uint64_t *)D.wide_index.p; W.wide_nm = (int32_t *)D.wide_nm.p; W.wide_lq = (int32_t *)D.wide_lq.p; W.wide_nc = (uint32_t *)D.wide_nc.p;
    W.sum = (DecodeSummary *)D.sum.p; W.carry = (DecodeCarry *)D.carry.p; W.tags = D.tags;
    if (D.profile) HIP_TRY(c, hipEventRecord(D.pe[2], c->stream));
    launch_decode_window(c->stream, W, (uint32_t *)D.scratch.p);
    if (D.profile) HIP_TRY(c, hipEventRecord(D.pe[3], c->stream));
    HIP_TRY(c, hipMemcpyAsync(D.h_sum, D.sum.p, sizeof(DecodeSummary), hipMemcpyDeviceToHost, c->stream));
    D.pending = true; D.pend_limited = limited;
    if (D.profile) { D.prof_in += compressed_bytes; D.prof_out += total; }
    // the caller's buffer is free again once the copy is through (the copy engine works beside the kernels)
    HIP_TRY(c, hipStreamSynchronize(D.copy_stream));
    if (!D.pipelined) return decode_finish(c, out);
    return RSQC_OK;
}

int rsqc_decode_end(rsqc_ctx *c, rsqc_decode_info *out) {
    if (!c) return RSQC_ERR_ARG;
    DecodeState &D = c->dec;
    if (!D.active) return fail(c, RSQC_ERR_ARG, "rsqc_decode_begin must precede rsqc_decode_end");
    D.active = false;
    rsqc_decode_window last{};
    if (D.pending) { const int rcf = decode_finish(c, &last); if (rcf) return rcf; }
    if (out) out->last = last;
    if (D.profile)
        fprintf(stderr, "[decode] %llu calls, %.1f MB in, %.1f MB inflated: copy %.1f ms, inflate %.1f ms (%.2f GB/s out), frame+parse %.1f ms, in the calls %.1f ms of %.1f ms between begin and end\n",
                (unsigned long long)D.prof_calls, D.prof_in / 1e6, D.prof_out / 1e6, D.ms_copy, D.ms_inflate, D.ms_inflate > 0 ? D.prof_out / D.ms_inflate / 1e6 : 0.0,
                D.ms_parse, D.ms_call, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - D.prof_t0).count());
    if (out) {
        D.bad_ptrs.clear();
        for (auto &n : D.bad_names) D.bad_ptrs.push_back(n.c_str());
        out->records = D.records; out->unsorted = D.unsorted ? 1 : 0;
        out->n_bad_refid = (int32_t)std::min<uint64_t>(D.n_bad, 0x7fffffff);
        out->bad_refid = D.bad_ptrs.data();
    }
    if (D.tail) { D.tail = 0; return fail(c, RSQC_ERR_INPUT, "truncated BAM record"); }
    return RSQC_OK;
}

const char *rsqc_strerror(int code) {
    switch (code) {
    case RSQC_OK: return "ok";
    case RSQC_ERR_ARG: return "bad argument or call order";
    case RSQC_ERR_HIP: return "HIP runtime error";
    case RSQC_ERR_BAD_CIGAR: return "Unrecognized Cigar Op";
    case RSQC_ERR_CAPACITY: return "device-side capacity exceeded";
    case RSQC_ERR_NO_DEVICE: return "no usable HIP device (this library has no CPU path)";
    case RSQC_ERR_EMPTY_MEDIAN: return "Cannot compute median of an empty list";
    case RSQC_ERR_INPUT: return "corrupt or truncated BAM input";
    default: return "unknown error";
    }
}

const char *rsqc_last_error(rsqc_ctx *c) { return c ? c->last_error.c_str() : ""; }

static const char *const kCounterNames[RSQC_N_COUNTERS] = {
    "Alternative Alignments", "Supplementary Alignments", "Failed Vendor QC", "Low Mapping Quality",
    "Chimeric Fragments_auto", "Chimeric Fragments_tag", "Unique Mapping, Vendor QC Passed Reads",
    "Unpaired Reads", "Mapped Reads", "Mapped Duplicate Reads", "Mapped Unique Reads",
    "Total Mapped Pairs", "End 1 Mapped Reads", "End 1 Mismatches", "End 1 Bases", "Duplicate Pairs",
    "Unique Fragments", "End 2 Mapped Reads", "End 2 Mismatches", "End 2 Bases", "Mismatched Bases",
    "Total Bases", "High Quality Reads", "Low Quality Reads", "Reads used for Intron/Exon counts",
    "Alignment Blocks", "Non-Globin Reads", "Non-Globin Duplicate Reads", "Intronic Reads",
    "Intragenic Reads", "HQ Intronic Reads", "HQ Intragenic Reads", "Intergenic Reads",
    "HQ Intergenic Reads", "Exonic Reads", "HQ Exonic Reads", "Ambiguous Reads", "HQ Ambiguous Reads",
    "rRNA Reads", "End 1 Sense", "End 1 Antisense", "End 2 Sense", "End 2 Antisense",
    "Total Alignments", "Filtered by tag: 0", "Filtered by tag: 1", "Filtered by tag: 2",
    "Filtered by tag: 3", "Filtered by tag: 4", "Split Reads",
};
static_assert(sizeof(kCounterNames) / sizeof(kCounterNames[0]) == RSQC_N_COUNTERS, "one name per counter");

const char *rsqc_counter_name(int counter) {
    return (counter >= 0 && counter < RSQC_N_COUNTERS) ? kCounterNames[counter] : "";
}

const char *rsqc_version(void) { return "RNASeQC 2.4.3 (rnaseqc_amd 0.1, MI355X/gfx950)"; }

uint64_t rsqc_qname_hash(const char *name, size_t len) {
    uint64_t h = 0xCBF29CE484222325ull;                    // FNV-1a 64
    for (size_t i = 0; i < len; ++i) { h ^= (uint8_t)name[i]; h *= 0x100000001B3ull; }
    h ^= h >> 33; h *= 0xFF51AFD7ED558CCDull; h ^= h >> 33; h *= 0xC4CEB9FE1A85EC53ull; h ^= h >> 33;   // fmix64
    return h;
}

uint32_t rsqc_qname_hash2(const char *name, size_t len) { return rsqc::bam_qname_hash2((const uint8_t *)name, (uint32_t)len); }

}  // extern "C"
