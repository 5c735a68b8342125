#include "bam.hpp"
#include "../rsqc_bamrec.h"

#include <sched.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <dlfcn.h>
#include <zlib.h>

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <stdexcept>

#include <chrono>

namespace rsqc_host {

namespace {
double g_t_read = 0, g_t_frame_blocks = 0, g_t_inflate = 0, g_t_frame_rec = 0, g_t_parse = 0, g_t_merge = 0, g_t_move = 0;
inline double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
}

void HostBatch::clear() {
    unsorted = false; bad_refid.clear();
    core.clear(); aux.clear(); qh2.clear(); cigar.clear(); seg_tid.clear(); seg_start.clear();
    wide_index.clear(); wide_nm.clear(); wide_lq.clear(); wide_ncig.clear();
}

bool HostBatch::keep_leading_contig(int32_t tid) {
    if (seg_tid.empty()) return false;
    const size_t total = core.size();
    size_t k = 0;                                                  // records of the leading segment(s) with this tid
    if (seg_tid[0] == tid) k = seg_tid.size() > 1 ? (size_t)seg_start[1] : total;
    if (k >= total) return false;
    const size_t c_end = core[k].cigar_off;
    core.resize(k); aux.resize(k); qh2.resize(k); cigar.resize(c_end);
    if (k == 0) { seg_tid.clear(); seg_start.clear(); }
    else { seg_tid.resize(1); seg_start.resize(1); }
    while (!wide_index.empty() && wide_index.back() >= k) { wide_index.pop_back(); wide_nm.pop_back(); wide_lq.pop_back(); wide_ncig.pop_back(); }
    return true;
}

rsqc_batch HostBatch::view() {
    if (seg_start.size() == seg_tid.size()) seg_start.push_back(core.size());
    else seg_start.back() = core.size();
    rsqc_batch b{};
    b.n = core.size(); b.file_index_base = file_index_base;
    b.core = core.data(); b.aux = aux.data(); b.qhash2 = qh2.data(); b.cigar = cigar.data(); b.n_cigar_total = cigar.size();
    b.n_seg = (uint32_t)seg_tid.size(); b.seg_tid = seg_tid.data(); b.seg_start = seg_start.data();
    b.n_wide = (uint32_t)wide_index.size(); b.wide_index = wide_index.data(); b.wide_nm = wide_nm.data();
    b.wide_l_qseq = wide_lq.data(); b.wide_n_cigar = wide_ncig.data();
    return b;
}

BamReader::~BamReader() {
    if (producer_started_) {
        { std::lock_guard<std::mutex> lk(mu_); stop_ = true; }
        cv_.notify_all();
        producer_.join();
    }
    delete pool_inflate_;
    if (map_) munmap(const_cast<uint8_t *>(map_), map_size_);
    if (fp_) fclose(fp_);
    delete pool_;
    if (getenv("RSQC_HOST_PROFILE"))
        fprintf(stderr, "[bam] read %.3f  frame-blocks %.3f  inflate %.3f  move %.3f  frame-records %.3f  parse %.3f  merge %.3f s\n",
                g_t_read, g_t_frame_blocks, g_t_inflate, g_t_move, g_t_frame_rec, g_t_parse, g_t_merge);
}

void BamReader::set_tags(const std::string &chimeric, const std::vector<std::string> &filters) {
    ch_tag_ = chimeric; filter_tags_ = filters;
}

static inline uint32_t le32(const uint8_t *p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24); }
static inline uint16_t le16(const uint8_t *p) { return (uint16_t)(p[0] | (p[1] << 8)); }

// ------------------------------------------------------------------ fork-join pool
WorkPool::WorkPool(int threads) {
    for (int i = 1; i < threads; ++i) workers_.emplace_back([this] { worker(); });
}
WorkPool::~WorkPool() {
    { std::lock_guard<std::mutex> lk(mu_); stop_ = true; }
    cv_work_.notify_all();
    for (auto &t : workers_) t.join();
}
void WorkPool::worker() {
    uint64_t seen = 0;
    for (;;) {
        const std::function<void(size_t)> *fn;
        size_t n;
        {
            std::unique_lock<std::mutex> lk(mu_);
            cv_work_.wait(lk, [&] { return stop_ || epoch_ != seen; });
            if (stop_) return;
            seen = epoch_; fn = fn_; n = n_tasks_;
        }
        std::string err;
        for (;;) {
            const size_t t = next_.fetch_add(1);
            if (t >= n) break;
            try { (*fn)(t); } catch (std::exception &e) { err = e.what(); }
        }
        {
            std::lock_guard<std::mutex> lk(mu_);
            if (!err.empty() && error_.empty()) error_ = err;
            if (--running_ == 0) cv_done_.notify_all();
        }
    }
}
void WorkPool::run(size_t n_tasks, const std::function<void(size_t)> &fn) {
    if (n_tasks == 0) return;
    if (workers_.empty() || n_tasks == 1) { for (size_t t = 0; t < n_tasks; ++t) fn(t); return; }
    {
        std::lock_guard<std::mutex> lk(mu_);
        fn_ = &fn; n_tasks_ = n_tasks; next_.store(0); running_ = workers_.size(); error_.clear(); ++epoch_;
    }
    cv_work_.notify_all();
    std::string err;
    for (;;) {
        const size_t t = next_.fetch_add(1);
        if (t >= n_tasks) break;
        try { fn(t); } catch (std::exception &e) { err = e.what(); }
    }
    std::unique_lock<std::mutex> lk(mu_);
    cv_done_.wait(lk, [&] { return running_ == 0; });
    if (err.empty()) err = error_;
    if (!err.empty()) throw std::runtime_error(err);
}

// CPUs this process may actually use: the affinity mask, capped by the cgroup's CFS quota (cpu.max of cgroup v2,
// cpu.cfs_quota_us / cpu.cfs_period_us of v1).  More runnable threads than that are not faster, they are throttled:
// the whole group is frozen for the rest of every period once the quota is spent.
int effective_cpus() {
    int n = (int)std::max(1u, std::thread::hardware_concurrency());
    cpu_set_t set;
    if (sched_getaffinity(0, sizeof set, &set) == 0) n = std::min(n, std::max(1, CPU_COUNT(&set)));
    auto quota_of = [](const char *path, const char *path2) -> double {
        FILE *f = fopen(path, "r");
        if (!f) return 0.0;
        char a[64] = {0}; long long period = 0, quota = -1;
        if (path2 == nullptr) {                                   // "max 100000" | "1600000 100000"
            if (fscanf(f, "%63s %lld", a, &period) == 2 && strcmp(a, "max") != 0) quota = atoll(a);
            fclose(f);
        } else {
            if (fscanf(f, "%lld", &quota) != 1) quota = -1;
            fclose(f);
            FILE *g = fopen(path2, "r");
            if (g) { if (fscanf(g, "%lld", &period) != 1) period = 0; fclose(g); }
        }
        return (quota > 0 && period > 0) ? (double)quota / (double)period : 0.0;
    };
    double q = quota_of("/sys/fs/cgroup/cpu.max", nullptr);
    if (q <= 0.0) q = quota_of("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "/sys/fs/cgroup/cpu/cpu.cfs_period_us");
    if (q > 0.0) n = std::min(n, std::max(1, (int)(q + 0.5)));
    return n;
}

bool BamReader::set_threads(int n) {
    if (n < 1) n = 1;
    if (n == n_threads_ && pool_) return true;
    if (producer_started_) return false;           // fixed once decoding has begun (open() starts it): call before open()
    delete pool_; delete pool_inflate_;
    n_threads_ = n;
    // n threads in two pools (n = 1: one thread each).  With libdeflate and the one-pass parser, inflating a group costs
    // ~1.5x what framing + parsing + copying it costs (measured per thread), so the producer side gets 60 % of them
    // (RSQC_HOST_INFLATE_THREADS overrides: the parser keeps the rest, at least one).
    int n_inflate = n <= 2 ? 1 : (n * 3 + 2) / 5;
    if (const char *e = getenv("RSQC_HOST_INFLATE_THREADS")) n_inflate = atoi(e);
    n_inflate = std::max(1, std::min(n_inflate, std::max(1, n - 1)));
    pool_ = new WorkPool(std::max(1, n - n_inflate));
    pool_inflate_ = new WorkPool(n_inflate);
    return true;
}

// ---- by-contig reading (counterpart of an index-driven region reader) -------------------------------
// <bam>.bai: per reference, the metadata pseudo-bin 37450 {ref_beg, ref_end, n_mapped, n_unmapped} when the indexer wrote
// it (samtools does), else the smallest chunk begin / largest chunk end over its bins; n_records is 0 when unknown.
bool BamReader::load_index(const std::string &bai_path) {
    FILE *f = fopen(bai_path.c_str(), "rb");
    if (!f) return false;
    std::vector<uint8_t> d;
    { uint8_t tmp[1 << 16]; size_t g; while ((g = fread(tmp, 1, sizeof tmp, f)) > 0) d.insert(d.end(), tmp, tmp + g); }
    fclose(f);
    size_t p = 0;
    auto need = [&](size_t k) { return p + k <= d.size(); };
    auto u32 = [&]() { const uint32_t v = le32(d.data() + p); p += 4; return v; };
    auto u64 = [&]() { const uint64_t v = (uint64_t)le32(d.data() + p) | ((uint64_t)le32(d.data() + p + 4) << 32); p += 8; return v; };
    if (!need(8) || memcmp(d.data(), "BAI\1", 4) != 0) return false;
    p = 4;
    const uint32_t n_ref = u32();
    std::vector<ContigRange> idx(n_ref);
    for (uint32_t r = 0; r < n_ref; ++r) {
        if (!need(4)) return false;
        const uint32_t n_bin = u32();
        uint64_t lo = ~0ull, hi = 0; bool meta = false;
        for (uint32_t b = 0; b < n_bin; ++b) {
            if (!need(8)) return false;
            const uint32_t bin = u32(), n_chunk = u32();
            if (!need(16 * (size_t)n_chunk)) return false;
            if (bin == 37450 && n_chunk == 2) {
                idx[r].beg = u64(); idx[r].end = u64(); idx[r].n_records = u64(); idx[r].n_records += u64(); meta = true;
            } else for (uint32_t c = 0; c < n_chunk; ++c) { const uint64_t cb = u64(), ce = u64(); lo = std::min(lo, cb); hi = std::max(hi, ce); }
        }
        if (!meta && lo != ~0ull) { idx[r].beg = lo; idx[r].end = hi; idx[r].n_records = 0; }
        idx[r].present = meta || lo != ~0ull;
        if (!need(4)) return false;
        const uint32_t n_intv = u32();
        if (!need(8 * (size_t)n_intv)) return false;
        p += 8 * (size_t)n_intv;
    }
    index_ = std::move(idx);
    n_no_coor_ = need(8) ? u64() : 0;
    return true;
}

// Restarts the stream at a BGZF virtual offset and stops inflating behind `voff_end` (0 = end of file).  Memory-mapped
// files only; must follow open().  The caller reads the records it wants (e.g. ContigRange::n_records, or until the
// segment table shows another contig).
bool BamReader::seek(uint64_t voff, uint64_t voff_end) {
    if (!map_) return false;
    if (producer_started_) {
        { std::lock_guard<std::mutex> lk(mu_); stop_ = true; }
        cv_.notify_all();
        producer_.join();
        stop_ = false; producer_started_ = false;
    }
    stage_state_ = kEmpty; stage_bytes_ = 0; stage_error_.clear();
    buf_.clear(); pos_ = 0; eof_ = false; file_eof_ = false;
    cpos_ = (size_t)(voff >> 16);
    skip_ = (size_t)(voff & 0xFFFF);
    // the block that holds the end is still inflated -- unless the range ends exactly on its first byte (htslib writes such an end
    // when the last record of a contig fills its block): that block holds nothing of the range, and when it starts with the
    // first part of a record larger than a block, framing it would report a truncated record instead of the end of the range
    climit_ = voff_end ? (size_t)(voff_end >> 16) + ((voff_end & 0xFFFF) ? 1 : 0) : 0;
    if (cpos_ > map_size_) return false;
    return true;
}

// libdeflate (whole-buffer inflate, 2-3x zlib on BGZF-sized blocks) when the shared library is on the system: it ships
// without headers in this image, so its three entry points are bound at run time; zlib otherwise (RSQC_HOST_ZLIB=1 forces it).
namespace {
struct LibDeflate {
    void *(*alloc)() = nullptr;
    int (*decompress)(void *, const void *, size_t, void *, size_t, size_t *) = nullptr;   // 0 = LIBDEFLATE_SUCCESS
    void (*release)(void *) = nullptr;
    uint32_t (*crc)(uint32_t, const void *, size_t) = nullptr;                              // libdeflate_crc32 (carry-less multiply)
    bool ok = false;
    LibDeflate() {
        if (const char *e = getenv("RSQC_HOST_ZLIB")) if (atoi(e)) return;
        void *h = dlopen("libdeflate.so.0", RTLD_NOW | RTLD_LOCAL);
        if (!h) h = dlopen("libdeflate.so", RTLD_NOW | RTLD_LOCAL);
        if (!h) return;
        alloc = (void *(*)())dlsym(h, "libdeflate_alloc_decompressor");
        decompress = (int (*)(void *, const void *, size_t, void *, size_t, size_t *))dlsym(h, "libdeflate_deflate_decompress");
        release = (void (*)(void *))dlsym(h, "libdeflate_free_decompressor");
        crc = (uint32_t (*)(uint32_t, const void *, size_t))dlsym(h, "libdeflate_crc32");
        ok = alloc && decompress && release && crc;
    }
};
const LibDeflate &libdeflate() { static const LibDeflate d; return d; }
}  // namespace

bool bgzf_inflate_block(const uint8_t *in, size_t in_len, uint8_t *out, size_t out_len, uint32_t crc) {
    if (!out_len) return true;
    const LibDeflate &ld = libdeflate();
    if (ld.ok) {
        static thread_local void *d = nullptr;                      // (one decompressor per thread, kept)
        if (!d) d = ld.alloc();
        if (!d) return false;
        return ld.decompress(d, in, in_len, out, out_len, nullptr) == 0 && ld.crc(0, out, out_len) == crc;
    }
    z_stream zs{};
    if (inflateInit2(&zs, -15) != Z_OK) return false;
    zs.next_in = const_cast<uint8_t *>(in); zs.avail_in = (uInt)in_len;
    zs.next_out = out; zs.avail_out = (uInt)out_len;
    const int rc = inflate(&zs, Z_FINISH);
    const bool ok = rc == Z_STREAM_END && zs.avail_out == 0 && (uint32_t)crc32(crc32(0L, Z_NULL, 0), out, (uInt)out_len) == crc;
    inflateEnd(&zs);
    return ok;
}

// BGZF blocks are independent deflate streams whose uncompressed size sits in the trailer, so a group of
// blocks is framed sequentially (headers only) and inflated in parallel straight into place.
bool BamReader::produce_group(RawVec<uint8_t> &dst, size_t head) {
    size_t GROUP_BYTES = (size_t)64 << 20;              // uncompressed bytes per group
    if (const char *e = getenv("RSQC_HOST_GROUP_BYTES")) GROUP_BYTES = std::max<size_t>(1, (size_t)atoll(e));   // (tests: many small groups)
    const size_t READ_CHUNK = (size_t)16 << 20;
    struct Blk { size_t coff, clen, out; uint32_t isize, crc; };
    std::vector<Blk> blks;
    size_t total = 0;
    for (;;) {
        // frame complete blocks of the compressed window
        while (total < GROUP_BYTES) {
            const size_t avail = (map_ ? map_size_ : cbuf_.size()) - cpos_;
            if (avail < 18) break;
            const uint8_t *h = (map_ ? map_ : cbuf_.data()) + cpos_;
            if (h[0] != 0x1f || h[1] != 0x8b || h[2] != 8 || !(h[3] & 4)) throw std::runtime_error("not a BGZF block");
            const uint16_t xlen = le16(h + 10);
            if (avail < 12 + (size_t)xlen) break;
            uint32_t bsize = 0;
            for (size_t o = 0; o + 4 <= xlen;) {
                const uint8_t *x = h + 12 + o;
                const uint16_t slen = le16(x + 2);
                if (x[0] == 'B' && x[1] == 'C' && slen == 2 && o + 6 <= xlen) bsize = (uint32_t)le16(x + 4) + 1;
                o += 4 + (size_t)slen;
            }
            if (!bsize) throw std::runtime_error("BGZF block without BC field");
            if (bsize < 12u + xlen + 8u) throw std::runtime_error("bad BGZF block size");
            if (avail < bsize) break;
            const size_t clen = bsize - xlen - 12 - 8;
            const uint32_t isize = le32(h + bsize - 4);
            if (isize > 65536u) throw std::runtime_error("BGZF block with ISIZE above 64 KiB (corrupt trailer)");   // before any buffer is sized from it
            if (climit_ && cpos_ >= climit_) break;                   // by-contig reading: nothing to inflate past the range
            blks.push_back(Blk{cpos_ + 12 + xlen, clen, total, isize, le32(h + bsize - 8)});
            total += isize;
            cpos_ += bsize;
        }
        if (total >= GROUP_BYTES || file_eof_ || map_) break;
        // need more compressed bytes: drop the consumed prefix only when no framed block still points into it
        if (blks.empty() && cpos_ > 0) { cbuf_.erase_front(cpos_); cpos_ = 0; }
        const double tr = now_s();
        const size_t old = cbuf_.size();
        cbuf_.resize(old + READ_CHUNK);
        const size_t got = fread(cbuf_.data() + old, 1, READ_CHUNK, fp_);
        cbuf_.resize(old + got);
        g_t_read += now_s() - tr;
        if (got == 0) file_eof_ = true;
    }
    if (blks.empty()) {
        if (climit_ && cpos_ >= climit_) return false;            // the end of a seek() range is an end of stream
        if ((map_ ? map_size_ : cbuf_.size()) - cpos_ != 0) throw std::runtime_error("truncated BGZF block");
        return false;
    }
    dst.resize(head + total);
    stage_bytes_ = total;
    const uint8_t *cdata = map_ ? map_ : cbuf_.data();
    uint8_t *odata = dst.data() + head;
    const double ti = now_s();
    // tasks of 4 consecutive blocks (fine enough to balance across ~100 threads): one z_stream per task
    const size_t per = 4, n_tasks = (blks.size() + per - 1) / per;
    const LibDeflate &ld = libdeflate();
    pool_inflate_->run(n_tasks, [&](size_t t) {
        const size_t b0 = t * per, b1 = std::min(blks.size(), b0 + per);
        if (ld.ok) {
            void *d = ld.alloc();
            if (!d) throw std::runtime_error("libdeflate: out of memory");
            for (size_t k = b0; k < b1; ++k) {
                const Blk &bk = blks[k];
                if (!bk.isize) continue;
                // exact fill required; the block's CRC-32 is checked like htslib's bgzf reader does
                if (ld.decompress(d, cdata + bk.coff, bk.clen, odata + bk.out, bk.isize, nullptr) != 0 ||
                    ld.crc(0, odata + bk.out, bk.isize) != bk.crc) {
                    ld.release(d);
                    throw std::runtime_error("BGZF inflate failed (corrupt block)");
                }
            }
            ld.release(d);
            return;
        }
        z_stream zs{};
        if (inflateInit2(&zs, -15) != Z_OK) throw std::runtime_error("zlib init failed");
        for (size_t k = b0; k < b1; ++k) {
            const Blk &bk = blks[k];
            if (!bk.isize) continue;
            if (k != b0) inflateReset(&zs);
            zs.next_in = const_cast<uint8_t *>(cdata + bk.coff); zs.avail_in = (uInt)bk.clen;
            zs.next_out = odata + bk.out; zs.avail_out = bk.isize;
            const int rc = inflate(&zs, Z_FINISH);
            if (rc != Z_STREAM_END || zs.avail_out != 0 || (uint32_t)crc32(crc32(0L, Z_NULL, 0), odata + bk.out, bk.isize) != bk.crc) {
                inflateEnd(&zs); throw std::runtime_error("BGZF inflate failed (corrupt block)");
            }
        }
        inflateEnd(&zs);
    });
    g_t_inflate += now_s() - ti;

    return true;
}

// ---- producer thread: keeps ONE inflated group ahead of the parser
void BamReader::producer_main() {
    for (;;) {
        {
            std::unique_lock<std::mutex> lk(mu_);
            cv_.wait(lk, [&] { return stop_ || stage_state_ == kEmpty; });
            if (stop_) return;
        }
        StageState next = kReady;
        std::string err;
        try { if (!produce_group(stage_, kHead)) next = kEof; }
        catch (std::exception &e) { next = kError; err = e.what(); }
        {
            std::lock_guard<std::mutex> lk(mu_);
            stage_state_ = next; stage_error_ = err;
        }
        cv_.notify_all();
        if (next != kReady) return;
    }
}
void BamReader::start_producer() {
    if (producer_started_) return;
    if (!pool_) {
        // two pools (inflate ahead, frame + parse; split in set_threads) with TWICE as many threads as CPUs the process
        // may use: the pools take turns blocking on each other, so about half of the threads are runnable at a time.  The
        // count follows the cgroup quota, not the hardware thread count -- measured on the 256-thread GPU box under its
        // 16-CPU quota (tools/decode_sweep.py, 100 M records): 32 threads 79 M reads/s, 16: 64-72 M, 64: 54 M, 128: 31 M
        int n = std::min(2 * effective_cpus(), 96);
        if (const char *e = getenv("RSQC_HOST_THREADS")) n = atoi(e);
        set_threads(n);
    }
    producer_started_ = true;
    producer_ = std::thread([this] { producer_main(); });
}

// consumer side: take the group the producer prepared, put the unread tail in front of it, let the producer go on
bool BamReader::fill_group() {
    if (eof_) return false;
    start_producer();
    {
        std::unique_lock<std::mutex> lk(mu_);
        cv_.wait(lk, [&] { return stage_state_ != kEmpty; });
        if (stage_state_ == kError) throw std::runtime_error(stage_error_);
        if (stage_state_ == kEof) { eof_ = true; return false; }
    }
    const double tm = now_s();
    const size_t tail = buf_.size() - pos_, got = stage_bytes_;
    if (tail <= kHead) {
        if (tail) memcpy(stage_.data() + kHead - tail, buf_.data() + pos_, tail);
        buf_.swap(stage_);                                        // (pointer swap)
        buf_.resize(kHead + got);
        pos_ = kHead - tail;
        if (skip_) { pos_ += std::min(skip_, got); skip_ = 0; }   // seek(): the first block starts before the wanted record
    } else {                                                     // a record larger than the head room: grow in place
        if (pos_ > 0) { buf_.erase_front(pos_); pos_ = 0; }
        const size_t base = buf_.size();
        buf_.resize(base + got);
        memcpy(buf_.data() + base, stage_.data() + kHead, got);
    }
    g_t_move += now_s() - tm;
    {
        std::lock_guard<std::mutex> lk(mu_);
        stage_state_ = kEmpty;
    }
    cv_.notify_all();
    return true;
}

bool BamReader::fill(size_t need) {
    while (buf_.size() - pos_ < need) { if (!fill_group()) break; }
    return buf_.size() - pos_ >= need;
}

bool BamReader::open(const std::string &path) {
    fp_ = fopen(path.c_str(), "rb");
    if (!fp_) return false;
    {
        struct stat st;
        if (fstat(fileno(fp_), &st) == 0 && S_ISREG(st.st_mode) && st.st_size > 0 && !getenv("RSQC_HOST_NO_MMAP")) {
            void *m = mmap(nullptr, (size_t)st.st_size, PROT_READ, MAP_PRIVATE, fileno(fp_), 0);
            if (m != MAP_FAILED) {
                map_ = (const uint8_t *)m; map_size_ = (size_t)st.st_size;
                (void)madvise(m, map_size_, MADV_SEQUENTIAL);
            }
        }
    }
    try {
        if (!fill(12) || memcmp(buf_.data() + pos_, "BAM\1", 4) != 0) return false;
        const uint32_t l_text = le32(buf_.data() + pos_ + 4);
        if (!fill(12 + (size_t)l_text)) return false;
        const uint32_t n_ref = le32(buf_.data() + pos_ + 8 + l_text);
        pos_ += 12 + l_text;
        for (uint32_t i = 0; i < n_ref; ++i) {
            if (!fill(4)) return false;
            const uint32_t l_name = le32(buf_.data() + pos_);
            if (!fill(8 + (size_t)l_name)) return false;
            names_.emplace_back((const char *)buf_.data() + pos_ + 4, l_name ? l_name - 1 : 0);
            pos_ += 8 + l_name;
        }
    } catch (std::exception &) { return false; }
    return true;
}

// ---- record framing ---------------------------------------------------------------------------------
// A BAM record can only be located by hopping from the previous one (block_size), one dependent cache miss per
// record.  The window is therefore cut into chunks that are framed (and parsed) IN PARALLEL from a guessed record
// start (the first offset where two consecutive records pass a structural check), and the guesses are then verified
// sequentially: chunk c's guess is accepted only if the TRUE chain of chunk c-1 (which starts at a verified
// boundary) ends exactly on it; otherwise chunk c is redone from the true position.  The result is exact --
// a wrong guess costs time, never correctness.
namespace {
inline bool plausible_record(const uint8_t *buf, size_t p, size_t end, int32_t n_ref) { return rsqc::bam_plausible(buf, p, end, n_ref); }
}  // namespace

// One pass per chunk: a worker hops through the records of its chunk from the guessed start AND parses each one while
// it is in cache, into chunk-local arrays (a record is ~190 bytes of which framing needs 6 and parsing ~100: framing
// first and parsing later read every record's lines from memory twice).  After the sequential verification of the
// guesses the chunk-local arrays are copied to their final place in parallel (40 bytes per record).
namespace {
struct ChunkOut {
    size_t start = 0, end = 0; bool guessed = false;
    std::vector<uint32_t> cig_end;                       // per record: CIGAR words of the chunk up to and including it
    std::vector<rsqc_rec_core> core;                     // cigar_off relative to the chunk
    std::vector<rsqc_rec_aux> aux;
    std::vector<uint32_t> qh2;                           // second name hash (rsqc_batch.qhash2)
    std::vector<uint32_t> cig;
    std::vector<std::pair<uint32_t, int32_t>> segs;      // (local record index, tid): contig changes inside the chunk (+ its first record)
    std::vector<uint32_t> widx; std::vector<int32_t> wnm, wlq; std::vector<uint32_t> wnc;
    int32_t last_tid = 0;
    // what the reference's loop reports on stderr: records that reach src/RNASeQC.cpp:333 with a RefID outside the header
    // (their names), and positions that go backwards inside a contig (:354); judged on primary, mapped, QC-passed records
    std::vector<std::pair<uint32_t, std::string>> bad_ref;   // (local record index, QNAME)
    bool unsorted = false, have_q = false; int32_t first_tid = 0, first_pos = 0, q_tid = 0, q_pos = 0;
    void clear() { core.clear(); aux.clear(); qh2.clear(); cig.clear(); cig_end.clear(); segs.clear(); widx.clear(); wnm.clear(); wlq.clear(); wnc.clear(); last_tid = 0;
                   bad_ref.clear(); unsorted = false; have_q = false; }
    size_t size() const { return core.size(); }
    // keep the first k records
    void truncate(size_t k) {
        if (k >= core.size()) return;
        const uint32_t nc = k ? cig_end[k - 1] : 0u;
        core.resize(k); aux.resize(k); qh2.resize(k); cig_end.resize(k); cig.resize(nc);
        while (!segs.empty() && segs.back().first >= k) segs.pop_back();
        while (!widx.empty() && widx.back() >= k) { widx.pop_back(); wnm.pop_back(); wlq.pop_back(); wnc.pop_back(); }
        last_tid = segs.empty() ? 0 : segs.back().second;
        // the order / RefID diagnostics describe parsed records: redo them for the kept ones (the dropped records come
        // back with the next batch)
        while (!bad_ref.empty() && bad_ref.back().first >= k) bad_ref.pop_back();
        unsorted = false; have_q = false;
        size_t sg = 0;
        for (size_t r = 0; r < k; ++r) {
            while (sg + 1 < segs.size() && segs[sg + 1].first <= r) ++sg;
            const int32_t tid = segs.empty() ? -1 : segs[sg].second;
            if (aux[r].flag & (RSQC_FSECONDARY | RSQC_FQCFAIL | RSQC_FSUPP | RSQC_FUNMAP)) continue;
            if (tid < 0 || tid >= n_ref) continue;                 // (names of undefined RefIDs are reported when their batch is read)
            if (!have_q) { first_tid = tid; first_pos = core[r].pos; }
            else if (q_tid == tid && q_pos > core[r].pos) unsorted = true;
            have_q = true; q_tid = tid; q_pos = core[r].pos;
        }
    }
    int32_t n_ref = 0;
};
using TagSpec = rsqc::BamTagSpec;

// hop from `p` while records are complete and start before `limit`, parsing each into `o`; returns the first start
// >= limit (or the start of the first incomplete record)
size_t frame_and_parse(const uint8_t *buf, size_t p, size_t limit, size_t end, const TagSpec &tags, ChunkOut &o) {
    bool have_last = false; int32_t last_tid = 0;
    o.n_ref = tags.n_ref;
    while (p < limit && p + 4 <= end) {
        const uint32_t block_size = le32(buf + p);
        if (block_size < 32) throw std::runtime_error("bad BAM record");
        if (p + 4 + (size_t)block_size > end) break;
        // Records are found by hopping, one dependent cache miss after the other, in data another core has just inflated.
        // The record after the next one most likely has this record's size: fetch its first two lines and its tail (aux)
        {
            const uint8_t *guess = buf + p + 2 * (4 + (size_t)block_size);
            __builtin_prefetch(guess, 0, 1); __builtin_prefetch(guess + 64, 0, 1); __builtin_prefetch(guess + block_size - 32, 0, 1);
        }
        // (the record itself: rsqc_bamrec.h, the code the device decode runs as well)
        rsqc::BamRecOut ro;
        if (!rsqc::bam_parse_record(buf + p, block_size, tags, ro)) throw std::runtime_error("bad BAM record");
        const int32_t tid = ro.tid, pos = ro.core.pos;
        const uint32_t k = (uint32_t)o.core.size();
        // a chunk's first record always opens a provisional segment; whether it continues the previous chunk's contig
        // is decided at merge time
        if (!have_last || last_tid != tid) o.segs.emplace_back(k, tid);
        have_last = true; last_tid = tid;
        if (rsqc::bam_flag_judged(ro.aux.flag)) {
            if (tid < 0 || tid >= tags.n_ref) { if (o.bad_ref.size() < 64) o.bad_ref.emplace_back(k, std::string((const char *)buf + p + 36, ro.qname_len)); }
            else {
                if (!o.have_q) { o.first_tid = tid; o.first_pos = pos; }
                else if (o.q_tid == tid && o.q_pos > pos) o.unsorted = true;
                o.have_q = true; o.q_tid = tid; o.q_pos = pos;
            }
        }
        ro.core.cigar_off = (uint32_t)o.cig.size();
        if (ro.wide) { o.widx.push_back(k); o.wnm.push_back(ro.nm); o.wlq.push_back(ro.l_seq); o.wnc.push_back(ro.n_ops); }
        const size_t c0 = o.cig.size();
        o.cig.resize(c0 + ro.n_ops);
        const uint8_t *ops = buf + p + ro.ops_off;
        for (uint32_t ci = 0; ci < ro.n_ops; ++ci) o.cig[c0 + ci] = le32(ops + 4 * (size_t)ci);
        o.cig_end.push_back((uint32_t)o.cig.size());
        o.core.push_back(ro.core); o.aux.push_back(ro.aux); o.qh2.push_back(ro.qhash2);
        p += 4 + (size_t)block_size;
    }
    o.last_tid = last_tid;
    return p;
}
}  // namespace

size_t BamReader::read_batch(HostBatch &out, size_t max_records) {
    size_t n = 0;
    if (!pool_) set_threads(1);
    const int32_t n_ref = (int32_t)names_.size();
    TagSpec tags{};
    tags.n_ref = n_ref;
    if (ch_tag_.size() == 2) { tags.have_ch = 1; tags.ch0 = (uint8_t)ch_tag_[0]; tags.ch1 = (uint8_t)ch_tag_[1]; }
    for (size_t fi = 0; fi < filter_tags_.size() && fi < RSQC_MAX_FILTER_TAGS; ++fi) {
        // a tag name that is not two characters long can never match; keep its bit position
        tags.f0[tags.n_filter] = filter_tags_[fi].size() == 2 ? filter_tags_[fi][0] : '\0';
        tags.f1[tags.n_filter] = filter_tags_[fi].size() == 2 ? filter_tags_[fi][1] : '\0';
        ++tags.n_filter;
    }
    // diagnostic (tools/decode_sweep.py --inflate-only): consume the inflated stream without framing or parsing it
    static const bool drain_only = getenv("RSQC_HOST_INFLATE_ONLY") != nullptr;
    if (drain_only) { while (fill_group()) pos_ = buf_.size(); return 0; }
    static thread_local std::vector<ChunkOut> tl_chunks;         // capacity is reused from group to group
    std::vector<ChunkOut> &chunks = tl_chunks;                   // (the workers must see THIS thread's instance)
    while (n < max_records) {
        if (buf_.size() - pos_ < 4 && !fill(4)) {
            if (buf_.size() - pos_ != 0) throw std::runtime_error("truncated BAM record");
            break;
        }
        const double tf = now_s();
        const uint8_t *bufp = buf_.data();
        const size_t end = buf_.size(), want = max_records - n;
        // ---- parallel speculative framing + parsing of [pos_, end): a BAM record can only be located by hopping from the
        //      previous one, so every chunk starts from a GUESSED record start (the first offset where two consecutive
        //      records pass a structural check)
        const size_t CH = (size_t)1 << 18;                       // 256 KB chunks: a few hundred per inflated group
        const size_t n_ch = std::max<size_t>(1, (end - pos_ + CH - 1) / CH);
        if (chunks.size() < n_ch) chunks.resize(n_ch);
        pool_->run(n_ch, [&](size_t c) {
            ChunkOut &o = chunks[c];
            o.clear();
            const size_t lo = pos_ + c * CH, hi = std::min(end, lo + CH);
            size_t p = lo;
            if (c > 0) {
                o.guessed = false;
                for (; p < hi; ++p) {
                    if (!plausible_record(bufp, p, end, n_ref)) continue;
                    const size_t q = p + 4 + le32(bufp + p);
                    if (q + 36 <= end ? plausible_record(bufp, q, end, n_ref) : true) { o.guessed = true; break; }
                }
                if (!o.guessed) { o.start = o.end = hi; return; }
            } else o.guessed = true;
            o.start = p;
            try { o.end = frame_and_parse(bufp, p, hi, end, tags, o); }
            catch (std::exception &) {
                if (c == 0) throw;                               // the first chunk starts at a true boundary: a real error
                o.clear(); o.guessed = false; o.start = o.end = hi;   // a wrong guess ran into garbage: re-done below from the true position
            }
        });
        // ---- sequential verification (exact): chunk c's guess is accepted only if the TRUE chain of chunk c-1 ends
        //      exactly on it; otherwise the chunk is redone from the true position.  A wrong guess costs time, never correctness.
        size_t truth = pos_;
        size_t total = 0;
        bool stop = false;
        size_t used = 0;                                         // chunks [0, used) carry records of this round
        for (size_t c = 0; c < n_ch && !stop; ++c) {
            ChunkOut &o = chunks[c];
            const size_t lo = pos_ + c * CH, hi = std::min(end, lo + CH);
            used = c + 1;
            if (truth >= hi) { o.clear(); o.start = o.end = truth; continue; }     // a long record spans the chunk
            if (!(o.guessed && o.start == truth)) {
                o.clear(); o.start = truth;
                o.end = frame_and_parse(bufp, truth, hi, end, tags, o);
            }
            (void)lo;
            if (total + o.size() >= want) {                                        // the batch ends inside this chunk
                const size_t keep = want - total;
                if (keep < o.size()) {
                    // the kept records end where record `keep` starts: hop there from the chunk's start
                    size_t q = o.start;
                    for (size_t k = 0; k < keep; ++k) q += 4 + (size_t)le32(bufp + q);
                    o.end = q;
                    o.truncate(keep);
                }
                stop = true;
            }
            total += o.size();
            truth = o.end;
            if (o.end < hi && !stop) break;                                        // incomplete record: the window ends here
        }
        for (size_t d = used; d < n_ch; ++d) chunks[d].clear();
        g_t_frame_rec += now_s() - tf;
        if (total == 0) {
            size_t need = 4;
            if (truth + 4 <= end) need = 4 + (size_t)le32(bufp + truth);
            if (truth != pos_) throw std::runtime_error("internal framing error");
            if (!fill(need)) throw std::runtime_error("truncated BAM record");
            continue;
        }
        // ---- prefix sums over chunks, then parallel copy to the final place
        const double tp = now_s();
        std::vector<size_t> rec0(used + 1, 0), cig0(used + 1, 0);
        for (size_t c = 0; c < used; ++c) { rec0[c + 1] = rec0[c] + chunks[c].size(); cig0[c + 1] = cig0[c] + chunks[c].cig.size(); }
        const size_t K = rec0[used], base = out.core.size(), cig_base = out.cigar.size();
        if (cig_base + cig0[used] > 0x3FFFFFF0ull) throw std::runtime_error("batch too large");
        out.core.resize(base + K); out.aux.resize(base + K); out.qh2.resize(base + K); out.cigar.resize(cig_base + cig0[used]);
        rsqc_rec_core *ocore = out.core.data(); rsqc_rec_aux *oaux = out.aux.data(); uint32_t *oqh2 = out.qh2.data(); uint32_t *ocig = out.cigar.data();
        pool_->run(used, [&](size_t c) {
            const ChunkOut &o = chunks[c];
            const size_t m = o.size();
            if (!m) return;
            const uint32_t shift = (uint32_t)(cig_base + cig0[c]);
            rsqc_rec_core *dc = ocore + base + rec0[c];
            for (size_t k = 0; k < m; ++k) { rsqc_rec_core v = o.core[k]; v.cigar_off += shift; dc[k] = v; }
            memcpy(oaux + base + rec0[c], o.aux.data(), m * sizeof(rsqc_rec_aux));
            memcpy(oqh2 + base + rec0[c], o.qh2.data(), m * sizeof(uint32_t));
            if (!o.cig.empty()) memcpy(ocig + cig_base + cig0[c], o.cig.data(), o.cig.size() * sizeof(uint32_t));
        });
        g_t_parse += now_s() - tp;
        const double tg = now_s();
        bool have_prev = !out.seg_tid.empty(); int32_t ptid = out.seg_tid.empty() ? 0 : out.seg_tid.back();
        for (size_t c = 0; c < used; ++c) {
            const ChunkOut &o = chunks[c];
            const uint64_t g0 = (uint64_t)(base + rec0[c]);
            for (size_t si = 0; si < o.segs.size(); ++si) {
                const auto &sg = o.segs[si];
                if (si == 0 && have_prev && ptid == sg.second) continue;            // the chunk continues the previous contig
                out.seg_tid.push_back(sg.second); out.seg_start.push_back(g0 + sg.first);
            }
            if (o.size()) { have_prev = true; ptid = o.last_tid; }
            for (auto &nm_ : o.bad_ref) if (out.bad_refid.size() < 64) out.bad_refid.push_back(nm_.second);
            if (o.have_q) {
                if (o.unsorted || (have_q_ && q_tid_ == o.first_tid && q_pos_ > o.first_pos)) out.unsorted = true;
                have_q_ = true; q_tid_ = o.q_tid; q_pos_ = o.q_pos;
            }
            for (size_t w = 0; w < o.widx.size(); ++w) {
                out.wide_index.push_back(g0 + o.widx[w]); out.wide_nm.push_back(o.wnm[w]); out.wide_lq.push_back(o.wlq[w]); out.wide_ncig.push_back(o.wnc[w]);
            }
        }
        g_t_merge += now_s() - tg;
        pos_ = truth;
        n += K; n_read_ += K;
    }
    return n;
}

}  // namespace rsqc_host
