// bam.hpp -- BGZF/BAM decode straight into the boundary's record batches (no htslib in this image).
// Replaces the SeqlibReader adapter (src/BamReader.{h,cpp}) for BAM input; CRAM is out of scope.
#pragma once

#include <atomic>
#include <condition_variable>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <functional>
#include <mutex>
#include <string>
#include <thread>
#include <utility>
#include <vector>

#include "../../../include/rnaseqc_amd.h"

namespace rsqc_host {

// growable array of trivially copyable T whose new elements are NOT value-initialised (the decode threads
// write every element; a zero-fill of hundreds of MB on one thread would be the bottleneck)
template <class T>
class RawVec {
public:
    RawVec() = default;
    RawVec(const RawVec &) = delete;
    RawVec &operator=(const RawVec &) = delete;
    ~RawVec() { release(p_, pinned_now_); }
    // page-locked storage (rsqc_host_alloc): rsqc_submit then copies by DMA without a staging pass.  Falls back to
    // ordinary memory when no device is present.  Call before the first growth.
    void use_pinned(bool on) { want_pinned_ = on; }
    void reserve(size_t n) { if (n > cap_) grow(n); }
    T *data() { return p_; }
    const T *data() const { return p_; }
    size_t size() const { return n_; }
    bool empty() const { return n_ == 0; }
    void clear() { n_ = 0; }
    T &operator[](size_t i) { return p_[i]; }
    const T &operator[](size_t i) const { return p_[i]; }
    void resize(size_t n) {
        if (n > cap_) {
            size_t c = cap_ ? cap_ : 1024;
            while (c < n) c += c / 2 + 1024;
            grow(c);
        }
        n_ = n;
    }
    void push_back(const T &v) { resize(n_ + 1); p_[n_ - 1] = v; }
    void swap(RawVec &o) { std::swap(p_, o.p_); std::swap(n_, o.n_); std::swap(cap_, o.cap_); std::swap(want_pinned_, o.want_pinned_); std::swap(pinned_now_, o.pinned_now_); }
    // drop the first k elements
    void erase_front(size_t k) { if (k >= n_) { n_ = 0; return; } memmove(p_, p_ + k, (n_ - k) * sizeof(T)); n_ -= k; }
private:
    static void release(T *p, bool pinned) { if (pinned) rsqc_host_free(p); else free(p); }
    void grow(size_t c) {
        T *q = nullptr; bool pin = false;
        if (want_pinned_) { q = (T *)rsqc_host_alloc(c * sizeof(T)); pin = q != nullptr; }
        if (!q) {
            if (!pinned_now_) { q = (T *)realloc(p_, c * sizeof(T)); if (!q) throw std::bad_alloc(); p_ = q; cap_ = c; return; }
            q = (T *)malloc(c * sizeof(T));
            if (!q) throw std::bad_alloc();
        }
        if (n_) memcpy(q, p_, n_ * sizeof(T));
        release(p_, pinned_now_);
        p_ = q; cap_ = c; pinned_now_ = pin;
    }
    T *p_ = nullptr;
    size_t n_ = 0, cap_ = 0;
    bool want_pinned_ = false, pinned_now_ = false;
};

struct HostBatch {                       // owns the arrays an rsqc_batch points to
    RawVec<rsqc_rec_core> core;
    RawVec<rsqc_rec_aux> aux;
    RawVec<uint32_t> qh2;                // second name hashes (rsqc_batch.qhash2)
    RawVec<uint32_t> cigar;
    std::vector<int32_t> seg_tid;
    std::vector<uint64_t> seg_start;
    std::vector<uint64_t> wide_index;
    std::vector<int32_t> wide_nm, wide_lq;
    std::vector<uint32_t> wide_ncig;
    uint64_t file_index_base = 0;
    // diagnostics of the decode (what the reference's loop prints on stderr): positions going backwards inside a contig
    // (src/RNASeQC.cpp:354-355) and names of records whose RefID the header does not define (:333-337)
    bool unsorted = false;
    std::vector<std::string> bad_refid;
    void clear();
    // keep only the leading records of contig `tid` (by-contig reading: the block that ends a contig's range may also
    // hold the first records of the next one); returns true when something was cut off
    bool keep_leading_contig(int32_t tid);
    size_t size() const { return core.size(); }
    rsqc_batch view();                   // closes the segment table
};

// CPUs the process may use: affinity mask capped by the cgroup CPU quota
int effective_cpus();
// one BGZF block's payload -> exactly out_len bytes, CRC-32 checked (libdeflate when the system has it, zlib otherwise)
bool bgzf_inflate_block(const uint8_t *in, size_t in_len, uint8_t *out, size_t out_len, uint32_t crc);

// fork-join helper: run(n, fn) calls fn(task) for task in [0, n) on the pool's threads and the caller
class WorkPool {
public:
    explicit WorkPool(int threads);
    ~WorkPool();
    int size() const { return (int)workers_.size() + 1; }
    void run(size_t n_tasks, const std::function<void(size_t)> &fn);
private:
    void worker();
    std::vector<std::thread> workers_;
    std::mutex mu_;
    std::condition_variable cv_work_, cv_done_;
    const std::function<void(size_t)> *fn_ = nullptr;
    size_t n_tasks_ = 0;
    std::atomic<size_t> next_{0};
    size_t running_ = 0;
    uint64_t epoch_ = 0;
    bool stop_ = false;
    std::string error_;
};

class BamReader {
public:
    bool open(const std::string &path);                 // false: cannot open / not a BAM
    const std::vector<std::string> &contigs() const { return names_; }
    // tags: the chimeric tag (2 chars) and up to RSQC_MAX_FILTER_TAGS filter tags
    void set_tags(const std::string &chimeric, const std::vector<std::string> &filters);
    // appends up to max_records records to `out`; returns the number appended (0 at EOF)
    size_t read_batch(HostBatch &out, size_t max_records);
    uint64_t records_read() const { return n_read_; }
    // decode threads in total (BGZF inflate + record parsing, split 60 / 40); default: RSQC_HOST_THREADS or twice the CPUs
    // the process may use (effective_cpus(): affinity capped by the cgroup quota).
    // Call BEFORE open(): open() starts the decode pipeline to read the header and the pools are fixed from then on
    // (returns false, and changes nothing, afterwards).
    bool set_threads(int n);
    // by-contig reading: the BAM index's per-reference ranges and a restart of the stream at a virtual offset
    struct ContigRange { uint64_t beg = 0, end = 0, n_records = 0; bool present = false; };
    bool load_index(const std::string &bai_path);
    const std::vector<ContigRange> &index() const { return index_; }
    uint64_t unplaced_records() const { return n_no_coor_; }
    bool seek(uint64_t voff, uint64_t voff_end);
    int inflate_threads() const { return pool_inflate_ ? pool_inflate_->size() : 0; }
    int parse_threads() const { return pool_ ? pool_->size() : 0; }
    ~BamReader();
private:
    bool fill(size_t need);              // make at least `need` decompressed bytes available
    bool fill_group();                   // swap in the next inflated group (prepared by the producer thread) behind the unread tail
    bool produce_group(RawVec<uint8_t> &dst, size_t head);   // producer side: frame + inflate one group at dst[head..)
    void producer_main();
    void start_producer();
    FILE *fp_ = nullptr;
    RawVec<uint8_t> buf_;                // decompressed stream window
    size_t pos_ = 0;
    RawVec<uint8_t> cbuf_;               // compressed window: [cpos_, cbuf_.size()) is unread (streams that cannot be mapped)
    const uint8_t *map_ = nullptr;       // the whole compressed file, memory-mapped (regular files): inflate reads the page cache directly
    size_t map_size_ = 0;
    size_t cpos_ = 0;
    size_t climit_ = 0;                  // seek(): compressed offset behind which nothing is framed (0 = none)
    size_t skip_ = 0;                    // seek(): bytes of the first inflated block that precede the wanted record
    std::vector<ContigRange> index_;
    uint64_t n_no_coor_ = 0;
    bool file_eof_ = false;
    bool eof_ = false;
    WorkPool *pool_ = nullptr;           // record framing + parsing (consumer side)
    WorkPool *pool_inflate_ = nullptr;   // BGZF inflate (producer thread): the next group is inflated while this one is parsed
    int n_threads_ = 0;
    // hand-over of one inflated group: the producer fills stage_[kHead, kHead + stage_bytes_), the consumer copies its
    // unread tail in front of it (into the head room) and swaps the buffers
    static constexpr size_t kHead = (size_t)4 << 20;
    RawVec<uint8_t> stage_;
    size_t stage_bytes_ = 0;
    enum StageState { kEmpty, kReady, kEof, kError } stage_state_ = kEmpty;
    std::string stage_error_;
    bool stop_ = false, producer_started_ = false;
    std::thread producer_;
    std::mutex mu_;
    std::condition_variable cv_;
    std::vector<std::string> names_;
    std::string ch_tag_ = "ch";
    std::vector<std::string> filter_tags_;
    uint64_t n_read_ = 0;
    bool have_q_ = false; int32_t q_tid_ = 0, q_pos_ = 0;      // last primary mapped record seen (sort check across batches)
};

}  // namespace rsqc_host
