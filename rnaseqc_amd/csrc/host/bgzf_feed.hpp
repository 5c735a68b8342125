// bgzf_feed.hpp -- host side of the device BAM decode (rsqc_decode_*): reads the file in large page-locked chunks and
// hops over the BGZF block headers (18 bytes per block); inflating, record framing and parsing happen on the GPU.
// Replaces, together with the device kernels, the reference's SeqlibReader loop (src/BamReader.{h,cpp}) for BAM input.
#pragma once

#include <atomic>
#include <cstdint>
#include <condition_variable>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../../include/rnaseqc_amd.h"

namespace rsqc_host {

class BgzfFeeder {
public:
    struct Chunk {
        uint8_t *data = nullptr; size_t cap = 0, bytes = 0; bool pinned = false;   // page-locked when a device is present
        std::vector<rsqc_bgzf_block> blocks;
        size_t total_bytes = 0;                                    // file bytes + the inflated bytes of the CPU's share behind them: what the GPU call uploads
        uint32_t skip = 0;                                         // inflated bytes of the first block that precede the first record
        uint64_t limit = 0;                                        // 0, or the inflated offset (from the chunk's first block) where the range ends
        bool last = false;                                         // the range / file ends with this chunk
    };
    BgzfFeeder() = default;
    ~BgzfFeeder();
    BgzfFeeder(const BgzfFeeder &) = delete;
    BgzfFeeder &operator=(const BgzfFeeder &) = delete;

    bool open(const std::string &path);
    // Virtual offset (coffset << 16 | uoffset) of the first alignment record: inflates the header's blocks on the host.
    // names (may be null): the reference sequence names of the header.  Throws std::runtime_error on a file that is not a BAM.
    uint64_t first_record_voffset(std::vector<std::string> *names = nullptr);
    // page-locks the three chunk buffers now (else: by the read-ahead thread when it first fills them -- page-locking takes
    // the HIP runtime's lock, and the thread that feeds the GPU stalls behind it)
    // max_out (0 = unknown): the inflated bytes a call holds at most -- with it the buffers are sized for what that takes in THIS file
    // (its compression sampled from the first megabytes) instead of for chunk_bytes; a later stretch of the file that compresses
    // less then makes smaller calls, not larger buffers
    void reserve(size_t chunk_bytes, uint64_t max_out = 0);
    // the blocks from virtual offset `beg` to `end` (0 = end of file); starts the read-ahead thread
    void start(uint64_t voff_beg, uint64_t voff_end, size_t chunk_bytes = (size_t)128 << 20, uint64_t max_out = (uint64_t)1024 << 20);
    // next chunk, or nullptr at the end; the previous chunk becomes reusable.  Throws on a malformed block header.
    Chunk *next();
    uint64_t file_size() const { return file_size_; }
    int read_threads = 4;
    // CPU share of the inflate work (device decode): the last blocks of every chunk are inflated here, by `threads` spare CPU
    // threads (libdeflate), and handed over as RSQC_BGZF_INFLATED; the share follows the consumer -- it grows while
    // next() finds a chunk waiting and shrinks when the consumer had to wait.  0 threads (default) = everything to the GPU.
    // call_out_bytes: the inflated bytes of the largest call (start()'s max_out): sizes the room behind the file bytes.
    void set_cpu_share(int threads, double initial_share = 0.15, double max_share = 0.5, uint64_t call_out_bytes = (uint64_t)1 << 30);
    double cpu_share() const { return share_.load(); }

private:
    void producer();
    bool fill(Chunk &c);
    int fd_ = -1; uint64_t file_size_ = 0;
    uint64_t cpos_ = 0, cend_ = 0; uint32_t skip_ = 0, uend_ = 0; bool done_ = false, has_end_ = false;
    size_t chunk_bytes_ = 0; uint64_t max_out_ = 0; int n_filled_ = 0;
    double ratio_ = 0;                 // inflated / file bytes of the chunk before (0: none yet)
    bool reserved_ = false;            // reserve() sized the buffers: fill() stays inside them
    double sample_ratio();             // inflated / file bytes over the first megabytes (0: unreadable)
    Chunk ring_[3];
    int head_ = 0, tail_ = 0, count_ = 0; Chunk *lent_ = nullptr;
    bool eof_ = false, stop_ = false; std::string error_;
    int cpu_threads_ = 0; double max_share_ = 0.5; size_t raw_cap_ = 0;
    std::atomic<double> share_{0.0};                                 // written by next() (the consumer), read by the read-ahead thread
    void *pool_ = nullptr;                                           // WorkPool of the CPU share
    std::mutex mu_; std::condition_variable cv_;
    std::thread th_;
};

}  // namespace rsqc_host
