#include "bgzf_feed.hpp"
#include "bam.hpp"

#include <fcntl.h>
#include <sys/stat.h>
#include <unistd.h>
#include <zlib.h>

#include <algorithm>
#include <atomic>
#include <cstdlib>
#include <cstring>
#include <stdexcept>

namespace rsqc_host {

namespace {
inline uint32_t le16(const uint8_t *p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8); }
inline uint32_t le32(const uint8_t *p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24); }

// BSIZE of the block whose header starts at h (avail bytes readable): 0 = not enough bytes yet; throws on a bad header
uint32_t block_size(const uint8_t *h, size_t avail) {
    if (avail < 18) return 0;
    if (h[0] != 0x1f || h[1] != 0x8b || h[2] != 8 || !(h[3] & 4)) throw std::runtime_error("not a BGZF block (bad gzip header)");
    const uint32_t xlen = le16(h + 10);
    if (avail < 12u + xlen) return 0;
    uint32_t bsize = 0;
    for (size_t o = 0; o + 4 <= xlen;) {
        const uint8_t *x = h + 12 + o;
        const uint32_t slen = le16(x + 2);
        if (x[0] == 'B' && x[1] == 'C' && slen == 2 && o + 6 <= xlen) bsize = le16(x + 4) + 1;
        o += 4 + (size_t)slen;
    }
    if (!bsize) throw std::runtime_error("BGZF block without BC field");
    if (bsize < 12u + xlen + 8u) throw std::runtime_error("bad BGZF block size");
    return bsize;
}
}  // namespace

void BgzfFeeder::set_cpu_share(int threads, double initial_share, double max_share, uint64_t call_out_bytes) {
    cpu_threads_ = std::max(0, threads);
    max_share_ = std::min(0.9, std::max(0.0, max_share));
    share_ = cpu_threads_ ? std::min(max_share_, std::max(0.0, initial_share)) : 0.0;
    raw_cap_ = cpu_threads_ ? (size_t)(max_share_ * (double)call_out_bytes) + (1u << 20) : 0;             // the CPU's share of the largest call, behind the file bytes
    if (cpu_threads_ && !pool_) pool_ = new WorkPool(cpu_threads_);
}

BgzfFeeder::~BgzfFeeder() {
    { std::lock_guard<std::mutex> lk(mu_); stop_ = true; }
    cv_.notify_all();
    if (th_.joinable()) th_.join();
    delete (WorkPool *)pool_;
    for (auto &c : ring_) if (c.data) { if (c.pinned) rsqc_host_free(c.data); else free(c.data); }
    if (fd_ >= 0) close(fd_);
}

bool BgzfFeeder::open(const std::string &path) {
    fd_ = ::open(path.c_str(), O_RDONLY);
    if (fd_ < 0) return false;
    struct stat st;
    if (fstat(fd_, &st) != 0 || !S_ISREG(st.st_mode)) return false;
    file_size_ = (uint64_t)st.st_size;
    return true;
}

uint64_t BgzfFeeder::first_record_voffset(std::vector<std::string> *names) {
    std::vector<uint8_t> raw, text;
    std::vector<std::pair<uint64_t, uint64_t>> blocks;                 // (file offset, inflated bytes before the block)
    uint64_t fpos = 0;
    auto need = [&](size_t n) {                                         // inflate blocks until `text` holds n bytes
        while (text.size() < n) {
            uint8_t head[18 + 65536];
            const ssize_t got = pread(fd_, head, sizeof head, (off_t)fpos);
            if (got < 18) throw std::runtime_error("truncated BAM header");
            const uint32_t bs = block_size(head, (size_t)got);
            if (!bs || bs > (uint32_t)got) throw std::runtime_error("truncated BAM header");
            const uint32_t xlen = le16(head + 10), isize = le32(head + bs - 4);
            if (isize > 65536u) throw std::runtime_error("BGZF block with ISIZE above 64 KiB (corrupt trailer)");
            blocks.emplace_back(fpos, (uint64_t)text.size());
            const size_t at = text.size();
            text.resize(at + isize);
            z_stream zs{};
            if (inflateInit2(&zs, -15) != Z_OK) throw std::runtime_error("zlib init failed");
            zs.next_in = head + 12 + xlen; zs.avail_in = bs - xlen - 20;
            zs.next_out = text.data() + at; zs.avail_out = isize;
            const int rc = inflate(&zs, Z_FINISH);
            inflateEnd(&zs);
            if (rc != Z_STREAM_END || zs.avail_out != 0) throw std::runtime_error("BGZF inflate failed (corrupt block)");
            fpos += bs;
        }
    };
    need(12);
    if (memcmp(text.data(), "BAM\1", 4) != 0) throw std::runtime_error("not a BAM file");
    const uint32_t l_text = le32(text.data() + 4);
    need(12 + (size_t)l_text);
    const uint32_t n_ref = le32(text.data() + 8 + l_text);
    size_t p = 12 + (size_t)l_text;
    for (uint32_t i = 0; i < n_ref; ++i) {
        need(p + 4);
        const uint32_t l_name = le32(text.data() + p);
        need(p + 8 + (size_t)l_name);
        if (names) names->emplace_back((const char *)text.data() + p + 4, l_name ? l_name - 1 : 0);
        p += 8 + (size_t)l_name;
    }
    // the block that holds inflated offset p (the next block when the header ends with its block)
    for (size_t b = blocks.size(); b-- > 0;)
        if (blocks[b].second <= p) {
            const uint64_t in_block = p - blocks[b].second;
            const uint64_t block_len = (b + 1 < blocks.size() ? blocks[b + 1].second : (uint64_t)text.size()) - blocks[b].second;
            if (in_block == block_len) return fpos << 16;                // (b is the last block inflated)
            return (blocks[b].first << 16) | in_block;
        }
    return 0;
}

// inflated bytes per file byte, the SMALLEST of three windows of the file (head, middle, tail: a file whose head compresses much better
// than its body -- a header-heavy or low-complexity start -- would otherwise size the buffers for calls that the body cannot fill).
// A window inside the file starts at the first position where two consecutive BGZF block headers parse.
double BgzfFeeder::sample_ratio() {
    if (fd_ < 0 || !file_size_) return 0;
    const size_t win = (size_t)std::min<uint64_t>(file_size_, (uint64_t)3 << 20);
    std::vector<uint8_t> buf(win);
    double best = 0;
    const uint64_t starts[3] = {0, file_size_ > 3 * (uint64_t)win ? file_size_ / 2 : 0, file_size_ > 3 * (uint64_t)win ? file_size_ - win : 0};
    for (int w = 0; w < 3; ++w) {
        if (w && starts[w] == 0) break;
        size_t got = 0;
        while (got < win) { const ssize_t g = pread(fd_, buf.data() + got, win - got, (off_t)(starts[w] + got)); if (g <= 0) break; got += (size_t)g; }
        size_t p = 0;
        if (w) {                                     // find a block boundary: two headers in a row
            bool found = false;
            for (; p + 64 < got; ++p) {
                if (buf[p] != 0x1f || buf[p + 1] != 0x8b || buf[p + 2] != 8 || !(buf[p + 3] & 4)) continue;     // (no exception per byte)
                uint32_t bs = 0;
                try { bs = block_size(buf.data() + p, got - p); } catch (std::exception &) { bs = 0; }
                if (!bs || p + bs + 28 > got) continue;
                uint32_t bs2 = 0;
                try { bs2 = block_size(buf.data() + p + bs, got - p - bs); } catch (std::exception &) { bs2 = 0; }
                if (bs2) { found = true; break; }
            }
            if (!found) continue;
        }
        uint64_t in = 0, out = 0;
        try {
            while (p < got) {
                const uint32_t bs = block_size(buf.data() + p, got - p);
                if (!bs || p + bs > got) break;
                in += bs; out += le32(buf.data() + p + bs - 4); p += bs;
            }
        } catch (std::exception &) { if (!w) return 0; }
        if (in > 65536) {
            const double r = (double)out / (double)in;
            if (best == 0 || r < best) best = r;
        }
    }
    return best;
}

// Page-locked footprint: the ring's three buffers of `cap` bytes each, cap <= chunk_bytes (the command line: 512 MB -> at most 1.5 GB, reached
// only by a file that compresses less than 2 x; a file of a real BAM's entropy, 3 x: 3 x 390 MB) + the CPU share's room when it is on.
void BgzfFeeder::reserve(size_t chunk_bytes, uint64_t max_out) {
    size_t cap = (size_t)std::min<uint64_t>(std::max<size_t>(chunk_bytes, (size_t)1 << 17), std::max<uint64_t>(file_size_, (uint64_t)1 << 17));
    if (max_out) {
        const double r = sample_ratio();
        if (r > 0) cap = std::min(cap, std::max<size_t>((size_t)((double)max_out / r * 1.15) + ((size_t)4 << 20), (size_t)1 << 17));
    }
    if (cpu_threads_) cap += raw_cap_ + 256;
    for (auto &c : ring_) {
        if (c.data && c.cap >= cap) continue;
        if (c.data) { if (c.pinned) rsqc_host_free(c.data); else free(c.data); }
        c.data = (uint8_t *)rsqc_host_alloc(cap + 64); c.pinned = c.data != nullptr;
        if (!c.data) c.data = (uint8_t *)malloc(cap + 64);
        if (!c.data) throw std::bad_alloc();
        c.cap = cap;
    }
    reserved_ = max_out != 0;           // (without it the buffers may still grow with the ramp, as in a sharded run's per-contig ranges)
}

void BgzfFeeder::start(uint64_t voff_beg, uint64_t voff_end, size_t chunk_bytes, uint64_t max_out) {
    // the producer of the previous range (a sharded run starts one range per contig on the same feeder) is told to stop and
    // joined BEFORE any of the state it reads is reset
    if (th_.joinable()) {
        { std::lock_guard<std::mutex> lk(mu_); stop_ = true; }
        cv_.notify_all();
        th_.join();
    }
    cpos_ = voff_beg >> 16; skip_ = (uint32_t)(voff_beg & 0xffff);
    cend_ = voff_end >> 16; uend_ = (uint32_t)(voff_end & 0xffff);
    has_end_ = voff_end != 0;
    chunk_bytes_ = std::max<size_t>(chunk_bytes, (size_t)1 << 17); max_out_ = max_out;      // (a chunk holds at least one whole block)
    done_ = cpos_ >= file_size_;
    // (a chunk never needs more than what is left of the file; the buffers are page-locked by the read-ahead thread the first
    //  time it fills them, so that pinning the second one overlaps the GPU's work on the first)
    chunk_bytes_ = (size_t)std::min<uint64_t>(chunk_bytes_, std::max<uint64_t>(file_size_ - std::min(file_size_, cpos_), (uint64_t)1 << 17));
    n_filled_ = 0; ratio_ = 0;
    head_ = tail_ = count_ = 0; lent_ = nullptr; eof_ = false; stop_ = false; error_.clear();
    th_ = std::thread([this] { producer(); });
}

bool BgzfFeeder::fill(Chunk &c) {
    c.blocks.clear(); c.bytes = 0; c.skip = 0; c.limit = 0; c.last = false;
    if (done_) return false;

    // the first chunks are small so that the GPU starts early; later ones are as large as allowed (a call should hold
    // thousands of blocks)
    const size_t ramp = std::min<size_t>(chunk_bytes_, ((size_t)24 << 20) << std::min(n_filled_, 8));
    ++n_filled_;
    // ... and no larger than what max_out_ inflated bytes take in the file at the ratio of the chunk before: what a call cannot
    // hold would be read again by the next one (a 12 x compressed file fills a 1 GB call from 90 MB)
    size_t cap_by_out = chunk_bytes_;
    if (ratio_ > 0) cap_by_out = (size_t)std::min<double>((double)chunk_bytes_, (double)max_out_ / ratio_ * 1.06 + (double)(1u << 20));
    size_t want = (size_t)std::min<uint64_t>(std::min(ramp, std::max<size_t>(cap_by_out, (size_t)1 << 17)), file_size_ - cpos_);
    {   // buffers sized by reserve() are not grown in the loop (page-locking stalls the thread that feeds the GPU): a smaller call instead
        const size_t room = cpu_threads_ ? raw_cap_ + 256 : 0;
        if (reserved_ && c.data && c.cap > room + ((size_t)1 << 17)) want = std::min(want, c.cap - room);
    }
    if (!c.data || c.cap < want + (cpu_threads_ ? raw_cap_ + 256 : 0)) {
        if (c.data) { if (c.pinned) rsqc_host_free(c.data); else free(c.data); }
        const size_t cap = std::min(chunk_bytes_, std::max(want, ramp * 4)) + (cpu_threads_ ? raw_cap_ + 256 : 0);      // (room for the next steps of the ramp)
        c.data = (uint8_t *)rsqc_host_alloc(cap + 64); c.pinned = c.data != nullptr;
        if (!c.data) c.data = (uint8_t *)malloc(cap + 64);               // (no device: the tests of the feeder alone)
        if (!c.data) throw std::bad_alloc();
        c.cap = cap;
    }
    // the slices of one chunk are read side by side (a single pread stream from the page cache is ~3-5 GB/s)
    const int T = std::max(1, std::min(read_threads, (int)(want >> 22) + 1));
    std::vector<std::thread> th;
    std::vector<ssize_t> got((size_t)T, 0);
    const size_t per = (want + (size_t)T - 1) / (size_t)T;
    auto job = [&](int t) {
        size_t off = (size_t)t * per; const size_t end = std::min(want, off + per);
        while (off < end) {
            const ssize_t g = pread(fd_, c.data + off, end - off, (off_t)(cpos_ + off));
            if (g <= 0) { got[(size_t)t] = -1; return; }
            off += (size_t)g;
        }
        got[(size_t)t] = 1;
    };
    for (int t = 1; t < T; ++t) th.emplace_back(job, t);
    job(0);
    for (auto &x : th) x.join();
    for (ssize_t g : got) if (g < 0) throw std::runtime_error("read error on the BAM file");
    size_t p = 0; uint64_t total_out = 0;
    while (p < want) {
        const uint32_t bs = block_size(c.data + p, want - p);
        if (!bs || p + bs > want) break;                               // the chunk ends inside this block
        const uint64_t coff = cpos_ + p;
        if (has_end_ && (coff > cend_ || (coff == cend_ && uend_ == 0))) { done_ = true; break; }
        const uint32_t xlen = le16(c.data + p + 10), isize = le32(c.data + p + bs - 4);
        if (isize > 65536u) throw std::runtime_error("BGZF block with ISIZE above 64 KiB (corrupt trailer)");
        if (total_out + isize > max_out_ && !c.blocks.empty()) break;
        c.blocks.push_back(rsqc_bgzf_block{(uint64_t)p + 12 + xlen, bs - xlen - 20, isize, le32(c.data + p + bs - 8), 0});
        total_out += isize; p += bs;
        if (has_end_ && coff == cend_) { c.limit = total_out - isize + uend_; done_ = true; break; }
    }
    if (c.blocks.empty() && !done_) {
        if (cpos_ + want >= file_size_) throw std::runtime_error("truncated BGZF block");
        throw std::runtime_error("BGZF block larger than the read chunk");
    }
    c.bytes = p; c.total_bytes = p; cpos_ += p;
    if (p) ratio_ = (double)total_out / (double)p;
    // ---- the CPU's share: the last blocks of the chunk, inflated here into the buffer behind the file bytes
    const double share = share_.load();
    if (cpu_threads_ && share > 0 && !c.blocks.empty()) {
        const size_t raw_off = (p + 255) & ~(size_t)255;
        const uint64_t want_raw = (uint64_t)(share * (double)total_out);
        size_t first = c.blocks.size(); uint64_t raw = 0;
        while (first > 0 && raw + c.blocks[first - 1].out_bytes <= want_raw && raw_off + raw + c.blocks[first - 1].out_bytes <= c.cap) { --first; raw += c.blocks[first].out_bytes; }
        if (first < c.blocks.size()) {
            std::vector<uint64_t> at(c.blocks.size() - first + 1, 0);
            for (size_t k = first; k < c.blocks.size(); ++k) at[k - first + 1] = at[k - first] + c.blocks[k].out_bytes;
            std::atomic<bool> bad{false};
            const size_t per_task = 4, n_tasks = (c.blocks.size() - first + per_task - 1) / per_task;
            ((WorkPool *)pool_)->run(n_tasks, [&](size_t t) {
                for (size_t k = first + t * per_task; k < std::min(c.blocks.size(), first + (t + 1) * per_task); ++k) {
                    const rsqc_bgzf_block &b = c.blocks[k];
                    if (!bgzf_inflate_block(c.data + b.in_offset, b.in_bytes, c.data + raw_off + at[k - first], b.out_bytes, b.crc32)) bad = true;
                }
            });
            if (bad) throw std::runtime_error("BGZF inflate failed (corrupt block)");
            for (size_t k = first; k < c.blocks.size(); ++k) {
                c.blocks[k].in_offset = raw_off + at[k - first]; c.blocks[k].in_bytes = c.blocks[k].out_bytes; c.blocks[k].flags = RSQC_BGZF_INFLATED;
            }
            c.total_bytes = raw_off + (size_t)raw;
        }
    }
    c.skip = skip_; skip_ = 0;
    if (cpos_ >= file_size_) done_ = true;
    c.last = done_;
    return !c.blocks.empty() || c.last;
}

void BgzfFeeder::producer() {
    for (;;) {
        Chunk *slot;
        {
            std::unique_lock<std::mutex> lk(mu_);
            cv_.wait(lk, [&] { return stop_ || count_ + (lent_ ? 1 : 0) < 3; });
            if (stop_) return;
            slot = &ring_[head_];
        }
        bool ok = false; std::string err;
        try { ok = fill(*slot); } catch (std::exception &e) { err = e.what(); }
        bool finished;
        {
            std::lock_guard<std::mutex> lk(mu_);
            if (!err.empty()) { error_ = err; eof_ = true; }
            else if (!ok) eof_ = true;
            else { head_ = (head_ + 1) % 3; ++count_; if (slot->last) eof_ = true; }
            finished = eof_ || stop_;                      // (decided under the lock: start() may be resetting the flags next)
        }
        cv_.notify_all();
        if (finished) return;
    }
}

BgzfFeeder::Chunk *BgzfFeeder::next() {
    std::unique_lock<std::mutex> lk(mu_);
    if (lent_) { lent_ = nullptr; cv_.notify_all(); }
    if (cpu_threads_) {                                             // the CPU's share follows the consumer (see set_cpu_share)
        if (count_ > 0) share_ = std::min(max_share_, share_.load() + 0.01);
        else if (!eof_) share_ = std::max(0.0, share_.load() - 0.04);
    }
    cv_.wait(lk, [&] { return count_ > 0 || eof_; });
    if (count_ == 0) {
        if (!error_.empty()) throw std::runtime_error(error_);
        return nullptr;
    }
    Chunk *c = &ring_[tail_];
    tail_ = (tail_ + 1) % 3; --count_;
    lent_ = c;
    cv_.notify_all();
    return c;
}

}  // namespace rsqc_host
