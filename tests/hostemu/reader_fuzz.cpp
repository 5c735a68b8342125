// reader_fuzz.cpp -- TEST HARNESS ONLY (never loaded by the product).
//
// The host BAM reader (rnaseqc_amd/csrc/host/bam.cpp: the RSQC_DECODE=host path of the CLI and the reader behind
// rnaseqc_amd/bamio.py) on files whose RECORDS are damaged behind valid BGZF blocks (the CRC-32 of a block only vouches for
// what the writer compressed), built with -fsanitize=address,undefined: the reader frames in parallel from guessed record
// starts and verifies the chain, so what is searched for here is a read or write outside its buffers, a loop that does
// not end, a thread that does not come back.  A damaged file must end in an exception or in a record count, never in a crash.
// A third of the files are damaged in the CONTAINER as well (BGZF headers, BSIZE, trailers, cut files), and every file also goes
// through the device decode's host side (host/bgzf_feed.cpp: block table, read-ahead thread, the CPU threads' share of the inflate).
//
//   reader_fuzz <cases> <seed> <tmp file>      exit 0 = nothing found
#include <signal.h>
#include <unistd.h>
#include <zlib.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <stdexcept>
#include <string>

#include "../../rnaseqc_amd/csrc/host/bam.hpp"
#include "../../rnaseqc_amd/csrc/host/bgzf_feed.hpp"
#include "fuzz_records.h"

static void bgzf_write(FILE *f, const uint8_t *p, size_t n) {              // one block (n <= 60000), zlib level 1; n = 0: the end-of-file marker
    uint8_t comp[70000];
    z_stream zs{};
    if (deflateInit2(&zs, 1, Z_DEFLATED, -15, 8, Z_DEFAULT_STRATEGY) != Z_OK) abort();
    zs.next_in = (Bytef *)p; zs.avail_in = (uInt)n; zs.next_out = comp; zs.avail_out = sizeof comp;
    if (deflate(&zs, Z_FINISH) != Z_STREAM_END) abort();
    const uint32_t clen = (uint32_t)zs.total_out, bsize = clen + 25, crc = (uint32_t)crc32(0, p, (uInt)n);
    deflateEnd(&zs);
    const uint8_t head[18] = {0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 0xff, 6, 0, 'B', 'C', 2, 0, (uint8_t)bsize, (uint8_t)(bsize >> 8)};
    fwrite(head, 1, 18, f); fwrite(comp, 1, clen, f);
    const uint32_t tail[2] = {crc, (uint32_t)n};
    fwrite(tail, 4, 2, f);
}

static const char *g_what = "";
static long g_case = -1;
static void on_alarm(int) {
    char msg[160];
    const int k = snprintf(msg, sizeof msg, "reader_fuzz: case %ld (%s) did not end\n", g_case, g_what);
    if (write(2, msg, (size_t)k) < 0) {}
    _exit(3);
}

int main(int argc, char **argv) {
    const long cases = argc > 1 ? atol(argv[1]) : 300;
    g_state = argc > 2 ? strtoull(argv[2], nullptr, 10) * 2 + 1 : 1;
    const std::string path = argc > 3 ? argv[3] : "/tmp/reader_fuzz.bam";
    signal(SIGALRM, on_alarm);
    long read_ok = 0, refused = 0, clean = 0;
    std::vector<uint8_t> base; uint64_t base_records = 0; int32_t n_ref = 3;
    for (long c = 0; c < cases; ++c) {
        g_case = c;
        if (c % 6 == 0) {
            n_ref = 1 + (int32_t)rnd(rnd(3) ? 4 : 300);
            base.clear(); base_records = 0;
            const uint32_t want = 2000 + rnd(rnd(5) ? 150000 : 1200000);
            while (base.size() < want) { put_record(base, n_ref, rnd(400) == 0); ++base_records; }
        }
        std::vector<uint8_t> w = base;
        auto at = [&]() { return rnd((uint32_t)w.size()); };
        bool damaged = true;
        switch (rnd(10)) {
        case 0: g_what = "clean"; damaged = false; break;
        case 1: g_what = "bit flips"; for (uint32_t k = 0, n = 1 + rnd(20); k < n; ++k) w[at()] ^= (uint8_t)(1u << rnd(8)); break;
        case 2: g_what = "bytes overwritten"; for (uint32_t k = 0, p = at(), n = 1 + rnd(64); k < n && p + k < w.size(); ++k) w[p + k] = (uint8_t)rnd(256); break;
        case 3: g_what = "cut short"; w.resize(rnd((uint32_t)w.size())); break;
        case 4: g_what = "garbage"; for (auto &b : w) b = (uint8_t)rnd(256); break;
        case 5: g_what = "zeros"; for (uint32_t k = 0, p = at(), n = 1 + rnd(20000); k < n && p + k < w.size(); ++k) w[p + k] = 0; break;
        case 6: g_what = "0xff"; for (uint32_t k = 0, p = at(), n = 1 + rnd(20000); k < n && p + k < w.size(); ++k) w[p + k] = 0xff; break;
        case 7: {   g_what = "a block_size field damaged";
            uint64_t p = 0; for (uint32_t hops = rnd(300); hops && p + 4 <= w.size(); --hops) { uint32_t bs; memcpy(&bs, &w[p], 4); if (p + 4 + bs + 4 > w.size()) break; p += 4 + (uint64_t)bs; }
            if (p + 4 <= w.size()) { const uint32_t v = rnd(4) == 0 ? rnd(40) : rnd(3) == 0 ? 0xFFFFFFF0u + rnd(16) : rnd(1u << (1 + rnd(27))); memcpy(&w[p], &v, 4); }
            break; }
        case 8: {   g_what = "l_name / n_cigar / l_seq damaged";
            uint64_t p = 0; for (uint32_t hops = rnd(300); hops && p + 4 <= w.size(); --hops) { uint32_t bs; memcpy(&bs, &w[p], 4); if (p + 4 + bs + 4 > w.size()) break; p += 4 + (uint64_t)bs; }
            if (p + 36 <= w.size()) { const uint32_t f = rnd(3); if (f == 0) w[p + 12] = (uint8_t)rnd(256); else if (f == 1) { w[p + 16] = (uint8_t)rnd(256); w[p + 17] = (uint8_t)rnd(256); } else { const uint32_t v = rnd(2) ? rnd() : 0x80000000u + rnd(100); memcpy(&w[p + 20], &v, 4); } }
            break; }
        default: g_what = "a slice of another place"; { const uint32_t n = 1 + rnd(3000), from = at(), to = at(); for (uint32_t k = 0; k < n && from + k < w.size() && to + k < w.size(); ++k) w[to + k] = w[from + k]; } break;
        }
        // header + records -> BGZF blocks of random sizes
        std::vector<uint8_t> file;
        file.insert(file.end(), {'B', 'A', 'M', 1}); put32(file, 0); put32(file, (uint32_t)n_ref);
        for (int32_t i = 0; i < n_ref; ++i) { const std::string nm = "c" + std::to_string(i); put32(file, (uint32_t)nm.size() + 1); file.insert(file.end(), nm.begin(), nm.end()); file.push_back(0); put32(file, 1u << 28); }
        if (rnd(40) == 0 && file.size() > 8) { file[rnd((uint32_t)file.size())] ^= 0x10; damaged = true; }         // now and then the header itself
        file.insert(file.end(), w.begin(), w.end());
        FILE *f = fopen(path.c_str(), "wb");
        if (!f) { perror("reader_fuzz"); return 2; }
        for (size_t o = 0; o < file.size();) { const size_t n = std::min<size_t>(file.size() - o, 1 + rnd(rnd(4) ? 60000 : 3000)); bgzf_write(f, file.data() + o, n); o += n; }
        bgzf_write(f, nullptr, 0);
        fclose(f);
        if (rnd(3) == 0) {                                                                  // the container: bytes of the file itself
            damaged = true;
            FILE *g = fopen(path.c_str(), "rb"); std::vector<uint8_t> raw; uint8_t tmp[65536]; size_t got;
            while ((got = fread(tmp, 1, sizeof tmp, g)) > 0) raw.insert(raw.end(), tmp, tmp + got);
            fclose(g);
            switch (rnd(4)) {
            case 0: for (uint32_t k = 0, n = 1 + rnd(6); k < n; ++k) raw[rnd((uint32_t)raw.size())] ^= (uint8_t)(1u << rnd(8)); break;
            case 1: raw.resize(rnd((uint32_t)raw.size())); break;
            case 2: { size_t p = 0; for (uint32_t hops = rnd(30); hops && p + 18 <= raw.size(); --hops) p += (size_t)(raw[p + 16] | (raw[p + 17] << 8)) + 1;     // a block header field
                      if (p + 18 <= raw.size()) raw[p + rnd(18)] = (uint8_t)rnd(256); break; }
            default: { size_t p = 0; for (uint32_t hops = rnd(30); hops && p + 18 <= raw.size(); --hops) p += (size_t)(raw[p + 16] | (raw[p + 17] << 8)) + 1;   // a trailer (CRC-32 / ISIZE)
                       const size_t e = p + 18 <= raw.size() ? p + (size_t)(raw[p + 16] | (raw[p + 17] << 8)) + 1 : 0; if (e >= 8 && e <= raw.size()) raw[e - 1 - rnd(8)] ^= (uint8_t)(1u << rnd(8)); break; }
            }
            g = fopen(path.c_str(), "wb"); if (!raw.empty()) fwrite(raw.data(), 1, raw.size(), g); fclose(g);
        }
        alarm(120);
        uint64_t n = 0; bool threw = false;
        try {
            rsqc_host::BamReader r;
            r.set_threads(2 + (int)rnd(6));
            if (!r.open(path)) threw = true;
            else {
                r.set_tags("ch", {"XF"});
                for (;;) { rsqc_host::HostBatch b; const size_t k = r.read_batch(b, 1 + rnd(rnd(3) ? 5000 : 50)); if (!k) break; n += k; (void)b.view(); }
            }
        } catch (const std::exception &) { threw = true; }
        // the same file through the feeder of the device decode
        uint64_t fed_blocks = 0; bool feeder_threw = false;
        try {
            rsqc_host::BgzfFeeder fd;
            if (!fd.open(path)) feeder_threw = true;
            else {
                std::vector<std::string> names;
                const uint64_t v = fd.first_record_voffset(&names);
                if (rnd(2)) fd.set_cpu_share(1 + (int)rnd(3), 0.3, 0.6, 1u << 20);
                fd.start(v, 0, (size_t)1 << (17 + rnd(4)), (uint64_t)1 << (18 + rnd(6)));
                while (auto *ch = fd.next()) fed_blocks += ch->blocks.size();
            }
        } catch (const std::exception &) { feeder_threw = true; }
        alarm(0);
        if (!damaged && (feeder_threw || fed_blocks == 0)) { fprintf(stderr, "reader_fuzz: case %ld: the feeder refused an undamaged file\n", c); return 1; }
        if (!damaged) {
            if (threw || n != base_records) { fprintf(stderr, "reader_fuzz: case %ld: an undamaged file of %llu records read as %llu%s\n", c, (unsigned long long)base_records, (unsigned long long)n, threw ? " and refused" : ""); return 1; }
            ++clean;
        }
        if (threw) ++refused; else ++read_ok;
    }
    unlink(path.c_str());
    printf("reader_fuzz: %ld cases: %ld files read to the end (%ld of them undamaged), %ld refused\n", cases, read_ok, clean, refused);
    return 0;
}
