// inflate_fuzz.cpp -- TEST HARNESS ONLY (never loaded by the product).
//
// The product's DEFLATE decoder (rnaseqc_amd/csrc/rsqc_inflate.h, what one wavefront runs per BGZF block) against
// damaged input, built with -fsanitize=address,undefined: whatever the bytes are, the decoder must stay inside the
// payload (+16 bytes it may look ahead), inside the ISIZE bytes of its output and inside its tables, must end, and must
// agree with zlib -- status 0 and the same bytes where zlib inflates the input to exactly ISIZE bytes with the right
// CRC-32; where zlib rejects the input, an error status, or (zlib also rejects code sets whose unused part is
// incomplete, which the decoder does not look at) status 0 with ISIZE bytes that pass the block's CRC-32: never wrong
// bytes.  On the GPU an out-of-bounds store or an endless loop is a dead device, not a wrong answer, and this container
// has no GPU: the host build is where that can be looked for.
//
//   inflate_fuzz <cases> <seed>      exit 0 = every case agreed; the sanitizers abort the process on a finding
#include <signal.h>
#include <unistd.h>
#include <zlib.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../../rnaseqc_amd/csrc/rsqc_inflate.h"

using namespace rsqc;

static uint64_t g_state = 1;
static uint32_t rnd() { g_state = g_state * 6364136223846793005ull + 1442695040888963407ull; return (uint32_t)(g_state >> 33); }
static uint32_t rnd(uint32_t n) { return n ? rnd() % n : 0; }

static std::vector<uint8_t> payload(uint32_t kind, uint32_t n) {
    std::vector<uint8_t> d(n);
    switch (kind) {
    case 0: for (auto &b : d) b = "ACGT"[rnd(4)]; break;                              // low entropy: short codes, many matches
    case 1: for (auto &b : d) b = (uint8_t)rnd(256); break;                           // incompressible: stored or long codes
    case 2: for (uint32_t i = 0; i < n; ++i) d[i] = (uint8_t)((i / 7u) * 31u); break; // runs (distance 1 .. 7)
    case 3: {                                                                          // a phrase repeated at long distances
        std::vector<uint8_t> ph(200 + rnd(3000));
        for (auto &b : ph) b = (uint8_t)rnd(256);
        for (uint32_t i = 0; i < n; ++i) d[i] = (rnd(50) == 0) ? (uint8_t)rnd(256) : ph[i % ph.size()];
        break; }
    default: for (uint32_t i = 0; i < n; ++i) d[i] = (uint8_t)((i & 64u) ? rnd(256) : 'A' + rnd(3)); break;   // mixed
    }
    return d;
}

static std::vector<uint8_t> deflate_raw(const std::vector<uint8_t> &d, int level, int strategy, bool two_blocks) {
    z_stream zs{};
    if (deflateInit2(&zs, level, Z_DEFLATED, -15, 9, strategy) != Z_OK) abort();
    std::vector<uint8_t> out(deflateBound(&zs, (uLong)d.size()) + 64);
    zs.next_out = out.data(); zs.avail_out = (uInt)out.size();
    const size_t half = two_blocks ? d.size() / 2 : d.size();
    zs.next_in = (Bytef *)d.data(); zs.avail_in = (uInt)half;
    if (two_blocks) { if (deflate(&zs, Z_FULL_FLUSH) != Z_OK) abort(); zs.next_in = (Bytef *)d.data() + half; zs.avail_in = (uInt)(d.size() - half); }
    if (deflate(&zs, Z_FINISH) != Z_STREAM_END) abort();
    out.resize(zs.total_out);
    deflateEnd(&zs);
    return out;
}

// zlib's verdict on a raw stream: true = it inflates to exactly n bytes (left in `out`), every payload byte needed
static bool zlib_inflates(const std::vector<uint8_t> &in, uint32_t n, std::vector<uint8_t> &out) {
    z_stream zs{};
    if (inflateInit2(&zs, -15) != Z_OK) abort();
    out.assign((size_t)n + 1, 0);
    zs.next_in = (Bytef *)in.data(); zs.avail_in = (uInt)in.size();
    zs.next_out = out.data(); zs.avail_out = n + 1;
    const int rc = inflate(&zs, Z_FINISH);
    const bool ok = rc == Z_STREAM_END && zs.total_out == n;
    if (getenv("FUZZ_VERBOSE")) fprintf(stderr, "zlib: rc %d total_out %lu avail_in %u msg %s\n", rc, zs.total_out, zs.avail_in, zs.msg ? zs.msg : "-");
    inflateEnd(&zs);
    out.resize(n);
    return ok;
}

static const char *g_what = "";
static long g_case = -1;
static void on_alarm(int) {
    char msg[160];
    const int k = snprintf(msg, sizeof msg, "inflate_fuzz: case %ld (%s) did not end\n", g_case, g_what);
    if (write(2, msg, (size_t)k) < 0) {}
    _exit(3);
}

int main(int argc, char **argv) {
    const long cases = argc > 1 ? atol(argv[1]) : 2000;
    g_state = argc > 2 ? strtoull(argv[2], nullptr, 10) * 2 + 1 : 1;
    signal(SIGALRM, on_alarm);
    InflateScratch *S = new InflateScratch;                                            // (heap: the sanitizer sees its end)
    inflate_crc_init(*S);
    long agreed_ok = 0, agreed_err = 0, lenient = 0;
    std::vector<uint8_t> base, comp;
    for (long c = 0; c < cases; ++c) {
        g_case = c;
        if (c % 8 == 0) {                                                              // a new clean stream every few cases
            static const int strat[4] = {Z_DEFAULT_STRATEGY, Z_FIXED, Z_HUFFMAN_ONLY, Z_RLE};
            static const int lvl[4] = {1, 6, 9, 0};
            base = payload(rnd(5), 1 + rnd(rnd(4) ? 6000 : 65536));
            comp = deflate_raw(base, lvl[rnd(4)], strat[rnd(4)], rnd(3) == 0);
        }
        std::vector<uint8_t> in = comp;
        uint32_t isize = (uint32_t)base.size();
        uint32_t crc = (uint32_t)crc32(0, base.data(), (uInt)base.size());
        switch (rnd(10)) {
        case 0: g_what = "clean"; break;
        case 1: g_what = "bit flip"; in[rnd((uint32_t)in.size())] ^= (uint8_t)(1u << rnd(8)); break;
        case 2: g_what = "bit flips"; for (uint32_t k = 0, n = 2 + rnd(8); k < n; ++k) in[rnd((uint32_t)in.size())] ^= (uint8_t)(1u << rnd(8)); break;
        case 3: g_what = "early bit flip"; in[rnd(std::min<uint32_t>((uint32_t)in.size(), 80))] ^= (uint8_t)(1u << rnd(8)); break;   // the block header / code lengths
        case 4: g_what = "bytes overwritten"; for (uint32_t k = 0, at = rnd((uint32_t)in.size()), n = 1 + rnd(16); k < n && at + k < in.size(); ++k) in[at + k] = (uint8_t)rnd(256); break;
        case 5: g_what = "cut short"; in.resize(rnd((uint32_t)in.size())); break;
        case 6: g_what = "garbage"; for (auto &b : in) b = (uint8_t)rnd(256); break;
        case 7: g_what = "garbage behind a dynamic header"; for (auto &b : in) b = (uint8_t)rnd(256); if (!in.empty()) in[0] = (uint8_t)((in[0] & ~7u) | 5u); break;
        case 8: g_what = "wrong ISIZE"; isize = rnd(4) ? isize + 1 + rnd(300) : (isize > 1 ? rnd(isize) : 0); break;
        default: g_what = "zero bytes"; for (uint32_t k = 0, at = rnd((uint32_t)in.size()), n = 1 + rnd(64); k < n && at + k < in.size(); ++k) in[at + k] = 0; break;
        }
        if (isize > 65536u) isize = 65536u;
        std::vector<uint8_t> want;
        const bool z_ok = zlib_inflates(in, isize, want) && (uint32_t)crc32(0, want.data(), (uInt)want.size()) == crc;
        // exact-size heap copies: the payload with the 16 bytes the decoder may look ahead, ISIZE bytes of output
        uint8_t *pin = (uint8_t *)malloc(in.size() + 16 + 3);
        memset(pin, 0, in.size() + 19);
        if (!in.empty()) memcpy(pin + 3, in.data(), in.size());                        // (an odd alignment on purpose)
        uint8_t *pout = (uint8_t *)malloc(isize ? isize : 1);
        alarm(20);
        const int rc = inflate_block(*S, pin + 3, (uint32_t)in.size(), pout, isize, crc);
        alarm(0);
        bool good;
        if (z_ok) { good = rc == 0 && (isize == 0 || memcmp(pout, want.data(), isize) == 0); agreed_ok += good; }
        else if (rc != 0) { good = true; ++agreed_err; }
        else { good = (uint32_t)crc32(0, pout, isize) == crc; lenient += good; }
        if (!good) {
            fprintf(stderr, "inflate_fuzz: case %ld (%s): zlib %s, decoder status %d (payload %zu bytes, ISIZE %u)\n", c, g_what, z_ok ? "inflates it" : "rejects it", rc, in.size(), isize);
            return 1;
        }
        free(pin); free(pout);
    }
    printf("inflate_fuzz: %ld cases: %ld inflated like zlib, %ld rejected like zlib, %ld that only zlib rejects (right bytes by the CRC)\n", cases, agreed_ok, agreed_err, lenient);
    delete S;
    return 0;
}
