// fuzz_records.h -- TEST HARNESS ONLY: a seeded generator of well-formed BAM alignment records (SAM spec 4.2) with random fields
// and aux data of every type, shared by the damaged-input searches (decode_fuzz.cpp, reader_fuzz.cpp).
#pragma once
#include <stdint.h>
#include <vector>

static uint64_t g_state = 1;
static uint32_t rnd() { g_state = g_state * 6364136223846793005ull + 1442695040888963407ull; return (uint32_t)(g_state >> 33); }
static uint32_t rnd(uint32_t n) { return n ? rnd() % n : 0; }

static void put32(std::vector<uint8_t> &v, uint32_t x) { for (int k = 0; k < 4; ++k) v.push_back((uint8_t)(x >> (8 * k))); }
static void put16(std::vector<uint8_t> &v, uint32_t x) { v.push_back((uint8_t)x); v.push_back((uint8_t)(x >> 8)); }

// one well-formed record (SAM spec 4.2) with random fields and aux data of every type
static void put_record(std::vector<uint8_t> &v, int32_t n_ref, bool huge) {
    std::vector<uint8_t> r;
    const uint32_t l_name = 1 + rnd(rnd(8) ? 30 : 254), n_cig = rnd(8) ? rnd(7) : rnd(300), l_seq = huge ? 50000 + rnd(150000) : rnd(rnd(6) ? 160 : 2000);
    put32(r, (uint32_t)((int32_t)rnd((uint32_t)n_ref + 1) - 1)); put32(r, rnd(1u << 28));
    r.push_back((uint8_t)l_name); r.push_back((uint8_t)rnd(61)); put16(r, rnd(65536));
    put16(r, n_cig); put16(r, rnd(4096)); put32(r, l_seq);
    put32(r, (uint32_t)((int32_t)rnd((uint32_t)n_ref + 1) - 1)); put32(r, rnd(1u << 28)); put32(r, rnd(2000) - 1000u);
    for (uint32_t k = 0; k + 1 < l_name; ++k) r.push_back((uint8_t)('!' + rnd(90)));
    r.push_back(0);
    for (uint32_t k = 0; k < n_cig; ++k) put32(r, (rnd(300) << 4) | rnd(9));
    for (uint32_t k = 0; k < (l_seq + 1) / 2 + l_seq; ++k) r.push_back((uint8_t)(rnd(4) ? rnd(256) : 0x11 * rnd(16)));   // (low-entropy stretches: what guesses trip over)
    for (uint32_t t = 0, nt = rnd(6); t < nt; ++t) {
        static const char *names[6] = {"NM", "ch", "XF", "CG", "MD", "zz"};
        const char *nm = names[rnd(6)];
        r.push_back((uint8_t)nm[0]); r.push_back((uint8_t)nm[1]);
        const char type = "AcCsSiIfdZHB"[rnd(12)];
        r.push_back((uint8_t)type);
        switch (type) {
        case 'A': case 'c': case 'C': r.push_back((uint8_t)rnd(256)); break;
        case 's': case 'S': put16(r, rnd(65536)); break;
        case 'i': case 'I': case 'f': put32(r, rnd()); break;
        case 'd': put32(r, rnd()); put32(r, rnd()); break;
        case 'Z': case 'H': for (uint32_t k = 0, n = rnd(40); k < n; ++k) r.push_back((uint8_t)('0' + rnd(40))); r.push_back(0); break;
        default: { const char st = "cCsSiIf"[rnd(7)]; const uint32_t es = (st == 'c' || st == 'C') ? 1 : (st == 's' || st == 'S') ? 2 : 4, cnt = rnd(50);
                   r.push_back((uint8_t)st); put32(r, cnt); for (uint32_t k = 0; k < es * cnt; ++k) r.push_back((uint8_t)rnd(256)); break; }
        }
    }
    put32(v, (uint32_t)r.size());
    v.insert(v.end(), r.begin(), r.end());
}

