// rsqc_inflate.h -- DEFLATE (RFC 1951) decoder for the payload of ONE BGZF block, written for ONE wavefront.
//
// The data format on the input side of the per-read path: a BAM file is a chain of BGZF blocks, each a
// gzip member of at most 64 KiB of output whose payload is an independent DEFLATE stream (SAM spec 4.1; the reference
// reads it through htslib's bgzf.c behind SeqLib, src/BamReader.cpp:12-20).  Independent blocks are the parallelism:
// one wave decodes one block, a few hundred thousand blocks per 100 M records.
//
// Shape of the decoder on CDNA4:
//  * the bit reader, the Huffman walk and the output position are WAVE-UNIFORM values (every lane computes the same
//    thing), so the compiler keeps them in SGPRs and the walk runs on the scalar unit -- which is what BOUNDS this kernel
//    (rocprofv3 --pmc: the CU's scalar issue port is ~97 % busy at five waves per SIMD, profiles/r2_decode_pmc_realistic_6M.txt):
//    the code below is written to keep scalar instructions per symbol down (raw-dword bit buffer, single-exit walk,
//    wave-uniform loop counts, a one-step decode of long codes), see DESIGN.md 6b;
//  * only WHERE a symbol starts is serial: every lane decodes the whole symbol that would start at its own bit offset of
//    the buffered input (both Huffman tables, extra bits, output length), the walk hops along those results with one
//    v_readlane per symbol, a prefix sum places the symbols, literals are stored by their own lanes (inflate_round);
//  * the compressed bytes arrive as one coalesced 256-byte vector load per 64 dwords (lane l holds dword l of the
//    window, the next window is already in flight) and are handed to the bit reader with v_readlane;
//  * the last 4 KiB of output live in an LDS ring, so a match is an LDS-to-LDS copy done by all lanes at once; the ring goes
//    out to HBM (and through the block's CRC-32) 2 KiB at a time, and the matches that reach further back than the
//    ring read the flushed bytes from HBM;
//  * LDS per wave: 4 KiB ring + 3.7 KiB of tables, so five waves share a SIMD (a round is a chain of dependent steps --
//    LDS look-ups, the walk, a prefix sum, LDS-to-LDS copies: other waves are what fills the gaps; one wave per SIMD is
//    2.6 times slower, profiles/r2_decode_variants.txt).
//
// The same source compiles for the host with a wave of ONE lane (tests/hostemu/decode_emu.cpp), which is how it is
// checked against zlib in the GPU-less container.
#pragma once

#include <stdint.h>
#include <string.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define RSQC_INF_FN __host__ __device__ __forceinline__
#else
#define RSQC_INF_FN inline
#endif

#if defined(__HIP_DEVICE_COMPILE__)
#define INF_W 64u
#define INF_LANE ((uint32_t)(threadIdx.x & 63u))
#define INF_UNI(x) ((uint32_t)__builtin_amdgcn_readfirstlane((int)(x)))
#else
#define INF_W 1u
#define INF_LANE 0u
#define INF_UNI(x) ((uint32_t)(x))
#endif
// a store done once per wave
#define INF_ST(stmt) do { if (INF_LANE == 0u) { stmt; } } while (0)

// One value per lane of the wave.  On the device that is a VGPR; the host build keeps an array of 64, so the tests run the
// same 64-lane logic (speculative table look-ups at 64 bit offsets, literals gathered one per lane) on the CPU.
#if defined(__HIP_DEVICE_COMPILE__)
typedef uint32_t InfVec;
#define INF_FOREACH(k) for (uint32_t k = INF_LANE, once_ = 1u; once_; once_ = 0u)              /* the body runs once: k = this lane */
#define INF_AT(vec, k) (vec)
#define INF_GET(vec, k) ((uint32_t)__builtin_amdgcn_readlane((int)(vec), (int)(k)))           /* k wave-uniform */
#define INF_SET(vec, k, x) ((vec) = (INF_LANE == (k)) ? (x) : (vec))
#define INF_GATHER(vec, idx) ((uint32_t)__builtin_amdgcn_ds_bpermute((int)((idx) << 2), (int)(vec)))   /* idx per lane, < 64: that lane's value of vec */
#else
struct InfVec { uint32_t v[64]; };
#define INF_FOREACH(k) for (uint32_t k = 0; k < 64u; ++k)
#define INF_AT(vec, k) ((vec).v[k])
#define INF_GET(vec, k) ((vec).v[k])
#define INF_SET(vec, k, x) ((vec).v[k] = (x))
#define INF_GATHER(vec, idx) ((vec).v[(idx) & 63u])
#endif

namespace rsqc {

// wave-wide helpers on lane vectors: which lanes hold a non-zero value, and the running (inclusive) sum over the lanes
#if defined(__HIP_DEVICE_COMPILE__)
__device__ __forceinline__ uint64_t inf_ballot(const InfVec &v) { return __ballot(v != 0u); }
#if !defined(INF_DPP_SCAN_CFG) || INF_DPP_SCAN_CFG
// The running sum as six DPP additions (row_shr 1 / 2 / 4 / 8 inside the rows of 16 lanes, then lane 15 of rows 0 and 2 into rows 1
// and 3, then lane 31 into rows 2 and 3) instead of six ds_bpermute round trips: +12.5 % / +8.7 % inflate rate on the realistic /
// the SURVEY 8(d) file (profiles/r3_decode_ab.txt; every block's CRC-32 and the report files are its check -- the host build
// runs the sequential form below).  -DINF_DPP_SCAN_CFG=0 builds the ds_bpermute form.
__device__ __forceinline__ InfVec inf_scan(InfVec v) {
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, false);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, false);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, false);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, false);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false);
    return v;
}
#else
__device__ __forceinline__ InfVec inf_scan(InfVec v) {
    for (uint32_t d = 1; d < 64u; d <<= 1) { const uint32_t x = (uint32_t)__shfl_up((int)v, d, 64); if (INF_LANE >= d) v += x; }
    return v;
}
#endif
#else
inline uint64_t inf_ballot(const InfVec &v) { uint64_t m = 0; for (uint32_t k = 0; k < 64u; ++k) if (v.v[k]) m |= 1ull << k; return m; }
inline InfVec inf_scan(InfVec v) { for (uint32_t k = 1; k < 64u; ++k) v.v[k] += v.v[k - 1]; return v; }
#endif

enum InflateStatus {
    INF_OK = 0,
    INF_ERR_BTYPE = 1,        // reserved block type
    INF_ERR_STORED = 2,       // LEN / NLEN mismatch
    INF_ERR_TABLE = 3,        // over-subscribed or malformed code lengths
    INF_ERR_SYMBOL = 4,       // a code that no symbol owns, or an invalid length / distance symbol
    INF_ERR_DISTANCE = 5,     // match reaching before the start of the block's output
    INF_ERR_OUTPUT = 6,       // more (or fewer) bytes than ISIZE
    INF_ERR_INPUT = 7,        // ran past the compressed payload
    INF_ERR_CRC = 8
};

// The LDS ring holds the most recent INF_RING bytes of output; a match that reaches further back than INF_NEAR reads the
// bytes from the stream in HBM, where they were flushed long before (INF_RING >= INF_FLUSH + 774 guarantees that, see
// inflate_copy).  A small ring is what lets several waves share a SIMD: the decoder is a chain of dependent scalar
// instructions, and the only way to fill the issue slots is more waves.
#ifndef INF_RING_BITS_CFG
#define INF_RING_BITS_CFG 12
#endif
constexpr uint32_t INF_RING_BITS = INF_RING_BITS_CFG, INF_RING = 1u << INF_RING_BITS, INF_RMASK = INF_RING - 1u;
constexpr uint32_t INF_NEAR = INF_RING - 258u;          // distances up to this are served from the ring
// (sizes measured as whole-library variants on one box, profiles/r2_decode_variants.txt: 2^8 / 2^7 entry tables + five waves
//  per SIMD beat 2^9 / 2^8 + four by 2 % / 6 % on the two files; 2^11 / 2^9 loses 25 %: the tables are rebuilt per block)
#ifndef INF_LBITS_CFG
#define INF_LBITS_CFG 8
#endif
#ifndef INF_DBITS_CFG
#define INF_DBITS_CFG 7
#endif
#ifndef INF_FLUSH_CFG
#define INF_FLUSH_CFG 2048
#endif
#ifndef INF_ROUND_BYTES_CFG
#define INF_ROUND_BYTES_CFG 1024
#endif
// The one-pass commit of a round (inflate_round<true>): a round whose output fits one byte per lane and whose matches all copy from
// before the round is committed in ONE vector pass.  Measured (profiles/r3_decode_ab.txt, same box, alternating builds, outputs
// identical): +7.4 % on the file with the entropy of a real BAM (35.1 -> 37.7 GB/s: most of its rounds qualify), -6 % on the
// SURVEY 8(d) file (131 -> 124 GB/s: its rounds are long matches and never qualify -- the cost there is the registers the path
// holds).  Both forms are therefore compiled (template parameter PAR) and rsqc_decode_submit picks per call by the call's
// compression ratio.  INF_PAR_COMMIT_CFG is the form the host tests and a caller without a preference get.
#ifndef INF_PAR_COMMIT_CFG
#define INF_PAR_COMMIT_CFG 1
#endif
// 1: a code longer than the fast table does not end the round: the walk decodes it where it stands (one step of the wave,
// inflate_symbol_slow), writes the result into that lane and goes on over the lanes behind it (+1.5 % / +3.8 %,
// profiles/r3_decode_ab.txt; the lane-parallel table build measured beside it was +-0 and is gone).
#ifndef INF_INWALK_CFG
#define INF_INWALK_CFG 1
#endif
// > 0: the walk hops 2^INF_VWALK_CFG symbols at a time.  Every lane knows where the symbol at its offset ends; that many rounds of
// pointer doubling on the vector side (three ds_bpermute each: the landing lane and the 64-bit mask of the starts on the way) give
// every lane the landing point and the starts of 2^n symbols, and the wave-uniform walk takes one step (three v_readlane + four
// scalar instructions) where it took 2^n (one v_readlane + ten each).  0: one symbol per step (rounds 2-4).
// Measured (round 5, profiles/r5_decode_walk_ab.txt, one box, outputs identical): 40.4 / 40.1 / 40.4 / 39.7 GB/s of inflated bytes at
// 0 / 1 / 2 / 3 on the realistic file, 122.4 / 123.8 / 123.0 / 121.8 on the SURVEY 8(d) file -- NOTHING.  The walk was the scalar
// port's largest single customer and it is not what holds the kernel: a round is ~800 issued instructions of which the walk was
// ~110, and what the scalar side gives up the vector side and six LDS round trips take back.  The product builds 0 (the form with
// three rounds of fuzzing behind it); the others stay under test (tests/test_device_decode_host.py).
#ifndef INF_VWALK_CFG
#define INF_VWALK_CFG 0
#endif
constexpr uint32_t INF_LBITS = INF_LBITS_CFG, INF_DBITS = INF_DBITS_CFG;
constexpr uint32_t INF_FLUSH = INF_FLUSH_CFG;          // the ring goes out to HBM (and through the CRC) in pieces of this size
static_assert(INF_RING >= INF_FLUSH + INF_ROUND_BYTES_CFG + 774u, "the far-match argument needs this (a round adds up to 1024 bytes before the next flush)");

// decoding tables + output history of one wave (LDS on the device)
struct InflateScratch {
    uint8_t ring[INF_RING];
    uint32_t lfast[1u << INF_LBITS];     // index: the next LBITS bits of the stream; an InflateEntry, 0 = longer code (or none)
    uint32_t dfast[1u << INF_DBITS];
    uint16_t lcount[16], dcount[16];     // codes per length
    uint16_t lsym[288], dsym[32];        // symbols by (length, symbol): canonical decoding of the codes the fast table does not hold
    uint16_t lfirst[16], dfirst[16];     // first canonical code of a length ...
    uint16_t lidx[16], didx[16];         // ... and the slot of its symbol in lsym / dsym
    uint8_t lens[320];                   // code lengths of the block being set up
    uint16_t offs[16];                   // first slot of a length in lsym / dsym while a table is built
    uint32_t crc_tab[256];               // CRC-32 (reflected 0xEDB88320), one byte per step
    uint32_t slot[64];                   // one-pass commit of a round: the symbol that starts at output byte i of the round
};

// ---- CRC-32 of the inflated bytes (the gzip member's trailer; htslib's bgzf reader checks it, so a corrupt block is an
// error in the reference).  The register is linear over GF(2): the state after A||B is (state after A, advanced over
// |B| zero bytes) xor (raw register of B started from 0), and advancing over n zero bytes is a multiplication by
// x^(8n) mod P.  So the lanes take consecutive pieces of a flushed chunk, and a six-level tree combines them.
constexpr uint32_t INF_CRC_POLY = 0xEDB88320u;
// a(x) * b(x) mod P, reflected representation (x^0 = 0x80000000)
RSQC_INF_FN constexpr uint32_t crc_mulmod(uint32_t a, uint32_t b) {
    uint32_t p = 0;
    for (uint32_t i = 0; i < 32u; ++i) {
        p ^= (a & (0x80000000u >> i)) ? b : 0u;
        b = (b >> 1) ^ ((b & 1u) ? INF_CRC_POLY : 0u);
    }
    return p;
}
// x^e mod P
RSQC_INF_FN constexpr uint32_t crc_xpow(uint64_t e) {
    uint32_t r = 0x80000000u, base = 0x40000000u;
    while (e) { if (e & 1u) r = crc_mulmod(r, base); base = crc_mulmod(base, base); e >>= 1; }
    return r;
}
RSQC_INF_FN void inflate_crc_init(InflateScratch &S) {
    for (uint32_t i = INF_LANE; i < 256u; i += INF_W) {
        uint32_t c = i;
        for (int k = 0; k < 8; ++k) c = (c >> 1) ^ ((c & 1u) ? INF_CRC_POLY : 0u);
        S.crc_tab[i] = c;
    }
}

// ---- input: dwords of the compressed payload -------------------------------------------------------------------
struct InflateIn {
    const uint32_t *base;    // dword-aligned address at or before the first payload byte
    uint32_t n_words;        // dwords that may be read
    uint32_t next;           // next dword index
    // The buffer is four RAW dwords of the stream, base[next - 4 .. next) (lo = the first two, hi = the last two), and the bit
    // offset `bo` of the first unconsumed bit in them: consuming bits is an addition, refilling moves whole dwords (register
    // moves) -- no 128-bit shifts by variable amounts on the scalar unit, which is what bounds this decoder.
    uint64_t lo, hi; uint32_t bo;
#if defined(__HIP_DEVICE_COMPILE__)
    uint32_t win, win_next;  // lane l: base[win_base + l] and base[win_base + 64 + l]
    uint32_t win_base;
    __device__ __forceinline__ uint32_t load(uint32_t i) const { return i < n_words ? __builtin_nontemporal_load(base + i) : 0u; }
    __device__ __forceinline__ void open_window(uint32_t at) {
        win_base = at & ~63u;
        win = load(win_base + INF_LANE);
        win_next = load(win_base + 64u + INF_LANE);
    }
    __device__ __forceinline__ uint32_t next32() {
        uint32_t k = next - win_base;
        if (k >= 64u) {                                  // (uniform) the next window becomes current, the one after it is issued
            win = win_next; win_base += 64u;
            win_next = load(win_base + 64u + INF_LANE);
            k -= 64u;
        }
        ++next;
        return (uint32_t)__builtin_amdgcn_readlane((int)win, (int)k);
    }
#else
    void open_window(uint32_t) {}
    uint32_t next32() { const uint32_t i = next++; return i < n_words ? base[i] : 0u; }
#endif
    // position the reader on payload byte `byte_pos` (relative to base)
    RSQC_INF_FN void seek(uint32_t byte_pos) {
        next = byte_pos >> 2;
        open_window(next);
        const uint32_t w0 = next32(), w1 = next32(), w2 = next32(), w3 = next32();
        lo = (uint64_t)w0 | ((uint64_t)w1 << 32); hi = (uint64_t)w2 | ((uint64_t)w3 << 32);
        bo = 8u * (byte_pos & 3u);
    }
    RSQC_INF_FN uint32_t avail() const { return 128u - bo; }                             // buffered bits not consumed yet
    RSQC_INF_FN void refill() {                                                          // afterwards 97 <= avail() <= 128
        while (bo >= 32u) { lo = (lo >> 32) | (hi << 32); hi = (hi >> 32) | ((uint64_t)next32() << 32); bo -= 32u; }
    }
    // the stream from bit `off` behind the first unconsumed one (bo + off < 128); only the first avail() - off bits are meaningful
    RSQC_INF_FN uint64_t bits_at(uint32_t off) const {
        const uint32_t sh = bo + off;
        if (sh >= 64u) return hi >> (sh - 64u);
        return sh ? (lo >> sh) | (hi << (64u - sh)) : lo;
    }
    RSQC_INF_FN uint32_t head32() const { return (uint32_t)bits_at(0); }                 // the next 32 bits (bo <= 96)
    RSQC_INF_FN uint32_t peek(uint32_t n) const { return head32() & ((1u << n) - 1u); }  // n < 32
    RSQC_INF_FN void drop(uint32_t n) { bo += n; }                                       // n <= avail()
    RSQC_INF_FN uint32_t take(uint32_t n) { const uint32_t v = peek(n); drop(n); return v; }
    RSQC_INF_FN uint32_t byte_pos() const { return next * 4u - avail() / 8u; }           // first byte not consumed yet (whole bytes left in the buffer)
};

// ---- Huffman tables ------------------------------------------------------------------------------------------
// A table entry says everything the decoder needs about the symbol, so that the symbol loop never goes back to the
// constant tables of RFC 1951 3.2.5:
//   bits 0-3 code length | bits 4-7 extra bits that follow | bits 8-23 value: the literal, the length base or the distance base
//   bit 31 literal | bit 30 end of block | bit 29 a symbol the format does not define (286, 287; distance 30, 31)
constexpr uint32_t INF_E_LITERAL = 0x80000000u, INF_E_END = 0x40000000u, INF_E_INVALID = 0x20000000u;
enum InflateTableKind { INF_T_PLAIN = 0, INF_T_LITLEN = 1, INF_T_DIST = 2 };      // PLAIN: (symbol << 4) | length, the code-length code
RSQC_INF_FN uint32_t inflate_entry(int kind, uint32_t sym, uint32_t len) {
    static const uint16_t kLenBase[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
    static const uint8_t kLenExtra[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
    static const uint16_t kDistBase[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
    static const uint8_t kDistExtra[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};
    if (kind == INF_T_PLAIN) return (sym << 4) | len;
    if (kind == INF_T_LITLEN) {
        if (sym < 256u) return INF_E_LITERAL | (sym << 8) | len;
        if (sym == 256u) return INF_E_END | len;
        if (sym >= 286u) return INF_E_INVALID | len;
        return ((uint32_t)kLenBase[sym - 257u] << 8) | ((uint32_t)kLenExtra[sym - 257u] << 4) | len;
    }
    if (sym >= 30u) return INF_E_INVALID | len;
    return ((uint32_t)kDistBase[sym] << 8) | ((uint32_t)kDistExtra[sym] << 4) | len;
}

// lens[0..n): code lengths.  Builds count[]/sym[] (canonical order) and the fast table of `fbits` bits.  false = the
// lengths over-subscribe the code space (zlib: "invalid code lengths set").  An incomplete set is accepted, as zlib
// accepts a single distance code; a code nobody owns is an error when the stream uses it.
RSQC_INF_FN bool inflate_build(const uint8_t *lens, uint32_t n, uint16_t *count, uint16_t *sym, uint16_t *first, uint16_t *index, uint32_t *fast, uint32_t fbits, uint16_t *offs, int kind) {
    for (uint32_t l = INF_LANE; l < 16u; l += INF_W) count[l] = 0;
    for (uint32_t k = INF_LANE; k < (1u << fbits); k += INF_W) fast[k] = 0;
    for (uint32_t s = 0; s < n; ++s) { const uint32_t l = INF_UNI(lens[s]); INF_ST(count[l]++); }
    if (INF_UNI(count[0]) == n) return true;                       // no codes at all: legal as long as none is used
    int32_t left = 1;
    uint32_t run = 0, fcode = 0;
    for (uint32_t l = 1; l <= 15u; ++l) {
        const uint32_t c = INF_UNI(count[l]);
        left = (left << 1) - (int32_t)c;
        if (left < 0) return false;
        INF_ST(offs[l] = (uint16_t)run; index[l] = (uint16_t)run; first[l] = (uint16_t)fcode);      // (fcode <= 2^l: the lengths do not over-subscribe)
        run += c;
        fcode = (fcode + c) << 1;
    }
    for (uint32_t s = 0; s < n; ++s) {
        const uint32_t l = INF_UNI(lens[s]);
        if (l) { INF_ST(sym[offs[l]] = (uint16_t)s; offs[l]++); }
    }
    // canonical codes in (length, symbol) order; the stream carries a code most significant bit first inside its
    // least-significant-bit-first bit order, so the table is indexed by the bit-reversed code
    uint32_t code = 0, idx = 0;
    for (uint32_t l = 1; l <= fbits; ++l) {
        const uint32_t c = INF_UNI(count[l]);
        for (uint32_t k = 0; k < c; ++k, ++idx, ++code) {
            const uint32_t s = INF_UNI(sym[idx]);
            uint32_t rev = 0;
            for (uint32_t b = 0; b < l; ++b) rev |= ((code >> b) & 1u) << (l - 1u - b);
            const uint32_t e = inflate_entry(kind, s, l);
            for (uint32_t t = rev + (INF_LANE << l); t < (1u << fbits); t += (INF_W << l)) fast[t] = e;
        }
        code <<= 1;
    }
    return true;
}

// a code longer than the fast table's index (or one nobody owns) at the head of `bits`: the symbol and its length, or 0xFFFF.
// Canonical decoding: the code of length l is the first l bits, most significant first; it is a symbol's iff it lies less
// than count[l] above first[l], and the shortest such l wins.  Lane l tests length l, so the whole search is one step of the
// wave instead of fifteen dependent ones on the scalar unit (every 20th symbol of a real file takes this path).
RSQC_INF_FN uint32_t inflate_bitrev32(uint32_t x) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_bitreverse32(x);
#else
    x = ((x >> 1) & 0x55555555u) | ((x & 0x55555555u) << 1); x = ((x >> 2) & 0x33333333u) | ((x & 0x33333333u) << 2);
    x = ((x >> 4) & 0x0F0F0F0Fu) | ((x & 0x0F0F0F0Fu) << 4); x = ((x >> 8) & 0x00FF00FFu) | ((x & 0x00FF00FFu) << 8);
    return (x >> 16) | (x << 16);
#endif
}
RSQC_INF_FN uint32_t inflate_symbol_slow(uint32_t bits, const uint16_t *count, const uint16_t *first, const uint16_t *index, const uint16_t *sym, uint32_t &len) {
    const uint32_t rev = inflate_bitrev32(bits);
    InfVec OWN;
    INF_FOREACH(k) {
        const uint32_t l = k < 1u ? 1u : k > 15u ? 15u : k;                 // lanes 1..15 test their own length (the others repeat one and are masked)
        INF_AT(OWN, k) = (k == l && (rev >> (32u - l)) - (uint32_t)first[l] < (uint32_t)count[l]) ? 1u : 0u;
    }
    const uint64_t own = inf_ballot(OWN);
    if (!own) { len = 0; return 0xFFFFu; }
    const uint32_t l = (uint32_t)__builtin_ctzll(own);
    len = l;
    return INF_UNI(sym[INF_UNI(index[l]) + (rev >> (32u - l)) - INF_UNI(first[l])]);
}
#define INF_BUILD inflate_build

// one symbol at the head of the reader (the block headers' code-length code)
RSQC_INF_FN uint32_t inflate_symbol(InflateIn &in, const uint32_t *fast, uint32_t fbits, const uint16_t *count, const uint16_t *first, const uint16_t *index, const uint16_t *sym) {
    const uint32_t e = INF_UNI(fast[in.peek(fbits)]);
    if (e) { in.drop(e & 15u); return e >> 4; }
    uint32_t len;
    const uint32_t s = inflate_symbol_slow(in.head32(), count, first, index, sym, len);
    in.drop(len);
    return s;
}

struct InflateOut {
    uint8_t *dst;            // the block's place in the inflated stream
    uint32_t out_len;        // ISIZE
    uint32_t pos, flushed;
    uint32_t crc;            // CRC register over the flushed bytes
};

#if defined(__HIP_DEVICE_COMPILE__)
__device__ __forceinline__ uint32_t inflate_lane_down(uint32_t v, uint32_t delta) { return (uint32_t)__shfl_down((int)v, delta, 64); }
#endif

// writes ring bytes [flushed, flushed + n) to the stream and runs them through the CRC; n <= INF_FLUSH
RSQC_INF_FN void inflate_flush(InflateScratch &S, InflateOut &o, uint32_t n) {
    for (uint32_t j = INF_LANE; j < n; j += INF_W) o.dst[o.flushed + j] = S.ring[(o.flushed + j) & INF_RMASK];
    // lane l: raw register of piece l of INF_W equal pieces; the chunk is right-aligned in them (leading zero bytes leave a
    // register that started from 0 at 0)
    const uint32_t plen = (n + INF_W - 1u) / INF_W, pad = plen * INF_W - n;
    uint32_t r = 0;
    for (uint32_t k = 0; k < plen; ++k) {
        const uint32_t v = INF_LANE * plen + k;
        if (v >= pad) { const uint32_t b = S.ring[(o.flushed + v - pad) & INF_RMASK]; r = S.crc_tab[(r ^ b) & 0xFFu] ^ (r >> 8); }
    }
#if defined(__HIP_DEVICE_COMPILE__)
    // r[l] = r[l] * x^(8 plen d) + r[l + d], d = 1, 2, .. 32
    if (n == INF_FLUSH) {                                           // a full chunk's six multipliers are compile-time constants (-5 % kernel time)
        constexpr uint32_t P = 8u * (INF_FLUSH / 64u);
        constexpr uint32_t M0 = crc_xpow(P), M1 = crc_xpow(2ull * P), M2 = crc_xpow(4ull * P), M3 = crc_xpow(8ull * P), M4 = crc_xpow(16ull * P), M5 = crc_xpow(32ull * P);
        r = crc_mulmod(r, M0) ^ inflate_lane_down(r, 1);
        r = crc_mulmod(r, M1) ^ inflate_lane_down(r, 2);
        r = crc_mulmod(r, M2) ^ inflate_lane_down(r, 4);
        r = crc_mulmod(r, M3) ^ inflate_lane_down(r, 8);
        r = crc_mulmod(r, M4) ^ inflate_lane_down(r, 16);
        r = crc_mulmod(r, M5) ^ inflate_lane_down(r, 32);
    } else {
        uint32_t m = crc_xpow(8ull * plen);
        for (uint32_t d = 1; d < 64u; d <<= 1) {
            const uint32_t hi = inflate_lane_down(r, d);
            r = crc_mulmod(r, m) ^ hi;
            m = crc_mulmod(m, m);
        }
    }
    r = INF_UNI(r);
#endif
    constexpr uint32_t kFullChunk = crc_xpow(8ull * INF_FLUSH);
    const uint32_t adv = (n == INF_FLUSH) ? kFullChunk : crc_xpow(8ull * n);
    o.crc = crc_mulmod(o.crc, adv) ^ r;
    o.flushed += n;
}

// out[pos + j] = out[pos + j - dist], j < len.
// Near (dist <= INF_NEAR): ring to ring.  A match that overlaps its own output (dist < len) repeats the dist bytes before
// pos: every lane reads byte (j mod dist) of that period, so all of it is ONE round of independent LDS reads -- a run of a
// single byte (dist 1, the bulk of SEQ/QUAL in low-entropy files) costs the same as any other match.  No write of the copy
// can land on a slot that a later pass still has to read: that needs dist >= INF_RING - 257.
// Far: the source left the ring; it is read back from the stream.  Those bytes have been flushed: a round starts with less
// than INF_FLUSH unflushed bytes and adds at most INF_ROUND_BYTES before the next flush, and the source ends before
// pos - dist + 258 < pos - INF_RING + 516 <= pos - INF_FLUSH - INF_ROUND_BYTES.
// The loads follow the flush's stores of the same wave: a workgroup-scope fence (a wait for the stores, the CU's vector
// cache is coherent within a workgroup) orders them.
#if defined(INF_STATS) && !defined(__HIP_DEVICE_COMPILE__)
struct InflateStats { unsigned long long near_matches, far_matches, match_bytes, rounds, round_symbols, slow_symbols, dist_hist[16], far_len_hist[10]; };
inline InflateStats &inflate_stats() { static InflateStats st{}; return st; }
#define INF_STAT(x) (x)
#else
#define INF_STAT(x) ((void)0)
#endif
RSQC_INF_FN void inflate_copy(InflateScratch &S, const InflateOut &o, uint32_t pos, uint32_t dist, uint32_t len) {
    INF_STAT((dist > INF_NEAR ? inflate_stats().far_matches : inflate_stats().near_matches)++); INF_STAT(inflate_stats().match_bytes += len); INF_STAT(inflate_stats().dist_hist[32 - __builtin_clz(dist | 1u) > 15 ? 15 : 32 - __builtin_clz(dist | 1u)]++);
    INF_STAT(dist > INF_NEAR ? inflate_stats().far_len_hist[len < 4 ? 0 : len <= 8 ? 1 : len <= 16 ? 2 : len <= 32 ? 3 : len <= 64 ? 4 : 5]++ : 0);
    // (the loops count wave-uniform passes of one byte per lane and predicate the lanes inside: a loop whose trip count differs per lane
    //  makes the compiler wrap the whole function in exec-mask bookkeeping, and this code is bound by the scalar unit)
    // len >= 3 (RFC 1951), so every loop runs at least once: do-while saves the entry test
    if (dist >= len) {                                                 // no overlap
        if (dist <= INF_NEAR) {                                        // the common case: ring to ring
            if (INF_LANE < len) S.ring[(pos + INF_LANE) & INF_RMASK] = S.ring[(pos + INF_LANE - dist) & INF_RMASK];      // (most matches fit one pass)
            for (uint32_t b = INF_W; b < len; b += INF_W) {
                const uint32_t j = b + INF_LANE;
                if (j < len) S.ring[(pos + j) & INF_RMASK] = S.ring[(pos + j - dist) & INF_RMASK];
            }
        } else {
#if defined(__HIP_DEVICE_COMPILE__)
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
#endif
            if (INF_LANE < len) S.ring[(pos + INF_LANE) & INF_RMASK] = o.dst[pos - dist + INF_LANE];
            for (uint32_t b = INF_W; b < len; b += INF_W) {
                const uint32_t j = b + INF_LANE;
                if (j < len) S.ring[(pos + j) & INF_RMASK] = o.dst[pos - dist + j];
            }
        }
    } else if (dist < 64u) {                                           // (dist < len <= 258: the source is in the ring)
        static const uint32_t kRecip[64] = {0, 65536, 32768, 21846, 16384, 13108, 10923, 9363, 8192, 7282, 6554, 5958, 5462, 5042, 4682, 4370, 4096, 3856, 3641, 3450, 3277,
                                            3121, 2979, 2850, 2731, 2622, 2521, 2428, 2341, 2260, 2185, 2115, 2048, 1986, 1928, 1873, 1821, 1772, 1725, 1681, 1639, 1599,
                                            1561, 1525, 1490, 1457, 1425, 1395, 1366, 1338, 1311, 1286, 1261, 1237, 1214, 1192, 1171, 1150, 1130, 1111, 1093, 1075, 1058, 1041};
        const uint32_t recip = kRecip[dist];                           // ceil(65536 / dist): j / dist == (j * recip) >> 16 for j < 1040, dist < 64 (tests: every pair)
        uint32_t b = 0;
        do {
            const uint32_t j = b + INF_LANE;
            const uint32_t r = j - ((j * recip) >> 16) * dist;
            if (j < len) S.ring[(pos + j) & INF_RMASK] = S.ring[(pos - dist + r) & INF_RMASK];
            b += INF_W;
        } while (b < len);
    } else {
        uint32_t b = 0;
        do {                                                            // len <= 258 < 5 * 64
            const uint32_t j = b + INF_LANE;
            uint32_t r = j;
            r -= (r >= dist) ? dist : 0u; r -= (r >= dist) ? dist : 0u; r -= (r >= dist) ? dist : 0u; r -= (r >= dist) ? dist : 0u;
            if (j < len) S.ring[(pos + j) & INF_RMASK] = S.ring[(pos - dist + r) & INF_RMASK];
            b += INF_W;
        } while (b < len);
    }
}

// ---- the symbols of a block ---------------------------------------------------------------------------------------
// What is serial in DEFLATE is only WHERE the next symbol starts.  Everything else about a symbol -- literal or match,
// its length and distance, how many bits it takes, how many bytes it puts out -- depends on nothing but the bits at its
// own offset.  So in a round every lane decodes, completely, "the symbol that would start at my bit offset" of the 97+
// buffered bits (both Huffman tables, extra bits included); the wave-uniform walk then only hops along those results
// (one v_readlane and eight scalar instructions per symbol) to mark the lanes that ARE symbol starts; a prefix sum over
// the marked lanes gives every symbol its place in the output; the literals are stored by their own lanes, the matches
// are copied one after the other by the whole wave.
enum { INF_K_LIT = 0, INF_K_MATCH = 1, INF_K_END = 2, INF_K_OTHER = 3 };      // OTHER: a code longer than the fast table, an undefined one, or bits not buffered yet
constexpr uint32_t INF_ROUND_BYTES = INF_ROUND_BYTES_CFG;   // output of one round at most (the ring keeps unflushed bytes: INF_FLUSH + this < INF_RING)

// one symbol the long way (a code longer than the fast table's index, or the symbol the buffered bits ended in).
// true = the block goes on; false = it ended (status untouched) or failed (status = the InflateStatus)
RSQC_INF_FN bool inflate_one_symbol(InflateScratch &S, InflateIn &bi, InflateOut &o, uint32_t &status) {
    bi.refill();
    uint32_t e = INF_UNI(S.lfast[bi.peek(INF_LBITS)]);
    if (!e) {
        uint32_t len;
        const uint32_t s = inflate_symbol_slow(bi.head32(), S.lcount, S.lfirst, S.lidx, S.lsym, len);
        if (s == 0xFFFFu) { status = INF_ERR_SYMBOL; return false; }
        e = inflate_entry(INF_T_LITLEN, s, len);
    }
    bi.drop(e & 15u);
    if (e & INF_E_LITERAL) {
        if (o.pos >= o.out_len) { status = INF_ERR_OUTPUT; return false; }
        INF_ST(S.ring[o.pos & INF_RMASK] = (uint8_t)(e >> 8));
        o.pos++;
        while (o.pos - o.flushed >= INF_FLUSH) inflate_flush(S, o, INF_FLUSH);
        return true;
    }
    if (e & INF_E_END) return false;
    if (e & INF_E_INVALID) { status = INF_ERR_SYMBOL; return false; }
    const uint32_t xb = (e >> 4) & 15u;
    const uint32_t mlen = ((e >> 8) & 0xFFFFu) + bi.take(xb);
    bi.refill();
    uint32_t f = INF_UNI(S.dfast[bi.peek(INF_DBITS)]);
    if (!f) {
        uint32_t dl;
        const uint32_t ds = inflate_symbol_slow(bi.head32(), S.dcount, S.dfirst, S.didx, S.dsym, dl);
        if (ds == 0xFFFFu) { status = INF_ERR_SYMBOL; return false; }
        f = inflate_entry(INF_T_DIST, ds, dl);
    }
    if (f & INF_E_INVALID) { status = INF_ERR_SYMBOL; return false; }
    bi.drop(f & 15u);
    const uint32_t dist = ((f >> 8) & 0xFFFFu) + bi.take((f >> 4) & 15u);
    if (dist > o.pos) { status = INF_ERR_DISTANCE; return false; }
    if (o.pos + mlen > o.out_len) { status = INF_ERR_OUTPUT; return false; }
    inflate_copy(S, o, o.pos, dist, mlen);
    o.pos += mlen;
    while (o.pos - o.flushed >= INF_FLUSH) inflate_flush(S, o, INF_FLUSH);
    return true;
}

// the symbol at bit offset `off` of the buffered bits when its code is longer than the fast table's index (wave-uniform; the walk of
// inflate_round stands on it).  false: not such a symbol -- the buffered bits end inside it, nobody owns the code
RSQC_INF_FN bool inflate_long_code_here(InflateScratch &S, InflateIn &bi, uint32_t off, uint32_t avail, uint32_t &kind, uint32_t &adv, uint32_t &ol, uint32_t &val) {
    const uint64_t w = bi.bits_at(off);
    if (INF_UNI(S.lfast[(uint32_t)w & ((1u << INF_LBITS) - 1u)])) return false;   // (not a long code: the symbol runs past the buffered bits, or is undefined)
    uint32_t len;
    const uint32_t sy = inflate_symbol_slow((uint32_t)w, S.lcount, S.lfirst, S.lidx, S.lsym, len);
    if (sy == 0xFFFFu) return false;
    const uint32_t e = inflate_entry(INF_T_LITLEN, sy, len);
    ol = 0; val = 0;
    if (e & INF_E_LITERAL) { kind = INF_K_LIT; adv = len; ol = 1; val = (e >> 8) & 0xFFu; }
    else if (e & INF_E_END) { kind = INF_K_END; adv = len; }
    else if (e & INF_E_INVALID) return false;
    else {
        const uint32_t xb = (e >> 4) & 15u;
        const uint64_t w1 = w >> len;
        const uint32_t mlen = ((e >> 8) & 0xFFFFu) + ((uint32_t)w1 & ((1u << xb) - 1u));
        const uint64_t w2 = w1 >> xb;
        uint32_t f = INF_UNI(S.dfast[(uint32_t)w2 & ((1u << INF_DBITS) - 1u)]);
        if (!f) {
            uint32_t dl;
            const uint32_t ds = inflate_symbol_slow((uint32_t)w2, S.dcount, S.dfirst, S.didx, S.dsym, dl);
            if (ds == 0xFFFFu) return false;
            f = inflate_entry(INF_T_DIST, ds, dl);
        }
        if (f & INF_E_INVALID) return false;
        const uint32_t l2 = f & 15u, db = (f >> 4) & 15u;
        kind = INF_K_MATCH; adv = len + xb + l2 + db; ol = mlen;
        val = ((f >> 8) & 0xFFFFu) + ((uint32_t)(w2 >> l2) & ((1u << db) - 1u));
    }
    return off + adv <= avail;                                              // (false: the symbol runs past the buffered bits)
}

// true = the block goes on; false = it ended (status untouched) or failed (status = the InflateStatus)
template <bool PAR = (INF_PAR_COMMIT_CFG != 0)>
RSQC_INF_FN bool inflate_round(InflateScratch &S, InflateIn &bi, InflateOut &o, uint32_t &status) {
    bi.refill();
    const uint32_t avail = bi.avail();
    InfVec PK, VAL, OL;                                                     // (kind << 8) | bits of the whole symbol; literal byte or distance; output bytes
    INF_FOREACH(k) {
        const uint64_t w = bi.bits_at(k);
        const uint32_t e = S.lfast[(uint32_t)w & ((1u << INF_LBITS) - 1u)];
        uint32_t kind = INF_K_OTHER, adv = 0, ol = 0, val = 0;
        if (e & INF_E_LITERAL) { kind = INF_K_LIT; adv = e & 15u; ol = 1; val = (e >> 8) & 0xFFu; }
        else if (e & INF_E_END) { kind = INF_K_END; adv = e & 15u; }
        else if (e && !(e & INF_E_INVALID)) {
            const uint32_t l1 = e & 15u, xb = (e >> 4) & 15u;
            const uint64_t w1 = w >> l1;
            const uint32_t mlen = ((e >> 8) & 0xFFFFu) + ((uint32_t)w1 & ((1u << xb) - 1u));
            const uint64_t w2 = w1 >> xb;
            const uint32_t f = S.dfast[(uint32_t)w2 & ((1u << INF_DBITS) - 1u)];
            if (f && !(f & INF_E_INVALID)) {
                const uint32_t l2 = f & 15u, db = (f >> 4) & 15u;
                kind = INF_K_MATCH; adv = l1 + xb + l2 + db; ol = mlen;
                val = ((f >> 8) & 0xFFFFu) + ((uint32_t)(w2 >> l2) & ((1u << db) - 1u));
            }
        }
        if (k + adv > avail) kind = INF_K_OTHER;                            // the symbol runs past the buffered bits
        INF_AT(PK, k) = (kind << 8) | adv; INF_AT(VAL, k) = val; INF_AT(OL, k) = ol;
    }
    // the walk: which lanes are symbol starts.  One exit test: a lane that ends the walk (END, OTHER) holds a value >= END << 8,
    // and off << 3 reaches that value exactly when the walk has left the 64 buffered offsets (off < 112: a symbol is at most 48 bits)
    uint64_t starts = 0;
    uint32_t off = 0;
#if INF_VWALK_CFG
    // J: where the walk stands after the symbols counted in (MLO, MHI) when it enters at this lane -- 128 + the lane it cannot take
    // (END, OTHER), or the offset >= 64 at which it has left the buffered offsets (< 112: a symbol is at most 48 bits); either
    // way a value >= 64 ends the walk, with ONE test per step.  The hops only go up, so what a lane holds never depends on a lane
    // below it: a long code decoded in place below (INF_INWALK_CFG) leaves the lanes above it valid.
    InfVec J, MLO, MHI;
    INF_FOREACH(k) {
        const uint32_t pk = INF_AT(PK, k);
        const bool take = pk < ((uint32_t)INF_K_END << 8);
        INF_AT(J, k) = take ? k + (pk & 0xFFu) : 128u + k;
        INF_AT(MLO, k) = (take && k < 32u) ? 1u << (k & 31u) : 0u;
        INF_AT(MHI, k) = (take && k >= 32u) ? 1u << (k & 31u) : 0u;
    }
#pragma unroll
    for (int lvl = 0; lvl < INF_VWALK_CFG; ++lvl) {
        InfVec J2, L2, H2;
        INF_FOREACH(k) {
            (void)k;
            const uint32_t j = INF_AT(J, k);
            const uint32_t jj = INF_GATHER(J, j & 63u), lo = INF_GATHER(MLO, j & 63u), hi = INF_GATHER(MHI, j & 63u);
            const bool in = j < 64u;
            INF_AT(J2, k) = in ? jj : j; INF_AT(L2, k) = INF_AT(MLO, k) | (in ? lo : 0u); INF_AT(H2, k) = INF_AT(MHI, k) | (in ? hi : 0u);
        }
        J = J2; MLO = L2; MHI = H2;
    }
    uint32_t a;
    for (;;) {
        do {
            starts |= (uint64_t)INF_GET(MLO, off) | ((uint64_t)INF_GET(MHI, off) << 32);
            off = INF_GET(J, off);
        } while (off < 64u);
        if (off >= 128u) off -= 128u;                                       // stands on a lane it cannot take
        a = INF_GET(PK, off & 63u);
#if INF_INWALK_CFG
        if (off >= 64u || (a >> 8) != (uint32_t)INF_K_OTHER) break;
        uint32_t kind, adv, ol = 0, val = 0;
        if (!inflate_long_code_here(S, bi, off, avail, kind, adv, ol, val)) break;
        a = (kind << 8) | adv;
        INF_SET(PK, off, a); INF_SET(VAL, off, val); INF_SET(OL, off, ol);
        if (kind == (uint32_t)INF_K_END) break;
        starts |= 1ull << off;
        off += adv;
        if (off >= 64u) break;
#else
        break;
#endif
    }
#else
    uint32_t a = INF_GET(PK, 0u);
#if INF_INWALK_CFG
    for (;;) {
#endif
    while ((a | (off << 3)) < ((uint32_t)INF_K_END << 8)) {
        starts |= 1ull << off;
        off += a & 0xFFu;
        a = INF_GET(PK, off & 63u);
    }
#if INF_INWALK_CFG
        // A lane the walk cannot take.  When it is a code the fast table does not hold, the symbol is decoded here -- its bits are
        // buffered, the lanes behind it have decoded what follows -- and the walk goes on; everything else (the buffered bits
        // end inside the symbol, a code nobody owns, the end of the block) ends the round as before.
        if (off >= 64u || (a >> 8) != (uint32_t)INF_K_OTHER) break;
        uint32_t kind, adv, ol = 0, val = 0;
        if (!inflate_long_code_here(S, bi, off, avail, kind, adv, ol, val)) break;
        a = (kind << 8) | adv;
        INF_SET(PK, off, a); INF_SET(VAL, off, val); INF_SET(OL, off, ol);
    }
#endif
#endif
    bool stopped = off < 64u;                                               // the walk met a symbol it cannot take, at bit offset off
    const uint32_t stopped_at = a;
    INF_STAT(inflate_stats().rounds++); INF_STAT(inflate_stats().round_symbols += (unsigned)__builtin_popcountll(starts)); INF_STAT(inflate_stats().slow_symbols += stopped ? 1 : 0);
    if (starts) {
        // every symbol's place in the output
        InfVec X, INC;
        INF_FOREACH(k) { INF_AT(X, k) = ((starts >> k) & 1ull) ? INF_AT(OL, k) : 0u; }
        INC = inf_scan(X);
        uint32_t total = INF_GET(INC, 63u);
        if (total > INF_ROUND_BYTES) {                                      // keep the round's output inside the ring: cut it at the symbol that crosses the line
            InfVec OVER;
            INF_FOREACH(k) { INF_AT(OVER, k) = (((starts >> k) & 1ull) && INF_AT(INC, k) > INF_ROUND_BYTES) ? 1u : 0u; }
            const uint32_t c = (uint32_t)__builtin_ctzll(inf_ballot(OVER));
            starts &= (1ull << c) - 1ull;
            total = INF_GET(INC, c) - INF_GET(X, c);
            off = c; stopped = false;                                       // (the lane index IS the bit offset: the next round starts at that symbol)
        }
        if (o.pos + total > o.out_len) { status = INF_ERR_OUTPUT; return false; }
        InfVec ISLIT, ISMATCH;
        INF_FOREACH(k) {
            const bool mine = (starts >> k) & 1ull;
            INF_AT(ISLIT, k) = (mine && (INF_AT(PK, k) >> 8) == (uint32_t)INF_K_LIT) ? 1u : 0u;
            INF_AT(ISMATCH, k) = (mine && (INF_AT(PK, k) >> 8) == (uint32_t)INF_K_MATCH) ? 1u : 0u;
        }
        uint64_t lit = inf_ballot(ISLIT), match = inf_ballot(ISMATCH);
        {                                                                   // a match that reaches before the block's first byte (all of them at once)
            InfVec BADM;
            INF_FOREACH(k) { (void)k; INF_AT(BADM, k) = (INF_AT(ISMATCH, k) && INF_AT(VAL, k) > o.pos + INF_AT(INC, k) - INF_AT(OL, k)) ? 1u : 0u; }
            if (inf_ballot(BADM)) { status = INF_ERR_DISTANCE; return false; }
        }
        // One pass for the whole round, one lane per OUTPUT byte, when the round puts out at most 64 bytes and no match reads what
        // this round writes (distance >= the match's end inside the round; 84 % of the rounds of a real file).  The symbols leave
        // (literal byte | distance) in the slot of their first output byte; a ballot over the slots is the mask of first bytes; a
        // byte lane finds its symbol as the highest first byte at or below it.  Every source is read before any byte is stored,
        // so a store cannot land on a ring slot that a match of the same round still has to read.
        bool one_pass = PAR && total <= 64u;                                       // (rounds of long matches -- a low-entropy file -- skip even the test)
        if (one_pass) {
            InfVec DEP;
            INF_FOREACH(k) { (void)k; INF_AT(DEP, k) = (INF_AT(ISMATCH, k) && INF_AT(VAL, k) < INF_AT(INC, k)) ? 1u : 0u; }
            one_pass = !inf_ballot(DEP);
        }
        if (one_pass) {
            InfVec FAR_;
            INF_FOREACH(k) { (void)k; INF_AT(FAR_, k) = (INF_AT(ISMATCH, k) && INF_AT(VAL, k) > INF_NEAR) ? 1u : 0u; }
            if (inf_ballot(FAR_)) {                                         // sources that left the ring are read from the stream (see inflate_copy)
#if defined(__HIP_DEVICE_COMPILE__)
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
#endif
            }
            const uint32_t r0 = o.pos;
            INF_FOREACH(k) { S.slot[k] = 0u; }
            INF_FOREACH(k) {
                if ((starts >> k) & 1ull)
                    S.slot[INF_AT(INC, k) - INF_AT(OL, k)] = 0x80000000u | (INF_AT(ISLIT, k) ? 0x40000000u : 0u) | (INF_AT(VAL, k) & 0xFFFFu);
            }
            // the slots were written by OTHER lanes of this wave: without a fence the compiler may forward a lane's own zero to the
            // read below (the accesses are plain LDS traffic of one wave, which the hardware runs in order -- what is needed is
            // that the compiler reloads).  This was the variant's failure on the device in profiles/r3_decode_ab.txt; the one-lane
            // host build cannot show it.
#if defined(__HIP_DEVICE_COMPILE__)
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#endif
            InfVec FIRST;
            INF_FOREACH(k) { INF_AT(FIRST, k) = S.slot[k] >> 31; }
            const uint64_t firsts = inf_ballot(FIRST);                      // bit 0 is set: the round's first symbol starts at its first byte
            InfVec BYTE_;
            INF_FOREACH(j) {
                uint32_t b = 0;
                if (j < total) {
                    const uint64_t le = firsts & ((2ull << j) - 1ull);
                    const uint32_t info = S.slot[63u - (uint32_t)__builtin_clzll(le)];
                    if (info & 0x40000000u) b = info & 0xFFu;
                    else {
                        const uint32_t dist = info & 0xFFFFu, src = r0 + j - dist;
                        b = (dist <= INF_NEAR) ? (uint32_t)S.ring[src & INF_RMASK] : (uint32_t)o.dst[src];
                    }
                }
                INF_AT(BYTE_, j) = b;
            }
            INF_FOREACH(j) { if (j < total) S.ring[(r0 + j) & INF_RMASK] = (uint8_t)INF_AT(BYTE_, j); }
        } else
        {
        while (match) {                                                     // in stream order: the literals before the next match, then the match
            const uint32_t m = (uint32_t)__builtin_ctzll(match);
            const uint64_t now = lit & ((1ull << m) - 1ull);
            INF_FOREACH(k) { if ((now >> k) & 1ull) S.ring[(o.pos + INF_AT(INC, k) - 1u) & INF_RMASK] = (uint8_t)INF_AT(VAL, k); }
            lit ^= now;
            const uint32_t mlen = INF_GET(OL, m);
            inflate_copy(S, o, o.pos + INF_GET(INC, m) - mlen, INF_GET(VAL, m), mlen);
            match &= match - 1ull;
        }
        INF_FOREACH(k) { if ((lit >> k) & 1ull) S.ring[(o.pos + INF_AT(INC, k) - 1u) & INF_RMASK] = (uint8_t)INF_AT(VAL, k); }      // the literals behind the last match
        }
        o.pos += total;
        while (o.pos - o.flushed >= INF_FLUSH) inflate_flush(S, o, INF_FLUSH);
    }
    bi.drop(off);
    if (!stopped) return true;
    if ((stopped_at >> 8) == (uint32_t)INF_K_END) { bi.drop(stopped_at & 0xFFu); return false; }
    return inflate_one_symbol(S, bi, o, status);
}

// Inflates `in_len` payload bytes at `in` into exactly `out_len` bytes at `dst` and checks their CRC-32 (inflate_crc_init
// has filled S.crc_tab).  Returns an InflateStatus (wave-uniform).
// The caller provides 16 readable bytes past the payload's end (the bit reader looks ahead by whole dwords).
template <bool PAR = (INF_PAR_COMMIT_CFG != 0)>
RSQC_INF_FN int inflate_block(InflateScratch &S, const uint8_t *in, uint32_t in_len, uint8_t *dst, uint32_t out_len, uint32_t crc32) {
    static const uint8_t kClOrder[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};

    const uintptr_t addr = (uintptr_t)in;
    const uint32_t lead = (uint32_t)(addr & 3u);
    InflateIn bi;
    bi.base = (const uint32_t *)(addr - lead);
    bi.n_words = (lead + in_len + 3u) / 4u + 2u;
    bi.seek(lead);
    InflateOut o{dst, out_len, 0u, 0u, 0xFFFFFFFFu};

    for (;;) {
        bi.refill();
        const uint32_t bfinal = bi.take(1), btype = bi.take(2);
        if (btype == 0u) {                                          // stored: LEN, ~LEN on a byte boundary, then the bytes
            bi.drop(bi.avail() & 7u);
            bi.refill();
            const uint32_t len = bi.take(16), nlen = bi.take(16);
            if ((len ^ nlen) != 0xFFFFu) return INF_ERR_STORED;
            if (o.pos + len > out_len) return INF_ERR_OUTPUT;
            const uint32_t at = bi.byte_pos();
            if (at + len > lead + in_len) return INF_ERR_INPUT;
            const uint8_t *src = (const uint8_t *)bi.base + at;
            for (uint32_t done = 0; done < len;) {
                const uint32_t step = INF_RING / 4u < 1024u ? INF_RING / 4u : 1024u, n = (len - done < step) ? len - done : step;      // (unflushed output stays inside the ring)
                for (uint32_t j = INF_LANE; j < n; j += INF_W) S.ring[(o.pos + j) & INF_RMASK] = src[done + j];
                o.pos += n; done += n;
                while (o.pos - o.flushed >= INF_FLUSH) inflate_flush(S, o, INF_FLUSH);
            }
            bi.seek(at + len);
        } else if (btype == 1u || btype == 2u) {
            uint32_t nlit, ndist;
            if (btype == 1u) {                                      // fixed code (RFC 1951 3.2.6)
                nlit = 288; ndist = 30;
                for (uint32_t s = INF_LANE; s < 320u; s += INF_W)
                    S.lens[s] = (uint8_t)(s < 144u ? 8 : s < 256u ? 9 : s < 280u ? 7 : s < 288u ? 8 : 5);
            } else {
                nlit = bi.take(5) + 257u; ndist = bi.take(5) + 1u;
                const uint32_t ncl = bi.take(4) + 4u;
                if (nlit > 286u || ndist > 30u) return INF_ERR_TABLE;
                for (uint32_t s = INF_LANE; s < 19u; s += INF_W) S.lens[s] = 0;
                for (uint32_t k = 0; k < ncl; ++k) {
                    bi.refill();
                    const uint32_t v = bi.take(3);
                    INF_ST(S.lens[kClOrder[k]] = (uint8_t)v);
                }
                // the code-length code is decoded with the distance tables' storage (7-bit codes fit the 8-bit fast table)
                if (!INF_BUILD(S.lens, 19, S.dcount, S.dsym, S.dfirst, S.didx, S.dfast, 7, S.offs, INF_T_PLAIN)) return INF_ERR_TABLE;
                uint32_t i = 0, prev = 0;
                while (i < nlit + ndist) {                         // (the code-length code's own lengths in lens[0..19) are not needed any more)
                    bi.refill();
                    const uint32_t s = inflate_symbol(bi, S.dfast, 7, S.dcount, S.dfirst, S.didx, S.dsym);
                    if (s < 16u) { INF_ST(S.lens[i] = (uint8_t)s); prev = s; ++i; continue; }
                    uint32_t rep, val = 0;
                    if (s == 16u) { if (i == 0u) return INF_ERR_TABLE; val = prev; rep = 3u + bi.take(2); }
                    else if (s == 17u) rep = 3u + bi.take(3);
                    else if (s == 18u) rep = 11u + bi.take(7);
                    else return INF_ERR_SYMBOL;
                    if (i + rep > nlit + ndist) return INF_ERR_TABLE;
                    for (uint32_t j = INF_LANE; j < rep; j += INF_W) S.lens[i + j] = (uint8_t)val;
                    i += rep; prev = val;
                }
                if (INF_UNI(S.lens[256]) == 0u) return INF_ERR_TABLE;                // no end-of-block code
            }
            if (!INF_BUILD(S.lens, nlit, S.lcount, S.lsym, S.lfirst, S.lidx, S.lfast, INF_LBITS, S.offs, INF_T_LITLEN)) return INF_ERR_TABLE;
            if (!INF_BUILD(S.lens + nlit, ndist, S.dcount, S.dsym, S.dfirst, S.didx, S.dfast, INF_DBITS, S.offs, INF_T_DIST)) return INF_ERR_TABLE;
            // ---- the symbols, in rounds (inflate_round)
            uint32_t status = INF_OK;
            while (inflate_round<PAR>(S, bi, o, status)) {}
            if (status) return (int)status;
        } else return INF_ERR_BTYPE;
        if (bfinal) break;
    }
    while (o.pos - o.flushed >= INF_FLUSH) inflate_flush(S, o, INF_FLUSH);
    inflate_flush(S, o, o.pos - o.flushed);                           // (the last, shorter piece)
    if (o.pos != out_len) return INF_ERR_OUTPUT;
    if (bi.byte_pos() > lead + in_len) return INF_ERR_INPUT;
    if ((o.crc ^ 0xFFFFFFFFu) != crc32) return INF_ERR_CRC;
    return INF_OK;
}

}  // namespace rsqc
