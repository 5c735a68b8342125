// rsqc_inflate.h -- DEFLATE (RFC 1951) decoder for the payload of ONE BGZF block, written for ONE wavefront.
//
// The data format on the input side of the per-read path: a BAM file is a chain of BGZF blocks, each a
// gzip member of at most 64 KiB of output whose payload is an independent DEFLATE stream (SAM spec 4.1; the reference
// reads it through htslib's bgzf.c behind SeqLib, src/BamReader.cpp:12-20).  Independent blocks are the parallelism:
// one wave decodes one block, a few hundred thousand blocks per 100 M records.
//
// Shape of the decoder on CDNA4:
//  * the bit reader, the Huffman walk and the output position are WAVE-UNIFORM values (every lane computes the same
//    thing; table entries come back from LDS through readfirstlane), so the compiler keeps them in SGPRs and the loop
//    runs on the scalar unit;
//  * the compressed bytes arrive as one coalesced 256-byte vector load per 64 dwords (lane l holds dword l of the
//    window, the next window is already in flight) and are handed to the bit reader with v_readlane;
//  * the last 32 KiB of output -- the whole DEFLATE history -- live in an LDS ring, so a match is an LDS-to-LDS copy done
//    by all lanes at once and never reads global memory back; the ring is written out to HBM 16 KiB at a time;
//  * LDS per wave: 32 KiB ring + 3.5 KiB of decoding tables = one wave per SIMD (4 per CU, 160 KiB).
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

namespace rsqc {

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

constexpr uint32_t INF_RING_BITS = 15, INF_RING = 1u << INF_RING_BITS, INF_RMASK = INF_RING - 1u;
constexpr uint32_t INF_LBITS = 10, INF_DBITS = 8;
constexpr uint32_t INF_FLUSH = 16384;

// decoding tables + output history of one wave (LDS on the device)
struct InflateScratch {
    uint8_t ring[INF_RING];
    uint16_t lfast[1u << INF_LBITS];     // index: next LBITS bits of the stream; (symbol << 4) | code length, 0 = longer code
    uint16_t dfast[1u << INF_DBITS];
    uint16_t lcount[16], dcount[16];     // codes per length
    uint16_t lsym[288], dsym[32];        // symbols by (length, symbol): canonical decoding of the codes the fast table does not hold
    uint8_t lens[320];                   // code lengths of the block being set up
    uint16_t offs[16];                   // first slot of a length in lsym / dsym while a table is built
    uint32_t crc_tab[256];               // CRC-32 (reflected 0xEDB88320), one byte per step
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
    uint64_t buf; uint32_t cnt;
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
        const uint32_t lead = byte_pos & 3u;
        const uint32_t w = next32();
        buf = (uint64_t)(w >> (8u * lead));
        cnt = 32u - 8u * lead;
    }
    RSQC_INF_FN void refill() { if (cnt < 32u) { buf |= (uint64_t)next32() << cnt; cnt += 32u; } }   // afterwards cnt >= 32
    RSQC_INF_FN uint32_t peek(uint32_t n) const { return (uint32_t)buf & ((1u << n) - 1u); }
    RSQC_INF_FN void drop(uint32_t n) { buf >>= n; cnt -= n; }
    RSQC_INF_FN uint32_t take(uint32_t n) { const uint32_t v = peek(n); drop(n); return v; }
    RSQC_INF_FN uint32_t byte_pos() const { return next * 4u - cnt / 8u; }    // first byte not consumed yet (whole bytes left in buf)
};

// ---- Huffman tables ------------------------------------------------------------------------------------------
// lens[0..n): code lengths.  Builds count[]/sym[] (canonical order) and the fast table of `fbits` bits.  false = the
// lengths over-subscribe the code space (zlib: "invalid code lengths set").  An incomplete set is accepted, as zlib
// accepts a single distance code; a code nobody owns is an error when the stream uses it.
RSQC_INF_FN bool inflate_build(const uint8_t *lens, uint32_t n, uint16_t *count, uint16_t *sym, uint16_t *fast, uint32_t fbits, uint16_t *offs) {
    for (uint32_t l = INF_LANE; l < 16u; l += INF_W) count[l] = 0;
    for (uint32_t k = INF_LANE; k < (1u << fbits); k += INF_W) fast[k] = 0;
    for (uint32_t s = 0; s < n; ++s) { const uint32_t l = INF_UNI(lens[s]); INF_ST(count[l]++); }
    if (INF_UNI(count[0]) == n) return true;                       // no codes at all: legal as long as none is used
    int32_t left = 1;
    uint32_t run = 0;
    for (uint32_t l = 1; l <= 15u; ++l) {
        const uint32_t c = INF_UNI(count[l]);
        left = (left << 1) - (int32_t)c;
        if (left < 0) return false;
        INF_ST(offs[l] = (uint16_t)run);
        run += c;
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
            const uint16_t e = (uint16_t)((s << 4) | l);
            for (uint32_t t = rev + (INF_LANE << l); t < (1u << fbits); t += (INF_W << l)) fast[t] = e;
        }
        code <<= 1;
    }
    return true;
}

// one symbol: the fast table, or bit by bit for a code longer than the table's index
RSQC_INF_FN uint32_t inflate_symbol(InflateIn &in, const uint16_t *fast, uint32_t fbits, const uint16_t *count, const uint16_t *sym) {
    const uint32_t e = INF_UNI(fast[in.peek(fbits)]);
    if (e) { in.drop(e & 15u); return e >> 4; }
    uint32_t code = 0, first = 0, index = 0;
    uint64_t b = in.buf;
    for (uint32_t l = 1; l <= 15u; ++l) {
        code |= (uint32_t)b & 1u; b >>= 1;
        const uint32_t c = INF_UNI(count[l]);
        if (code - first < c) { in.drop(l); return INF_UNI(sym[index + (code - first)]); }
        index += c; first = (first + c) << 1; code <<= 1;
    }
    return 0xFFFFu;                                                 // no symbol owns this code
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
    constexpr uint32_t kFullPiece = crc_xpow(8ull * (INF_FLUSH / 64u));
    uint32_t m = (n == INF_FLUSH) ? kFullPiece : crc_xpow(8ull * plen);                                // x^(8 plen): a constant for full chunks
    for (uint32_t d = 1; d < 64u; d <<= 1) {                                                          // r[l] = r[l] * x^(8 plen d) + r[l + d]
        const uint32_t hi = inflate_lane_down(r, d);
        r = crc_mulmod(r, m) ^ hi;
        m = crc_mulmod(m, m);
    }
    r = INF_UNI(r);
#endif
    constexpr uint32_t kFullChunk = crc_xpow(8ull * INF_FLUSH);
    const uint32_t adv = (n == INF_FLUSH) ? kFullChunk : crc_xpow(8ull * n);
    o.crc = crc_mulmod(o.crc, adv) ^ r;
    o.flushed += n;
}

// Inflates `in_len` payload bytes at `in` into exactly `out_len` bytes at `dst` and checks their CRC-32 (inflate_crc_init
// has filled S.crc_tab).  Returns an InflateStatus (wave-uniform).
// The caller provides 16 readable bytes past the payload's end (the bit reader looks ahead by whole dwords).
RSQC_INF_FN int inflate_block(InflateScratch &S, const uint8_t *in, uint32_t in_len, uint8_t *dst, uint32_t out_len, uint32_t crc32) {
    static const uint16_t kLenBase[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
    static const uint8_t kLenExtra[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
    static const uint16_t kDistBase[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
    static const uint8_t kDistExtra[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};
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
            bi.drop(bi.cnt & 7u);
            bi.refill();
            const uint32_t len = bi.take(16), nlen = bi.take(16);
            if ((len ^ nlen) != 0xFFFFu) return INF_ERR_STORED;
            if (o.pos + len > out_len) return INF_ERR_OUTPUT;
            const uint32_t at = bi.byte_pos();
            if (at + len > lead + in_len) return INF_ERR_INPUT;
            const uint8_t *src = (const uint8_t *)bi.base + at;
            for (uint32_t done = 0; done < len;) {
                const uint32_t n = (len - done < INF_FLUSH) ? len - done : INF_FLUSH;
                for (uint32_t j = INF_LANE; j < n; j += INF_W) S.ring[(o.pos + j) & INF_RMASK] = src[done + j];
                o.pos += n; done += n;
                if (o.pos - o.flushed >= INF_FLUSH) inflate_flush(S, o, INF_FLUSH);
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
                if (!inflate_build(S.lens, 19, S.dcount, S.dsym, S.dfast, 7, S.offs)) return INF_ERR_TABLE;
                uint32_t i = 0, prev = 0;
                while (i < nlit + ndist) {                         // (the code-length code's own lengths in lens[0..19) are not needed any more)
                    bi.refill();
                    const uint32_t s = inflate_symbol(bi, S.dfast, 7, S.dcount, S.dsym);
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
            if (!inflate_build(S.lens, nlit, S.lcount, S.lsym, S.lfast, INF_LBITS, S.offs)) return INF_ERR_TABLE;
            if (!inflate_build(S.lens + nlit, ndist, S.dcount, S.dsym, S.dfast, INF_DBITS, S.offs)) return INF_ERR_TABLE;
            for (;;) {
                bi.refill();
                uint32_t s = inflate_symbol(bi, S.lfast, INF_LBITS, S.lcount, S.lsym);
                if (s < 256u) {
                    if (o.pos >= out_len) return INF_ERR_OUTPUT;
                    INF_ST(S.ring[o.pos & INF_RMASK] = (uint8_t)s);
                    o.pos++;
                    if (o.pos - o.flushed >= INF_FLUSH) inflate_flush(S, o, INF_FLUSH);
                    continue;
                }
                if (s == 256u) break;
                s -= 257u;
                if (s >= 29u) return INF_ERR_SYMBOL;
                const uint32_t len = kLenBase[s] + bi.take(kLenExtra[s]);
                bi.refill();
                const uint32_t ds = inflate_symbol(bi, S.dfast, INF_DBITS, S.dcount, S.dsym);
                if (ds >= 30u) return INF_ERR_SYMBOL;
                const uint32_t dist = kDistBase[ds] + bi.take(kDistExtra[ds]);
                if (dist > o.pos) return INF_ERR_DISTANCE;
                if (o.pos + len > out_len) return INF_ERR_OUTPUT;
                // out[pos + j] = out[pos + j - dist].  A match that overlaps its own output (dist < len) is periodic with
                // period dist: the copy offset doubles (dist, 2 dist, ...) while the bytes written so far allow it, so that every
                // pass only reads bytes that exist and a run of one repeated byte takes ~10 passes instead of 258 steps
                uint32_t off = dist, done = 0;
                while (done < len) {
                    const uint32_t n = (len - done < off) ? len - done : off;
                    for (uint32_t j = INF_LANE; j < n; j += INF_W)
                        S.ring[(o.pos + done + j) & INF_RMASK] = S.ring[(o.pos + done + j - off) & INF_RMASK];
                    done += n;
                    if (off < 64u) off <<= 1;
                }
                o.pos += len;
                if (o.pos - o.flushed >= INF_FLUSH) inflate_flush(S, o, INF_FLUSH);
            }
        } else return INF_ERR_BTYPE;
        if (bfinal) break;
    }
    if (o.pos - o.flushed >= INF_FLUSH) inflate_flush(S, o, INF_FLUSH);
    inflate_flush(S, o, o.pos - o.flushed);
    if (o.pos != out_len) return INF_ERR_OUTPUT;
    if (bi.byte_pos() > lead + in_len) return INF_ERR_INPUT;
    if ((o.crc ^ 0xFFFFFFFFu) != crc32) return INF_ERR_CRC;
    return INF_OK;
}

}  // namespace rsqc
