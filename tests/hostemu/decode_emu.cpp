// decode_emu.cpp -- TEST HARNESS ONLY (never loaded by the product).
//
// Compiles the product's device-decode cores (rnaseqc_amd/csrc/rsqc_inflate.h: the DEFLATE decoder one wavefront runs
// per BGZF block; rsqc_bamrec.h: BAM record framing and parsing) with g++ as a wave of ONE lane, so that what the HIP
// kernels execute can be diffed against zlib and against the host BAM reader in the GPU-less build container.
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../../rnaseqc_amd/csrc/rsqc_inflate.h"

using namespace rsqc;

extern "C" __attribute__((visibility("default")))
int emu_inflate(const uint8_t *in, uint32_t in_len, uint8_t *out, uint32_t out_len, uint32_t crc32) {
    static InflateScratch S;
    inflate_crc_init(S);
    // the decoder may look 16 bytes past the payload
    std::vector<uint8_t> padded((size_t)in_len + 32, 0);
    memcpy(padded.data() + 3, in, in_len);                       // (an odd alignment on purpose)
    return inflate_block(S, padded.data() + 3, in_len, out, out_len, crc32);
}

// the 64-lane form of inflate_flush's CRC step (pieces, right alignment, six-level tree), with the lanes as an array:
// checks the arithmetic the device build uses where the one-lane host build takes a shortcut
extern "C" __attribute__((visibility("default")))
uint32_t emu_crc_wave64(const uint8_t *data, uint32_t n, uint32_t crc_before) {
    static InflateScratch S;
    inflate_crc_init(S);
    const uint32_t W = 64, plen = (n + W - 1u) / W, pad = plen * W - n;
    uint32_t r[64];
    for (uint32_t l = 0; l < W; ++l) {
        uint32_t x = 0;
        for (uint32_t k = 0; k < plen; ++k) {
            const uint32_t v = l * plen + k;
            if (v >= pad) x = S.crc_tab[(x ^ data[v - pad]) & 0xFFu] ^ (x >> 8);
        }
        r[l] = x;
    }
    uint32_t m = crc_xpow(8ull * plen);
    for (uint32_t d = 1; d < 64u; d <<= 1) {
        uint32_t nr[64];
        for (uint32_t l = 0; l < W; ++l) { const uint32_t hi = l + d < W ? r[l + d] : r[l]; nr[l] = crc_mulmod(r[l], m) ^ hi; }
        memcpy(r, nr, sizeof r);
        m = crc_mulmod(m, m);
    }
    return crc_mulmod(crc_before, crc_xpow(8ull * n)) ^ r[0];
}
