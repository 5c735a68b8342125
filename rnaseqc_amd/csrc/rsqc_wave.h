// rsqc_wave.h -- 64-lane wavefront helpers shared by the kernels (rsqc_kernels.hip, rsqc_k1.h).  Written against the
// HIP wave intrinsics (__ballot, __shfl*, mbcnt); tests/hostemu/k1_emu.cpp compiles the same source for the host on top
// of a 64-fiber wave emulation that provides those intrinsics.
#pragma once

namespace rsqc {

__device__ __forceinline__ int lane_id() { return (int)(threadIdx.x & 63u); }

// the value of the lane below (lane 0 keeps its own): ONE DPP move (wave_shr:1) instead of a trip through the LDS crossbar
__device__ __forceinline__ uint32_t lane_below(uint32_t v) {
#if defined(__HIP_DEVICE_COMPILE__)
    return (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x138, 0xf, 0xf, false);
#else
    return __shfl_up(v, 1, 64);
#endif
}

// bit-field helpers that map to ONE vector instruction each (v_bfe_u32 / v_bfe_i32 / v_bfi_b32).  The hardware takes the low
// five bits of the offset operand, so a 16-bit table repeated in both halves of `src` can be indexed by a value whose bit 4
// is garbage: bfe_u(0x01810181, cigar_word, 1) is "operation code in {M, =, X}" without masking the code out first.
__device__ __forceinline__ uint32_t bfe_u(uint32_t src, uint32_t off, uint32_t width) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_ubfe(src, off, width);
#else
    return (src >> (off & 31u)) & ((width & 31u) ? ((1u << (width & 31u)) - 1u) : 0u);
#endif
}
// the same, sign-extended: a 1-bit field becomes an all-ones / all-zeros select mask
__device__ __forceinline__ uint32_t bfe_m(uint32_t src, uint32_t off) {
#if defined(__HIP_DEVICE_COMPILE__)
    return (uint32_t)__builtin_amdgcn_sbfe((int)src, off, 1u);
#else
    return ((src >> (off & 31u)) & 1u) ? 0xFFFFFFFFu : 0u;
#endif
}
__device__ __forceinline__ uint32_t bfi(uint32_t mask, uint32_t a, uint32_t b) { return (mask & a) | (~mask & b); }   // v_bfi_b32

__device__ __forceinline__ uint32_t mask_rank(uint64_t m) {      // #set bits below this lane
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
}

template <class T>
__device__ __forceinline__ T wave_sum(T v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
// the value of one lane (wave-uniform index) in every lane: v_readlane, no trip through the LDS crossbar
__device__ __forceinline__ uint32_t lane_value(uint32_t v, int lane) {
#if defined(__HIP_DEVICE_COMPILE__)
    return (uint32_t)__builtin_amdgcn_readlane((int)v, lane);
#else
    return __shfl(v, lane, 64);
#endif
}
// maximum over the wave, in every lane.  Six DPP steps (row_shr 1 / 2 / 4 / 8 inside the rows of 16 lanes, then lane 15 of
// rows 0 and 2 into rows 1 and 3, then lane 31 into rows 2 and 3) leave it in lane 63 -- no ds_bpermute round trips, which
// were a chain of six dependent LDS instructions per tile.
// (all 64 lanes must be active: the result is read from lane 63)
__device__ __forceinline__ uint32_t wave_max_u32_full(uint32_t v) {
#if defined(__HIP_DEVICE_COMPILE__)
#define RSQC_DPP_MAX(ctrl, rmask) { const uint32_t t = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, ctrl, rmask, 0xf, false); v = t > v ? t : v; }
    RSQC_DPP_MAX(0x111, 0xf) RSQC_DPP_MAX(0x112, 0xf) RSQC_DPP_MAX(0x114, 0xf) RSQC_DPP_MAX(0x118, 0xf)
    RSQC_DPP_MAX(0x142, 0xa) RSQC_DPP_MAX(0x143, 0xc)
#undef RSQC_DPP_MAX
    return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
#else
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { uint32_t t = __shfl_xor(v, o, 64); v = t > v ? t : v; }
    return v;
#endif
}
// minimum over the wave (all 64 lanes active): the same six DPP steps (lanes without a source keep their own value: old = v)
__device__ __forceinline__ uint32_t wave_min_u32_full(uint32_t v) {
#if defined(__HIP_DEVICE_COMPILE__)
#define RSQC_DPP_MIN(ctrl, rmask) { const uint32_t t = (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, ctrl, rmask, 0xf, false); v = t < v ? t : v; }
    RSQC_DPP_MIN(0x111, 0xf) RSQC_DPP_MIN(0x112, 0xf) RSQC_DPP_MIN(0x114, 0xf) RSQC_DPP_MIN(0x118, 0xf)
    RSQC_DPP_MIN(0x142, 0xa) RSQC_DPP_MIN(0x143, 0xc)
#undef RSQC_DPP_MIN
    return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
#else
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { uint32_t t = __shfl_xor(v, o, 64); v = t < v ? t : v; }
    return v;
#endif
}
// sum over the wave (all 64 lanes active), in every lane: the same six DPP steps with an add (a ds_bpermute butterfly is six
// dependent LDS round trips and five address computations per step)
__device__ __forceinline__ uint32_t wave_sum_u32_full(uint32_t v) {
#if defined(__HIP_DEVICE_COMPILE__)
#define RSQC_DPP_ADD(ctrl, rmask) { v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, ctrl, rmask, 0xf, false); }
    RSQC_DPP_ADD(0x111, 0xf) RSQC_DPP_ADD(0x112, 0xf) RSQC_DPP_ADD(0x114, 0xf) RSQC_DPP_ADD(0x118, 0xf)
    RSQC_DPP_ADD(0x142, 0xa) RSQC_DPP_ADD(0x143, 0xc)
#undef RSQC_DPP_ADD
    return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
#else
    return wave_sum(v);
#endif
}
__device__ __forceinline__ uint32_t wave_max_u32(uint32_t v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { uint32_t t = __shfl_xor(v, o, 64); v = t > v ? t : v; }
    return v;
}
__device__ __forceinline__ uint32_t wave_min_u32(uint32_t v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { uint32_t t = __shfl_xor(v, o, 64); v = t < v ? t : v; }
    return v;
}
__device__ __forceinline__ uint32_t wave_inclusive_scan_u32(uint32_t v) {
    const int l = lane_id();
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { uint32_t t = __shfl_up(v, o, 64); if (l >= o) v += t; }
    return v;
}

// inclusive prefix sum over the wave (all 64 lanes active) on DPP moves: row_shr 1 / 2 / 4 / 8 inside the rows of 16 lanes, then
// row_bcast:15 (lane 15 of a row into the next row) and row_bcast:31 (lane 31 into rows 2 and 3) -- six vector instructions instead
// of six trips through the LDS crossbar
__device__ __forceinline__ uint32_t wave_inclusive_scan_u32_dpp(uint32_t v) {
#if defined(__HIP_DEVICE_COMPILE__)
#define RSQC_DPP_SCAN(ctrl, rmask) { v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, ctrl, rmask, 0xf, false); }
    RSQC_DPP_SCAN(0x111, 0xf) RSQC_DPP_SCAN(0x112, 0xf) RSQC_DPP_SCAN(0x114, 0xf) RSQC_DPP_SCAN(0x118, 0xf)
    RSQC_DPP_SCAN(0x142, 0xa) RSQC_DPP_SCAN(0x143, 0xc)
#undef RSQC_DPP_SCAN
    return v;
#else
    return wave_inclusive_scan_u32(v);
#endif
}
// the value of lane `src` (per-lane index, taken modulo 64): one ds_bpermute
__device__ __forceinline__ uint32_t lane_gather(uint32_t v, uint32_t src) {
#if defined(__HIP_DEVICE_COMPILE__)
    return (uint32_t)__builtin_amdgcn_ds_bpermute((int)((src & 63u) << 2), (int)v);
#else
    return __shfl(v, (int)(src & 63u), 64);
#endif
}

// One atomic per distinct key in the wave.  Must be called by all 64 lanes (converged).
template <class F>
__device__ __forceinline__ void wave_aggregate(bool valid, uint32_t key, uint64_t flagmask, F &&leader) {
    uint64_t todo = __ballot(valid);
    const int l = lane_id();
    while (todo) {
        const int lead = __ffsll((unsigned long long)todo) - 1;
        const uint32_t k0 = __shfl(key, lead, 64);
        const uint64_t same = __ballot(valid && key == k0);
        if (l == lead) leader(k0, (uint32_t)__popcll(same), (uint32_t)__popcll(same & flagmask));
        todo &= ~same;
    }
}

// Runs of equal keys in lane order.  The input is coordinate-sorted, so records that hit the
// same exon / gene / coverage slot sit in neighbouring lanes: merging each run into one atomic
// removes the same-address serialisation on highly expressed genes in O(1) instructions.
struct Run { bool head; uint32_t count; int end; uint64_t mask; };
// `vmask`: the lanes that take part, as a lane mask (rsqc_read.h, LaneMask: conditions of the per-record kernel are masks in
// scalar registers; a ballot of a derived per-lane bool would cost a v_cndmask + v_cmp round trip through a VGPR)
__device__ __forceinline__ Run make_run(uint64_t vmask, uint32_t key) {
    const int l = lane_id();
    const uint32_t pk = lane_below(key);
    const uint64_t neq = WaveSink::prim(pk != key).m;                // (lane 0 compares with itself: its bit of vmask << 1 is 0)
    const uint64_t headm = vmask & (~(vmask << 1) | neq);
    Run r;
    r.head = WaveSink::lane(LaneMask{headm});
    const uint64_t stop = headm | ~vmask;                            // lanes that end the run before them
    const uint64_t above = l == 63 ? 0ull : stop & ~((2ull << l) - 1ull);
    r.end = above ? __ffsll((unsigned long long)above) - 1 : 64;
    r.count = (uint32_t)(r.end - l);
    const uint64_t upto = r.end == 64 ? ~0ull : ((1ull << r.end) - 1ull);
    r.mask = upto & ~((1ull << l) - 1ull);
    return r;
}
__device__ __forceinline__ Run make_run(bool valid, uint32_t key) { return make_run(WaveSink::prim(valid).m, key); }

// The same runs for callers that need only the LENGTH of the run at its first lane (the per-record kernel's commit: one add of
// `count` per run).  make_run builds the run's end and lane mask from 64-bit per-lane shifts, subtractions and selects -- about
// 35 vector instructions, and the commit of ONE slot called it three times (exon, coverage +1, coverage -1).  Here the lanes
// that CONTINUE a run are a scalar mask, and a first lane counts the consecutive continuing lanes above it:
//     cont  = lanes whose key equals the key of the (valid) lane below            (one DPP move, one compare, scalar ands)
//     count = 1 + trailing ones of (cont >> (lane + 1))                           (one 64-bit shift, two nots, two ffbl, min, adds)
struct RunLite { bool head; uint32_t count; };
__device__ __forceinline__ uint64_t run_cont_mask(uint64_t vmask, uint32_t key) {
    const uint32_t pk = lane_below(key);
    return vmask & (vmask << 1) & ~WaveSink::prim(pk != key).m;      // (lane 0 has nothing below: bit 0 of vmask << 1 is 0)
}
__device__ __forceinline__ uint32_t run_length_at(uint64_t cont) {   // for a first lane: lanes of its run
    const uint64_t t = (cont >> 1) >> (uint32_t)lane_id();           // bit 0: does lane + 1 continue?  (bit 63 of cont >> 1 is 0: the count ends)
    const uint32_t lo = ~(uint32_t)t, hi = ~(uint32_t)(t >> 32);
#if defined(__HIP_DEVICE_COMPILE__)
    uint32_t a, fh;
    asm("v_ffbl_b32 %0, %1" : "=v"(a) : "v"(lo));                    // index of the lowest set bit, 0xFFFFFFFF for 0
    asm("v_ffbl_b32 %0, %1" : "=v"(fh) : "v"(hi));
    const uint32_t b2 = 32u + fh;
#else
    const uint32_t a = lo ? (uint32_t)__builtin_ctz(lo) : 0xFFFFFFFFu, b2 = 32u + (hi ? (uint32_t)__builtin_ctz(hi) : 0xFFFFFFFFu);
#endif
    return 1u + (a < b2 ? a : b2);
}
__device__ __forceinline__ RunLite make_run_lite(uint64_t vmask, uint32_t key) {
    const uint64_t cont = run_cont_mask(vmask, key);
    RunLite r;
    r.head = WaveSink::lane(LaneMask{vmask & ~cont});
    r.count = run_length_at(cont);
    return r;
}
// lanes of `m` among the `count` lanes that start at this lane (count <= 64 - lane)
__device__ __forceinline__ uint32_t run_popcount(uint64_t m, uint32_t count) {
    const uint64_t t = m >> (uint32_t)lane_id();
    const uint64_t keep = count >= 64u ? ~0ull : ((1ull << count) - 1ull);
    return (uint32_t)__popcll(t & keep);
}
// sum of v over the run that starts at this (head) lane
__device__ __forceinline__ double run_sum_f64(double v, const Run &r) {
    const int l = lane_id();
    double sc = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const double t = __shfl_up(sc, o, 64); if (l >= o) sc += t; }
    const double at_end = __shfl(sc, r.end - 1, 64);
    return at_end - (sc - v);
}

// integer variant: sum of v over the run that starts at this (head) lane
__device__ __forceinline__ uint32_t run_sum_u32(uint32_t v, const Run &r) {
    const uint32_t sc = wave_inclusive_scan_u32(v);
    const uint32_t at_end = __shfl(sc, r.end - 1, 64);
    return at_end - (sc - v);
}

#ifndef RSQC_COV_MERGE_MIN
#define RSQC_COV_MERGE_MIN 1
#endif
// cov[idx] += sign * (number of lanes of the run) with identical neighbouring slots merged into one atomic.
// Most tiles have no two neighbouring lanes on the same slot: one shuffle and one ballot decide that, and only
// then is the run structure built.
__device__ __forceinline__ void cov_add_merged(uint32_t *cov, uint64_t vmask, uint32_t idx, uint32_t sign) {
    const uint64_t cont = run_cont_mask(vmask, idx);                 // a lane on the slot of the lane below it
    if (__popcll(cont) < RSQC_COV_MERGE_MIN) {
        if (WaveSink::lane(LaneMask{vmask})) atomicAdd(&cov[idx], sign);
    } else {
        const uint32_t n = run_length_at(cont);
        if (WaveSink::lane(LaneMask{vmask & ~cont})) atomicAdd(&cov[idx], sign * n);
    }
}
__device__ __forceinline__ void cov_add_merged(uint32_t *cov, bool valid, uint32_t idx, uint32_t sign) {
    cov_add_merged(cov, WaveSink::prim(valid).m, idx, sign);
}

// last segment whose start <= i (wave-uniform i -> scalar loads)
__device__ __forceinline__ uint32_t find_segment(const DevBatch &b, uint64_t i) {
    uint32_t lo = 0, hi = b.n_seg;
    while (hi - lo > 1) { const uint32_t m = (lo + hi) >> 1; if (b.seg_start[m] <= i) lo = m; else hi = m; }
    return lo;
}


}  // namespace rsqc
