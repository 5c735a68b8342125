// rsqc_decode.hip -- device-side BAM decode: BGZF inflate, record framing and record parsing on MI355X (gfx950).
// The per-thread bodies are in rsqc_inflate.h / rsqc_bamrec.h / rsqc_decode.h (shared with the host emulation of the
// tests); this file holds the kernels around them and their launches.
//
// Input side of the per-read path (the reference: SeqLib/htslib behind src/BamReader.cpp:12-20, one
// thread).  What the host still does: read the file and hop over the BGZF block headers.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

#include "rsqc_inflate.h"
#include "rsqc_decode.h"
#include "rsqc_device.h"

namespace rsqc {

// ---- K0: one wave per BGZF block (work handed out by an atomic counter: block decode times differ) -----------------
// LDS: 8.7 KB per wave, registers capped for four waves per SIMD (sixteen per CU): the decoder is a chain of dependent scalar
// instructions, so the waves of a SIMD take turns in its issue slots.  WPW waves share a workgroup, each with its own scratch.
template <int WPW>
__global__ __launch_bounds__(64 * WPW) __attribute__((amdgpu_waves_per_eu(4, 4))) void bgzf_inflate_kernel(const uint8_t *__restrict__ in, const DevBgzfBlock *__restrict__ blk, uint32_t n_blk,
                                                                uint8_t *__restrict__ out, DecodeSummary *sum) {
    __shared__ InflateScratch SS[WPW];
    InflateScratch &S = SS[WPW == 1 ? 0 : (threadIdx.x >> 6)];
    inflate_crc_init(S);
    for (;;) {
        uint32_t b = 0;
        if (INF_LANE == 0u) b = atomicAdd(&sum->next_block, 1u);
        b = INF_UNI(b);
        if (b >= n_blk) break;
        const DevBgzfBlock k = blk[b];
        const int rc = inflate_block(S, in + k.in_off, k.in_len, out + k.out_off, k.out_len, k.crc);
        if (rc && INF_LANE == 0u) {
            atomicCAS(&sum->inflate_fail, 0u, ((b + 1u) << 4) | (uint32_t)rc);
            atomicOr(&sum->status, DEC_ST_INFLATE);
        }
    }
}

// ---- frame: one thread per segment -------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void bam_frame_kernel(DecodeWindow W) {
    const uint32_t s = blockIdx.x * 256u + threadIdx.x;
    if (s < W.n_seg) decode_frame_one(W, s);
}

// ---- chain: one workgroup ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void bam_chain_kernel(DecodeWindow W) {
    __shared__ uint32_t s_ok, s_consumed, s_bad;
    __shared__ uint32_t s_rec[1024], s_ops[1024];
    const uint32_t t = threadIdx.x, T = 1024;
    if (t == 0) { s_ok = 1; s_bad = 0; s_consumed = W.start; }
    __syncthreads();
    bool ok = true;
    for (uint32_t s = t; s < W.n_seg; s += T) ok = ok && decode_guess_confirmed(W, s);
    if (!ok) atomicAnd(&s_ok, 0u);
    __syncthreads();
    if (t == 0 && W.n_seg) {
        if (s_ok) s_consumed = W.seg[W.n_seg - 1].land;
        else { uint32_t bad = 0; s_consumed = bam_verify_chain(W.buf, W.seg, W.n_seg, W.start, DEC_SEG_BYTES, W.end, bad); s_bad = bad; }
    }
    __syncthreads();
    // exclusive sums of the per-segment counts: a contiguous run of segments per thread, the 1024 partial sums by thread 0
    const uint32_t per = (W.n_seg + T - 1) / T, lo = min(W.n_seg, t * per), hi = min(W.n_seg, lo + per);
    uint32_t nr = 0, no = 0;
    for (uint32_t s = lo; s < hi; ++s) { nr += W.seg[s].n_rec; no += W.seg[s].n_ops; }
    s_rec[t] = nr; s_ops[t] = no;
    __syncthreads();
    if (t == 0) {
        uint32_t a = 0, b = 0;
        for (uint32_t k = 0; k < T; ++k) { const uint32_t x = s_rec[k], y = s_ops[k]; s_rec[k] = a; s_ops[k] = b; a += x; b += y; }
        W.sum->n_rec = a; W.sum->n_ops = b; W.sum->consumed_end = s_consumed;
        if (s_bad) atomicOr(&W.sum->status, DEC_ST_BAD_RECORD);
    }
    __syncthreads();
    nr = s_rec[t]; no = s_ops[t];
    for (uint32_t s = lo; s < hi; ++s) { W.seg_rec0[s] = nr; W.seg_ops0[s] = no; nr += W.seg[s].n_rec; no += W.seg[s].n_ops; }
}

__global__ __launch_bounds__(256) void bam_offsets_kernel(DecodeWindow W) {
    if (W.sum->status) return;
    const uint32_t s = blockIdx.x * 256u + threadIdx.x;
    if (s < W.n_seg) decode_offsets_one(W, s);
}

// ---- parse: one thread per record, grid-stride (the record count is only known on the device) ---------------------
__global__ __launch_bounds__(256) void bam_parse_kernel(DecodeWindow W) {
    if (W.sum->status) return;
    const uint32_t n = W.sum->n_rec;
    uint32_t raise = 0; bool unsorted = false;
    for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < n; i += gridDim.x * 256u) raise |= decode_parse_one(W, i, unsorted);
    if (raise) atomicOr(&W.sum->status, raise);
    if (unsorted) W.sum->unsorted = 1u;
}

// ---- lists: one workgroup ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void bam_lists_kernel(DecodeWindow W) {
    __shared__ uint32_t s_seg[1024], s_wide[1024], s_bad[1024];
    __shared__ int32_t s_last[1024];
    if (W.sum->status) return;
    const uint32_t t = threadIdx.x, T = 1024, n = W.sum->n_rec;
    const uint32_t per = (n + T - 1) / T, lo = min(n, t * per), hi = min(n, lo + per);
    DecodeListCounts c;
    decode_lists_count(W, lo, hi, c);
    s_seg[t] = c.seg; s_wide[t] = c.wide; s_bad[t] = c.bad; s_last[t] = c.last_judged;
    __syncthreads();
    if (t == 0) {
        DecodeListCounts run{0, 0, 0, -1};
        for (uint32_t k = 0; k < T; ++k) {
            const uint32_t a = s_seg[k], b = s_wide[k], d = s_bad[k];
            s_seg[k] = run.seg; s_wide[k] = run.wide; s_bad[k] = run.bad;
            run.seg += a; run.wide += b; run.bad += d;
            if (s_last[k] >= 0) run.last_judged = s_last[k];
        }
        decode_lists_finish(W, n, run);
    }
    __syncthreads();
    decode_lists_write(W, lo, hi, DecodeListCounts{s_seg[t], s_wide[t], s_bad[t], -1});
}

// ---- launches ----------------------------------------------------------------------------------------------------
void launch_bgzf_inflate(hipStream_t s, const uint8_t *in, const DevBgzfBlock *blk, uint32_t n_blk, uint8_t *out, DecodeSummary *sum) {
    if (!n_blk) return;
    // four waves per SIMD of the chip; the counter feeds them
    static const int wpw = getenv("RSQC_INFLATE_WPW") ? atoi(getenv("RSQC_INFLATE_WPW")) : 4;
    static bool told = false;
    if (!told && getenv("RSQC_DECODE_PROFILE")) {
        int a = 0, b = 0;
        (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&a, bgzf_inflate_kernel<1>, 64, 0);
        (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&b, bgzf_inflate_kernel<4>, 256, 0);
        fprintf(stderr, "[decode] inflate kernel: %d workgroups of 1 wave or %d of 4 waves per CU; running %d waves per workgroup\n", a, b, wpw);
        told = true;
    }
    if (wpw == 1) {
        const uint32_t grid = n_blk < 256u * 16u ? n_blk : 256u * 16u;
        bgzf_inflate_kernel<1><<<grid, 64, 0, s>>>(in, blk, n_blk, out, sum);
    } else {
        const uint32_t need = (n_blk + 3u) / 4u, grid = need < 256u * 4u ? need : 256u * 4u;
        bgzf_inflate_kernel<4><<<grid, 256, 0, s>>>(in, blk, n_blk, out, sum);
    }
}
void launch_decode_window(hipStream_t s, const DecodeWindow &W) {
    if (W.n_seg) bam_frame_kernel<<<(W.n_seg + 255u) / 256u, 256, 0, s>>>(W);
    bam_chain_kernel<<<1, 1024, 0, s>>>(W);
    if (W.n_seg) bam_offsets_kernel<<<(W.n_seg + 255u) / 256u, 256, 0, s>>>(W);
    bam_parse_kernel<<<256 * 8, 256, 0, s>>>(W);
    bam_lists_kernel<<<1, 1024, 0, s>>>(W);
}

}  // namespace rsqc
