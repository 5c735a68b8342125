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

#ifndef RSQC_INFLATE_WAVES
#define RSQC_INFLATE_WAVES 5
#endif
// ---- K0: one wave per BGZF block (work handed out by an atomic counter: block decode times differ) -----------------
// LDS: 7.7 KB per wave, registers capped for five waves per SIMD (twenty per CU): the decoder is a chain of dependent
// instructions and LDS round trips, so the waves of a SIMD take turns in its issue slots (four waves per CU instead of
// sixteen: 2.6 times slower).  WPW waves share a workgroup, each with its own scratch.
// PAR: the one-pass commit of a round (rsqc_inflate.h) -- compiled both ways, picked per call by launch_bgzf_inflate.
template <int WPW, bool PAR>
__global__ __launch_bounds__(64 * WPW) __attribute__((amdgpu_waves_per_eu(RSQC_INFLATE_WAVES, RSQC_INFLATE_WAVES))) void bgzf_inflate_kernel(const uint8_t *__restrict__ in, const DevBgzfBlock *__restrict__ blk, uint32_t n_blk,
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
        const int rc = inflate_block<PAR>(S, in + k.in_off, k.in_len, out + k.out_off, k.out_len, k.crc);
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

// ---- exclusive sums of a workgroup of 256 threads (one value per thread; returns the total in `total`) --------------
__device__ __forceinline__ uint32_t block_scan_256(uint32_t v, uint32_t *lds /* [256] */, uint32_t &total) {
    const uint32_t t = threadIdx.x;
    lds[t] = v;
    __syncthreads();
    for (uint32_t d = 1; d < 256u; d <<= 1) {
        const uint32_t x = t >= d ? lds[t - d] : 0u;
        __syncthreads();
        lds[t] += x;
        __syncthreads();
    }
    total = lds[255];
    const uint32_t incl = lds[t];
    __syncthreads();
    return incl - v;
}

// ---- chain: are all guesses confirmed, and where do a segment's records and operations go.  Three launches: sums per
// workgroup of 256 segments (and the list of segments whose guess is not confirmed), one workgroup over those sums (after
// the exact repair of the listed segments: bam_repair_listed, a handful of walks), then the positions inside every workgroup.
constexpr uint32_t DEC_CHAIN_BLOCK = 256, DEC_BAD_LIST = 2048;
__global__ __launch_bounds__(256) void bam_chain_sums_kernel(DecodeWindow W, uint32_t *blk_rec, uint32_t *blk_ops, uint32_t *all_ok, uint32_t *bad_list) {
    __shared__ uint32_t lds[256];
    const uint32_t s = blockIdx.x * DEC_CHAIN_BLOCK + threadIdx.x;
    uint32_t nr = 0, no = 0;
    if (s < W.n_seg) {
        if (!decode_guess_confirmed(W, s)) { const uint32_t k = atomicAdd(all_ok + 1, 1u); if (k < DEC_BAD_LIST) bad_list[k] = s; atomicAnd(all_ok, 0u); }
        nr = W.seg[s].n_rec; no = W.seg[s].n_ops;
    }
    uint32_t tr, to;
    (void)block_scan_256(nr, lds, tr);
    (void)block_scan_256(no, lds, to);
    if (threadIdx.x == 0) { blk_rec[blockIdx.x] = tr; blk_ops[blockIdx.x] = to; }
}
__global__ __launch_bounds__(1024) void bam_chain_top_kernel(DecodeWindow W, uint32_t *blk_rec, uint32_t *blk_ops, uint32_t n_blk, uint32_t *all_ok, uint32_t *bad_list) {
    __shared__ uint32_t s_rec[1024], s_ops[1024], s_consumed, s_bad;
    const uint32_t t = threadIdx.x;
    if (t == 0) { s_bad = 0; s_consumed = W.n_seg ? W.seg[W.n_seg - 1].land : W.start; }
    __syncthreads();
    if (W.n_seg && !*all_ok) {                                          // (uniform) a wrong guess: walk again from the truth, then redo the sums
        if (t == 0) {
            uint32_t bad = 0;
            const uint32_t n_list = all_ok[1];
            if (n_list <= DEC_BAD_LIST) {
                for (uint32_t i = 1; i < n_list; ++i) {                  // (the list arrives in the order of the atomics: a few entries, insertion sort)
                    const uint32_t v = bad_list[i]; uint32_t j = i;
                    for (; j > 0 && bad_list[j - 1] > v; --j) bad_list[j] = bad_list[j - 1];
                    bad_list[j] = v;
                }
                s_consumed = bam_repair_listed(W.buf, W.seg, W.n_seg, W.start, DEC_SEG_BYTES, W.end, bad_list, n_list, bad);
            } else s_consumed = bam_verify_chain(W.buf, W.seg, W.n_seg, W.start, DEC_SEG_BYTES, W.end, bad);
            s_bad = bad;
        }
        __syncthreads();
        for (uint32_t b = t; b < n_blk; b += 1024u) {
            uint32_t nr = 0, no = 0;
            const uint32_t lo = b * DEC_CHAIN_BLOCK, hi = min(W.n_seg, lo + DEC_CHAIN_BLOCK);
            for (uint32_t s = lo; s < hi; ++s) { nr += W.seg[s].n_rec; no += W.seg[s].n_ops; }
            blk_rec[b] = nr; blk_ops[b] = no;
        }
        __syncthreads();
    }
    // exclusive sums over the workgroups' totals (at most 1024 of them: 2 GiB / 8 KiB / 256)
    s_rec[t] = t < n_blk ? blk_rec[t] : 0u; s_ops[t] = t < n_blk ? blk_ops[t] : 0u;
    __syncthreads();
    for (uint32_t d = 1; d < 1024u; d <<= 1) {
        const uint32_t x = t >= d ? s_rec[t - d] : 0u, y = t >= d ? s_ops[t - d] : 0u;
        __syncthreads();
        s_rec[t] += x; s_ops[t] += y;
        __syncthreads();
    }
    if (t < n_blk) { const uint32_t r = s_rec[t] - blk_rec[t], o = s_ops[t] - blk_ops[t]; blk_rec[t] = r; blk_ops[t] = o; }
    if (t == 0) {
        W.sum->n_rec = s_rec[1023]; W.sum->n_ops = s_ops[1023]; W.sum->consumed_end = s_consumed;
        if (s_bad) atomicOr(&W.sum->status, DEC_ST_BAD_RECORD);
    }
}
__global__ __launch_bounds__(256) void bam_chain_place_kernel(DecodeWindow W, const uint32_t *blk_rec, const uint32_t *blk_ops) {
    __shared__ uint32_t lds[256];
    const uint32_t s = blockIdx.x * DEC_CHAIN_BLOCK + threadIdx.x;
    uint32_t nr = 0, no = 0;
    if (s < W.n_seg) { nr = W.seg[s].n_rec; no = W.seg[s].n_ops; }
    uint32_t tot;
    const uint32_t r0 = block_scan_256(nr, lds, tot), o0 = block_scan_256(no, lds, tot);
    if (s < W.n_seg) { W.seg_rec0[s] = blk_rec[blockIdx.x] + r0; W.seg_ops0[s] = blk_ops[blockIdx.x] + o0; }
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

// ---- lists: the marks, in record order, become the batch's segment and wide tables.  Same three-launch shape: counts per
// workgroup (256 threads x 32 consecutive records), one workgroup over those, then the writes.
constexpr uint32_t DEC_LIST_PER_THREAD = 32, DEC_LIST_BLOCK = 256 * DEC_LIST_PER_THREAD;
struct DecodeListBlock { uint32_t seg, wide, bad; int32_t last_judged; };
__global__ __launch_bounds__(256) void bam_lists_count_kernel(DecodeWindow W, DecodeListBlock *blk) {
    __shared__ uint32_t lds[256];
    __shared__ int32_t s_last;
    if (W.sum->status) return;
    const uint32_t n = W.sum->n_rec;
    const uint32_t first = blockIdx.x * DEC_LIST_BLOCK;
    if (first >= n) return;                                             // (the grid is sized for the most records the window can hold)
    const uint32_t lo = min(n, first + threadIdx.x * DEC_LIST_PER_THREAD), hi = min(n, lo + DEC_LIST_PER_THREAD);
    if (threadIdx.x == 0) s_last = -1;
    DecodeListCounts c;
    decode_lists_count(W, lo, hi, c);
    uint32_t ts, tw, tb;
    (void)block_scan_256(c.seg, lds, ts); (void)block_scan_256(c.wide, lds, tw); (void)block_scan_256(c.bad, lds, tb);
    if (c.last_judged >= 0) atomicMax(&s_last, c.last_judged);
    __syncthreads();
    if (threadIdx.x == 0) blk[blockIdx.x] = DecodeListBlock{ts, tw, tb, s_last};
}
__global__ __launch_bounds__(1024) void bam_lists_top_kernel(DecodeWindow W, DecodeListBlock *blk) {
    __shared__ uint32_t s_seg[1024], s_wide[1024], s_bad[1024];
    __shared__ int32_t s_last;
    if (W.sum->status) return;
    const uint32_t t = threadIdx.x, n = W.sum->n_rec, n_blk = (n + DEC_LIST_BLOCK - 1) / DEC_LIST_BLOCK;
    const uint32_t per = (n_blk + 1023u) / 1024u, lo = min(n_blk, t * per), hi = min(n_blk, lo + per);
    if (t == 0) s_last = -1;
    __syncthreads();
    uint32_t a = 0, b = 0, d = 0; int32_t last = -1;
    for (uint32_t k = lo; k < hi; ++k) { a += blk[k].seg; b += blk[k].wide; d += blk[k].bad; if (blk[k].last_judged >= 0) last = blk[k].last_judged; }
    s_seg[t] = a; s_wide[t] = b; s_bad[t] = d;
    if (last >= 0) atomicMax(&s_last, last);
    __syncthreads();
    for (uint32_t st = 1; st < 1024u; st <<= 1) {
        const uint32_t x = t >= st ? s_seg[t - st] : 0u, y = t >= st ? s_wide[t - st] : 0u, z = t >= st ? s_bad[t - st] : 0u;
        __syncthreads();
        s_seg[t] += x; s_wide[t] += y; s_bad[t] += z;
        __syncthreads();
    }
    uint32_t ra = s_seg[t] - a, rb = s_wide[t] - b, rd = s_bad[t] - d;          // this thread's run of workgroups starts here
    for (uint32_t k = lo; k < hi; ++k) {
        const DecodeListBlock x = blk[k];
        blk[k] = DecodeListBlock{ra, rb, rd, x.last_judged};
        ra += x.seg; rb += x.wide; rd += x.bad;
    }
    if (t == 0) decode_lists_finish(W, n, DecodeListCounts{s_seg[1023], s_wide[1023], s_bad[1023], s_last});
}
__global__ __launch_bounds__(256) void bam_lists_write_kernel(DecodeWindow W, const DecodeListBlock *blk) {
    __shared__ uint32_t lds[256];
    if (W.sum->status) return;
    const uint32_t n = W.sum->n_rec;
    const uint32_t first = blockIdx.x * DEC_LIST_BLOCK;
    if (first >= n) return;
    const uint32_t lo = min(n, first + threadIdx.x * DEC_LIST_PER_THREAD), hi = min(n, lo + DEC_LIST_PER_THREAD);
    DecodeListCounts c;
    decode_lists_count(W, lo, hi, c);
    uint32_t tot;
    const uint32_t s0 = block_scan_256(c.seg, lds, tot), w0 = block_scan_256(c.wide, lds, tot), b0 = block_scan_256(c.bad, lds, tot);
    if (c.seg | c.wide | c.bad) {
        const DecodeListBlock base = blk[blockIdx.x];
        decode_lists_write(W, lo, hi, DecodeListCounts{base.seg + s0, base.wide + w0, base.bad + b0, -1});
    }
}

// ---- launches ----------------------------------------------------------------------------------------------------
void launch_bgzf_inflate(hipStream_t s, const uint8_t *in, const DevBgzfBlock *blk, uint32_t n_blk, uint8_t *out, DecodeSummary *sum, bool one_pass) {
    if (!n_blk) return;
    // four waves per SIMD of the chip; the counter feeds them
    static const int wpw = getenv("RSQC_INFLATE_WPW") ? atoi(getenv("RSQC_INFLATE_WPW")) : 4;
    static bool told = false;
    if (!told && getenv("RSQC_DECODE_PROFILE")) {
        int a = 0, b = 0;
        (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&a, bgzf_inflate_kernel<1, true>, 64, 0);
        (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&b, bgzf_inflate_kernel<4, true>, 256, 0);
        fprintf(stderr, "[decode] inflate kernel: %d workgroups of 1 wave or %d of 4 waves per CU; running %d waves per workgroup\n", a, b, wpw);
        told = true;
    }
    if (wpw == 1) {
        const uint32_t grid = n_blk < 256u * 16u ? n_blk : 256u * 16u;
        if (one_pass) bgzf_inflate_kernel<1, true><<<grid, 64, 0, s>>>(in, blk, n_blk, out, sum);
        else bgzf_inflate_kernel<1, false><<<grid, 64, 0, s>>>(in, blk, n_blk, out, sum);
    } else {
        const uint32_t need = (n_blk + 3u) / 4u, grid = need < 256u * 8u ? need : 256u * 8u;
        if (one_pass) bgzf_inflate_kernel<4, true><<<grid, 256, 0, s>>>(in, blk, n_blk, out, sum);
        else bgzf_inflate_kernel<4, false><<<grid, 256, 0, s>>>(in, blk, n_blk, out, sum);
    }
}
void launch_decode_window(hipStream_t s, const DecodeWindow &W, uint32_t *scratch) {
    // scratch: DEC_SCRATCH_WORDS words: [0] the all-guesses-confirmed flag, then the per-workgroup sums of the two scans
    const uint32_t seg_blocks = (W.n_seg + DEC_CHAIN_BLOCK - 1) / DEC_CHAIN_BLOCK;
    uint32_t *all_ok = scratch, *blk_rec = scratch + 16, *blk_ops = blk_rec + 1024, *bad_list = blk_ops + 1024;
    DecodeListBlock *lblk = (DecodeListBlock *)(bad_list + DEC_BAD_LIST);
    (void)hipMemsetAsync(all_ok, 0xff, 4, s);                                // [0] every guess confirmed so far
    (void)hipMemsetAsync(all_ok + 1, 0, 4, s);                               // [1] entries of bad_list
    if (W.n_seg) bam_frame_kernel<<<(W.n_seg + 255u) / 256u, 256, 0, s>>>(W);
    if (seg_blocks) bam_chain_sums_kernel<<<seg_blocks, 256, 0, s>>>(W, blk_rec, blk_ops, all_ok, bad_list);
    bam_chain_top_kernel<<<1, 1024, 0, s>>>(W, blk_rec, blk_ops, seg_blocks, all_ok, bad_list);
    if (seg_blocks) bam_chain_place_kernel<<<seg_blocks, 256, 0, s>>>(W, blk_rec, blk_ops);
    if (W.n_seg) bam_offsets_kernel<<<(W.n_seg + 255u) / 256u, 256, 0, s>>>(W);
    bam_parse_kernel<<<256 * 8, 256, 0, s>>>(W);
    // (the record count is on the device: grids for the most records the window's bytes can hold; surplus workgroups leave at once)
    const uint32_t max_rec = (W.end - W.start) / 36u + 1u, list_blocks = (max_rec + DEC_LIST_BLOCK - 1) / DEC_LIST_BLOCK;
    bam_lists_count_kernel<<<list_blocks, 256, 0, s>>>(W, lblk);
    bam_lists_top_kernel<<<1, 1024, 0, s>>>(W, lblk);
    bam_lists_write_kernel<<<list_blocks, 256, 0, s>>>(W, lblk);
}

}  // namespace rsqc
