// wavemu.h -- TEST HARNESS ONLY (never loaded by the product).
//
// A SIMT emulation for the host: every lane of a workgroup is a fiber (ucontext) on ONE operating-system thread; the
// wave intrinsics the kernels are written against (__ballot, __shfl*, mbcnt, readfirstlane, __syncthreads, atomics on LDS
// and "global" memory, threadIdx / blockIdx) are provided on top of a cooperative scheduler, so that the UNMODIFIED source
// of a HIP kernel (rnaseqc_amd/csrc/rsqc_k1.h, rsqc_wave.h) runs on the CPU with 64-lane wavefronts and can be diffed
// against the oracle in the GPU-less build container.  A collective (ballot / shuffle / barrier) parks the calling lane
// until every lane of its wave (workgroup) has arrived; lanes of a wave must therefore reach the same collectives --
// which is what the hardware requires of converged code as well.  Atomics are plain read-modify-writes (one thread).
#pragma once
#define RSQC_WAVE_EMU 1

#include <stdint.h>
#include <string.h>
#include <ucontext.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <functional>
#include <vector>

#define __global__
#define __device__
#define __host__
#define __shared__ static
#define __forceinline__ inline
#define __launch_bounds__(...)

struct uint2 { uint32_t x, y; };
struct uint4 { uint32_t x, y, z, w; };
struct int4 { int32_t x, y, z, w; };
struct dim3 { uint32_t x = 1, y = 1, z = 1; };
static inline uint2 make_uint2(uint32_t x, uint32_t y) { return uint2{x, y}; }
static inline uint4 make_uint4(uint32_t x, uint32_t y, uint32_t z, uint32_t w) { return uint4{x, y, z, w}; }
typedef void *hipStream_t;

namespace wavemu {

constexpr int kWave = 64;
struct WaveState { uint64_t xchg[2][kWave]; uint64_t gen = 0; int arrived = 0; };
struct LaneState { ucontext_t ctx; bool done = false; char *stack = nullptr; };
struct Block {
    int n_threads = 0;
    std::vector<LaneState> lanes;
    std::vector<WaveState> waves;
    uint64_t block_gen = 0; int block_arrived = 0;
    ucontext_t sched;
    int cur = 0;                                   // running lane (thread index in the block)
    std::function<void()> body;
};
inline Block *&cur_block() { static Block *b = nullptr; return b; }
inline void yield() { Block *b = cur_block(); swapcontext(&b->lanes[(size_t)b->cur].ctx, &b->sched); }
inline int tid() { return cur_block()->cur; }
inline WaveState &wave() { Block *b = cur_block(); return b->waves[(size_t)(b->cur / kWave)]; }
inline int lanes_in_wave() { Block *b = cur_block(); const int w = b->cur / kWave; const int left = b->n_threads - w * kWave; return left < kWave ? left : kWave; }
// parks the lane until all lanes of its wave have called; returns the parity of the exchange buffer of this collective
inline int wave_arrive() {
    WaveState &w = wave();
    const uint64_t g = w.gen;
    if (++w.arrived == lanes_in_wave()) { w.arrived = 0; ++w.gen; }
    else while (w.gen == g) yield();
    return (int)(g & 1u);
}
inline void block_sync() {
    Block *b = cur_block();
    const uint64_t g = b->block_gen;
    if (++b->block_arrived == b->n_threads) { b->block_arrived = 0; ++b->block_gen; }
    else while (b->block_gen == g) yield();
}
inline void fiber_main() {
    Block *b = cur_block();
    b->body();
    b->lanes[(size_t)b->cur].done = true;
    swapcontext(&b->lanes[(size_t)b->cur].ctx, &b->sched);
}
// runs `body` as n_threads lanes to completion
inline void run_block(int n_threads, const std::function<void()> &body) {
    Block blk;
    blk.n_threads = n_threads; blk.body = body;
    blk.lanes.resize((size_t)n_threads); blk.waves.resize((size_t)((n_threads + kWave - 1) / kWave));
    cur_block() = &blk;
    constexpr size_t kStack = 256 * 1024;
    for (int t = 0; t < n_threads; ++t) {
        LaneState &L = blk.lanes[(size_t)t];
        L.stack = (char *)malloc(kStack);
        getcontext(&L.ctx);
        L.ctx.uc_stack.ss_sp = L.stack; L.ctx.uc_stack.ss_size = kStack; L.ctx.uc_link = &blk.sched;
        makecontext(&L.ctx, (void (*)())fiber_main, 0);
    }
    for (;;) {
        bool any = false;
        for (int t = 0; t < n_threads; ++t) {
            if (blk.lanes[(size_t)t].done) continue;
            any = true; blk.cur = t;
            swapcontext(&blk.sched, &blk.lanes[(size_t)t].ctx);
        }
        if (!any) break;
    }
    for (int t = 0; t < n_threads; ++t) free(blk.lanes[(size_t)t].stack);
    cur_block() = nullptr;
}

struct Idx { uint32_t x = 0, y = 0, z = 0; };
inline Idx &block_idx() { static Idx i; return i; }
inline Idx &grid_dim() { static Idx i; return i; }
struct ThreadIdxProxy { struct X { operator uint32_t() const { return (uint32_t)tid(); } } x; };
struct BlockDimProxy { struct X { operator uint32_t() const { return (uint32_t)cur_block()->n_threads; } } x; };
struct BlockIdxProxy { struct X { operator uint32_t() const { return block_idx().x; } } x; };
struct GridDimProxy { struct X { operator uint32_t() const { return grid_dim().x; } } x; };

template <class T> inline uint64_t to_bits(T v) { uint64_t b = 0; static_assert(sizeof(T) <= 8, "shuffle of at most 8 bytes"); memcpy(&b, &v, sizeof(T)); return b; }
template <class T> inline T from_bits(uint64_t b) { T v; memcpy(&v, &b, sizeof(T)); return v; }
}  // namespace wavemu

static wavemu::ThreadIdxProxy threadIdx;
static wavemu::BlockDimProxy blockDim;
static wavemu::BlockIdxProxy blockIdx;
static wavemu::GridDimProxy gridDim;

// ---- wave collectives ---------------------------------------------------------------------------------------------
inline unsigned long long __ballot(bool c) {
    wavemu::WaveState &w = wavemu::wave();
    const int l = wavemu::tid() % wavemu::kWave;
    w.xchg[w.gen & 1u][l] = c ? 1u : 0u;
    const int par = wavemu::wave_arrive();
    unsigned long long m = 0;
    for (int i = 0; i < wavemu::lanes_in_wave(); ++i) m |= (unsigned long long)(w.xchg[par][i] & 1u) << i;
    return m;
}
template <class T> inline T wavemu_exchange(T v, int src_lane) {
    wavemu::WaveState &w = wavemu::wave();
    const int l = wavemu::tid() % wavemu::kWave;
    w.xchg[w.gen & 1u][l] = wavemu::to_bits(v);
    const int par = wavemu::wave_arrive();
    return (src_lane >= 0 && src_lane < wavemu::kWave) ? wavemu::from_bits<T>(w.xchg[par][src_lane]) : v;
}
template <class T> inline T __shfl(T v, int src, int width = 64) { (void)width; return wavemu_exchange(v, src & 63); }
template <class T> inline T __shfl_up(T v, unsigned delta, int width = 64) { (void)width; const int l = wavemu::tid() % 64; return wavemu_exchange(v, l >= (int)delta ? l - (int)delta : l); }
template <class T> inline T __shfl_xor(T v, int mask, int width = 64) { (void)width; const int l = wavemu::tid() % 64; return wavemu_exchange(v, l ^ mask); }
inline void __syncthreads() { wavemu::block_sync(); }
// a one-wave "workgroup" orders its LDS traffic with a fence + wave barrier (rsqc_k3.h, k3_sync<64>): the lanes are fibers here,
// so the wave barrier is where they wait for each other; fences order nothing on one thread
#define __builtin_amdgcn_fence(...) ((void)0)
inline void __builtin_amdgcn_wave_barrier() { (void)wavemu::wave_arrive(); }
inline void __threadfence_block() {}
inline double __longlong_as_double(long long v) { double d; memcpy(&d, &v, 8); return d; }
using std::isnan; using std::isinf;
inline uint32_t wavemu_mbcnt_lo(uint32_t m, uint32_t add) { const int l = wavemu::tid() % 64; return add + (uint32_t)__builtin_popcount(l >= 32 ? m : (m & ((1u << l) - 1u))); }
inline uint32_t wavemu_mbcnt_hi(uint32_t m, uint32_t add) { const int l = wavemu::tid() % 64; return add + (uint32_t)(l > 32 ? __builtin_popcount(m & ((1u << (l - 32)) - 1u)) : 0); }
#define __builtin_amdgcn_mbcnt_lo wavemu_mbcnt_lo
#define __builtin_amdgcn_mbcnt_hi wavemu_mbcnt_hi
#define __builtin_amdgcn_readfirstlane(x) (x)      /* only applied to wave-uniform values */
inline int __popcll(unsigned long long x) { return __builtin_popcountll(x); }
inline int __ffsll(unsigned long long x) { return __builtin_ffsll((long long)x); }

// ---- atomics (one OS thread: plain read-modify-write) ---------------------------------------------------------------
template <class T> inline T wavemu_add(T *p, T v) { const T o = *p; *p = o + v; return o; }
inline uint32_t atomicAdd(uint32_t *p, uint32_t v) { return wavemu_add(p, v); }
inline int atomicAdd(int *p, int v) { return wavemu_add(p, v); }
inline unsigned long long atomicAdd(unsigned long long *p, unsigned long long v) { return wavemu_add(p, v); }
inline double atomicAdd(double *p, double v) { return wavemu_add(p, v); }
inline uint32_t atomicCAS(uint32_t *p, uint32_t cmp, uint32_t v) { const uint32_t o = *p; if (o == cmp) *p = v; return o; }
inline uint32_t atomicMax(uint32_t *p, uint32_t v) { const uint32_t o = *p; if (v > o) *p = v; return o; }
inline uint32_t atomicMin(uint32_t *p, uint32_t v) { const uint32_t o = *p; if (v < o) *p = v; return o; }
inline int atomicExch(int *p, int v) { const int o = *p; *p = v; return o; }
inline uint32_t atomicExch(uint32_t *p, uint32_t v) { const uint32_t o = *p; *p = v; return o; }
inline unsigned long long atomicCAS(unsigned long long *p, unsigned long long cmp, unsigned long long v) { const unsigned long long o = *p; if (o == cmp) *p = v; return o; }
inline unsigned long long atomicExch(unsigned long long *p, unsigned long long v) { const unsigned long long o = *p; *p = v; return o; }
