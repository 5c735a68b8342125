// rsqc_k1.h -- K1, the per-record kernel of the default rule set since round 3 (included by rsqc_kernels.hip, which
// provides the wave helpers).  Replaces src/RNASeQC.cpp:254-369 and src/Expression.cpp:26-67,106-117,308-458.
//
// The shape follows from two measurements of the round-2 kernel: the vector pipe was 63 % of its time
// (1 083 VALU instructions per 64-record tile) and the rest was exposed latency of a chain of dependent index loads
// at 4 waves per SIMD.  Both came from running ONE predicated program over records of every shape -- four block rounds
// and eight commit slots for a stream in which 58 % of the records have one block.
//
//   classify_ei_kernel     every WAVE streams its own contiguous range of records.  Per 64-record tile ("phase A") it
//       unpacks the record words, walks the first operations of the CIGAR, runs the gate cascade with its counters
//       (WaveSink) and the Read-Length inputs, and then SORTS the surviving records by block count into per-wave LDS
//       queues: one block, two blocks.  Whenever a queue holds 64 records the wave runs the feature stage of exactly
//       that shape on them (`k1e_process<NB>`): all lanes busy, one / two look-up rounds, two / four commit slots, no
//       block-count predicates.  Records with longer CIGARs (more than 4 operations or more than 2 blocks: 9 % of an
//       RNA-seq file) wait in a third queue as (record index, flags, position, CIGAR offset) and are processed 64 at a time by
//       `k1e_process3` (round 6): the eight operations come back from the caches in two loads, a bit-field walk over them
//       captures up to THREE blocks, and the record takes the three-block feature stage and commit.  What that stage cannot
//       take -- more than 8 operations, more than 3 blocks: 0.6 % of the records -- is listed in the workgroup's own region of
//       the deferred list and taken by `classify_long_kernel` behind this kernel with the general walk and FAST_BLOCKS blocks
//       (rounds 3-5 ran that code inside the tile loop: it alone held the kernel at 126 VGPRs = four waves per SIMD).
//       A record whose blocks meet an interval covered by more than two exons goes to the general code (classify_slow_kernel).
//
// The feature stage reads the ELEMENTARY-INTERVAL index (rsqc_read.h: EiEntry / EiRank): a block costs two 16-byte
// rank-word loads (independent, no walk) and one round of entry loads, instead of bin -> rows -> rows further down.
#pragma once

namespace rsqc {

// A rarely taken branch that loads (wide-table values, the contig of a boundary tile, operations past the eighth) ends with
// this: the loaded value is waited for INSIDE the branch.  Left to the compiler, the wait lands at the first use after the
// branches join -- on the common path -- and, vmcnt being one in-order counter, becomes a wait for everything the wave has
// in flight, the record words staged for the next tile and the previous tile's atomics included.
#if defined(__HIP_DEVICE_COMPILE__)
#define K1E_LANDED(x) asm volatile("" : "+v"(x))
#ifdef K1E_PIN_ARGS                                      /* (A/B build; measured 1-2 % slower: the pinned scalars are spilled to VGPR lanes instead) */
#define K1E_PIN(x) asm volatile("" : "+s"(x))            /* a wave-uniform value the compiler may not re-derive: it stays in SGPRs */
#else
#define K1E_PIN(x) (void)(x)                             /* (A/B build) */
#endif
#else
#define K1E_LANDED(x) (void)(x)
#define K1E_PIN(x) (void)(x)
#endif

// Everything the two kernels are given, as ONE by-value struct: its layout is the kernel-argument segment's.  The hot loop
// uses a few dozen of these words; the rest (wide table, BED candidates, overflow list, error word, fall-back targets of the
// LDS tables, what the epilogue flushes) is read through k1e_lazy_args() at the point of use with scalar loads -- held in
// SGPRs for the whole kernel they are what pushed the round-2 kernel to 155 spilled scalars, every one of which costs
// v_writelane / v_readlane pairs on the vector pipe.
struct K1Args { DevAnnotation a; DevParams p; DevBatch b; DevAccum acc; };
#if defined(__HIPCC__)
__device__ __forceinline__ const K1Args *k1e_lazy_args() {
#if defined(__HIP_DEVICE_COMPILE__)
    const K1Args *q = (const K1Args *)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(q));                  // opaque: the loads stay where they are written
    return q;
#else
    return nullptr;                              // (host pass of the compiler: never called)
#endif
}
#else
static const K1Args *g_k1e_args = nullptr;       // (host emulation: set by the harness)
static unsigned long long g_k1e_coarse_hits = 0;
static unsigned long long g_k1e_ucache_hits = 0;     // (test harness: ... of which the interval came from the wave's one-entry cache)
static unsigned long long g_k1e_uniform_calls = 0;   // (test harness: one-block feature-stage calls answered by the wave-uniform path)
static unsigned long long g_k1e_uniform2_calls = 0;  // (test harness: two-block calls answered by k1e_uniform2)
static inline const K1Args *k1e_lazy_args() { return g_k1e_args; }
#endif

// Streams that are read ONCE (the core half-records, the CIGAR pool) and written once (the pairs) can be marked non-temporal: the line is
// the first to leave the XCD's L2, which then keeps what the feature stages come BACK for -- the auxiliary half-records (name hashes), the
// rank words and interval entries.  K1E_NT: bit 0 the record / CIGAR loads, bit 1 the pair stores (A/B, call r6f).
#ifndef K1E_NT
#define K1E_NT 0
#endif
template <class T> __device__ __forceinline__ T k1e_ld32_stream(const T *base, uint32_t idx) {
#if defined(__HIP_DEVICE_COMPILE__)
    if (K1E_NT & 1) {
        typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
        static_assert(sizeof(T) == 16, "16-byte streams");
        const u32x4 v = __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(reinterpret_cast<const char *>(base) + (uint32_t)(idx * 16u)));
        T r; __builtin_memcpy(&r, &v, 16); return r;
    }
#endif
    return ld32(base, idx);
}
// eight CIGAR operations of one record in TWO loads (a 16-byte global load needs only dword alignment; eight separate
// dword gathers were eight trips through the address unit)
struct alignas(4) Cig4 { uint32_t v[4]; };
__device__ __forceinline__ void k1e_load_cigar8(const uint32_t *cigar, uint32_t off, uint32_t (&c)[8]) {
    const char *const at = reinterpret_cast<const char *>(cigar) + (uint32_t)(off * 4u);       // buffers carry 32 bytes of slack
#if defined(__HIP_DEVICE_COMPILE__)
    if (K1E_NT & 1) {
        typedef uint32_t u32x4a __attribute__((ext_vector_type(4), aligned(4)));
        const u32x4a l4 = __builtin_nontemporal_load(reinterpret_cast<const u32x4a *>(at)), h4 = __builtin_nontemporal_load(reinterpret_cast<const u32x4a *>(at + 16));
        c[0] = l4.x; c[1] = l4.y; c[2] = l4.z; c[3] = l4.w; c[4] = h4.x; c[5] = h4.y; c[6] = h4.z; c[7] = h4.w;
        return;
    }
#endif
    const Cig4 lo = *reinterpret_cast<const Cig4 *>(at), hi = *reinterpret_cast<const Cig4 *>(at + 16);
#pragma unroll
    for (int k = 0; k < 4; ++k) { c[k] = lo.v[k]; c[4 + k] = hi.v[k]; }
}

// (instruction-count experiments, `make variant DEFS=-DK1E_STOP_AT=<mark>`: phase A ends at that section mark with its results
//  kept alive, no feature stage -- wrong results by design, never in the product build)
#if defined(K1E_STOP_AT) && defined(__HIP_DEVICE_COMPILE__)
template <class T> __device__ __forceinline__ int k1e_keep_v(T x) { asm volatile("" :: "v"(x)); return 0; }
template <class T> __device__ __forceinline__ int k1e_keep_s(T x) { asm volatile("" :: "s"(x)); return 0; }
template <class... T> __device__ __forceinline__ void k1e_sink_v(T... x) { const int d[] = {k1e_keep_v(x)...}; (void)d; }
template <class... T> __device__ __forceinline__ void k1e_sink_s(T... x) { const int d[] = {k1e_keep_s(x)...}; (void)d; }
#define K1E_STOP(n, V, S) if (K1E_STOP_AT == (n)) { k1e_sink_v V; k1e_sink_s S; break; }
#else
#define K1E_STOP(n, V, S)
#endif
#if defined(K1E_STAGE_MARKS)
#define K1E_AMARK(sec)                             /* (phase A's marks off: its time lands in [8], or in [0] for a tile without a stage call) */
#else
#define K1E_AMARK(sec) RSQC_MARK(sec)
#endif
constexpr int K1E_WAVES = RSQC_K1_THREADS / 64;
#ifndef K1E_PIECE_RECORDS
#define K1E_PIECE_RECORDS 256
#endif
constexpr uint64_t K1E_PIECE = K1E_PIECE_RECORDS;   // records a wave takes from its workgroup's range at a time (4 tiles)
#ifndef K1E_QCAP_N
#define K1E_QCAP_N 128
#endif
constexpr int K1E_QCAP = K1E_QCAP_N;                 // per-wave queue slots: < 64 left over + <= 64 of the next tile
#ifndef K1E_ESLOTS_N
#define K1E_ESLOTS_N 256      /* (round 5 measured 512 / 256 against 256 / 128: no difference, call r5p; the smaller tables are what lets five workgroups share a CU) */
#endif
#ifndef K1E_GSLOTS_N
#define K1E_GSLOTS_N 128
#endif
constexpr int K1E_ESLOTS = K1E_ESLOTS_N, K1E_GSLOTS = K1E_GSLOTS_N;
constexpr uint32_t K1E_HQ = 1u << 16;         // item word `flhq`: the record's flag word | K1E_HQ when high quality
// (timing experiments, `make variant DEFS=-DK1E_ABL=<bits>`: 1 no feature stage, 2 no LDS table updates, 4 no coverage atomics,
//  8 no pairs, 16 no long-CIGAR kernel -- wrong results by design; the product build has K1E_ABL == 0 and none of it)
#ifndef K1E_ABL
#define K1E_ABL 0
#endif
#ifndef K1E_WALK_SKIP
#define K1E_WALK_SKIP 1       /* operations 5-7 of the staged eight are walked only by tiles in which some record has more than five (call r6o: K1 2.27 -> 2.18 ms; 0: rounds 6a-6n) */
#endif
#ifndef K1E_EXON_RUNSUM
#define K1E_EXON_RUNSUM 1     /* 0: exon fractions of multi-block records lane by lane (rounds 3-5), for an A/B */
#endif

// Workgroup-local accumulators: a workgroup streams a short genomic window, so it touches a handful of neighbouring
// exons and genes; direct-mapped LDS tables take every update and each distinct key costs ONE global atomic when the
// workgroup retires.  A key that finds its slot taken by another key goes straight to memory.
struct K1eTables {
    // the seven sum-type counters, 64 bits each (sum slot k1e_sum_slot(counter); the others count records: cnt32).  A table of all
    // RSQC_N_COUNTERS of them took the structure past 32 000 bytes -- gfx950 hands LDS out in granules of 1 280 bytes, so a workgroup
    // of more than 25 granules leaves room for FOUR per CU, not five (call r6b: 32 176 bytes ran at four, measured by wave-cycles)
    unsigned long long cnt[8];
    uint32_t cnt32[64];                          // one-per-record counters (a workgroup sees < 2^32 records)
    double eval[K1E_ESLOTS];
    uint32_t ekey[K1E_ESLOTS];                   // exon id
    unsigned long long gval[K1E_GSLOTS];         // low word: records, high word: records that are not duplicates
    uint32_t gkey[K1E_GSLOTS];
    uint32_t rl[3];
    uint32_t pairs;                              // pairs in the workgroup's chunk
    uint32_t frags;                              // fragment-size candidates in the workgroup's region (BED runs)
    uint32_t piece;                              // next piece of the workgroup's range to hand to a wave
    uint32_t defer;                              // records in the workgroup's region of the deferred list (-> classify_long_kernel)
    // The last elementary interval a wave's uniform path (k1e_uniform1) looked up: [lo, hi) and its index entry.  The queued records are
    // neighbours in the sorted stream, and so are consecutive CALLS: on the contract workload 56 % of the one-block calls lie in the
    // interval of the call before them (tools/uniform_tiles.py) -- those take their interval from here, two LDS reads, instead of two
    // dependent scalar loads from the index.  One entry per wave (a wave reads what it wrote itself: no tearing), emptied when the
    // wave's stream enters another contig.  (The one-entry form of the interval window in LDS that north_star names.)
    alignas(16) uint32_t ucache[K1E_WAVES][12];  // EiEntry (8 words), lo, hi, 2 unused
    // --bed: a wave's cursor into its contig's start-sorted BED rows (classify_ei_kernel<true>, phase A): the segment it belongs to, the
    // start of the last row that starts at or before the last tile's end / the start of the row behind it / the running max of `end`
    // up to the former
    int32_t bedc[K1E_WAVES][4];
    static __device__ __forceinline__ constexpr int sum_slot(int c) {
        return c == RSQC_C_END1_MISMATCHES ? 0 : c == RSQC_C_END1_BASES ? 1 : c == RSQC_C_END2_MISMATCHES ? 2 : c == RSQC_C_END2_BASES ? 3 :
               c == RSQC_C_MISMATCHED_BASES ? 4 : c == RSQC_C_TOTAL_BASES ? 5 : c == RSQC_C_ALIGNMENT_BLOCKS ? 6 : 7;      // (7: no sum counter, stays 0)
    }
    __device__ __forceinline__ void init(uint32_t pairs0) {
        if (threadIdx.x < 8) cnt[threadIdx.x] = 0ull;
        if (threadIdx.x < 64) cnt32[threadIdx.x] = 0u;
        for (int c = threadIdx.x; c < K1E_ESLOTS; c += blockDim.x) { ekey[c] = 0xFFFFFFFFu; eval[c] = 0.0; }
        for (int c = threadIdx.x; c < K1E_GSLOTS; c += blockDim.x) { gkey[c] = 0xFFFFFFFFu; gval[c] = 0ull; }
        if (threadIdx.x == 0) { rl[0] = 0u; rl[1] = 0xFFFFFFFFu; rl[2] = 0u; pairs = pairs0; piece = 0u; frags = 0u; defer = 0u; }
        if (threadIdx.x < K1E_WAVES) { ucache[threadIdx.x][8] = 1u; ucache[threadIdx.x][9] = 0u; bedc[threadIdx.x][0] = -1; }   // [1, 0): holds no position; no segment
    }
    __device__ __forceinline__ void exon_add(uint32_t eid, double frac) {
        const uint32_t slot = eid & (K1E_ESLOTS - 1);
        const uint32_t old = atomicCAS(&ekey[slot], 0xFFFFFFFFu, eid);
        if (old == 0xFFFFFFFFu || old == eid) atomicAdd(&eval[slot], frac);
        else atomicAdd(&k1e_lazy_args()->acc.exon_acc[eid], frac);
    }
    // n records, nd of them not duplicates
    __device__ __forceinline__ void gene_add(uint32_t g, uint32_t n, uint32_t nd) {
        const uint32_t slot = g & (K1E_GSLOTS - 1);
        const uint32_t old = atomicCAS(&gkey[slot], 0xFFFFFFFFu, g);
        if (old == 0xFFFFFFFFu || old == g) atomicAdd(&gval[slot], (unsigned long long)n | ((unsigned long long)nd << 32));
        else {
            const DevAccum &acc = k1e_lazy_args()->acc;
            atomicAdd(&acc.gene_reads[g], (unsigned long long)n); if (nd) atomicAdd(&acc.gene_unique[g], (unsigned long long)nd);
        }
    }
    // after a __syncthreads(): every table goes to memory, one atomic per distinct key
    // n_records: records of the workgroup's range (-> Total Alignments; the gate cascade runs LEAN, rsqc_read.h)
    __device__ __forceinline__ void flush(unsigned long long n_records) {
        const DevAccum &acc = k1e_lazy_args()->acc;
        for (int c = threadIdx.x; c < RSQC_N_COUNTERS; c += blockDim.x) {
            unsigned long long v = cnt[sum_slot(c)] + (unsigned long long)cnt32[c];
            if (c == RSQC_C_TOTAL_ALIGNMENTS) v += n_records;
            if (c == RSQC_C_MAPPED_UNIQUE_READS) v += (unsigned long long)cnt32[RSQC_C_MAPPED_READS] - cnt32[RSQC_C_MAPPED_DUPLICATE_READS];
            if (c == RSQC_C_UNIQUE_FRAGMENTS) v += (unsigned long long)cnt32[RSQC_C_END1_MAPPED_READS] - cnt32[RSQC_C_DUPLICATE_PAIRS];
            if (c == RSQC_C_LOW_QUALITY_READS) v += (unsigned long long)cnt32[RSQC_C_READS_USED] - cnt32[RSQC_C_HIGH_QUALITY_READS];
            if (v) atomicAdd(&acc.counters[c], v);
        }
        for (int c = threadIdx.x; c < K1E_ESLOTS; c += blockDim.x)
            if (ekey[c] != 0xFFFFFFFFu) atomicAdd(&acc.exon_acc[ekey[c]], eval[c]);
        for (int c = threadIdx.x; c < K1E_GSLOTS; c += blockDim.x)
            if (gkey[c] != 0xFFFFFFFFu) {
                const unsigned long long v = gval[c];
                atomicAdd(&acc.gene_reads[gkey[c]], v & 0xFFFFFFFFull);
                if (v >> 32) atomicAdd(&acc.gene_unique[gkey[c]], v >> 32);
            }
    }
};

// The per-wave queues are structures of ARRAYS of dwords (round 6): a queue write or read of one field is a ds_write_b32 /
// ds_read_b32 over 64 consecutive dwords -- conflict-free -- where the 16-byte rows of rounds 3-5 (ds_write_b128: 8 lanes per LDS
// cycle, and the ring's wrap put rows of one instruction on the same banks) showed SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE = 48 %.
struct K1eQueue1 { uint32_t bs[K1E_QCAP], lf[K1E_QCAP], idx[K1E_QCAP];       // one block: start, len (16 bits) | flag bits 0-11 << 16 | high quality << 28, record index
#ifdef K1E_COARSE
                   uint32_t cz[K1E_QCAP];                                       // coarse-table answer (0 = none)
#endif
};
struct K1eQueue2 { uint32_t bs0[K1E_QCAP], len0[K1E_QCAP], bs1[K1E_QCAP], len1[K1E_QCAP], idx[K1E_QCAP], flhq[K1E_QCAP]; };   // two blocks
// three blocks (round 6: captured by phase A's walk over the eight staged operations), lengths in 16 bits each (a longer block -- never
// seen in RNA-seq -- is classify_long_kernel's): lens01 = len0 | len1 << 16, lf2 = len2 | flag bits 0-11 << 16 | high quality << 28.
// 84 slots (what the workgroup's LDS budget leaves; a ring of any size: the slot index wraps by one compare): the ring is emptied at 64,
// so a tile's records find it full only when more than 20 of them have three blocks; those go to the deferred list instead.  (Call r6c, 64
// slots of 8 words: the surplus was a third of the deferred records.)
#ifndef K1E_Q3CAP_N
#define K1E_Q3CAP_N 84
#endif
constexpr int K1E_Q3CAP = K1E_Q3CAP_N;
static_assert(K1E_Q3CAP >= 64 && K1E_Q3CAP <= 128, "the three-block ring holds at least one full call");
struct K1eQueue3 { uint32_t bs0[K1E_Q3CAP], bs1[K1E_Q3CAP], bs2[K1E_Q3CAP], lens01[K1E_Q3CAP], lf2[K1E_Q3CAP], idx[K1E_Q3CAP]; };
__device__ __forceinline__ uint32_t k1e_wrap3(uint32_t s) { return s >= (uint32_t)K1E_Q3CAP ? s - (uint32_t)K1E_Q3CAP : s; }    // (s < 2 * K1E_Q3CAP)
struct K1eShared {
    K1eTables T;
    K1eQueue1 q1[K1E_WAVES];
    K1eQueue2 q2[K1E_WAVES];
    K1eQueue3 q3[K1E_WAVES];
#ifdef K1E_LDS_PAD
    char pad[K1E_LDS_PAD];                       // (occupancy experiments: `make variant DEFS=-DK1E_LDS_PAD=12288` leaves room for three workgroups per CU)
#endif
};

#if defined(__HIP_DEVICE_COMPILE__)
#define K1E_GLOBAL(T, p) ((__attribute__((address_space(1))) T *)(p))      /* a pointer known to be global memory */
#else
#define K1E_GLOBAL(T, p) (p)
#endif
// the pair chunk a wave writes to: its workgroup's own in classify_ei_kernel; in classify_long_kernel the chunk of the K1 workgroup
// whose deferred records the workgroup is working on (wave-uniform, an SGPR)
// where a workgroup's (gene, name) pairs go: the three streams of DevAccum and the chunk capacity
struct K1ePairDst { uint64_t pairs; uint32_t cap; };
#ifndef K1E_LAZYPAIR
#define K1E_LAZYPAIR 1        /* 1: read at the head of every commit with scalar loads; 0: held across the tile loop (the tree's form) */
#endif
// The two words straight from the kernel-argument segment, as SCALAR loads in wave-uniform code.  (Read through k1e_lazy_args()
// they become flat VECTOR loads followed by a wait for every outstanding memory operation of the wave -- fine in the rare paths
// that function serves, 21 % of the kernel when it sat in the commit: profiles/r4_k1_variants.txt, r4n2.)
__device__ __forceinline__ K1ePairDst k1e_pair_dst() {
    K1ePairDst d;
#if defined(__HIP_DEVICE_COMPILE__)
    const void *q = __builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("s_load_dwordx2 %0, %2, %3\n\ts_load_dword %1, %2, %4\n\ts_waitcnt lgkmcnt(0)"
                 : "=&s"(d.pairs), "=&s"(d.cap)
                 : "s"(q), "n"(offsetof(K1Args, acc) + offsetof(DevAccum, pairs)), "n"(offsetof(K1Args, acc) + offsetof(DevAccum, pair_chunk_cap)));
#else
    const DevAccum &a = k1e_lazy_args()->acc;
    d.pairs = (uint64_t)(uintptr_t)a.pairs; d.cap = a.pair_chunk_cap;
#endif
    return d;
}
// "is this lane 0" / "is this lane below N" as CONSTANT lane masks (WaveSink::lane of a literal): written as `l == 0` the test is
// one v_cmp whose 64-bit result the compiler computes once, hoists out of the tile loop, spills to a VGPR lane with the other
// long-lived scalars and brings back with two v_readlane at each of its nine uses per tile -- vector instructions all
#ifndef K1E_CONSTMASK
#define K1E_CONSTMASK 1       /* 0: the compare form, for an A/B */
#endif
__device__ __forceinline__ bool k1e_first_lane() { return K1E_CONSTMASK ? WaveSink::lane(LaneMask{1ull}) : lane_id() == 0; }
template <int N> __device__ __forceinline__ bool k1e_lane_below() { static_assert(N > 0 && N < 64, "lanes"); return K1E_CONSTMASK ? WaveSink::lane(LaneMask{(1ull << N) - 1ull}) : lane_id() < N; }

// ---- (gene, name) pairs of the lanes of `m` into the workgroup's chunk: one LDS slot reservation per wave, one coalesced 16-byte store per lane ----
// LANE_CHUNK (classify_long_kernel's fall-back, see there): `chunk` is a per-lane value -- the chunk of the K1 workgroup that listed the
// record -- and the slot a returning memory atomic per pair on that chunk's count.  (As the kernel's only path this cost 0.6 ms, call r6d:
// the lanes of a call share a few chunks, and same-address returning atomics complete at ~88 per microsecond.)
template <bool LANE_CHUNK = false>
__device__ __forceinline__ void k1e_emit_pairs(K1eTables &T, uint64_t m, uint32_t g, uint64_t qhash, uint32_t qh2, const K1ePairDst &pd, uint32_t chunk) {
    if (LANE_CHUNK) {
        if (WaveSink::lane(LaneMask{m})) {
            const uint32_t slot = atomicAdd(&k1e_lazy_args()->acc.pair_chunk_count[chunk], 1u);
            if (slot < pd.cap) K1E_GLOBAL(PairRec, (PairRec *)(uintptr_t)pd.pairs)[(size_t)chunk * pd.cap + slot] = PairRec{g, qh2, qhash};
            else atomicExch(k1e_lazy_args()->acc.error, RSQC_ERR_CAPACITY);
        }
        return;
    }
    const int lead = __ffsll((unsigned long long)m) - 1;
    uint32_t base = 0;
    if (lane_id() == lead) base = atomicAdd(&T.pairs, (uint32_t)__popcll(m));
    base = lane_value(base, lead);
    if (WaveSink::lane(LaneMask{m})) {
        const uint32_t slot = base + mask_rank(m);
        const size_t chunk_at = (size_t)chunk * pd.cap;
        if (slot < pd.cap) {
#if defined(__HIP_DEVICE_COMPILE__)
            if (K1E_NT & 2) {
                typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
                const u32x4 v = {g, qh2, (uint32_t)qhash, (uint32_t)(qhash >> 32)};
                __builtin_nontemporal_store(v, reinterpret_cast<u32x4 *>(K1E_GLOBAL(PairRec, (PairRec *)(uintptr_t)pd.pairs) + (chunk_at + slot)));
            } else
#endif
            K1E_GLOBAL(PairRec, (PairRec *)(uintptr_t)pd.pairs)[chunk_at + slot] = PairRec{g, qh2, qhash};          // (one 16-byte store)
        } else atomicExch(k1e_lazy_args()->acc.error, RSQC_ERR_CAPACITY);
    }
}

// ---- commit: what exon_metrics_ei returned goes to the accumulators ------------------------------------------------
// exonCounts[eid] += len / aligned (src/Expression.cpp:345, Metrics.cpp:59-66) and the per-gene counters go to the
// workgroup's LDS tables; per-base coverage goes to memory as a difference array (+1 at the block's first base, -1 after
// its last), identical neighbouring slots merged into one atomic; (gene, qname-hash) pairs go to the workgroup's chunk.
template <int NB, bool LANE_CHUNK = false>
__device__ __forceinline__ void k1e_commit(uint32_t *cov_diff, K1eTables &T, const EiOut &eo, const uint32_t (&len)[NB], uint32_t fl,
                                           uint64_t qhash, uint32_t qh2, const K1ePairDst &held, uint32_t chunk) {
    typedef WaveSink WS;
    const K1ePairDst pd = K1E_LAZYPAIR ? k1e_pair_dst() : held;
    const uint64_t notdup = WS::prim((fl & RSQC_FDUP) == 0).m;
    double inv_aligned = 1.0;                    // one block: len / aligned is exactly 1
    uint32_t aligned_of = 0; (void)aligned_of;
    if (NB > 1) {
        uint32_t aligned = 0;
#pragma unroll
        for (int b = 0; b < NB; ++b) aligned += len[b];
        inv_aligned = 1.0 / (double)(aligned ? aligned : 1u);     // (a slot adds len * (1 / aligned): within 1 ulp of len / aligned)
        aligned_of = aligned;
    }
#pragma unroll
    for (int k = 0; k < 2 * NB; ++k) {
        const uint64_t has = WS::prim(((eo.cmask >> k) & 1u) != 0).m;
        if (has == 0ull) continue;
        const uint32_t ln = len[k >> 1];
        const uint64_t hvm = has & WS::prim(ln > 0).m;
        const bool hv = WS::lane(LaneMask{hvm});
        // The input is coordinate-sorted, so the lanes of a tile that hit one exon / gene sit next to each other: the first lane
        // of a run of equal keys adds for the whole run (64 LDS atomics on one address take 64 passes of the LDS, one takes one).
        // One-block records add exactly 1 each, so their runs need no sum; fractions of longer records go lane by lane.
        if (NB == 1) {
            const RunLite r = make_run_lite(hvm, eo.eid[k]);
            if (r.head && !(K1E_ABL & 2)) T.exon_add(eo.eid[k], (double)r.count);
            // a one-block record that commits slot 0 is counted to that exon's gene (hit[0], see exon_metrics_ei: the first gene of
            // the set is the gene of the block's first containing exon): the exon's run serves the gene counters too
            if (k == 0) {
                const uint64_t nd = hvm & notdup;
                if (r.head && !(K1E_ABL & 2)) T.gene_add(eo.hit[0], r.count, run_popcount(nd, r.count));
            }
        } else if (!(K1E_ABL & 2)) {
#if K1E_EXON_RUNSUM
            // Neighbouring lanes that hit one exon with the same aligned length (the rule: reads of one length) add through their
            // first lane: (sum of block lengths) * (1 / aligned), one LDS atomic per run.  Lane by lane, the 64 adds of a tile inside
            // one exon are 64 passes of the LDS on one address -- ALL of the kernel's LDS bank conflicts (call r6b: 1.4e8 conflict
            // cycles with the table updates, 0 without).  A run whose lanes differ in aligned length adds lane by lane as before.
            const uint64_t cont = run_cont_mask(hvm, eo.eid[k]);
            if (cont == 0ull) { if (hv) T.exon_add(eo.eid[k], (double)ln * inv_aligned); }
            else {
                const uint32_t al = aligned_of;
                const uint64_t mixed_len = cont & WS::prim(lane_below(al) != al).m;          // a lane that continues a run with another aligned length
                if (mixed_len != 0ull) { if (hv) T.exon_add(eo.eid[k], (double)ln * inv_aligned); }
                else {
                    const uint32_t sc = wave_inclusive_scan_u32_dpp(hv ? ln : 0u);           // prefix sums of the lengths over the wave
                    const uint32_t cnt_run = run_length_at(cont);
                    const uint32_t at_end = lane_gather(sc, (uint32_t)lane_id() + cnt_run - 1u);
                    const bool head = WS::lane(LaneMask{hvm & ~cont});
                    if (head) T.exon_add(eo.eid[k], (double)(at_end - (sc - ln)) * inv_aligned);
                }
            }
#else
            if (hv) T.exon_add(eo.eid[k], (double)ln * inv_aligned);
#endif
        }
        const uint32_t base = hv ? eo.cidx[k] : 0u;
        if (!(K1E_ABL & 4)) {
            cov_add_merged(cov_diff, hvm, base, 1u);
            cov_add_merged(cov_diff, hvm, base + ln, 0xFFFFFFFFu);
        }
    }
#pragma unroll
    for (int k = 0; k < FAST_SET; ++k) {
        const uint64_t m = WS::prim(eo.n_hit > k).m;
        if (m == 0ull) break;
        const uint32_t g = eo.hit[k];
        if (!(K1E_ABL & 8)) k1e_emit_pairs<LANE_CHUNK>(T, m, g, qhash, qh2, pd, chunk);
        if (NB > 1 || k > 0) {
            const RunLite r = make_run_lite(m, g);
            const uint64_t nd = m & notdup;
            if (r.head && !(K1E_ABL & 2)) T.gene_add(g, r.count, run_popcount(nd, r.count));
        }
    }
}

// `index`: record index, | K1E_OVF_LONG when the general code also has to count the record's blocks and check its
// operations (a long-CIGAR straggler of a boundary tile, which classify_ei_kernel did not walk to the end)
constexpr uint64_t K1E_OVF_LONG = 1ull << 63;
// One reservation per wave call (the lanes of a call that overflow take consecutive slots): the list has ONE counter, and a returning
// memory atomic on one address completes at about 88 per microsecond chip-wide (MI355X_MICROARCH.md, "dequeue").
// `stage` / `stage_n` (classify_long_kernel): the workgroup's LDS buffer in front of the list -- its 36 k entries of the contract workload,
// one memory atomic each, were 0.42 ms of that kernel on their own (call r6b); the workgroup moves its buffer to the list when it retires.
constexpr uint32_t K1E_OVF_STAGE = 1024;
__device__ __forceinline__ void k1e_overflow(bool over, uint64_t index, unsigned long long *stage = nullptr, uint32_t *stage_n = nullptr) {
    const uint64_t m = WaveSink::prim(over).m;
    if (m == 0ull) return;
    const int lead = __ffsll((unsigned long long)m) - 1;
    const uint32_t n = (uint32_t)__popcll(m);
    uint32_t base = 0;
    bool staged = false;
    if (stage) {
        if (lane_id() == lead) base = atomicAdd(stage_n, n);
        base = lane_value(base, lead);
        staged = base + n <= K1E_OVF_STAGE;                  // (wave-uniform; a full buffer: straight to the list, the counter keeps the overshoot)
        if (staged && over) stage[base + mask_rank(m)] = index;
        if (!staged && lane_id() == lead) atomicAdd(stage_n, 0u - n);
    }
    if (!staged) {
        const DevAccum &acc = k1e_lazy_args()->acc;
        if (lane_id() == lead) base = atomicAdd(acc.ovf_count, n);
        base = lane_value(base, lead);
        if (over) {
            const uint32_t slot = base + mask_rank(m);
            if (slot < acc.ovf_cap) acc.ovf_index[slot] = index;
            else atomicExch(acc.error, RSQC_ERR_CAPACITY);
        }
    }
}

// ---- one-block tiles whose blocks ALL lie in ONE elementary interval (round 5) ------------------------------------------------------
// The queued records are neighbours in the sorted stream: on the contract workload two thirds of the one-block feature-stage calls
// have every block inside one interval (the body of an exon of an expressed gene, a stretch of intron; tools/uniform_tiles.py).  Such a
// call needs ONE index look-up, and it is a SCALAR one: the interval of the first lane's block from its rank word (s_load, scalar
// shift and popcount), the interval's entry and the start of the next one in one 64-byte scalar load, then two compares per lane
// decide whether every block lies inside [start, next start).  If so, everything exon_metrics_ei derives from the index --
// containing exons, gene set, class flags, globin, the commit slots -- is a property of the WAVE: the counters are popcounts of
// the lanes' quality / flag masks written to a counter lane chosen by a scalar, the exon and gene tables take one add per call
// instead of one per run of lanes, and only coverage slots and (gene, name) pairs remain per-lane work.  Unstranded runs only
// (--stranded makes the containing exons a per-lane property again); an interval under more than two exons (EIM_DEEP) and a
// contig without features take the general path, as does any call that fails the test.
#ifndef K1E_UNIFORM1
#define K1E_UNIFORM1 1
#endif
#ifndef K1E_UCACHE
#define K1E_UCACHE 1                         /* the uniform path asks the wave's last interval (LDS) before the index */
#endif
struct K1eInterval { EiEntry S; int32_t next_pos; };          // (wave-uniform)
// what one lane of the wave stored to LDS is read by all of them behind this point (the hardware runs a wave's LDS traffic in order;
// this keeps the COMPILER from moving a lane's loads across another lane's store, and lines the lanes of the host emulation up)
__device__ __forceinline__ void k1e_wave_lds_visible() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
// the interval of position x0 and the start of the one behind it, by scalar loads
__device__ __forceinline__ K1eInterval k1e_interval_of(const DevAnnotation &a, const ContigInfo &ci, int32_t x0) {
    const int32_t top = (int32_t)(ci.rk_words << 6) - 1;
    const int32_t xc = x0 < 0 ? 0 : (x0 > top ? top : x0);
    const uint32_t word = ci.rk_base + ((uint32_t)xc >> 6);
    K1eInterval r;
#if defined(__HIP_DEVICE_COMPILE__)
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    typedef uint32_t u32x16 __attribute__((ext_vector_type(16)));
    u32x4 w;
    asm volatile("s_load_dwordx4 %0, %1, %2\n\ts_waitcnt lgkmcnt(0)" : "=s"(w) : "s"(a.ei_rank), "s"(word * 16u));
    const uint64_t t = (((uint64_t)w.y << 32) | (uint64_t)w.x) << (63u - ((uint32_t)xc & 63u));
    const uint32_t j = w.z + (uint32_t)__popcll(t) - 1u;
    u32x16 e;                                                   // entry j and the first words of entry j + 1 (the table ends with a terminator entry)
    asm volatile("s_load_dwordx16 %0, %1, %2\n\ts_waitcnt lgkmcnt(0)" : "=s"(e) : "s"(a.ei), "s"(j * 32u));
    r.S = EiEntry{(int32_t)e.s0, e.s1, e.s2, e.s3, e.s4, e.s5, e.s6, e.s7};
    r.next_pos = (int32_t)e.s8;
#else
    const EiRank w = a.ei_rank[word];
    const uint64_t t = (((uint64_t)w.hi << 32) | (uint64_t)w.lo) << (63u - ((uint32_t)xc & 63u));
    const uint32_t j = w.rank + (uint32_t)__builtin_popcountll(t) - 1u;
    r.S = a.ei[j]; r.next_pos = a.ei[j + 1].pos;
#endif
    return r;
}
// one lane per counter, chosen at run time (WaveSink::add takes the lane as a template argument: v_writelane with a constant lane
// select; with the value AND the lane in scalar registers the instruction would read two of them, one more than gfx9's constant bus
// carries, and routing the lane through M0 means clobbering a register the compiler reserves -- a compare and a select instead)
__device__ __forceinline__ void k1e_count_at(WaveSink &cnt, int counter, uint32_t n) {
    cnt.vec = lane_id() == counter ? n : cnt.vec;
}
// class_counts_b (src/Expression.cpp:407-457) of a call whose class flags `cf` and gene set (`any_gene`: it is not empty) are the WAVE's:
// one class for all of its records -- the counters are popcounts of the lanes' quality / flag masks
__device__ __forceinline__ void k1e_uniform_class_counts(const DevParams &p, uint32_t cf, bool any_gene, uint32_t flhq, uint64_t onm,
                                                         uint32_t n_on, uint32_t n_hq, WaveSink &cnt) {
    typedef WaveSink WS;
    const bool exonic = (cf & CF_EXONIC) != 0, intragenic = (cf & CF_INTRAGENIC) != 0;
    const int cls = !exonic ? (intragenic ? RSQC_C_INTRONIC_READS : RSQC_C_INTERGENIC_READS) : (any_gene ? RSQC_C_EXONIC_READS : RSQC_C_AMBIGUOUS_READS);
    static_assert(RSQC_C_HQ_INTRONIC_READS == RSQC_C_INTRONIC_READS + 2 && RSQC_C_HQ_INTERGENIC_READS == RSQC_C_INTERGENIC_READS + 1 &&
                  RSQC_C_HQ_EXONIC_READS == RSQC_C_EXONIC_READS + 1 && RSQC_C_HQ_AMBIGUOUS_READS == RSQC_C_AMBIGUOUS_READS + 1, "counter order");
    k1e_count_at(cnt, cls, n_on); k1e_count_at(cnt, cls + (cls == RSQC_C_INTRONIC_READS ? 2 : 1), n_hq);
    if ((!exonic && intragenic) || (exonic && any_gene)) { k1e_count_at(cnt, RSQC_C_INTRAGENIC_READS, n_on); k1e_count_at(cnt, RSQC_C_HQ_INTRAGENIC_READS, n_hq); }
    if (cf & CF_RIBOSOMAL) k1e_count_at(cnt, RSQC_C_RRNA_READS, n_on);
    const bool plus = (cf & CF_PLUS) != 0, minus = (cf & CF_MINUS) != 0;
    if (plus != minus) {                                                                        // :445-457
        const uint64_t one = p.unpaired ? onm : (onm & WS::prim((flhq & RSQC_FPAIRED) != 0).m);
        const uint64_t revm = WS::prim((flhq & RSQC_FREVERSE) != 0).m, sense = minus ? revm : ~revm;
        const uint64_t end1 = p.unpaired ? ~0ull : WS::prim((flhq & RSQC_FREAD1) != 0).m;
        RSQC_COUNT(cnt, RSQC_C_END1_SENSE, LaneMask{one & end1 & sense}); RSQC_COUNT(cnt, RSQC_C_END1_ANTISENSE, LaneMask{one & end1 & ~sense});
        RSQC_COUNT(cnt, RSQC_C_END2_SENSE, LaneMask{one & ~end1 & sense}); RSQC_COUNT(cnt, RSQC_C_END2_ANTISENSE, LaneMask{one & ~end1 & ~sense});
    }
}
// true: the call was handled here.  `onm`: lanes that hold a record (the low n lanes).
__device__ __forceinline__ bool k1e_uniform1(const DevAnnotation &a, const DevParams &p, uint32_t *cov_diff, const ContigInfo &ci, K1eTables &T,
                                             int32_t bs, uint32_t len, uint32_t flhq, uint64_t qhash, uint32_t qh2, uint64_t onm,
                                             WaveSink &cnt, const K1ePairDst &held, int wave) {
    typedef WaveSink WS;
    if (p.stranded != RSQC_STRAND_UNKNOWN || ci.rk_words == 0u) return false;
    uint32_t *const uc = T.ucache[wave]; (void)uc;
    EiEntry S;
    bool cached = false;
#if K1E_UCACHE
    {   // the interval of the call before this one (K1eTables::ucache): the same one more often than not
        const int32_t lo = (int32_t)__builtin_amdgcn_readfirstlane((int)uc[8]), hi = (int32_t)__builtin_amdgcn_readfirstlane((int)uc[9]);
        cached = (WS::prim(bs < lo || (int32_t)(bs + (int32_t)len) >= hi).m & onm) == 0ull;
        if (cached) {
            const uint4 e0 = *reinterpret_cast<const uint4 *>(uc), e1 = *reinterpret_cast<const uint4 *>(uc + 4);
#define K1E_U(x) ((uint32_t)__builtin_amdgcn_readfirstlane((int)(x)))
            S = EiEntry{(int32_t)K1E_U(e0.x), K1E_U(e0.y), K1E_U(e0.z), K1E_U(e0.w), K1E_U(e1.x), K1E_U(e1.y), K1E_U(e1.z), K1E_U(e1.w)};
#undef K1E_U
#if defined(RSQC_WAVE_EMU)
            if (lane_id() == 0) ++g_k1e_ucache_hits;
#endif
        }
    }
#endif
    if (!cached) {
        const K1eInterval iv = k1e_interval_of(a, ci, (int32_t)lane_value((uint32_t)bs, 0));      // (lane 0 holds a record: n >= 1)
        S = iv.S;
        if (S.mask & EIM_DEEP) return false;
        const int32_t lo = S.pos, hi = iv.next_pos > S.pos ? iv.next_pos : 0x7FFFFFFF;              // (next_pos <= pos: the contig's last interval)
        if ((WS::prim(bs < lo || (int32_t)(bs + (int32_t)len) >= hi).m & onm) != 0ull) return false;
#if K1E_UCACHE
        if (k1e_first_lane()) {
            *reinterpret_cast<uint4 *>(uc) = make_uint4((uint32_t)S.pos, S.mask, S.eidA, S.eidB);
            *reinterpret_cast<uint4 *>(uc + 4) = make_uint4(S.gfA, S.cdA, S.gfB, S.cdB);
            *reinterpret_cast<uint2 *>(uc + 8) = make_uint2((uint32_t)lo, (uint32_t)hi);
        }
        k1e_wave_lds_visible();
#endif
    }
#if defined(RSQC_WAVE_EMU)
    if (lane_id() == 0) ++g_k1e_uniform_calls;
#endif
    // ---- exon_metrics_ei<1> with js == je == j1 in every lane: every value below is wave-uniform --------------------------------
    const bool cA = S.eidA != EI_NONE, cB = S.eidB != EI_NONE, c1 = cA && cB;
    const uint32_t g0 = S.gfA & ROW_GENE_MASK, g1 = S.gfB & ROW_GENE_MASK;
    const uint32_t gX = cA ? g0 : g1, gfX = cA ? S.gfA : S.gfB;
    const bool va = cA || cB, vb = c1 && g1 != gX;                                              // genes.front(), src/Expression.cpp:363-367
    const bool globin = (va && ((gfX >> ROW_FLAG_SHIFT) & ROWF_GLOBIN) != 0) || (vb && ((S.gfB >> ROW_FLAG_SHIFT) & ROWF_GLOBIN) != 0);
    const uint64_t hqm = WS::prim((flhq & K1E_HQ) != 0).m & onm, dupm = WS::prim((flhq & RSQC_FDUP) != 0).m;
    const uint32_t n_on = (uint32_t)__popcll(onm), n_hq = (uint32_t)__popcll(hqm);
    if (!globin) {                                                                              // :363,395-404
        RSQC_COUNT(cnt, RSQC_C_NON_GLOBIN_READS, LaneMask{onm}); RSQC_COUNT(cnt, RSQC_C_NON_GLOBIN_DUPLICATE_READS, LaneMask{onm & dupm});
    }
    k1e_uniform_class_counts(p, ei_class_flags(S.mask, RSQC_STRAND_UNKNOWN), va, flhq, onm, n_on, n_hq, cnt);
    // ---- k1e_commit<1>: high-quality records of a call that has a gene (:377-392) -------------------------------------------------
    if (!va || hqm == 0ull) return true;
    const uint64_t hvm = hqm & WS::prim(len > 0).m;                                              // (a zero-length block commits nothing)
    if (hvm == 0ull) return true;
    const K1ePairDst pd = K1E_LAZYPAIR ? k1e_pair_dst() : held;
    const uint32_t n0 = (uint32_t)__popcll(hvm), nd0 = (uint32_t)__popcll(hvm & ~dupm);
    const bool first = k1e_first_lane();
    if (!(K1E_ABL & 2)) {
        if (first) { T.exon_add(cA ? S.eidA : S.eidB, (double)n0); T.gene_add(gX, n0, nd0); }
        if (first && c1) T.exon_add(S.eidB, (double)n0);
        if (first && vb) T.gene_add(g1, n0, nd0);
    }
    if (!(K1E_ABL & 4)) {
        const uint32_t base = (cA ? S.cdA : S.cdB) + (uint32_t)bs;
        cov_add_merged(cov_diff, hvm, base, 1u); cov_add_merged(cov_diff, hvm, base + len, 0xFFFFFFFFu);
        if (c1) { const uint32_t b1 = S.cdB + (uint32_t)bs; cov_add_merged(cov_diff, hvm, b1, 1u); cov_add_merged(cov_diff, hvm, b1 + len, 0xFFFFFFFFu); }
    }
    if (!(K1E_ABL & 8)) {
        k1e_emit_pairs(T, hvm, gX, qhash, qh2, pd, blockIdx.x);
        if (vb) k1e_emit_pairs(T, hvm, g1, qhash, qh2, pd, blockIdx.x);
    }
    return true;
}

// ---- two-block calls whose 64 records cross ONE junction (round 6, third session) ---------------------------------------------------
// The analogue of k1e_uniform1 for the two-block queue.  Its records are neighbours in the sorted stream of a spliced gene: on the contract
// workload 70 % of the two-block calls have every first block inside ONE elementary interval (the exon in front of the junction) and every
// second block inside ONE other (the exon behind it).  Two scalar look-ups (both rank words in flight together, then both entries) and four
// compares per lane decide that; then exon_metrics_ei<2> is a computation on SCALARS -- containing exons, the intersection of the two gene
// sets, class flags, globin, the commit slots -- and k1e_commit<2> shrinks to one exon add per block and slot (the sum of the lanes' block
// lengths times 1 / aligned when the lanes share one aligned length -- reads of one length; lane by lane otherwise), one gene add per gene,
// and the per-lane coverage slots and (gene, name) pairs.
// A block that ENDS on its interval's last base has be = bs + len on the next interval's first position (blocks are [bs, bs + len], both
// ends inclusive: src/Expression.cpp:111) -- every first block of a spliced read does.  Such a block takes the next interval's class bits
// too (ei_resolve: m_je), its containing exons stay the first interval's (find(max(bs, be - 1)), src/GTF.cpp:181-186): the call stays
// uniform when ALL of its lanes touch the next interval or none does.
// MEASURED AND LEFT OFF (call r6x, profiles/r6_k1_variants.txt): K1 2.231 / 2.244 ms with the path, 2.203 without, same box.  The path trades
// ~150 vector instructions of a general two-block call for ~400 SCALAR ones (the gene-set logic and the commit's bookkeeping on SGPRs), and
// the CU's one scalar port is the kernel's tighter floor (1.17 ms against 0.73 ms for the vector pipes) -- the same reason
// k1e_uniform1 measures +-0.  Opt-in build (-DK1E_UNIFORM2=1); the host emulation's superset build runs it (tests/hostemu, build_k1).
#ifndef K1E_UNIFORM2
#define K1E_UNIFORM2 0
#endif
struct K1eInterval2 { EiEntry S[2]; int32_t next_pos[2]; uint32_t next_mask[2]; };          // (wave-uniform)
__device__ __forceinline__ K1eInterval2 k1e_intervals_of(const DevAnnotation &a, const ContigInfo &ci, int32_t x0, int32_t x1) {
    const int32_t top = (int32_t)(ci.rk_words << 6) - 1;
    const int32_t xc0 = x0 < 0 ? 0 : (x0 > top ? top : x0), xc1 = x1 < 0 ? 0 : (x1 > top ? top : x1);
    const uint32_t word0 = ci.rk_base + ((uint32_t)xc0 >> 6), word1 = ci.rk_base + ((uint32_t)xc1 >> 6);
    K1eInterval2 r;
#if defined(__HIP_DEVICE_COMPILE__)
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    typedef uint32_t u32x16 __attribute__((ext_vector_type(16)));
    u32x4 w0, w1;
    asm volatile("s_load_dwordx4 %0, %2, %3\n\ts_load_dwordx4 %1, %2, %4\n\ts_waitcnt lgkmcnt(0)"
                 : "=&s"(w0), "=&s"(w1) : "s"(a.ei_rank), "s"(word0 * 16u), "s"(word1 * 16u));
    const uint64_t t0 = (((uint64_t)w0.y << 32) | (uint64_t)w0.x) << (63u - ((uint32_t)xc0 & 63u));
    const uint64_t t1 = (((uint64_t)w1.y << 32) | (uint64_t)w1.x) << (63u - ((uint32_t)xc1 & 63u));
    const uint32_t j0 = w0.z + (uint32_t)__popcll(t0) - 1u, j1 = w1.z + (uint32_t)__popcll(t1) - 1u;
    u32x16 e0, e1;                                              // entry j and the first words of entry j + 1 (the table ends with a terminator entry)
    asm volatile("s_load_dwordx16 %0, %2, %3\n\ts_load_dwordx16 %1, %2, %4\n\ts_waitcnt lgkmcnt(0)"
                 : "=&s"(e0), "=&s"(e1) : "s"(a.ei), "s"(j0 * 32u), "s"(j1 * 32u));
    r.S[0] = EiEntry{(int32_t)e0.s0, e0.s1, e0.s2, e0.s3, e0.s4, e0.s5, e0.s6, e0.s7}; r.next_pos[0] = (int32_t)e0.s8; r.next_mask[0] = e0.s9;
    r.S[1] = EiEntry{(int32_t)e1.s0, e1.s1, e1.s2, e1.s3, e1.s4, e1.s5, e1.s6, e1.s7}; r.next_pos[1] = (int32_t)e1.s8; r.next_mask[1] = e1.s9;
#else
    const int32_t xc[2] = {xc0, xc1}; const uint32_t word[2] = {word0, word1};
    for (int b = 0; b < 2; ++b) {
        const EiRank w = a.ei_rank[word[b]];
        const uint64_t t = (((uint64_t)w.hi << 32) | (uint64_t)w.lo) << (63u - ((uint32_t)xc[b] & 63u));
        const uint32_t j = w.rank + (uint32_t)__builtin_popcountll(t) - 1u;
        r.S[b] = a.ei[j]; r.next_pos[b] = a.ei[j + 1].pos; r.next_mask[b] = a.ei[j + 1].mask;
    }
#endif
    return r;
}
// true: the call was handled here.  `onm`: lanes that hold a record (the low n lanes).
__device__ __forceinline__ bool k1e_uniform2(const DevAnnotation &a, const DevParams &p, uint32_t *cov_diff, const ContigInfo &ci, K1eTables &T,
                                             const int32_t (&bs)[2], const uint32_t (&len)[2], uint32_t flhq, uint64_t qhash, uint32_t qh2,
                                             uint64_t onm, WaveSink &cnt, const K1ePairDst &held) {
    typedef WaveSink WS;
    if (p.stranded != RSQC_STRAND_UNKNOWN || ci.rk_words == 0u) return false;
    const K1eInterval2 iv = k1e_intervals_of(a, ci, (int32_t)lane_value((uint32_t)bs[0], 0), (int32_t)lane_value((uint32_t)bs[1], 0));   // (lane 0 holds a record: n >= 1)
    if ((iv.S[0].mask | iv.S[1].mask) & EIM_DEEP) return false;
    uint32_t mask = 0;
#pragma unroll
    for (int b = 0; b < 2; ++b) {
        const int32_t lo = iv.S[b].pos, hi = iv.next_pos[b] > iv.S[b].pos ? iv.next_pos[b] : 0x7FFFFFFF;    // (next_pos <= pos: the contig's last interval)
        const int32_t be = bs[b] + (int32_t)len[b];
        if ((WS::prim(bs[b] < lo || bs[b] >= hi || be > hi).m & onm) != 0ull) return false;
        const uint64_t touch = WS::prim(be == hi).m & onm;
        if (touch != 0ull && touch != onm) return false;
        mask |= iv.S[b].mask | (touch != 0ull ? iv.next_mask[b] & ~EIM_DEEP : 0u);
    }
#if defined(RSQC_WAVE_EMU)
    if (lane_id() == 0) ++g_k1e_uniform2_calls;
#endif
    // ---- exon_metrics_ei<2> with js == j1 the looked-up interval of each block in every lane: every value below is wave-uniform ---------
    uint32_t eid[4], cd[4], con = 0, ma = 0, mb = 0, la = 0, lb = 0;
    bool va = false, vb = false, ga = false, gb = false;
#pragma unroll
    for (int b = 0; b < 2; ++b) {
        const EiEntry &S = iv.S[b];
        const bool cA = S.eidA != EI_NONE, cB = S.eidB != EI_NONE, c0 = cA || cB, c1 = cA && cB;
        const uint32_t g0 = S.gfA & ROW_GENE_MASK, g1 = S.gfB & ROW_GENE_MASK;
        const uint32_t gX = cA ? g0 : g1, gfX = cA ? S.gfA : S.gfB;
        eid[2 * b] = cA ? S.eidA : S.eidB; cd[2 * b] = cA ? S.cdA : S.cdB; eid[2 * b + 1] = S.eidB; cd[2 * b + 1] = S.cdB;
        con |= (c0 ? 1u : 0u) << (2 * b) | (c1 ? 1u : 0u) << (2 * b + 1);
        if (b == 0) {                                                                           // genes.front(), :363-367
            la = gX; va = c0; ga = c0 && ((gfX >> ROW_FLAG_SHIFT) & ROWF_GLOBIN) != 0;
            lb = g1; vb = c1 && g1 != gX; gb = vb && ((S.gfB >> ROW_FLAG_SHIFT) & ROWF_GLOBIN) != 0;
            ma = (c0 ? 1u : 0u) | ((c1 && g1 == gX) ? 2u : 0u); mb = vb ? 2u : 0u;
        } else {                                                                                // set_intersection, :368-374
            const bool a_in = (c0 && la == gX) || (c1 && la == g1), b_in = (c0 && lb == gX) || (c1 && lb == g1);
            va = va && a_in; vb = vb && b_in;
            ma |= ((c0 && la == gX) ? 1u : 0u) << 2 | ((c1 && la == g1) ? 1u : 0u) << 3;
            mb |= ((c0 && lb == gX) ? 1u : 0u) << 2 | ((c1 && lb == g1) ? 1u : 0u) << 3;
        }
    }
    ga = ga && va; gb = gb && vb;
    const bool any_gene = va || vb;
    const uint64_t hqm = WS::prim((flhq & K1E_HQ) != 0).m & onm, dupm = WS::prim((flhq & RSQC_FDUP) != 0).m;
    const uint32_t n_on = (uint32_t)__popcll(onm), n_hq = (uint32_t)__popcll(hqm);
    if (!(ga || gb)) {                                                                          // :363,395-404
        RSQC_COUNT(cnt, RSQC_C_NON_GLOBIN_READS, LaneMask{onm}); RSQC_COUNT(cnt, RSQC_C_NON_GLOBIN_DUPLICATE_READS, LaneMask{onm & dupm});
    }
    k1e_uniform_class_counts(p, ei_class_flags(mask, RSQC_STRAND_UNKNOWN), any_gene, flhq, onm, n_on, n_hq, cnt);
    // ---- k1e_commit<2>: high-quality records of a call that has a gene (:377-392) -------------------------------------------------
    if (!any_gene || hqm == 0ull) return true;
    const uint32_t cmask = con & ((va ? ma : 0u) | (vb ? mb : 0u));
    const K1ePairDst pd = K1E_LAZYPAIR ? k1e_pair_dst() : held;
    const uint32_t aligned = len[0] + len[1];
    const bool first = k1e_first_lane();
#pragma unroll
    for (int b = 0; b < 2; ++b) {
        const uint32_t bits = (cmask >> (2 * b)) & 3u;                                          // (bit 1 only with bit 0: con, and B is the same gene's or the second gene's)
        if (bits == 0u) continue;
        const uint32_t ln = len[b];
        const uint64_t hvm = hqm & WS::prim(ln > 0).m;                                          // (a zero-length block commits nothing)
        if (hvm == 0ull) continue;
        const bool hv = WS::lane(LaneMask{hvm});
        if (!(K1E_ABL & 2)) {
            // exonCounts[eid] += len / aligned (src/Expression.cpp:345): the lanes share the exon; when they also share the aligned length
            // (the first committing lane's) the call adds (sum of their block lengths) * (1 / aligned) once, as k1e_commit does per run
            const uint32_t al0 = lane_value(aligned, __ffsll((unsigned long long)hvm) - 1);
            if ((hvm & WS::prim(aligned != al0).m) == 0ull) {
                const uint32_t tot = wave_sum_u32_full(hv ? ln : 0u);
                const double frac = (double)tot * (1.0 / (double)al0);
                if (first && (bits & 1u)) T.exon_add(eid[2 * b], frac);
                if (first && (bits & 2u)) T.exon_add(eid[2 * b + 1], frac);
            } else if (hv) {
                const double frac = (double)ln * (1.0 / (double)aligned);
                if (bits & 1u) T.exon_add(eid[2 * b], frac);
                if (bits & 2u) T.exon_add(eid[2 * b + 1], frac);
            }
        }
        if (!(K1E_ABL & 4)) {
#pragma unroll
            for (int j = 0; j < 2; ++j)
                if (bits & (1u << j)) {
                    const uint32_t base = cd[2 * b + j] + (uint32_t)bs[b];
                    cov_add_merged(cov_diff, hvm, base, 1u); cov_add_merged(cov_diff, hvm, base + ln, 0xFFFFFFFFu);
                }
        }
    }
    const uint64_t m = hqm & WS::prim(aligned > 0).m;                                           // (exon_metrics_ei: hits only with aligned > 0)
    if (m == 0ull) return true;
    const uint32_t n0 = (uint32_t)__popcll(m), nd0 = (uint32_t)__popcll(m & ~dupm);
    const uint32_t hit0 = va ? la : lb;
    if (!(K1E_ABL & 8)) k1e_emit_pairs(T, m, hit0, qhash, qh2, pd, blockIdx.x);
    if (first && !(K1E_ABL & 2)) T.gene_add(hit0, n0, nd0);
    if (va && vb) {
        if (!(K1E_ABL & 8)) k1e_emit_pairs(T, m, lb, qhash, qh2, pd, blockIdx.x);
        if (first && !(K1E_ABL & 2)) T.gene_add(lb, n0, nd0);
    }
    return true;
}

constexpr uint32_t K1E_TAB_BLOCK3 = CIG_BLOCK_SET | (CIG_BLOCK_SET << 16), K1E_TAB_REF3 = CIG_REF_SET | (CIG_REF_SET << 16), K1E_TAB_BAD3 = 0xFE00FE00u;   // (k1e_walk3; the same tables as k1e_walk's)
// ---- the feature stage of 64 queued records of NB blocks each (n < 64 only when a queue is drained) -----------------
template <int NB>
__device__ __forceinline__ void k1e_process(const DevAnnotation &a, const DevParams &p, const rsqc_rec_aux *aux, uint32_t *cov_diff,
                                            const ContigInfo &ci, K1eShared &S, int wave, uint32_t head, uint32_t n,
                                            const uint32_t *qh2col, const K1ePairDst &held) {
    static_assert(NB >= 1 && NB <= 3, "queues of one-, two- and three-block records");
    const int l = lane_id();
    const bool on = (uint32_t)l < n;
    const uint32_t slot = NB == 3 ? k1e_wrap3(head + (uint32_t)l) : ((head + (uint32_t)l) & (K1E_QCAP - 1));
    int32_t bs[NB]; uint32_t len[NB]; uint32_t idx, flhq, pre0 = 0u;
    if (NB == 1) {
        const K1eQueue1 &q = S.q1[wave];
        const uint32_t lf = q.lf[slot];
        bs[0] = (int32_t)q.bs[slot]; len[0] = lf & 0xFFFFu; idx = q.idx[slot];
#ifdef K1E_COARSE
        pre0 = q.cz[slot];
#endif
        flhq = ((lf >> 16) & 0xFFFu) | ((lf >> 28) << 16);                  // flag bits 0-11, K1E_HQ
    } else if (NB == 2) {
        const K1eQueue2 &q = S.q2[wave];
        bs[0] = (int32_t)q.bs0[slot]; len[0] = q.len0[slot]; bs[NB - 1] = (int32_t)q.bs1[slot]; len[NB - 1] = q.len1[slot]; idx = q.idx[slot]; flhq = q.flhq[slot];
    } else {
        const K1eQueue3 &q = S.q3[wave];
        const uint32_t l01 = q.lens01[slot], lf = q.lf2[slot];
        bs[0] = (int32_t)q.bs0[slot]; len[0] = l01 & 0xFFFFu; bs[NB > 1 ? 1 : 0] = (int32_t)q.bs1[slot]; len[NB > 1 ? 1 : 0] = l01 >> 16;
        bs[NB - 1] = (int32_t)q.bs2[slot]; len[NB - 1] = lf & 0xFFFFu; idx = q.idx[slot];
        flhq = ((lf >> 16) & 0xFFFu) | ((lf >> 28) << 16);                  // flag bits 0-11, K1E_HQ
    }
    if (!on) { idx = 0u; flhq = 0u; pre0 = 0u; }
#if defined(RSQC_WAVE_EMU)
    if (pre0) ++g_k1e_coarse_hits;                 // (test harness: records answered by the coarse table)
#endif
    // the name hash is only needed by records that are counted to a gene: it comes back from the record array (the lines
    // were streamed through this CU's caches a few tiles ago) instead of riding through the queue.
    // (Call r6o: issued BEHIND the entry loads of the two- and three-block stages instead -- loads return in order, so in front of the rank words
    //  these gathers are waited for by the first look-up round -- K1 2.27 -> 2.30 ms: their latency is then exposed at the pair stores.)
    const uint2 qh = ld32(reinterpret_cast<const uint2 *>(aux), idx * 2u);
    const uint32_t qh2 = qh2col ? ld32(qh2col, idx) : 0u;        // (uniform branch: the batch carries second name hashes or it does not)
    K1E_SMARK(1);                                          // [1] queue entries read
    WaveSink cnt;
    const uint64_t onm = n >= 64u ? ~0ull : (1ull << n) - 1ull;
    bool done = false;
    if (NB == 1 && K1E_UNIFORM1) done = k1e_uniform1(a, p, cov_diff, ci, S.T, bs[0], len[0], flhq, (uint64_t)qh.x | ((uint64_t)qh.y << 32), qh2, onm, cnt, held, wave);
    if (NB == 2 && K1E_UNIFORM2) {
        const int32_t b2[2] = {bs[0], bs[NB - 1]}; const uint32_t l2[2] = {len[0], len[NB - 1]};
        done = k1e_uniform2(a, p, cov_diff, ci, S.T, b2, l2, flhq, (uint64_t)qh.x | ((uint64_t)qh.y << 32), qh2, onm, cnt, held);
    }
    if (!done) {
        EiOut eo; bool over = false;
        exon_metrics_ei<NB, WaveSink>(a, p, ci, flhq & 0xFFFFu, bs, len, (flhq & K1E_HQ) != 0, eo, over, cnt, on, (uint32_t)NB, pre0);
        k1e_overflow(on && over, (uint64_t)idx);
        k1e_commit<NB>(cov_diff, S.T, eo, len, flhq, (uint64_t)qh.x | ((uint64_t)qh.y << 32), qh2, held, blockIdx.x);
        K1E_SMARK(6);                                      // [6] commit (LDS tables, coverage atomics, pairs)
    } else { K1E_SMARK(7); }                               // [7] a one- or two-block call answered by a wave-uniform path, whole
    if (k1e_lane_below<RSQC_N_COUNTERS>() && cnt.vec) atomicAdd(&S.T.cnt32[l], cnt.vec);
}

// ---- the first eight operations of a CIGAR on bit fields, THREE blocks captured (round 6) -------------------------------------------
// Same idiom as k1e_walk below (operation class = a bit of a 16-bit table indexed by the operation word, select masks, v_bfi), but
// forwards, with the captured blocks in a three-deep shift register (a block pushes the two before it down): after the walk
// (b0, l0) is the LAST block seen, (b1, l1) the one before it, (b2, l2) the one before that -- for a record of at most three blocks,
// the only kind the caller uses them for, all of its blocks.  No per-operation arrays stay live (the backward selects of k1e_walk
// need the eight starts and eight class masks at once: sixteen registers this stage does not have at five waves per SIMD).
struct Walk3 { uint32_t ref_len, nb, bad; uint32_t b0, b1, b2, l0, l1, l2; };
// `cigar`: the record's operations in memory, for the reference length of a record with more than eight of them (rare; such a record's
// blocks and legality are classify_long_kernel's to count / check)
__device__ __forceinline__ void k1e_walk3(int32_t pos, uint32_t n, const uint32_t (&c)[8], const uint32_t *cigar, Walk3 &w) {
    uint32_t cur = (uint32_t)pos + 1u;                                     // 1-based position of the next reference base
    uint32_t b0 = 0u, b1 = 0u, b2 = 0u, l0 = 0u, l1 = 0u, l2 = 0u, nbneg = 0u, bad = 0u;
    const uint32_t live = n >= 8u ? 0xFFu : ((1u << n) - 1u);              // bit k: operation k exists
    auto op = [&](int k) {
        const uint32_t ck = bfi(bfe_m(live, (uint32_t)k), c[k], 5u);       // past the end: a hard clip of length 0 (no block, no reference, legal)
        const uint32_t len = ck >> 4, blk = bfe_m(K1E_TAB_BLOCK3, ck);
        bad |= bfe_u(K1E_TAB_BAD3, ck, 1u);                                // Expression.cpp:61-63
        nbneg += blk;
        b2 = bfi(blk, b1, b2); l2 = bfi(blk, l1, l2);
        b1 = bfi(blk, b0, b1); l1 = bfi(blk, l0, l1);
        b0 = bfi(blk, cur, b0); l0 = bfi(blk, len, l0);
        cur += len & bfe_m(K1E_TAB_REF3, ck);
    };
#if K1E_WALK_SKIP
    // operations 5, 6, 7 exist in few records (four blocks, three blocks with clips or an indel): a tile without one skips them (a scalar
    // branch; for the lanes of a tile that takes them the masked operations are the same no-ops as before)
#pragma unroll
    for (int k = 0; k < 5; ++k) op(k);
    if (__ballot(n > 5u) != 0ull) {
#pragma unroll
        for (int k = 5; k < 8; ++k) op(k);
    }
#else
#pragma unroll
    for (int k = 0; k < 8; ++k) op(k);
#endif
    if (__ballot(n > 8) != 0ull) {
        for (uint32_t i = 8; i < n; ++i) { const uint32_t cx = cigar[i]; cur += (cx >> 4) & bfe_m(K1E_TAB_REF3, cx); }
        K1E_LANDED(cur);
    }
    w.ref_len = cur - ((uint32_t)pos + 1u); w.nb = 0u - nbneg; w.bad = bad;
    w.b0 = b0; w.b1 = b1; w.b2 = b2; w.l0 = l0; w.l1 = l1; w.l2 = l2;
}

// a record for classify_long_kernel: the workgroup's own region of the deferred list (slot = first record of its range + an LDS
// counter: a record is listed at most once), no memory atomic.  Entry: record index | high quality << 31.
__device__ __forceinline__ void k1e_defer(K1eTables &T, uint64_t m, uint32_t idx, bool hq, uint32_t wg_beg) {
    if (m == 0ull) return;
    const int lead = __ffsll((unsigned long long)m) - 1;
    uint32_t base = 0;
    if (lane_id() == lead) base = atomicAdd(&T.defer, (uint32_t)__popcll(m));
    base = lane_value(base, lead);
    if (WaveSink::lane(LaneMask{m})) k1e_lazy_args()->acc.defer_index[wg_beg + base + mask_rank(m)] = idx | (hq ? 0x80000000u : 0u);
}

// ---- 64 deferred records (classify_long_kernel): record words, CIGAR and name hashes come from memory, the CIGAR is walked in
// full -- every block counted, the first FAST_BLOCKS captured -- and the record takes the feature stage with its own block count ----
template <bool LANE_CHUNK>
__device__ __forceinline__ void k1e_long_call(const DevAnnotation &a, const DevParams &p, const DevBatch &b, uint32_t *cov_diff,
                                              const ContigInfo &ci, K1eTables &T, uint32_t idx, bool hq, bool on0, uint32_t &sum_blk,
                                              const K1ePairDst &held, uint32_t chunk, unsigned long long *stage, uint32_t *stage_n) {
    const int l = lane_id();
    if (!on0) idx = 0u;
    const int4 cv = ld32(reinterpret_cast<const int4 *>(b.core), idx);
    const int4 av = ld32(reinterpret_cast<const int4 *>(b.aux), idx);
    const uint32_t fl = (uint32_t)av.z & 0xFFFFu;
    uint32_t cg[8];
    k1e_load_cigar8(b.cigar, (uint32_t)cv.w, cg);
    uint32_t n_cigar = (uint32_t)av.w >> 24;
    bool ok = true;
    if (on0 && n_cigar == RSQC_NCIGAR_ESCAPE) {
        const DevBatch &bw = k1e_lazy_args()->b;
        uint32_t lo = 0, hi = bw.n_wide;
        while (lo < hi) { uint32_t m = (lo + hi) >> 1; if (bw.wide_index[m] < idx) lo = m + 1; else hi = m; }
        if (lo >= bw.n_wide || bw.wide_index[lo] != idx) ok = false; else n_cigar = bw.wide_n_cigar[lo];
        K1E_LANDED(n_cigar);
    }
    CigarWalk cw; Blocks B;
    cw.ref_len = 0; cw.nblocks = 0; cw.aligned = 0; cw.bad = false;
#pragma unroll
    for (int k = 0; k < FAST_BLOCKS; ++k) { B.bs[k] = 0; B.len[k] = 0; }
    const uint32_t nc = (on0 && ok) ? n_cigar : 0u;
#pragma unroll
    for (int k = 0; k < 8; ++k) cigar_op(cg[k], cv.x, cw, B, (uint32_t)k < nc);
    if (__ballot(nc > 8) != 0ull) {
        const uint32_t *cp = b.cigar + (uint32_t)cv.w;
        for (uint32_t i = 8; i < nc; ++i) cigar_op(cp[i], cv.x, cw, B);
        K1E_LANDED(cw.nblocks); K1E_LANDED(cw.ref_len);
    }
    const bool longc = nc > 8;                                // classify_ei_kernel left legality and the block count of these to this kernel
    if (longc && cw.bad) atomicExch(k1e_lazy_args()->acc.error, RSQC_ERR_BAD_CIGAR);
    const bool on = on0 && ok && !(longc && cw.bad);
    sum_blk += (on && longc) ? cw.nblocks : 0u;                                       // src/RNASeQC.cpp:360
    WaveSink cnt;
    {
        const bool none = on && cw.nblocks == 0;
        RSQC_COUNT(cnt, RSQC_C_INTERGENIC_READS, none); RSQC_COUNT(cnt, RSQC_C_HQ_INTERGENIC_READS, none && hq);
    }
    const bool fast = on && cw.nblocks >= 1 && cw.nblocks <= (uint32_t)FAST_BLOCKS;
    const uint64_t qhash = (uint64_t)(uint32_t)av.x | ((uint64_t)(uint32_t)av.y << 32);
    const uint32_t qh2 = b.qhash2 ? ld32(b.qhash2, idx) : 0u;
    EiOut eo; bool over = false;
    exon_metrics_ei<FAST_BLOCKS, WaveSink>(a, p, ci, fl, B.bs, B.len, hq, eo, over, cnt, fast, cw.nblocks);
    k1e_overflow(on && cw.nblocks >= 1 && (over || !fast), (uint64_t)idx, stage, stage_n);
    k1e_commit<FAST_BLOCKS, LANE_CHUNK>(cov_diff, T, eo, B.len, fl, qhash, qh2, held, chunk);
    if (k1e_lane_below<RSQC_N_COUNTERS>() && cnt.vec) atomicAdd(&T.cnt32[l], cnt.vec);
}

// the record range of workgroup `block` of `grid` (the same split for the kernel and for what reads its per-workgroup regions)
__device__ __forceinline__ void k1e_wg_range(uint32_t n_rec, uint32_t grid, uint32_t block, uint32_t &beg, uint32_t &end) {
    const uint32_t total_waves = grid * K1E_WAVES;
    const uint32_t per_wave = (((n_rec + total_waves - 1u) / total_waves) + 63u) & ~63u;
    const uint64_t b64 = (uint64_t)block * K1E_WAVES * per_wave, e64 = b64 + (uint64_t)K1E_WAVES * per_wave;
    beg = b64 < (uint64_t)n_rec ? (uint32_t)b64 : n_rec;
    end = e64 < (uint64_t)n_rec ? (uint32_t)e64 : n_rec;
}
// packs the workgroups' candidate regions (FragCandidates::chunk_count) into the dense list the pairing stage reads
__global__ void __launch_bounds__(256)
frag_compact_kernel(FragCandidates src, FragCandidates dst, uint32_t n_rec, uint32_t k1_grid) {
    __shared__ uint32_t s_part[4], s_off;
    uint32_t before = 0;
    for (uint32_t k = threadIdx.x; k < blockIdx.x; k += blockDim.x) before += src.chunk_count[k];
    before = wave_sum(before);
    if (lane_id() == 0) s_part[threadIdx.x >> 6] = before;
    __syncthreads();
    if (threadIdx.x == 0) s_off = s_part[0] + s_part[1] + s_part[2] + s_part[3];
    __syncthreads();
    const uint32_t off = s_off, mine = src.chunk_count[blockIdx.x];
    uint32_t beg, end;
    k1e_wg_range(n_rec, k1_grid, blockIdx.x, beg, end);
    for (uint32_t j = threadIdx.x; j < mine; j += blockDim.x) {
        dst.file_index[off + j] = src.file_index[beg + j]; dst.qhash[off + j] = src.qhash[beg + j];
        dst.name[off + j] = src.name[beg + j]; dst.endpos[off + j] = src.endpos[beg + j]; dst.flag_size[off + j] = src.flag_size[beg + j];
        dst.h2[off + j] = src.h2[beg + j];
    }
    if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) *dst.count = off + mine;
}

#ifndef K1E_MINW
#define K1E_MINW 5            /* waves per SIMD the register allocation aims at: 96 VGPRs (rounds 3-5: 4 -- the long-CIGAR stage inside the tile loop needed 126) */
#endif
// BED: the run has a BED (--bed: fragment-size candidates).  A template parameter, so that the instance of runs without one carries
// none of its state (the candidate cursor alone took the kernel from 123 to 128 VGPRs and into scratch).
#ifndef K1E_MINW_BED
#define K1E_MINW_BED 4           /* the --bed instance: 113 VGPRs, no scratch; held to 96 it spills 60 bytes and runs 3.59 instead of 3.22 ms (call r6d) */
#endif
template <bool BED>
__global__ void __launch_bounds__(RSQC_K1_THREADS, BED ? K1E_MINW_BED : K1E_MINW)
classify_ei_kernel(K1Args A) {
    __shared__ K1eShared S;
    const DevAnnotation &a = A.a; const DevBatch &b = A.b; const DevAccum &acc = A.acc;
    // The run parameters and the CIGAR pool's address are used by every tile.  Left in the kernel-argument segment the compiler
    // re-loads them where it needs them (s_load + s_waitcnt lgkmcnt(0): five such stalls per tile in round 3's listing, and
    // lgkmcnt(0) also waits for every LDS operation in flight); as opaque scalars they live in SGPRs for the whole kernel.
    DevParams p = A.p;
    // (K1E_LAZYPAIR = 0: the pair destination held across the tile loop, the tree's form)
    const K1ePairDst held = {(uint64_t)(uintptr_t)acc.pairs, acc.pair_chunk_cap};
    K1E_PIN(p.mapq_threshold); K1E_PIN(p.base_mismatch); K1E_PIN(p.chimeric_distance); K1E_PIN(p.stranded); K1E_PIN(p.unpaired);
    K1E_PIN(p.exclude_chimeric); K1E_PIN(p.n_filter_tags);
    const uint32_t *cigar_pool = b.cigar; K1E_PIN(cigar_pool);
    uint32_t *tile_span = acc.tile_span; K1E_PIN(tile_span);
    const int l = lane_id();
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    S.T.init(0u);
#ifdef RSQC_K1_PROF
    if (threadIdx.x < 48) s_prof_acc[threadIdx.x] = 0ull;
    if (l == 0) s_prof_last[wave] = __builtin_amdgcn_s_memtime();
#endif
    if (blockIdx.x == 0 && threadIdx.x == 0) *k1e_lazy_args()->acc.pair_slow_count = 0u;     // written only by the slow kernel, which runs after this one
    __syncthreads();

    // the seven sum-type counters stay per-lane sums, reduced every 31 tiles
    uint32_t sum_e1mm = 0, sum_e1b = 0, sum_e2mm = 0, sum_e2b = 0, sum_mm = 0, sum_b = 0, sum_blk = 0;
    int pending = 0;
    uint32_t l_span = 0u, l_lmin = 0xFFFFFFFFu, l_lmax = 0u;
    auto flush_counts = [&]() {
        const uint32_t s0 = wave_sum_u32_full(sum_e1mm), s1 = wave_sum_u32_full(sum_e1b), s2 = wave_sum_u32_full(sum_e2mm), s3 = wave_sum_u32_full(sum_e2b),
                       s4 = wave_sum_u32_full(sum_mm), s5 = wave_sum_u32_full(sum_b), s6 = wave_sum_u32_full(sum_blk);
        if (k1e_first_lane() && s0) atomicAdd(&S.T.cnt[K1eTables::sum_slot(RSQC_C_END1_MISMATCHES)], (unsigned long long)s0);
        if (k1e_first_lane() && s1) atomicAdd(&S.T.cnt[K1eTables::sum_slot(RSQC_C_END1_BASES)], (unsigned long long)s1);
        if (k1e_first_lane() && s2) atomicAdd(&S.T.cnt[K1eTables::sum_slot(RSQC_C_END2_MISMATCHES)], (unsigned long long)s2);
        if (k1e_first_lane() && s3) atomicAdd(&S.T.cnt[K1eTables::sum_slot(RSQC_C_END2_BASES)], (unsigned long long)s3);
        if (k1e_first_lane() && s4) atomicAdd(&S.T.cnt[K1eTables::sum_slot(RSQC_C_MISMATCHED_BASES)], (unsigned long long)s4);
        if (k1e_first_lane() && s5) atomicAdd(&S.T.cnt[K1eTables::sum_slot(RSQC_C_TOTAL_BASES)], (unsigned long long)s5);
        if (k1e_first_lane() && s6) atomicAdd(&S.T.cnt[K1eTables::sum_slot(RSQC_C_ALIGNMENT_BLOCKS)], (unsigned long long)s6);
        sum_e1mm = sum_e1b = sum_e2mm = sum_e2b = sum_mm = sum_b = sum_blk = 0;
        pending = 0;
    };

    // A WORKGROUP owns a contiguous range of records (the same one as before: its pair chunk is sized for it); its waves take
    // the range in PIECES of K1E_PIECE records from an LDS counter.  With one fixed quarter per wave the four waves finished
    // up to 100 us apart (dense and sparse stretches cost differently) and waited for each other at the final barrier:
    // 10.8 % of the kernel (profiles/r3_k1_sections_v2.txt).  A piece is still a contiguous, coordinate-sorted run.
    // record indices are 32-bit inside the kernel (a batch holds fewer than 2^31 records, rsqc_api.cpp: run_batch): 64-bit
    // indices cost a register pair, a v_cmp_*_u64 and a v_lshl_add_u64 wherever a lane touches one
    const uint32_t n_rec = (uint32_t)b.n;
    uint32_t wg_beg, wg_end;
    k1e_wg_range(n_rec, gridDim.x, blockIdx.x, wg_beg, wg_end);
    constexpr uint32_t NONE = 0xFFFFFFFFu;
    auto take_piece = [&]() -> uint32_t {                 // first record of the next unclaimed piece, NONE when the range is used up
        uint32_t c = 0;
        if (k1e_first_lane()) c = atomicAdd(&S.T.piece, 1u);
        c = lane_value(c, 0);
        const uint64_t at = (uint64_t)wg_beg + (uint64_t)c * K1E_PIECE;
        return at < (uint64_t)wg_end ? (uint32_t)at : NONE;
    };
    auto tile_after = [&](uint32_t t) -> uint32_t {       // the tile this wave works on after tile t
        if (t == NONE) return NONE;
        const uint32_t nx = t + 64u;
        if (nx < wg_end && ((nx - wg_beg) % (uint32_t)K1E_PIECE) != 0u) return nx;
        return take_piece();
    };
    uint32_t w0 = wg_beg < wg_end ? take_piece() : NONE, w1 = tile_after(w0), w2 = tile_after(w1);
    const uint32_t wbeg = w0 == NONE ? n_rec : w0, wend = wg_end;
    uint32_t seg = wbeg < n_rec ? find_segment(b, wbeg) : 0u;
    int32_t u_tid = -1;
    ContigInfo u_ci = {0, 0, 0, 0, 0, 0, 0, 0};
    uint32_t seg_next = NONE;                             // first record of the next segment: kept in scalar registers, so that the
                                                          // per-tile boundary tests are compares (a scalar load per tile also waits,
                                                          // through the shared lgkmcnt, for every LDS operation in flight)
    auto load_contig = [&]() {
        u_tid = b.n_seg ? b.seg_tid[seg] : -1;
        if (u_tid >= 0 && u_tid < a.n_contigs) u_ci = a.contig[u_tid];
        else u_ci = ContigInfo{0, 0, 0, 0, 0, 0, 0, 0};
        seg_next = seg + 1 < b.n_seg ? (uint32_t)b.seg_start[seg + 1] : NONE;
        if (k1e_first_lane()) { S.T.ucache[wave][8] = 1u; S.T.ucache[wave][9] = 0u; }        // (the cached interval belongs to the contig the wave leaves)
        k1e_wave_lds_visible();
    };
    load_contig();
    // A batch of several file ranges (rsqc_batch.seg_file_index: the contigs of one shard) keeps the Read-Length inputs per SEGMENT
    // (DevAccum::rl_seg): the wave hands its lane extremes over whenever it leaves a segment, and at its end.  (Rare path; a batch
    // without ranges has rl_seg == null and keeps the batch-level extremes below.)
    auto flush_rl_seg = [&](uint32_t sg) {
        uint32_t *const rs = k1e_lazy_args()->acc.rl_seg;
        if (!rs) return;
        const uint32_t ws = wave_max_u32(l_span), wmn = wave_min_u32(l_lmin), wmx = wave_max_u32(l_lmax);
        if (k1e_first_lane() && wmn != 0xFFFFFFFFu) { atomicMax(&rs[3u * sg], ws); atomicMin(&rs[3u * sg + 1u], wmn); atomicMax(&rs[3u * sg + 2u], wmx); }
        l_span = 0u; l_lmin = 0xFFFFFFFFu; l_lmax = 0u;
    };
    uint32_t h1 = 0, c1 = 0, h2 = 0, c2 = 0, h3 = 0, c3 = 0;   // queue heads and fills (wave-uniform)

    // Record words and eight CIGAR words per record are staged ONE TILE AHEAD; the CIGAR address of the tile after that
    // comes with them (it is the fourth word of the core record).  The staged loads are issued at the TOP of a tile and
    // waited for at the END of its phase A, right before the feature stages: on gfx9 loads, stores and atomics retire through
    // ONE in-order counter (vmcnt), so a wave that waits for a load also waits for every memory operation it issued before
    // it.  Placed there, that wait finds the coverage atomics of the previous tile's feature stage a whole phase A old, and
    // the feature stage's own atomics are never waited for by the record stream (round 2: 69 of the loop's 96 waits were
    // vmcnt(0) with atomics in flight -- the exposed latency of its T = a + b / waves).
    const int4 zero4 = {0, 0, 0, 0};
    int4 cur_cv = zero4, cur_av = zero4;
    uint32_t cg[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    uint32_t nx_co = 0;                                                   // CIGAR offset of this lane's record of the NEXT tile
    uint32_t rk_pf = 0, rk_pf2 = 0; (void)rk_pf; (void)rk_pf2;                                       // rank-table words prefetched for the NEXT tile's records (K1E_RANK_PREFETCH; never read)
    uint32_t cur_cz = 0; (void)cur_cz;                                                  // coarse-table word of this lane's record (DevAnnotation::ei_coarse): 0, or 1 + the
                                                                          // interval that covers the 1024 breakpoint-free positions around its start
    const int4 *const core4 = reinterpret_cast<const int4 *>(b.core), *const aux4 = reinterpret_cast<const int4 *>(b.aux);
    const uint32_t *const core1 = reinterpret_cast<const uint32_t *>(b.core);
    // Staged loads are UNCONDITIONAL: a lane past the end of the range (or a wave without a next tile) reads the batch's last
    // record instead and is switched off by `valid` -- zeroing nine registers and branching around every load cost more than
    // the ignored load (the kernel is not launched on an empty batch).
    const uint32_t last_rec = n_rec - 1u;
    // (addresses stay scalar base + 32-bit lane offset: the tile's first record, then min(lane, records left - 1))
    auto tile_base = [&](uint32_t w) -> uint32_t { return w == NONE ? 0u : w; };
    auto lane_off = [&](uint32_t wb) -> uint32_t { const uint32_t left = last_rec - wb; return (uint32_t)l < left ? (uint32_t)l : left; };
    { const uint32_t wb = tile_base(w0), lo = lane_off(wb); cur_cv = k1e_ld32_stream(core4 + wb, lo); cur_av = ld32(aux4 + wb, lo); }
    { const uint32_t wb = tile_base(w1); nx_co = ld32(core1 + 4 * (size_t)wb, 4u * lane_off(wb) + 3u); }
    k1e_load_cigar8(cigar_pool, (uint32_t)cur_cv.w, cg);
    // (the wait of the first tile's words sits here, not in the loop: the compiler places a wait where ANY path into an
    //  instruction has the load pending, and a wait inside the loop is executed by every tile)
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("" :: "v"(cur_cv.x), "v"(cur_cv.y), "v"(cur_cv.z), "v"(cur_cv.w), "v"(cur_av.x), "v"(cur_av.y), "v"(cur_av.z), "v"(cur_av.w), "v"(nx_co));
    asm volatile("" :: "v"(cg[0]), "v"(cg[1]), "v"(cg[2]), "v"(cg[3]), "v"(cg[4]), "v"(cg[5]), "v"(cg[6]), "v"(cg[7]));
#endif
    for (; w0 != NONE; w0 = w1, w1 = w2, w2 = tile_after(w2)) {
        RSQC_MARK(0);
        const uint32_t i = w0 + (uint32_t)l;
        const bool valid = i < wend;
        if (seg_next <= w0) {                               // (the queues were emptied by the tile before the boundary)
            flush_rl_seg(seg);
            while (seg + 1 < b.n_seg && b.seg_start[seg + 1] <= w0) ++seg;
            load_contig();
        }
        const bool mixed = seg_next != NONE && seg_next - w0 < 64u;           // a contig boundary inside the tile
        // ---- the next tile's words start their trip now -----------------------------------------------------------------
        int4 n_cv, n_av; uint32_t n_cg[8]; uint32_t n_co;
        { const uint32_t wb = tile_base(w1), lo = lane_off(wb); n_cv = k1e_ld32_stream(core4 + wb, lo); n_av = ld32(aux4 + wb, lo); }
        k1e_load_cigar8(cigar_pool, nx_co, n_cg);
        { const uint32_t wb = tile_base(w2); n_co = ld32(core1 + 4 * (size_t)wb, 4u * lane_off(wb) + 3u); }
        WaveSink cnt;
        // ---- phase A: record words, CIGAR, gate cascade ----------------------------------------------------------
        K1E_AMARK(1);
        // The run's switches (--unpaired, chimeric exclusion, configured tag filters) are the same for every tile: the conditions the
        // cascade derives from them are 64-bit lane masks that the compiler computes ONCE, keeps in scalar register pairs across
        // the tile loop, spills to VGPR lanes under the cascade's pressure and reloads with two v_readlane each, every tile.
        // Opaque per tile, they are re-derived with two scalar instructions where they are used and occupy nothing in between.
        DevParams pt = p;
#if defined(__HIP_DEVICE_COMPILE__) && !defined(K1E_NO_OPAQUE_SWITCHES)
        asm volatile("" : "+s"(pt.unpaired), "+s"(pt.exclude_chimeric), "+s"(pt.n_filter_tags));
#endif
        do {                                                  // (K1E_STOP leaves through `break`)
        Record r; uint32_t cur_cigar_off; (void)cur_cigar_off;
        {
            const int4 cv = cur_cv, av = cur_av;                                  // (zero for lanes past the range)
            r.pos = cv.x; r.mpos = cv.y; r.isize = cv.z; cur_cigar_off = (uint32_t)cv.w;
            r.cigar = cigar_pool + (uint32_t)cv.w;
            r.qhash = (uint64_t)(uint32_t)av.x | ((uint64_t)(uint32_t)av.y << 32);
            r.flag = (uint32_t)av.z & 0xFFFFu; r.l_qseq = (int32_t)((uint32_t)av.z >> 16);
            r.mapq = (uint32_t)av.w & 0xFFu; r.nm = (int32_t)(((uint32_t)av.w >> 8) & 0xFFu);
            r.tagbits = ((uint32_t)av.w >> 16) & 0xFFu; r.n_cigar = (uint32_t)av.w >> 24;
        }
        typedef WaveSink WS; typedef WaveSink::B WB;
        uint32_t bad_wide = 0;                                // (an integer, so that its lane mask below is one compare)
        if (valid && (r.l_qseq == RSQC_LQSEQ_ESCAPE || r.nm == RSQC_NM_ESCAPE || r.n_cigar == RSQC_NCIGAR_ESCAPE)) {
            const DevBatch &bw = k1e_lazy_args()->b;
            uint32_t lo = 0, hi = bw.n_wide;                    // wide table is sorted by record index
            while (lo < hi) { uint32_t m = (lo + hi) >> 1; if (bw.wide_index[m] < i) lo = m + 1; else hi = m; }
            if (lo >= bw.n_wide || bw.wide_index[lo] != i) bad_wide = 1u;
            else { r.l_qseq = bw.wide_l_qseq[lo]; r.nm = bw.wide_nm[lo]; r.n_cigar = bw.wide_n_cigar[lo]; }
            K1E_LANDED(r.l_qseq); K1E_LANDED(r.nm); K1E_LANDED(r.n_cigar);
        }
        r.tid = u_tid;
        uint32_t my_seg = seg;                                // (the lane's own segment: differs from the wave's only in a boundary tile)
        if (mixed && valid) { uint32_t s2 = seg; while (s2 + 1 < b.n_seg && b.seg_start[s2 + 1] <= i) ++s2; r.tid = b.seg_tid[s2]; my_seg = s2; K1E_LANDED(r.tid); }
        if (bad_wide) atomicExch(k1e_lazy_args()->acc.error, RSQC_ERR_ARG);
        const WB lane_on = WS::prim(valid) && !WS::prim(bad_wide != 0u);
        if (!WS::lane(lane_on)) r.n_cigar = 0;
        K1E_AMARK(2);
        K1E_STOP(2, (r.pos, r.mpos, r.isize, r.flag, r.l_qseq, r.mapq, r.nm, r.tagbits, r.n_cigar, cur_cigar_off), (lane_on.m))
        Walk3 w2;                                             // all eight staged operations, up to three blocks captured (last block first)
        k1e_walk3(r.pos, r.n_cigar, cg, r.cigar, w2);
        K1E_AMARK(3);
        K1E_STOP(3, (r.mpos, r.isize, r.flag, r.l_qseq, r.mapq, r.nm, r.tagbits, r.n_cigar, cur_cigar_off, w2.ref_len, w2.nb, w2.bad, w2.b0, w2.b1, w2.b2, w2.l0, w2.l1, w2.l2), (lane_on.m))
        const bool shortc = r.n_cigar <= 8;                    // blocks and legality are known here; longer CIGARs: classify_long_kernel
        CigarWalk cw;
        cw.ref_len = w2.ref_len; cw.nblocks = shortc ? w2.nb : 0u; cw.aligned = 0; cw.bad = shortc && w2.bad != 0u;
        RecordCounters rc; WB hq = lane_on;
        const WB go = gate_cascade_b<false, WaveSink, true>(a, pt, r, cw, rc, hq, cnt, lane_on);    // (a lane without a record leaves with every output 0)
        K1E_AMARK(4);
        K1E_STOP(4, (r.flag, r.n_cigar, cur_cigar_off, w2.nb, w2.b0, w2.b1, w2.b2, w2.l0, w2.l1, w2.l2, rc.e1_mm, rc.e1_bases, rc.e2_mm, rc.e2_bases, rc.mm, rc.bases, rc.blocks, rc.rl_span, rc.rl_lqseq, rc.rl_eligible, rc.error, rc.frag_candidate, cnt.vec), (go.m, hq.m))
        // --bed (src/RNASeQC.cpp:372): the block tests of fragmentSizeMetrics walk the CIGAR again and chase the BED rows per lane --
        // divergent code with dependent loads, and as soon as ONE lane of the tile takes it the wave does (round 4: 4.2 instead of
        // 2.6 ms).  Most tiles lie nowhere near a BED interval: a wave-level test first -- does ANY interval of the contig overlap
        // the span [first candidate start, last candidate end + 1] of this tile?  The answer comes from a cursor into the contig's
        // start-sorted rows that the wave moves along with the sorted stream (scalar registers; a binary search only when the tile's
        // end passes the next interval's start): in steady state two wave reductions and no memory access.
        bool bed_near = false;
        if (BED) {
            const bool cand = rc.frag_candidate != 0u;
            const uint32_t t_lo = wave_min_u32_full(cand ? (uint32_t)(r.pos + 1) : 0x7FFFFFFFu), t_hi = wave_max_u32_full(cand ? (uint32_t)rc.endpos + 1u : 0u);
            if (t_hi != 0u) {                                 // (some lane holds a candidate)
                if (mixed) bed_near = true;                   // (a contig boundary inside the tile: no shortcut)
                else {
                    // the cursor lives in LDS (K1eTables::bedc: this wave's four words), not in registers: the kernel has no scalar
                    // registers left (held across the tile loop the seven words of rounds 4-5 were spilled to VGPRs and from there to
                    // scratch: 32 bytes per lane, written and read in every tile)
                    int32_t *const bc = S.T.bedc[wave];
                    int32_t bed_cur = (int32_t)__builtin_amdgcn_readfirstlane(bc[1]), bed_nxt = (int32_t)__builtin_amdgcn_readfirstlane(bc[2]), bed_pm = (int32_t)__builtin_amdgcn_readfirstlane(bc[3]);
                    const bool fresh = (uint32_t)__builtin_amdgcn_readfirstlane(bc[0]) != seg;        // first candidate tile on this contig
                    __builtin_amdgcn_wave_barrier();             // every lane has read the cursor before the first lane may rewrite it (no instruction:
                                                                 // an ordering point for the compiler -- and where the host emulation's lanes, which run one after the other, line up)
                    if (fresh || (int32_t)t_hi >= bed_nxt || (int32_t)t_hi < bed_cur) {   // ... or the tile's end passed the next row's start (or the stream went backwards)
                        const K1Args *q = k1e_lazy_args();
                        uint32_t bed_lo = 0u, bed_hi = 0u;
                        if (u_tid >= 0 && u_tid < q->a.n_contigs) { bed_lo = q->a.bed_range[u_tid]; bed_hi = q->a.bed_range[u_tid + 1]; }
                        uint32_t lo2 = bed_lo, hi2 = bed_hi;                          // first row with start > t_hi
                        while (lo2 < hi2) { const uint32_t m = lo2 + ((hi2 - lo2) >> 1); if (q->a.bed_start[m] <= (int32_t)t_hi) lo2 = m + 1; else hi2 = m; }
                        // (a contig without rows: nothing before, nothing behind -- never near, never searched again)
                        bed_cur = lo2 > bed_lo ? q->a.bed_start[lo2 - 1] : (int32_t)0x80000000;
                        bed_pm = lo2 > bed_lo ? q->a.bed_pmax[lo2 - 1] : (int32_t)0x80000000;
                        bed_nxt = lo2 < bed_hi ? q->a.bed_start[lo2] : (int32_t)0x7FFFFFFF;
                        if (k1e_first_lane()) { bc[0] = (int32_t)seg; bc[1] = bed_cur; bc[2] = bed_nxt; bc[3] = bed_pm; }
                        k1e_wave_lds_visible();
                    }
                    // a row starts at or before the span's end and one of those reaches its start (bed_pm = INT_MIN: no such row).  "- 1": a
                    // zero-length first block sits one position before the record's first base and bed_interval_of tests it too
                    bed_near = bed_pm >= (int32_t)t_lo - 1;
                }
            }
        }
        // (Call r6f: the candidates of such tiles through a fourth per-wave ring and a 64-lane stage of their own -- record words and CIGAR
        //  re-gathered there -- measured SLOWER than these few divergent lanes: K1 3.31 instead of 3.17-3.22 ms, step 6.76 instead of 6.61.)
        if (bed_near && rc.frag_candidate) {                  // src/RNASeQC.cpp:372
            const K1Args *q = k1e_lazy_args();
            const int32_t name = bed_interval_of(q->a, r);
            if (name >= 0) {
                const FragCandidates &fr = q->acc.frag;
                const uint32_t slot = wg_beg + atomicAdd(&S.T.frags, 1u);       // the workgroup's own region (<= one candidate per record)
                {
                    // (name hash, mate position and insert size come back from the record arrays -- cache lines this wave streamed a
                    //  moment ago -- instead of living in registers from the top of phase A to this rare branch)
                    const int4 co = reinterpret_cast<const int4 *>(q->b.core)[i];
                    const uint2 qh = reinterpret_cast<const uint2 *>(q->b.aux)[2u * i];
                    fr.file_index[slot] = batch_file_index(q->b, my_seg, i); fr.qhash[slot] = (uint64_t)qh.x | ((uint64_t)qh.y << 32);
                    fr.h2[slot] = q->b.qhash2 ? q->b.qhash2[i] : 0u;                 // (the name is 96 bits on every path that keys on it)
                    fr.name[slot] = name; fr.endpos[slot] = rc.endpos;
                    const bool fok = !(r.flag & RSQC_FMREVERSE) && (r.flag & RSQC_FREVERSE) && r.pos != co.y;
                    const uint32_t sz = (uint32_t)(co.z < 0 ? -(int64_t)co.z : (int64_t)co.z);
                    fr.flag_size[slot] = (sz & 0x7FFFFFFFu) | (fok ? 0x80000000u : 0u);
                }
            }
        }
        if (rc.error) atomicExch(k1e_lazy_args()->acc.error, rc.error);
        sum_e1mm += rc.e1_mm; sum_e1b += rc.e1_bases; sum_e2mm += rc.e2_mm; sum_e2b += rc.e2_bases;
        sum_mm += rc.mm; sum_b += rc.bases; sum_blk += rc.blocks;
        const WB big_any = WS::prim((rc.bases | rc.mm | rc.blocks) >= (1u << 26));
        {   // Read-Length inputs: per-tile max span + batch-level extremes
            uint32_t sp = rc.rl_span;                        // (0 unless the record reaches src/RNASeQC.cpp:275)
            const uint32_t wsp = wave_max_u32_full(sp);
            if (k1e_first_lane()) tile_span[w0 >> 6] = wsp;
            uint32_t lq = (uint32_t)rc.rl_lqseq; bool elig = rc.rl_eligible != 0u;
            if (mixed) {                                      // a batch of several file ranges keeps these per SEGMENT: the records behind
                uint32_t *const rs = k1e_lazy_args()->acc.rl_seg;                 // the boundary go to their own segment's slots
                if (rs && valid && my_seg != seg) {
                    if (elig) { atomicMax(&rs[3u * my_seg], sp); atomicMin(&rs[3u * my_seg + 1u], lq); atomicMax(&rs[3u * my_seg + 2u], lq); }
                    sp = 0u; lq = 0u; elig = false;
                }
            }
            l_span = sp > l_span ? sp : l_span;
            l_lmin = (elig && lq < l_lmin) ? lq : l_lmin; l_lmax = lq > l_lmax ? lq : l_lmax;    // (rl_lqseq is 0 for the others)
        }
        K1E_AMARK(5);
        K1E_STOP(5, (r.flag, r.n_cigar, cur_cigar_off, w2.nb, w2.b0, w2.b1, w2.b2, w2.l0, w2.l1, w2.l2, cnt.vec), (go.m, hq.m, big_any.m))
        const uint32_t flhq = r.flag | (WS::lane(hq) ? K1E_HQ : 0u);
        // ---- sort by shape ------------------------------------------------------------------------------------------
        // (stragglers of a boundary tile take the general code, which finds their contig itself; for one with a long CIGAR it
        //  also counts the blocks and checks the operations, K1E_OVF_LONG)
        const WB walked = WS::prim(r.n_cigar <= 8u);
        const WB nb0 = WS::prim(w2.nb == 0u), nb1 = WS::prim(w2.nb == 1u), nb2 = WS::prim(w2.nb == 2u), nb3 = WS::prim(w2.nb == 3u);
        // one-block records whose length fits the queue's 16 bits (a longer block -- never seen in RNA-seq -- is classify_long_kernel's)
        const WB fits = WS::prim(w2.l0 < 65536u), fits3 = WS::prim((w2.l0 | w2.l1 | w2.l2) < 65536u);
        const WB shape = walked && (nb0 || (nb1 && fits) || nb2 || (nb3 && fits3));
        WB mine = go;                                        // stragglers of a boundary tile: general code
        if (mixed) {
            mine = go && WS::prim(r.tid == u_tid);
            k1e_overflow(WS::lane(go && !mine), WS::lane(walked) ? (uint64_t)i : ((uint64_t)i | K1E_OVF_LONG));
        }
        const WB simple = mine && shape, listed = mine && !shape;
        {   // no block at all (clips / insertions only): intergenic, src/Expression.cpp:407-441 with no feature seen
            const WB none = simple && nb0;
            RSQC_COUNT(cnt, RSQC_C_INTERGENIC_READS, none); RSQC_COUNT(cnt, RSQC_C_HQ_INTERGENIC_READS, none && hq);
        }
        K1E_AMARK(6);
        // ---- the staged words have landed (see above).  In the -DK1E_COARSE build the wait sits HERE, in front of the queue writes:
        //      the coarse word of this tile's records (loaded a tile ago, i.e. older than the staged words) is then complete without
        //      a wait of its own -- a wait anywhere earlier in phase A would also wait for the previous feature stage's atomics.
        //      The default build waits behind the queue writes and the counters (below): the LDS traffic overlaps the landing ------
#if defined(K1E_COARSE)                               /* (only that build consumes a loaded word in the queue writes) */
#if defined(__HIP_DEVICE_COMPILE__)
        asm volatile("; K1E staged words landed" :: "v"(n_cv.x), "v"(n_cv.y), "v"(n_cv.z), "v"(n_cv.w), "v"(n_av.x), "v"(n_av.y), "v"(n_av.z), "v"(n_av.w), "v"(n_co));
        asm volatile("" :: "v"(n_cg[0]), "v"(n_cg[1]), "v"(n_cg[2]), "v"(n_cg[3]), "v"(n_cg[4]), "v"(n_cg[5]), "v"(n_cg[6]), "v"(n_cg[7]));
#endif
#endif
        const uint64_t m1 = (simple && nb1).m, m2 = (simple && nb2).m, m3all = (simple && nb3).m;
        // the three-block ring has 64 slots: the records it has no room for join the deferred ones
        const uint32_t room3 = (uint32_t)K1E_Q3CAP - c3;
        const uint64_t m3 = m3all & WS::prim(mask_rank(m3all) < room3).m;
        k1e_defer(S.T, listed.m | (m3all & ~m3), (uint32_t)i, WS::lane(hq), wg_beg);
        if (WS::lane(LaneMask{m1})) {
            const uint32_t slot = (h1 + c1 + mask_rank(m1)) & (K1E_QCAP - 1);
            K1eQueue1 &q = S.q1[wave];
            q.bs[slot] = w2.b0; q.lf[slot] = w2.l0 | ((flhq & 0xFFFu) << 16) | ((flhq >> 16) << 28); q.idx[slot] = (uint32_t)i;
#ifdef K1E_COARSE
            // the coarse word answers the block's look-ups when the block ends inside the 1024 positions it speaks for
            const bool inside = ((w2.b0 + w2.l0) >> 9) - (((uint32_t)(r.pos + 1)) >> 9) <= 1u;
            q.cz[slot] = inside ? cur_cz : 0u;
#endif
        }
        if (WS::lane(LaneMask{m2})) {                           // (the walk holds the blocks last-first)
            const uint32_t slot = (h2 + c2 + mask_rank(m2)) & (K1E_QCAP - 1);
            K1eQueue2 &q = S.q2[wave];
            q.bs0[slot] = w2.b1; q.len0[slot] = w2.l1; q.bs1[slot] = w2.b0; q.len1[slot] = w2.l0; q.idx[slot] = (uint32_t)i; q.flhq[slot] = flhq;
        }
        if (WS::lane(LaneMask{m3})) {
            const uint32_t slot = k1e_wrap3(h3 + c3 + mask_rank(m3));
            K1eQueue3 &q = S.q3[wave];
            q.bs0[slot] = w2.b2; q.bs1[slot] = w2.b1; q.bs2[slot] = w2.b0; q.lens01[slot] = w2.l2 | (w2.l1 << 16);
            q.lf2[slot] = w2.l0 | ((flhq & 0xFFFu) << 16) | ((flhq >> 16) << 28); q.idx[slot] = (uint32_t)i;
        }
        c1 += (uint32_t)__popcll(m1); c2 += (uint32_t)__popcll(m2); c3 += (uint32_t)__popcll(m3);
        __builtin_amdgcn_wave_barrier();                     // the queue entries are read by OTHER lanes of the wave (no instruction: an ordering point)
        if (k1e_lane_below<RSQC_N_COUNTERS>() && cnt.vec) atomicAdd(&S.T.cnt32[l], cnt.vec);
        if (++pending == 31 || WS::any(big_any)) flush_counts();
        } while (0);
        K1E_AMARK(7);
#if !defined(K1E_COARSE)                              /* the landing wait behind the queue writes and the counters: 2.59 -> 2.52 ms (profiles/r4_k1_variants.txt) */
#if defined(__HIP_DEVICE_COMPILE__)
        asm volatile("; K1E staged words landed" :: "v"(n_cv.x), "v"(n_cv.y), "v"(n_cv.z), "v"(n_cv.w), "v"(n_av.x), "v"(n_av.y), "v"(n_av.z), "v"(n_av.w), "v"(n_co));
        asm volatile("" :: "v"(n_cg[0]), "v"(n_cg[1]), "v"(n_cg[2]), "v"(n_cg[3]), "v"(n_cg[4]), "v"(n_cg[5]), "v"(n_cg[6]), "v"(n_cg[7]));
        asm volatile("" :: "v"(rk_pf), "v"(rk_pf2));           // (the previous tile's prefetch: older than the staged words, so this asks for no wait of its own)
#endif
#endif
        cur_cv = n_cv; cur_av = n_av; nx_co = n_co;
        // (Measured and left off, profiles/r5_k1_variants.txt call r5c: touching the rank-table word of every record of the NEXT tile here,
        //  a tile ahead of the feature stage that reads it -- a cold HBM line per ~256 positions of a 775 MB table streamed once --
        //  with a load nobody reads made the kernel 1-6 % SLOWER: loads return in order, so every wait behind the prefetch, the next
        //  feature stage's first look-up included, now also waits for the slowest line of the wave's prefetch.)
#ifndef K1E_RANK_PREFETCH
#define K1E_RANK_PREFETCH 0
#endif
#if K1E_RANK_PREFETCH
        if (!(w1 == NONE || seg_next <= w1) && u_ci.rk_words != 0u) {
            const int32_t top = (int32_t)(u_ci.rk_words << 6) - 1;
            const int32_t x = n_cv.x + 1, xc = x < 0 ? 0 : (x > top ? top : x);
            rk_pf = ld32(reinterpret_cast<const uint32_t *>(a.ei_rank), (u_ci.rk_base + ((uint32_t)xc >> 6)) * 4u);
#if K1E_RANK_PREFETCH > 1
            const int32_t x2 = x + 192, xc2 = x2 < 0 ? 0 : (x2 > top ? top : x2);
            rk_pf2 = ld32(reinterpret_cast<const uint32_t *>(a.ei_rank), (u_ci.rk_base + ((uint32_t)xc2 >> 6)) * 4u);
#endif
        }
#endif
        // the next tile's positions are here: its coarse-table words start their trip now and are looked at when that tile sorts
        // its records by shape, a feature stage and most of a phase A later (a read in an empty stretch of the genome then needs
        // no rank word).  Not across a contig boundary: the table is addressed through THIS tile's contig.
        cur_cz = 0u;
#ifdef K1E_COARSE                                    /* opt-in build: -0.7 GB of HBM traffic per 102 M records, +3 % time (profiles/r4_k1_variants.txt) */
        if (!(w1 == NONE || seg_next <= w1) && u_ci.rk_words != 0u) {
            const int32_t top = (int32_t)(u_ci.rk_words << 6) - 1;
            const int32_t x = n_cv.x + 1, xc = x < 0 ? 0 : (x > top ? top : x);
            cur_cz = ld32(a.ei_coarse, (u_ci.rk_base >> 3) + ((uint32_t)xc >> 9));
        }
#endif
#pragma unroll
        for (int k = 0; k < 8; ++k) cg[k] = n_cg[k];
        // ---- a full tile of one shape: its feature stage.  The queues are emptied before the stream leaves the contig
        //      (the queued records belong to it) and at the end of the range ------------------------------------------------
        const bool leaving = w1 == NONE || seg_next <= w1;    // (NONE is the largest index)             // the wave's next tile lies in another segment (or there is none)
        const uint32_t thr = leaving ? 1u : 64u;
        while (c1 >= thr) {
            const uint32_t take = c1 < 64u ? c1 : 64u;
            RSQC_MARK(8);
            if (!(K1E_ABL & 1)) k1e_process<1>(a, pt, b.aux, acc.cov_diff, u_ci, S, wave, h1, take, b.qhash2, held);
            h1 = (h1 + take) & (K1E_QCAP - 1); c1 -= take;
            RSQC_MARK(9);                          // [9] one-block tiles
        }
        while (c2 >= thr) {
            const uint32_t take = c2 < 64u ? c2 : 64u;
            RSQC_MARK(8);
            if (!(K1E_ABL & 1)) k1e_process<2>(a, pt, b.aux, acc.cov_diff, u_ci, S, wave, h2, take, b.qhash2, held);
            h2 = (h2 + take) & (K1E_QCAP - 1); c2 -= take;
            RSQC_MARK(10);                         // [10] two-block tiles
        }
        while (__builtin_expect(c3 >= thr, 0)) {
            const uint32_t take = c3 < 64u ? c3 : 64u;
            RSQC_MARK(8);
            // The three-block stage is the widest of the loop (three blocks' index words and six commit slots) and the tile loop's own
            // per-lane state -- six of the seven counter sums, three Read-Length extremes -- would push it past the 96 registers of five waves
            // per SIMD (12-28 bytes of scratch in every build tried, reloaded through vmcnt in front of the stage's atomics).  That
            // state is PARKED in LDS for the duration of the call instead: the free part of the wave's own one- and two-block rings
            // (a ring holds fewer than 64 entries here, so the 64 slots behind its fill are unused), one dword per lane and value.
            {
                const uint32_t s1 = (h1 + c1 + (uint32_t)l) & (K1E_QCAP - 1), s2 = (h2 + c2 + (uint32_t)l) & (K1E_QCAP - 1);
                K1eQueue1 &p1 = S.q1[wave]; K1eQueue2 &p2 = S.q2[wave];
                p1.bs[s1] = sum_e1mm; p1.lf[s1] = sum_e1b; p1.idx[s1] = sum_e2mm;
                p2.bs0[s2] = sum_e2b; p2.len0[s2] = sum_mm; p2.bs1[s2] = sum_b; p2.len1[s2] = l_span; p2.idx[s2] = l_lmin; p2.flhq[s2] = l_lmax;
                if (!(K1E_ABL & 16)) k1e_process<3>(a, pt, b.aux, acc.cov_diff, u_ci, S, wave, h3, take, b.qhash2, held);
#if defined(__HIP_DEVICE_COMPILE__)
                asm volatile("" ::: "memory");                  // (the parked values are read back from LDS, not kept in registers across the call)
#endif
                sum_e1mm = p1.bs[s1]; sum_e1b = p1.lf[s1]; sum_e2mm = p1.idx[s1];
                sum_e2b = p2.bs0[s2]; sum_mm = p2.len0[s2]; sum_b = p2.bs1[s2]; l_span = p2.len1[s2]; l_lmin = p2.idx[s2]; l_lmax = p2.flhq[s2];
            }
            h3 = k1e_wrap3(h3 + take); c3 -= take;
            RSQC_MARK(11);                         // [11] three-block tiles
        }
    }
    flush_counts();
    flush_rl_seg(seg);
    {
        const uint32_t ws = wave_max_u32(l_span), wmn = wave_min_u32(l_lmin), wmx = wave_max_u32(l_lmax);
        if (k1e_first_lane()) { atomicMax(&S.T.rl[0], ws); atomicMin(&S.T.rl[1], wmn); atomicMax(&S.T.rl[2], wmx); }
    }
    RSQC_MARK(12);
    __syncthreads();
    S.T.flush(wg_beg < wg_end ? wg_end - wg_beg : 0ull);
#ifdef RSQC_K1_PROF
    RSQC_MARK(13);                                 // [13] workgroup epilogue (barrier + flush of the LDS tables)
    __syncthreads();
    if (threadIdx.x < 48 && s_prof_acc[threadIdx.x]) atomicAdd(&g_k1_prof[threadIdx.x], s_prof_acc[threadIdx.x]);
#endif
    if (threadIdx.x == 0) {
        const K1Args *q = k1e_lazy_args();
        atomicMax(&q->acc.rl_stats[0], S.T.rl[0]); atomicMin(&q->acc.rl_stats[1], S.T.rl[1]); atomicMax(&q->acc.rl_stats[2], S.T.rl[2]);
        q->acc.pair_chunk_count[blockIdx.x] = S.T.pairs < q->acc.pair_chunk_cap ? S.T.pairs : q->acc.pair_chunk_cap;
        if (BED) q->acc.frag.chunk_count[blockIdx.x] = S.T.frags;
    }
    // the workgroup's deferred records move from its region to the dense list classify_long_kernel reads: one memory atomic per workgroup
    const uint32_t n_def = S.T.defer;                          // (final since the barrier in front of the flush)
    if (n_def != 0u) {
        const K1Args *q = k1e_lazy_args();
        if (threadIdx.x == 0) S.T.piece = atomicAdd(q->acc.defer_total, n_def);
        __syncthreads();
        const uint32_t at = S.T.piece;
        for (uint32_t j = threadIdx.x; j < n_def; j += blockDim.x) q->acc.defer_list[at + j] = q->acc.defer_index[wg_beg + j];
    }
}

// ---- the records classify_ei_kernel deferred (more than eight operations, more than three blocks, the three-block ring's surplus:
// about 0.7 % of an RNA-seq file) -----------------------------------------------------------------------------------------------------
// One WAVE per call of 64 entries of the dense list, calls taken grid-stride.  Every workgroup of THIS kernel owns a pair chunk of its
// own behind the K1 grid's (chunk k1_grid + blockIdx.x, slots from an LDS counter: no memory atomic); only when that chunk is nearly full
// -- an input whose records are all deferred -- do a call's pairs go into the chunks of the K1 workgroups that listed the records, which
// are sized for FAST_SET pairs of EVERY record of their ranges, deferred ones included (per-lane chunk, one returning memory atomic per
// pair).  Exon / gene / counter updates go to this workgroup's LDS tables.  A call whose records span contigs runs once per contig with
// the other lanes switched off.
// (Call r6a: one workgroup per K1 workgroup's region -- 0.67 ms, the regions of multi-exon genes hold thousands of records.  Call r6b: one
//  wave per call, 0.42 ms -- the 36 k entries it adds to the general kernel's list, one returning atomic each on ONE counter.  Call r6c:
//  those staged in LDS, calls padded per K1 workgroup: 0.245 ms, 18 k calls of 45 us at 2 calls per wave.  Call r6d: dense list, the
//  three-block ring's surplus gone (10 k calls), but every pair a returning memory atomic on its K1 chunk's count: 0.61 ms.)
__global__ void __launch_bounds__(RSQC_K1_THREADS)              // (103 VGPRs: four waves per SIMD; held to 96 it spills 24 bytes)
classify_long_kernel(K1Args A, uint32_t k1_grid) {
    __shared__ K1eTables T;
    __shared__ unsigned long long s_stage[K1E_OVF_STAGE];       // records for the general kernel, moved to its list at the end (k1e_overflow)
    __shared__ uint32_t s_stage_n;
    const DevAnnotation &a = A.a; const DevBatch &b = A.b; const DevAccum &acc = A.acc;
    const K1ePairDst held = {(uint64_t)(uintptr_t)acc.pairs, acc.pair_chunk_cap};
    const int l = lane_id();
    T.init(0u);
    if (threadIdx.x == 0) s_stage_n = 0u;
    __syncthreads();
    uint32_t sum_blk = 0;
    const uint32_t n_rec = (uint32_t)b.n;
    const uint32_t total = *acc.defer_total;
    const uint32_t n_waves = gridDim.x * (uint32_t)K1E_WAVES;
    const uint32_t total_waves = k1_grid * (uint32_t)K1E_WAVES;
    const uint32_t per_wg = (uint32_t)K1E_WAVES * ((((n_rec + total_waves - 1u) / total_waves) + 63u) & ~63u);      // records per K1 workgroup (as in k1e_wg_range)
    for (uint32_t call = blockIdx.x * (uint32_t)K1E_WAVES + (threadIdx.x >> 6); call * 64u < total; call += n_waves) {
        const uint32_t at = call * 64u + (uint32_t)l;
        const bool on = at < total;
        const uint32_t e = acc.defer_list[on ? at : call * 64u];
        const uint32_t idx = e & 0x7FFFFFFFu; const bool hq = on && (e >> 31) != 0u;
        const uint32_t chunk = idx / per_wg;                    // the K1 workgroup that listed the record
        // the contigs the call spans (its entries are not sorted: several waves of a K1 workgroup list into one region)
        const uint32_t seg_lo = find_segment(b, wave_min_u32(idx)), seg_hi = find_segment(b, wave_max_u32(idx));
        for (uint32_t sg = seg_lo; sg <= seg_hi; ++sg) {
            const bool mine = on && (seg_lo == seg_hi || (idx >= (uint32_t)b.seg_start[sg] && (sg + 1 >= b.n_seg || idx < (uint32_t)b.seg_start[sg + 1])));
            if (__ballot(mine) == 0ull) continue;
            const int32_t tid = b.n_seg ? b.seg_tid[sg] : -1;
            const ContigInfo ci = (tid >= 0 && tid < a.n_contigs) ? a.contig[tid] : ContigInfo{0, 0, 0, 0, 0, 0, 0, 0};
            // (room for this call's pairs in the workgroup's own chunk even if the three other waves are emitting theirs: 4 x 64 x FAST_SET)
            const bool own = (uint32_t)__builtin_amdgcn_readfirstlane((int)T.pairs) + 4u * 64u * (uint32_t)FAST_SET <= acc.pair_chunk_cap;
            if (own) k1e_long_call<false>(a, A.p, b, acc.cov_diff, ci, T, idx, hq, mine, sum_blk, held, k1_grid + blockIdx.x, s_stage, &s_stage_n);
            else k1e_long_call<true>(a, A.p, b, acc.cov_diff, ci, T, idx, hq, mine, sum_blk, held, chunk, s_stage, &s_stage_n);
        }
    }
    {
        const uint32_t s6 = wave_sum(sum_blk);
        if (k1e_first_lane() && s6) atomicAdd(&T.cnt[K1eTables::sum_slot(RSQC_C_ALIGNMENT_BLOCKS)], (unsigned long long)s6);
    }
    __syncthreads();
    T.flush(0ull);
    if (threadIdx.x == 0) acc.pair_chunk_count[k1_grid + blockIdx.x] = T.pairs < acc.pair_chunk_cap ? T.pairs : acc.pair_chunk_cap;
    const uint32_t n_st = s_stage_n;                            // (final since the barrier)
    if (n_st != 0u) {
        if (threadIdx.x == 0) T.piece = atomicAdd(acc.ovf_count, n_st);
        __syncthreads();
        const uint32_t at = T.piece;
        for (uint32_t j = threadIdx.x; j < n_st; j += blockDim.x) {
            if (at + j < acc.ovf_cap) acc.ovf_index[at + j] = s_stage[j];
            else atomicExch(acc.error, RSQC_ERR_CAPACITY);
        }
    }
}

}  // namespace rsqc
