// rsqc_k4.h -- K4, geneFragmentCounts (src/Expression.cpp:383-387): the number of distinct read names among the records counted
// to a gene.  Device code only (kernels + their helpers), included by rsqc_kernels.hip -- and, unmodified, by the host SIMT
// emulation of the tests (tests/hostemu/k4_emu.cpp), which runs these kernels on the CPU against a std::set.
#pragma once

namespace rsqc {

// An open-addressing table per gene (round 1's first form) is bound by the chip's rate of random memory-side CAS
// operations (one per distinct fragment).  Streaming passes over PARTITIONS instead: a gene with n counted records owns
// ceil(n / RSQC_K4_PART_READS) partitions (chosen by the high word of the name hash), each with a key list of fixed
// capacity laid out by frag_layout_kernel from the final geneCounts:
//   frag_local_kernel   per pair chunk: pairs already seen in the chunk are dropped (LDS window), the others' 96-bit name
//                       identities (FragKey: rsqc_rec_aux::qhash + rsqc_batch.qhash2) are APPENDED to their partition's list (space for one pass's keys of a partition is reserved with
//                       ONE memory atomicAdd; positions inside the reservation come from an LDS counter)
//   frag_count_kernel   one workgroup per partition: its keys go through an LDS hash set (64-bit words claimed by CAS, the
//                       second hash beside them, equal words with different second hashes set aside and counted exactly);
//                       the number of distinct keys is added to geneFragmentCounts
// A partition expects <= RSQC_K4_PART_READS keys and has room for RSQC_K4_SUB_CAP; one that overflows (never with
// rsqc_qname_hash values) reports RSQC_ERR_CAPACITY instead of miscounting.

// the partition of a key inside its gene comes from the key's HIGH word (the keys are rsqc_qname_hash values: FNV-1a + fmix64;
// one multiply-xorshift on top keeps a caller's weaker hash from piling up), the slot inside the partition's set from the low word
__device__ __forceinline__ uint32_t frag_part_hash(uint64_t key) {
    uint32_t h = (uint32_t)(key >> 32);
    h ^= h >> 15; h *= 0x2C1B3C6Du; h ^= h >> 12;
    return h;
}
__device__ __forceinline__ uint32_t frag_parts_of(unsigned long long reads) {
    return (uint32_t)((reads + RSQC_K4_PART_READS - 1) / RSQC_K4_PART_READS);
}
// per gene: part_first[g] = its first partition (part_first[n_genes] = partition count) and the offset of its key lists (a
// multiple of 16 entries, kept / 16 in the gene's row); partition k of the gene has capacity frag_cap_of(reads) and starts at
// offset + k * capacity
__device__ __forceinline__ uint32_t frag_cap_of(unsigned long long reads) {
    return frag_parts_of(reads) == 1 ? (uint32_t)reads : (uint32_t)RSQC_K4_SUB_CAP;
}
// list entries a gene owns: partitions x capacity, rounded up to 16 so that the gene's offset fits the 32-bit field of its row
__device__ __forceinline__ unsigned long long frag_space_of(unsigned long long reads) {
    return ((unsigned long long)frag_parts_of(reads) * frag_cap_of(reads) + 15ull) & ~15ull;
}
// Two launches of 1024-thread workgroups, 1024 genes each (coalesced loads): (1) per-workgroup totals, (2) every workgroup
// adds the totals before it (a few dozen values) to its own scan.  (One workgroup walking all the genes serially took
// 0.24 ms at 56 202 genes: 55 dependent loads per thread, twice.)
__global__ void __launch_bounds__(1024)
frag_layout_totals_kernel(const unsigned long long *gene_reads, uint32_t n_genes, unsigned long long *blk_space, uint32_t *blk_parts, int *error) {
    __shared__ unsigned long long w_space[16];
    __shared__ uint32_t w_parts[16];
    const uint32_t g = blockIdx.x * 1024u + threadIdx.x;
    unsigned long long space = 0; uint32_t parts = 0;
    if (g < n_genes) {
        const unsigned long long n = gene_reads[g];
        if (n > 0xFFFFFFF0ull) atomicExch(error, RSQC_ERR_CAPACITY);
        parts = frag_parts_of(n); space = frag_space_of(n);
    }
    space = wave_sum(space); parts = wave_sum(parts);
    if (lane_id() == 0) { w_space[threadIdx.x >> 6] = space; w_parts[threadIdx.x >> 6] = parts; }
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned long long s = 0; uint32_t p = 0;
        for (int w = 0; w < 16; ++w) { s += w_space[w]; p += w_parts[w]; }
        blk_space[blockIdx.x] = s; blk_parts[blockIdx.x] = p;
    }
}
__global__ void __launch_bounds__(1024)
frag_layout_kernel(const unsigned long long *gene_reads, uint32_t n_genes, const unsigned long long *blk_space, const uint32_t *blk_parts,
                   uint32_t *part_first, uint4 *ginfo, uint32_t *cursor, uint4 *part_info, uint32_t *full_n) {
    __shared__ unsigned long long w_space[16];
    __shared__ uint32_t w_parts[16];
    __shared__ unsigned long long s_base; __shared__ uint32_t p_base;
    const int l = lane_id(), wv = (int)(threadIdx.x >> 6);
    if (wv == 0) {                                                     // offsets of this workgroup: totals of the ones before it
        unsigned long long s = 0; uint32_t p = 0;
        for (uint32_t k = (uint32_t)l; k < blockIdx.x; k += 64) { s += blk_space[k]; p += blk_parts[k]; }
        s = wave_sum(s); p = wave_sum(p);
        if (l == 0) { s_base = s; p_base = p; }
    }
    const uint32_t g = blockIdx.x * 1024u + threadIdx.x;
    unsigned long long space = 0; uint32_t parts = 0, cap = 0;
    if (g < n_genes) { const unsigned long long n = gene_reads[g]; parts = frag_parts_of(n); cap = frag_cap_of(n); space = frag_space_of(n); }
    unsigned long long isp = space; uint32_t ipt = parts;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const unsigned long long ts = __shfl_up(isp, o, 64); const uint32_t tp = __shfl_up(ipt, o, 64);
        if (l >= o) { isp += ts; ipt += tp; }
    }
    if (l == 63) { w_space[wv] = isp; w_parts[wv] = ipt; }
    __syncthreads();
    unsigned long long bs = s_base; uint32_t bp = p_base;
    for (int w = 0; w < wv; ++w) { bs += w_space[w]; bp += w_parts[w]; }
    const uint32_t pf = bp + ipt - parts; const unsigned long long gb = bs + isp - space;
    if (g < n_genes) { part_first[g] = pf; ginfo[g] = make_uint4(pf, parts, cap, (uint32_t)(gb >> 4)); }
    if (g == n_genes - 1) part_first[n_genes] = bp + ipt;
    if (g == 0) *full_n = 0u;
    // per partition: fill cursor = 0 and what the counting kernel needs in one load {gene, capacity, list offset}.  Most genes
    // own one partition; the wave walks the partitions of its larger genes together
    if (parts >= 1) { cursor[pf] = 0u; part_info[pf] = make_uint4(g, cap, (uint32_t)gb, (uint32_t)(gb >> 32)); }
    unsigned long long big = __ballot(parts > 1);
    while (big) {
        const int src = __ffsll(big) - 1; big &= big - 1;
        const uint32_t pf_s = lane_value(pf, src), n_s = lane_value(parts, src), cap_s = lane_value(cap, src), g_s = lane_value(g, src);
        const unsigned long long gb_s = (unsigned long long)lane_value((uint32_t)gb, src) | ((unsigned long long)lane_value((uint32_t)(gb >> 32), src) << 32);
        for (uint32_t k = 1u + (uint32_t)l; k < n_s; k += 64) {
            const unsigned long long off = gb_s + (unsigned long long)k * cap_s;
            cursor[pf_s + k] = 0u; part_info[pf_s + k] = make_uint4(g_s, cap_s, (uint32_t)off, (uint32_t)(off >> 32));
        }
    }
}

// frag_local_kernel: every (gene, name hash) pair of a chunk goes to the key list of its partition (partition = gene's first +
// hash-scaled index).  512 threads take 2048 pairs per pass (4 per thread): the pairs of a pass that share a partition reserve
// their list slots with ONE memory atomic (ranks inside the pass come from a small LDS table keyed by partition id).
// The kernel is a chain of dependent gathers, so it is laid out as a pipeline: the pairs of pass k + 1 and the per-gene rows of
// pass k are in flight while pass k - 1's ranks are taken.
// What bounds it is the memory side: one returning atomic and one scattered 12-byte store (FragKey) per pair of a gene with many
// partitions.  The two mates of a fragment sit a few hundred records apart, i.e. in the same chunk, so half of the pairs are
// repeats the counting kernel would throw away: a direct-mapped LDS window over the WHOLE chunk (one 64-bit + one 32-bit exchange per pair,
// never cleared between passes, no probing: a newer pair simply replaces an older one) drops a pair whose word is already
// there.  (A per-pass table with probing removed 12 % of the keys for 18 % of the kernel; the window removes the mates.)
// The word is key ^ f(gene), compared together with the pair's second name hash (rsqc_batch.qhash2, a parallel 32-bit window):
// a pair is dropped only against its own (gene, 96-bit name identity).
#ifndef RSQC_K4L_THREADS
#define RSQC_K4L_THREADS 512
#endif
#ifndef RSQC_K4L_U
#define RSQC_K4L_U 4
#endif
#define RSQC_K4L_PIECE (RSQC_K4L_U * RSQC_K4L_THREADS)
#define RSQC_K4L_GSLOTS RSQC_K4L_PIECE
constexpr int k4l_log2(unsigned v) { return v <= 1 ? 0 : 1 + k4l_log2(v >> 1); }
#ifndef RSQC_K4L_WIN
#define RSQC_K4L_WIN 2048
#endif
#ifndef RSQC_K4L_CHUNKS
#define RSQC_K4L_CHUNKS 2                        /* pair chunks a workgroup of frag_local_kernel takes as one run (A/B: 1) */
#endif
// workgroups of frag_local_kernel that take chunks (the launch adds the sharers of the dense region behind them)
inline uint32_t frag_local_chunk_wgs(uint32_t n_chunks) { return (n_chunks + RSQC_K4L_CHUNKS - 1u) / RSQC_K4L_CHUNKS; }
struct K4LocalShared {
    uint32_t gkey[RSQC_K4L_GSLOTS], gcnt[RSQC_K4L_GSLOTS];      // keyed by partition id: pairs of the pass, then their first list slot
    unsigned long long win[RSQC_K4L_WIN];       // direct-mapped window of the chunk's recent (gene, key) words: see the kernel
    uint32_t win2[RSQC_K4L_WIN];                // ... and their second name hash (rsqc_batch.qhash2; all 0 without it)
};

#ifdef __HIPCC__
#ifndef RSQC_K4L_WAVES
#define RSQC_K4L_WAVES 6
#endif
#define RSQC_K4L_OCC __attribute__((amdgpu_waves_per_eu(RSQC_K4L_WAVES, RSQC_K4L_WAVES)))   // three workgroups per CU (<= 80 VGPRs): at 90 two fit, +20 % time
#else
#define RSQC_K4L_OCC
#endif
__global__ void __launch_bounds__(RSQC_K4L_THREADS) RSQC_K4L_OCC
frag_local_kernel(const PairRec *pairs, uint32_t chunk_cap,
                  const uint32_t *chunk_count, uint32_t n_chunks, uint32_t slow_base, uint32_t slow_cap,
                  const uint4 *ginfo, uint32_t *cursor, FragKey *list, int *error) {
    __shared__ K4LocalShared S;
    // A workgroup takes RSQC_K4L_CHUNKS neighbouring chunks as ONE run of pairs (round 6: K1 runs 5120 workgroups + 1024 of the long
    // kernel -- one workgroup of this kernel per chunk cost 0.03 ms per thousand chunks in launches, window clears and tails; neighbouring
    // chunks are neighbouring genomic ranges, so the window de-dup carries over).
    constexpr int CH = RSQC_K4L_CHUNKS;
    const uint32_t n_cwg = (n_chunks + (uint32_t)CH - 1u) / (uint32_t)CH;      // workgroups that take chunks; the rest share the dense region
    uint32_t cbase[CH], cend[CH];                                              // first pair of chunk k, end of chunk k in the run
    uint32_t count, piece0 = 0; constexpr uint32_t piece_step = 1;
    if (blockIdx.x < n_cwg) {
        uint32_t run = 0;
#pragma unroll
        for (int k = 0; k < CH; ++k) {
            const uint32_t c = blockIdx.x * (uint32_t)CH + (uint32_t)k;
            const uint32_t n = c < n_chunks ? (chunk_count[c] < chunk_cap ? chunk_count[c] : chunk_cap) : 0u;
            cbase[k] = c * chunk_cap; run += n; cend[k] = run;
        }
        count = run;
    } else {
#pragma unroll
        for (int k = 0; k < CH; ++k) { cbase[k] = slow_base; cend[k] = 0u; }
        count = chunk_count[n_chunks] < slow_cap ? chunk_count[n_chunks] : slow_cap;
        cend[CH - 1] = count;
#pragma unroll
        for (int k = 0; k < CH - 1; ++k) cend[k] = 0u;
    }
    if (count == 0u) return;                                               // (empty chunks: a sparse stretch, idle workgroups of classify_long_kernel)
    constexpr int U = RSQC_K4L_PIECE / RSQC_K4L_THREADS;
    constexpr uint32_t NONE = 0xFFFFFFFFu;
    uint32_t n_pieces = (count + RSQC_K4L_PIECE - 1) / RSQC_K4L_PIECE;
    if (blockIdx.x >= n_cwg) {
        // a dense region shared by several workgroups (a batch's slow-path region, the arena of retired batches): each takes a
        // CONTIGUOUS run of passes, so that its window sees neighbouring records
        const uint32_t sharers = gridDim.x - n_cwg, me = blockIdx.x - n_cwg;
        const uint32_t per = (n_pieces + sharers - 1) / sharers;
        piece0 = me * per < n_pieces ? me * per : n_pieces;
        n_pieces = piece0 + per < n_pieces ? piece0 + per : n_pieces;
    }
    auto pair_at = [&](uint32_t j) -> uint32_t {                           // pair j of the run -> its index in the pair buffer
        uint32_t at = cbase[CH - 1] + (j - (CH > 1 ? cend[CH - 2] : 0u));
#pragma unroll
        for (int k = CH - 2; k >= 0; --k) if (j < cend[k]) at = cbase[k] + (j - (k > 0 ? cend[k - 1] : 0u));
        return at;
    };
    auto load_piece = [&](uint32_t piece, uint32_t (&g)[U], uint64_t (&key)[U], uint32_t (&h2)[U]) {      // one 16-byte load per pair
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const uint32_t j = piece * RSQC_K4L_PIECE + (uint32_t)u * RSQC_K4L_THREADS + threadIdx.x;
            const bool ok = piece < n_pieces && j < count;
            PairRec r{NONE, 0u, 0ull};
            if (ok) r = pairs[pair_at(j)];
            g[u] = r.gene; key[u] = r.hash; h2[u] = r.h2;
        }
    };
#ifdef RSQC_K1_PROF
    if (blockIdx.x == 1000 && n_chunks > 1000) {                           // (diagnostic: one chunk, as K1 wrote it)
        for (uint32_t i = threadIdx.x; i < count && i < 32768u; i += blockDim.x) { g_dbg_pair_hash[i] = pairs[pair_at(i)].hash; g_dbg_pair_gene[i] = pairs[pair_at(i)].gene; }
        if (threadIdx.x == 0) g_dbg_pair_count = count;
    }
#endif
    uint32_t g[U]; uint64_t key[U]; uint32_t h2[U];
    load_piece(piece0, g, key, h2);
    for (int i = threadIdx.x; i < RSQC_K4L_WIN; i += blockDim.x) { S.win[i] = 0ull; S.win2[i] = 0u; }
    RSQC_FIN_BEGIN
    for (uint32_t piece = piece0; piece < n_pieces; piece += piece_step) {
        RSQC_FIN_SECT(32, 0);
        uint4 gi[U];
#pragma unroll
        for (int u = 0; u < U; ++u) { const uint32_t gq = g[u] != NONE ? g[u] : 0u; gi[u] = ginfo[gq]; }
        uint32_t gn[U]; uint64_t keyn[U]; uint32_t h2n[U];
        load_piece(piece + piece_step, gn, keyn, h2n);
        __syncthreads();                                                   // (the previous pass has read its list slots)
        for (int i = threadIdx.x; i < RSQC_K4L_GSLOTS; i += blockDim.x) { S.gkey[i] = NONE; S.gcnt[i] = 0u; }
        __syncthreads();
        RSQC_FIN_SECT(32, 1);
        uint32_t gp[U], gslot[U], rank[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {                                      // rank of the pair among the pass's pairs of its partition
            gslot[u] = NONE; rank[u] = 0; gp[u] = 0;
            if (key[u] == 0) key[u] = 0x9e3779b97f4a7c15ull;
            if (g[u] == NONE) continue;
            {
                unsigned long long lk = key[u] ^ (((unsigned long long)g[u] << 32) | (unsigned long long)(g[u] * 0x9E3779B1u));
                if (lk == 0ull) lk = 1ull;
                static_assert((RSQC_K4L_WIN & (RSQC_K4L_WIN - 1)) == 0, "window size");
                const uint32_t ws = ((((uint32_t)lk ^ (uint32_t)(lk >> 32)) * 0x9E3779B1u) >> 12) & (RSQC_K4L_WIN - 1);
                // (both words are exchanged: a pair is dropped only when the slot held ITS 96 bits -- two concurrent writers of one
                //  slot can at worst make each other survive, which the exact count behind this stage absorbs)
                const bool seen1 = atomicExch(&S.win[ws], lk) == lk, seen2 = atomicExch(&S.win2[ws], h2[u]) == h2[u];
                if (seen1 && seen2) { g[u] = NONE; continue; }                          // its mate went through this chunk already
            }
            gp[u] = gi[u].x + (gi[u].y > 1 ? (uint32_t)(((unsigned long long)frag_part_hash(key[u]) * gi[u].y) >> 32) : 0u);
            static_assert((RSQC_K4L_GSLOTS & (RSQC_K4L_GSLOTS - 1)) == 0, "slot hash");
            uint32_t sl = (gp[u] * 0x9E3779B1u) >> (32 - k4l_log2(RSQC_K4L_GSLOTS));
#pragma unroll 1
            for (int probe = 0; probe < 16; ++probe) {
                const uint32_t o = atomicCAS(&S.gkey[sl], NONE, gp[u]);
                if (o == NONE || o == gp[u]) { gslot[u] = sl; rank[u] = atomicAdd(&S.gcnt[sl], 1u); break; }
                sl = (sl + 1) & (RSQC_K4L_GSLOTS - 1);
            }
        }
        RSQC_FIN_SECT(32, 3);
        __syncthreads();
        RSQC_FIN_SECT(32, 4);
        {                                                                  // one reservation per partition of the pass, all in flight
            constexpr int R = RSQC_K4L_GSLOTS / RSQC_K4L_THREADS;
            uint32_t pk[R], pc[R], pr[R];
#pragma unroll
            for (int j = 0; j < R; ++j) { pk[j] = S.gkey[j * RSQC_K4L_THREADS + threadIdx.x]; pc[j] = S.gcnt[j * RSQC_K4L_THREADS + threadIdx.x]; }
#pragma unroll
            for (int j = 0; j < R; ++j) pr[j] = pk[j] != NONE ? atomicAdd(&cursor[pk[j]], pc[j]) : 0u;
#pragma unroll
            for (int j = 0; j < R; ++j) S.gcnt[j * RSQC_K4L_THREADS + threadIdx.x] = pr[j];
        }
        __syncthreads();
        RSQC_FIN_SECT(32, 5);
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (g[u] == NONE) continue;
            const uint32_t at = gslot[u] != NONE ? S.gcnt[gslot[u]] + rank[u] : atomicAdd(&cursor[gp[u]], 1u);   // (crowded table)
            const uint32_t cap = gi[u].z;
            if (at < cap) {
                const unsigned long long where = ((unsigned long long)gi[u].w << 4) + (unsigned long long)(gp[u] - gi[u].x) * cap + at;
                list[where] = FragKey{(uint32_t)key[u], (uint32_t)(key[u] >> 32), h2[u]};   // (one 12-byte store)
            } else atomicExch(error, RSQC_ERR_CAPACITY);
        }
        RSQC_FIN_SECT(32, 6);
#ifdef RSQC_K1_PROF
        if (threadIdx.x == 0) atomicAdd(&g_fin_prof[32 + 15], 1ull);
#endif
#pragma unroll
        for (int u = 0; u < U; ++u) { g[u] = gn[u]; key[u] = keyn[u]; h2[u] = h2n[u]; }
    }
}

// frag_count_kernel: one workgroup per partition at a time: its keys go through an LDS set sized to the partition, the number
// of distinct keys is added to the gene.  Three partitions are in flight per workgroup: the row {gene, capacity, list
// offset} + fill count of the one after next, the keys of the next one (all loads issued together), the set of the current one.
// (Measured against one WAVE per partition with partitions a quarter of the size: the counting got 15 % faster, the
// scatter in front of it 40 % slower -- it pays one returning atomic per partition touched by a pass.)
// Two instances: SLOTS = PART_SLOTS / 2 (16 KB of LDS, eight workgroups per CU) takes the partitions whose keys fit it at half
// load -- with the window de-dup in front nearly all of them --, SLOTS = PART_SLOTS (32 KB) the fuller ones; each skips the
// other's partitions: the first instance walks all partitions and LISTS the fuller ones (`full_list`, counter zeroed by
// frag_layout_kernel), the second walks that list.
template <int SLOTS>
__global__ void __launch_bounds__(RSQC_K4_COUNT_THREADS)
frag_count_kernel(const uint32_t *n_parts_at, const uint32_t *cursor, const uint4 *part_info, const FragKey *list,
                  unsigned long long *gene_frag, uint32_t *full_list, uint32_t *full_n, int *error) {
    constexpr bool LISTED = SLOTS == RSQC_K4_PART_SLOTS;
    __shared__ unsigned long long s_keys[SLOTS];
    __shared__ uint32_t s_h2[SLOTS];                                       // second hash of the slot's owner
    // two names with one 64-bit hash and different second hashes (never seen outside the crafted fixture): set aside here and
    // counted by one thread
    constexpr uint32_t OVF = 32;
    __shared__ unsigned long long s_ovk[OVF];
    __shared__ uint32_t s_ov2[OVF], s_ovn;
    __shared__ uint32_t s_fresh[2];
    constexpr int KPT = (SLOTS / 2) / RSQC_K4_COUNT_THREADS;                // keys per thread of the fullest list of this instance
    constexpr uint32_t N_LO = SLOTS == RSQC_K4_PART_SLOTS ? (uint32_t)SLOTS / 4u : 0u;   // this instance: N_LO < keys <= SLOTS / 2
    const uint32_t n_parts = LISTED ? *full_n : *n_parts_at;               // (iterations: partitions, or entries of the list)
    struct Row { uint32_t fill; uint4 info; bool fuller; };
    auto row_of = [&](uint32_t i) -> Row {
        Row r; r.fill = 0u; r.info = make_uint4(0u, 0u, 0u, 0u); r.fuller = false;
        if (i < n_parts) {
            const uint32_t w = LISTED ? full_list[i] : i;
            r.fill = cursor[w]; r.info = part_info[w];
            const uint32_t n = r.fill < r.info.y ? r.fill : r.info.y;
            r.fuller = n > (uint32_t)SLOTS / 2u;
            if (n <= N_LO || r.fuller) r.fill = 0u;                        // (the other instance's, or empty)
        }
        return r;
    };
    auto keys_of = [&](const Row &r, unsigned long long (&kv)[KPT], uint32_t (&k2)[KPT]) {
        const uint32_t n = r.fill < r.info.y ? r.fill : r.info.y;
        const unsigned long long off = (unsigned long long)r.info.z | ((unsigned long long)r.info.w << 32);
        const FragKey *keys = list + off;
#pragma unroll
        for (int j = 0; j < KPT; ++j) {
            const uint32_t i = (uint32_t)j * RSQC_K4_COUNT_THREADS + threadIdx.x;
            FragKey e{0u, 0u, 0u};
            if (i < n) e = keys[i];
            kv[j] = (unsigned long long)e.lo | ((unsigned long long)e.hi << 32); k2[j] = e.h2;
        }
    };
    uint32_t w = blockIdx.x;
    Row cur = row_of(w), nxt = row_of(w + gridDim.x);
    unsigned long long kv[KPT], kvn[KPT]; uint32_t k2[KPT], k2n[KPT];
    keys_of(cur, kv, k2);
    if (threadIdx.x < 2) s_fresh[threadIdx.x] = 0u;
    if (threadIdx.x == 0) s_ovn = 0u;
    uint32_t round = 0;
    // Two barriers per partition (round 6, third session; rounds 2-6: four).  What the workgroup owes a partition AFTER its inserts -- the sum of the
    // waves' fresh keys, the distinct set-aside entries, the one add to the gene -- is settled by thread 0 behind the NEXT partition's barrier B
    // (every wave is past the previous partition's second-hash checks when it arrives there), while the other threads insert:
    //     clear s_keys | B | insert (thread 0 first settles the partition before) | C | second-hash checks, fresh keys into s_fresh[round & 1]
    // No barrier is needed in front of the clear: the checks behind C read s_h2 and the set-aside list only, s_h2 is written again behind the
    // next B, and a wave clears s_keys only after every wave has passed C, i.e. has finished its inserts.
    bool owe = false; uint32_t owe_gene = 0u;                               // (uniform) a partition waits to be settled
    auto settle = [&]() {                                                   // thread 0, behind a barrier that follows the partition's checks
        const uint32_t cell = (round - 1u) & 1u;
        uint32_t t = s_fresh[cell]; s_fresh[cell] = 0u;
        if (s_ovn) {
            const uint32_t m = s_ovn < OVF ? s_ovn : OVF;
            for (uint32_t a = 0; a < m; ++a) {
                bool first = true;
                for (uint32_t b2 = 0; b2 < a; ++b2) if (s_ovk[b2] == s_ovk[a] && s_ov2[b2] == s_ov2[a]) first = false;
                if (first) ++t;
            }
            s_ovn = 0u;                                                     // (the current partition's entries come behind its barrier C)
        }
        if (t) atomicAdd(&gene_frag[owe_gene], (unsigned long long)t);
    };
#if defined(__HIP_DEVICE_COMPILE__)
#define K4_UNI(x) ((uint32_t)__builtin_amdgcn_readfirstlane((int)(x)))     /* a value every thread loaded from one address: kept in a scalar register */
#else
#define K4_UNI(x) ((uint32_t)(x))
#endif
    RSQC_FIN_BEGIN
    while (w < n_parts) {
        RSQC_FIN_SECT(48, 0);
        const Row nn = row_of(w + 2u * gridDim.x);
        keys_of(nxt, kvn, k2n);
        if (!LISTED && cur.fuller && threadIdx.x == 0) full_list[atomicAdd(full_n, 1u)] = w;
        if (cur.fill != 0u) {                                               // (uniform)
            const uint32_t gene = K4_UNI(cur.info.x), cap = K4_UNI(cur.info.y), fill = K4_UNI(cur.fill);
            const uint32_t n = fill < cap ? fill : cap;
            // the set is sized to the partition: the smallest power of two >= 2 n, at least 64 (most partitions hold a few hundred keys)
            uint32_t slots = 64;
            while (slots < 2 * n && slots < (uint32_t)SLOTS) slots <<= 1;
            const uint32_t smask = slots - 1;
            for (uint32_t i = threadIdx.x; i < slots; i += blockDim.x) s_keys[i] = 0ull;
            __syncthreads();                                                // B
            if (owe && threadIdx.x == 0) settle();
            RSQC_FIN_SECT(48, 1);
            uint32_t fresh = 0;
            constexpr uint32_t NO_SLOT = 0xFFFFFFFFu;
            // returns the slot whose owner has the same 64-bit key (its second hash is compared after the barrier below: the owner
            // may not have written it yet), NO_SLOT when the key was placed (or the set is full)
            auto insert = [&](unsigned long long k, uint32_t h2) -> uint32_t {
                // (the keys are fmix64 outputs and the partition was chosen from the HIGH word: the low word is as good as a
                //  fresh hash inside the partition; one multiply spreads neighbouring values anyway)
                uint32_t slot = (((uint32_t)k * 0x9E3779B1u) >> 16) & smask;
#pragma unroll 1
                for (uint32_t probe = 0; probe < slots; ++probe) {
                    const unsigned long long old = atomicCAS(&s_keys[slot], 0ull, k);
                    if (old == 0ull) { s_h2[slot] = h2; ++fresh; return NO_SLOT; }
                    if (old == k) return slot;
                    slot = (slot + 1) & smask;
                }
                atomicExch(error, RSQC_ERR_CAPACITY);
                return NO_SLOT;
            };
            uint32_t same[KPT];
#pragma unroll
            for (int j = 0; j < KPT; ++j) same[j] = ((uint32_t)j * RSQC_K4_COUNT_THREADS + threadIdx.x < n) ? insert(kv[j], k2[j]) : NO_SLOT;
            static_assert(RSQC_K4_SUB_CAP <= RSQC_K4_PART_SLOTS / 2 && RSQC_K4_PART_READS <= RSQC_K4_PART_SLOTS / 2, "a list fits the registers of its instance");
            if (n > (uint32_t)KPT * RSQC_K4_COUNT_THREADS) atomicExch(error, RSQC_ERR_CAPACITY);   // (cannot happen: capacities are <= SUB_CAP)
            // second hashes of the entries whose 64-bit key was there already (the owner's is in s_h2 once every wave is past its
            // inserts): an entry that differs is set aside; thread 0 counts the distinct ones when it settles the partition
            __syncthreads();                                                // C
#pragma unroll
            for (int j = 0; j < KPT; ++j)
                if (same[j] != NO_SLOT && s_h2[same[j]] != k2[j]) {
                    const uint32_t at = atomicAdd(&s_ovn, 1u);
                    if (at < OVF) { s_ovk[at] = kv[j]; s_ov2[at] = k2[j]; } else atomicExch(error, RSQC_ERR_CAPACITY);
                }
            RSQC_FIN_SECT(48, 2);
            fresh = wave_sum_u32_full(fresh);                               // (every lane is here: the branch above is uniform)
            // the per-partition total alternates between two LDS cells: the partition after this one adds to the other cell, and this
            // one's is read and cleared behind that partition's barrier B
            if (lane_id() == 0 && fresh) atomicAdd(&s_fresh[round & 1], fresh);
            owe = true; owe_gene = gene;
            ++round;
            RSQC_FIN_SECT(48, 3);
#ifdef RSQC_K1_PROF
            if (threadIdx.x == 0) { atomicAdd(&g_fin_prof[48 + 15], 1ull); atomicAdd(&g_fin_prof[48 + 14], (unsigned long long)n); }
#endif
        }
        w += gridDim.x; cur = nxt; nxt = nn;
#pragma unroll
        for (int j = 0; j < KPT; ++j) { kv[j] = kvn[j]; k2[j] = k2n[j]; }
    }
    if (owe) {                                                              // (uniform) the last partition of the workgroup
        __syncthreads();
        if (threadIdx.x == 0) settle();
    }
#undef K4_UNI
}

}  // namespace rsqc
