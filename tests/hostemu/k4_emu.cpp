// k4_emu.cpp -- TEST HARNESS ONLY (never loaded by the product).
//
// The fragment-counting KERNELS -- rnaseqc_amd/csrc/rsqc_k4.h (frag_layout_totals / frag_layout / frag_local / the two
// instances of frag_count), unmodified -- compiled for the host on top of the 64-lane fiber emulation of wavemu.h and run
// on seeded (gene, name hash) pairs against a std::set per gene: partition layout, the chunk-wide de-dup window, ranks and
// list reservations, the dense "arena" form of the input, the split of the partitions between the two counting instances.
#include "wavemu.h"

#include <algorithm>
#include <set>
#include <vector>

#include "../../rnaseqc_amd/csrc/rsqc_read.h"
#include "../../rnaseqc_amd/csrc/rsqc_device.h"
#include "../../rnaseqc_amd/csrc/rsqc_wave.h"
#define RSQC_FIN_STAMP(sec)
#define RSQC_FIN_SECT(base, sec)
#define RSQC_FIN_BEGIN
#include "../../rnaseqc_amd/csrc/rsqc_k4.h"

using namespace rsqc;

namespace {
struct Rng { uint64_t s; uint64_t next() { s += 0x9E3779B97F4A7C15ull; uint64_t z = s; z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; return z ^ (z >> 31); }
             uint32_t below(uint32_t n) { return (uint32_t)((next() >> 32) * (uint64_t)n >> 32); } };
}  // namespace

// mode 0: the pairs sit in `n_chunks` chunks (+ the slow-path region); mode 1: one dense list (retired batches).
// hot_reads: pairs of gene 0 (names mostly unique: its partitions fill beyond 1024 keys and go to the second counting instance);
// returns 0 when every gene's count equals the size of its name set, else 1 + the first gene that differs; -code on a device error.
extern "C" __attribute__((visibility("default")))
int k4emu_run(uint64_t seed, int n_genes, int n_chunks, int n_names, int hot_reads, int mode, uint64_t *stats /*[6]*/) {
    // mode bit 0: dense-list form; bit 1: names with second hashes (rsqc_batch.qhash2), some sharing their 64-bit key; without it every second hash is 0
    const bool has2 = (mode & 2) != 0; mode &= 1;
    Rng R{seed};
    struct Pair { uint32_t g; uint64_t key; uint32_t h2; };
    std::vector<Pair> stream;
    // names: one or two records each; the second sits a random distance behind the first (inside or beyond the window, sometimes in
    // another chunk); one name in sixteen is also counted to a second gene with the same key; a few keys are 0
    for (int i = 0; i < n_names; ++i) {
        const uint32_t g = 1u + R.below((uint32_t)n_genes - 1u);
        uint64_t key = R.next(); if (i % 997 == 0) key = 0ull;
        uint32_t h2 = has2 ? (uint32_t)R.next() : 0u;
        // (96-bit identity) one name in 53 takes the 64-bit key -- and the gene -- of an earlier, different name: one more fragment
        if (has2 && i % 53 == 52 && !stream.empty()) { const Pair &o = stream[R.below((uint32_t)stream.size())]; stream.push_back(Pair{o.g, o.key, h2}); if (R.below(2)) stream.push_back(Pair{o.g, o.key, h2}); continue; }
        stream.push_back(Pair{g, key, h2});
        if (R.below(16) == 0) stream.push_back(Pair{1u + R.below((uint32_t)n_genes - 1u), key, h2});
    }
    for (int i = 0; i < hot_reads; ++i) stream.push_back(Pair{0u, R.next(), has2 ? (uint32_t)R.next() : 0u});
    // shuffle lightly, then add the mates at their distances
    for (size_t i = stream.size(); i > 1; --i) std::swap(stream[i - 1], stream[R.below((uint32_t)i)]);
    {
        std::vector<Pair> with_mates; with_mates.reserve(stream.size() * 2);
        std::vector<std::pair<size_t, Pair>> later;
        for (size_t i = 0; i < stream.size(); ++i) {
            with_mates.push_back(stream[i]);
            const bool hot = stream[i].g == 0u;
            if (R.below(hot ? 8u : 2u) == 0u) {
                const uint32_t k = R.below(4);
                const size_t d = k == 0 ? 1 + R.below(8) : k == 1 ? 1 + R.below(600) : k == 2 ? 1 + R.below(5000) : 1 + R.below(60000);
                later.push_back({with_mates.size() + d, stream[i]});
            }
        }
        std::sort(later.begin(), later.end(), [](const std::pair<size_t, Pair> &a, const std::pair<size_t, Pair> &b) { return a.first > b.first; });
        for (auto &m : later) { const size_t at = std::min(m.first, with_mates.size()); with_mates.insert(with_mates.begin() + (long)at, m.second); }
        stream.swap(with_mates);
    }
    const size_t n_pairs = stream.size();
    const uint32_t G = (uint32_t)n_genes;
    std::vector<unsigned long long> gene_reads(G, 0ull);
    std::vector<std::set<std::pair<uint64_t, uint32_t>>> names(G);
    for (const Pair &p : stream) { gene_reads[p.g]++; names[p.g].insert({p.key == 0ull ? 0x9e3779b97f4a7c15ull : p.key, p.h2}); }

    // ---- the plan's arrays, sized like rsqc_api.cpp sizes them
    const uint64_t parts_bound = n_pairs / RSQC_K4_PART_READS + G + 1;
    const uint64_t keys_bound = 2 * n_pairs + (uint64_t)RSQC_K4_SUB_CAP * std::min<uint64_t>(parts_bound, n_pairs / RSQC_K4_PART_READS + 1) + 16ull * G + 16;
    const uint32_t lay_blocks = (G + 1023u) / 1024u;
    std::vector<uint4> ginfo(G + 1), part_info(parts_bound);
    std::vector<uint32_t> part_first(G + 2), cursor(parts_bound, 0xDEADBEEFu), full_list(parts_bound), blk_parts(lay_blocks);
    std::vector<unsigned long long> blk_space(lay_blocks), gene_frag(G, 0ull);
    std::vector<FragKey> list(keys_bound, FragKey{0xABABABABu, 0xABABABABu, 0xCDCDCDCDu});
    uint32_t full_n = 0xDEADBEEFu; int error = 0;

    wavemu::grid_dim().x = lay_blocks;
    for (uint32_t b = 0; b < lay_blocks; ++b) { wavemu::block_idx().x = b; wavemu::run_block(1024, [&]() { frag_layout_totals_kernel(gene_reads.data(), G, blk_space.data(), blk_parts.data(), &error); }); }
    for (uint32_t b = 0; b < lay_blocks; ++b) { wavemu::block_idx().x = b; wavemu::run_block(1024, [&]() { frag_layout_kernel(gene_reads.data(), G, blk_space.data(), blk_parts.data(), part_first.data(), ginfo.data(), cursor.data(), part_info.data(), &full_n); }); }
    if (error) return -error;
    const uint32_t n_parts = part_first[G];
    if (n_parts > parts_bound || full_n != 0u) return -1000;

    // ---- the pairs as K1 leaves them
    std::vector<PairRec> pairs; std::vector<uint32_t> counts;
    uint32_t chunk_cap = 0, slow_base = 0, slow_cap = 0, grid = 0, nch = 0;
    if (mode == 0) {
        nch = (uint32_t)n_chunks;
        const size_t slow_n = n_pairs / 50;                                     // the tail goes to the slow-path region
        const size_t body = n_pairs - slow_n;
        std::vector<size_t> cut(nch + 1, 0);
        for (uint32_t c = 1; c < nch; ++c) cut[c] = R.below((uint32_t)body + 1u);
        cut[nch] = body; std::sort(cut.begin(), cut.end());
        for (uint32_t c = 0; c < nch; ++c) chunk_cap = std::max<uint32_t>(chunk_cap, (uint32_t)(cut[c + 1] - cut[c]));
        chunk_cap += 7; slow_base = nch * chunk_cap; slow_cap = (uint32_t)slow_n + 5;
        pairs.assign((size_t)slow_base + slow_cap, PairRec{0xFFFFFFF0u, 0xDEADu, 0ull});
        counts.assign(nch + 1, 0u);
        for (uint32_t c = 0; c < nch; ++c) {
            counts[c] = (uint32_t)(cut[c + 1] - cut[c]);
            for (size_t i = cut[c]; i < cut[c + 1]; ++i) pairs[(size_t)c * chunk_cap + (i - cut[c])] = PairRec{stream[i].g, stream[i].h2, stream[i].key};
        }
        counts[nch] = (uint32_t)slow_n;
        for (size_t i = 0; i < slow_n; ++i) pairs[slow_base + i] = PairRec{stream[body + i].g, stream[body + i].h2, stream[body + i].key};
        grid = frag_local_chunk_wgs(nch) + 32u;
    } else {
        nch = 0; chunk_cap = 0; slow_base = 0; slow_cap = (uint32_t)n_pairs;
        pairs.resize(n_pairs); counts.assign(1, (uint32_t)n_pairs);
        for (size_t i = 0; i < n_pairs; ++i) pairs[i] = PairRec{stream[i].g, stream[i].h2, stream[i].key};
        grid = (uint32_t)std::min<uint64_t>(4096, n_pairs / 1024 + 1);
        grid = std::min<uint32_t>(grid, 24u);                                 // (emulation time; any sharing is legal)
    }
    wavemu::grid_dim().x = grid;
    for (uint32_t b = 0; b < grid; ++b) {
        wavemu::block_idx().x = b;
        wavemu::run_block(RSQC_K4L_THREADS, [&]() { frag_local_kernel(pairs.data(), chunk_cap, counts.data(), nch, slow_base, slow_cap, ginfo.data(), cursor.data(), list.data(), &error); });
    }
    if (error) return -error;
    uint64_t kept = 0; uint32_t fuller = 0;
    for (uint32_t w = 0; w < n_parts; ++w) { kept += cursor[w]; if (cursor[w] > part_info[w].y) return -1001; if (cursor[w] > (uint32_t)RSQC_K4_PART_SLOTS / 4u) fuller++; }

    const uint32_t cgrid = 16;
    wavemu::grid_dim().x = cgrid;
    for (uint32_t b = 0; b < cgrid; ++b) { wavemu::block_idx().x = b; wavemu::run_block(RSQC_K4_COUNT_THREADS, [&]() { frag_count_kernel<RSQC_K4_PART_SLOTS / 2>(part_first.data() + G, cursor.data(), part_info.data(), list.data(), gene_frag.data(), full_list.data(), &full_n, &error); }); }
    if (full_n != fuller) return -1002;
    wavemu::grid_dim().x = 4;
    for (uint32_t b = 0; b < 4; ++b) { wavemu::block_idx().x = b; wavemu::run_block(RSQC_K4_COUNT_THREADS, [&]() { frag_count_kernel<RSQC_K4_PART_SLOTS>(part_first.data() + G, cursor.data(), part_info.data(), list.data(), gene_frag.data(), full_list.data(), &full_n, &error); }); }
    if (error) return -error;
    if (stats) { stats[0] = n_pairs; stats[1] = kept; stats[2] = n_parts; stats[3] = fuller; stats[4] = 0; for (uint32_t g = 0; g < G; ++g) stats[4] += names[g].size(); stats[5] = chunk_cap; }
    for (uint32_t g = 0; g < G; ++g) if (gene_frag[g] != (unsigned long long)names[g].size()) return 1 + (int)g;
    return 0;
}
