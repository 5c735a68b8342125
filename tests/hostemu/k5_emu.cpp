// k5_emu.cpp -- TEST HARNESS ONLY (never loaded by the product).
//
// The fragment-size KERNELS -- rnaseqc_amd/csrc/rsqc_k5.h (partition by name hash, per-bucket LDS sort + replay of the
// reference's state machine, radix select of the first N samples by file index, size histogram + compaction), unmodified --
// compiled for the host on top of the 64-lane fiber emulation of wavemu.h and run on seeded candidates against a literal
// std::map walk in file order (src/Expression.cpp:509-538 with the --fragment-samples cut-off of src/RNASeQC.cpp:372-376).
// The host side of rsqc_fragsize.hip (launch order, the digit loop of the selection) is restated here with plain memory.
#include "wavemu.h"

#include <algorithm>
#include <map>
#include <vector>

#include "../../rnaseqc_amd/csrc/rsqc_read.h"
#include "../../rnaseqc_amd/csrc/rsqc_device.h"
#include "../../rnaseqc_amd/csrc/rsqc_k5.h"

using namespace rsqc;

namespace {
struct Rng { uint64_t s; uint64_t next() { s += 0x9E3779B97F4A7C15ull; uint64_t z = s; z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; return z ^ (z >> 31); }
             uint32_t below(uint32_t n) { return (uint32_t)((next() >> 32) * (uint64_t)n >> 32); } };
template <class F> void launch(uint32_t grid, int threads, F &&body) {
    wavemu::grid_dim().x = grid;
    for (uint32_t b = 0; b < grid; ++b) { wavemu::block_idx().x = b; wavemu::run_block(threads, body); }
}
}  // namespace

// n_names names with 1-4 candidate records each (BED interval, end position, flags and sizes at random, a few sizes beyond the
// direct table), emitted in a shuffled order like K1's atomic slots.  Returns 0 when samples kept, their (size, count) pairs and
// the kept (file index, size) set equal the literal walk; 1000 + k for check k; -code for a device error.
extern "C" __attribute__((visibility("default")))
int k5emu_run(uint64_t seed, int n_names, uint32_t max_samples, int hot /* records of ONE extra name: a bucket beyond the LDS sort; -1: a file as sequencers write
                 them -- one or two candidates per name -- plus, one name in 997, a DIFFERENT name crafted to share the 64-bit mix of the pairing set (rsqc_k5.h,
                 pair_bucket_hashed) with an earlier one */, uint64_t *stats /*[7]: candidates, samples, kept, distinct sizes, listed buckets, buckets paired through the set, buckets sorted*/) {
    Rng R{seed};
    const bool pairs_only = hot < 0; if (pairs_only) hot = 0;
    g_k5_hashed_buckets = 0; g_k5_sorted_buckets = 0;
    struct Cand { uint64_t file, q; uint32_t h2; int32_t name, endpos; uint32_t flag_size; };
    std::vector<Cand> cands;
    uint64_t file = 1000;
    for (int i = 0; i < n_names; ++i) {
        uint64_t q = R.next();
        if (i % 311 == 0) q = ~0ull;                                  // (the padding key of the LDS sort, as a real name)
        uint32_t h2 = (uint32_t)R.next();
        if (pairs_only) {
            if (i % 997 == 996 && !cands.empty()) {        // another name on an earlier name's mix: q' ^ h2' K = q ^ h2 K (the set must hand the bucket to the sort)
                const Cand &o = cands[R.below((uint32_t)cands.size())];
                // (... AND its bucket: the buckets are cut on the high bits of q, so the two mixes' difference must leave them alone)
                uint64_t d = 0;
                do { h2 = (uint32_t)R.next(); d = ((uint64_t)o.h2 * 0x9E3779B97F4A7C15ull) ^ ((uint64_t)h2 * 0x9E3779B97F4A7C15ull); } while (h2 == o.h2 || (d >> 50) != 0);
                q = o.q ^ d;
            }
            const int k2 = 1 + (int)R.below(2);
            const int32_t iv2 = (int32_t)R.below(50);
            for (int j = 0; j < k2; ++j) {
                Cand c;
                c.q = q; c.h2 = h2; c.name = R.below(5) == 0 ? iv2 + 1 : iv2; c.endpos = 1000 + (int32_t)R.below(400);
                c.flag_size = (80u + R.below(700)) | (R.below(4) ? 0x80000000u : 0u);
                c.file = 0; cands.push_back(c);
            }
            continue;
        }
        if (i % 313 == 0 && !cands.empty()) { const Cand &o = cands[R.below((uint32_t)cands.size())]; q = o.q; h2 = o.h2; }   // a name that comes back much later
        if (i % 47 == 0 && !cands.empty()) { q = cands[R.below((uint32_t)cands.size())].q; h2 = (uint32_t)R.next() | 1u; }  // ANOTHER name with the same 64-bit hash (differs in the second hash)
        const int k = 1 + (int)R.below(R.below(8) == 0 ? 4 : 2);
        const int32_t iv = (int32_t)R.below(50);
        for (int j = 0; j < k; ++j) {
            Cand c;
            c.q = q; c.h2 = h2; c.name = R.below(5) == 0 ? iv + 1 : iv; c.endpos = 1000 + (int32_t)R.below(400);
            const uint32_t size = R.below(97) == 0 ? (1u << 20) + R.below(5000) * 1000u : 80u + R.below(700);
            c.flag_size = size | (R.below(4) ? 0x80000000u : 0u);
            c.file = 0; cands.push_back(c);
        }
    }
    if (hot > 0) {                                                    // one name with `hot` records (stripped / constant read names): all in one bucket
        const uint64_t q = R.next(); const uint32_t h2 = (uint32_t)R.next();
        for (int j = 0; j < hot; ++j) {
            Cand c; c.q = q; c.h2 = h2; c.name = (int32_t)R.below(3); c.endpos = 1000 + (int32_t)R.below(400);
            c.flag_size = (80u + R.below(700)) | (R.below(4) ? 0x80000000u : 0u); c.file = 0; cands.push_back(c);
        }
    }
    // file order = a shuffle of the candidates (mates of a name end up a random distance apart); indices are unique and sparse
    for (size_t i = cands.size(); i > 1; --i) std::swap(cands[i - 1], cands[R.below((uint32_t)i)]);
    for (auto &c : cands) { file += 1 + R.below(3); c.file = file; }
    // the literal walk, file order
    std::map<std::pair<uint64_t, uint32_t>, std::pair<int32_t, int32_t>> open_names;          // a name = (64-bit hash, second hash)
    std::vector<std::pair<uint64_t, uint32_t>> want_samples;
    for (const Cand &c : cands) {
        auto it = open_names.find({c.q, c.h2});
        if (it == open_names.end()) open_names[{c.q, c.h2}] = {c.name, c.endpos};
        else if (it->second.first == c.name) {
            if (!(c.flag_size >> 31) || c.endpos <= it->second.second) continue;
            want_samples.push_back({c.file, c.flag_size & 0x7FFFFFFFu});
            open_names.erase(it);
        }
    }
    const uint32_t want_keep = (uint32_t)std::min<size_t>(want_samples.size(), max_samples);
    std::sort(want_samples.begin(), want_samples.end());
    std::map<int64_t, uint64_t> want_hist;
    for (uint32_t i = 0; i < want_keep; ++i) want_hist[(int64_t)want_samples[i].second]++;
    // emission order differs from file order
    for (size_t i = cands.size(); i > 1; --i) std::swap(cands[i - 1], cands[R.below((uint32_t)i)]);
    const uint32_t n = (uint32_t)cands.size();
    std::vector<uint64_t> c_file(n), c_q(n); std::vector<int32_t> c_name(n), c_end(n); std::vector<uint32_t> c_fs(n), c_h2(n);
    for (uint32_t i = 0; i < n; ++i) { c_file[i] = cands[i].file; c_q[i] = cands[i].q; c_name[i] = cands[i].name; c_end[i] = cands[i].endpos; c_fs[i] = cands[i].flag_size; c_h2[i] = cands[i].h2; }
    FragCandidates fc{c_file.data(), c_q.data(), c_name.data(), c_end.data(), c_fs.data(), nullptr, n, nullptr, c_h2.data()};

    int error = 0;
    const uint32_t nb = std::max<uint32_t>(1u, n / PB_MEAN);
    std::vector<uint32_t> count(nb + 1, 0u), off(nb + 1, 0xDEADu), cursor(nb + 1, 0xDEADu), perm(n, 0xFFFFFFFFu), big_list(PB_BIG_MAX + 1, 0u), big_idx(2 * (size_t)n + 16, 0xDEADu);
    const uint32_t G = (n + 255) / 256;
    launch(G, 256, [&]() { pair_bucket_count_kernel(c_q.data(), n, nb, count.data()); });
    launch(1, 1024, [&]() { pair_bucket_scan_kernel(count.data(), nb, off.data(), cursor.data(), big_list.data(), &error); });
    launch(G, 256, [&]() { pair_bucket_scatter_kernel(c_q.data(), n, nb, cursor.data(), perm.data()); });
    if (error) return -error;
    if (off[nb] != n) return 1001;
    std::vector<uint64_t> s_file(n + 1), k_file(n + 1); std::vector<uint32_t> s_size(n + 1), k_size(n + 1);
    uint32_t ns = 0;
    launch(nb, PB_THREADS, [&]() { frag_replay_kernel(fc, off.data(), perm.data(), s_file.data(), s_size.data(), &ns); });
    launch(3, 1024, [&]() { frag_replay_big_kernel(fc, off.data(), perm.data(), big_list.data(), big_idx.data(), s_file.data(), s_size.data(), &ns); });
    if ((hot > (int)PB_CAP) != (big_list[0] > 0)) return 1007;
    if (ns != want_samples.size()) return 1002;
    const uint32_t keep = std::min(ns, max_samples);
    uint32_t n_kept = 0;
    {   // the select as rsqc_fragsize.hip runs it: planned and decided on the "device", no counters read back in between
        uint64_t state[2] = {~0ull, ~0ull};
        std::vector<uint32_t> h(256, 0u);
        launch(1, 64, [&]() { sample_plan_kernel(&ns, max_samples, state); });
        const uint32_t ns_bound = n / 2u + 1u;
        for (int shift = 56; shift >= 0; shift -= 8) {
            launch(std::min<uint32_t>(1024u, (ns_bound + 255u) / 256u), 256, [&]() { sample_digit_hist_kernel(s_file.data(), &ns, shift, state, h.data()); });
            launch(1, 256, [&]() { sample_digit_pick_kernel(h.data(), shift, state); });
        }
        launch((ns_bound + 1023) / 1024, 1024, [&]() { sample_keep_kernel(s_file.data(), s_size.data(), &ns, state, k_file.data(), k_size.data(), &n_kept); });
        if (n_kept != keep) return 1003;
    }
    {
        std::vector<std::pair<uint64_t, uint32_t>> got(n_kept);
        for (uint32_t i = 0; i < n_kept; ++i) got[i] = {k_file[i], k_size[i]};
        std::sort(got.begin(), got.end());
        for (uint32_t i = 0; i < n_kept; ++i) if (got[i] != want_samples[i]) return 1004;
    }
    std::vector<uint32_t> table(SIZE_TABLE, 0u), out_size(SIZE_TABLE), out_count(SIZE_TABLE), big(n + 1);
    uint32_t n_big = 0, n_out = 0xDEADu, top = 0;
    launch(std::min<uint32_t>(512u, (std::min(n / 2u + 1u, max_samples) + 255u) / 256u), 256, [&]() { size_hist_kernel(k_size.data(), &n_kept, table.data(), big.data(), &n_big, &top); });
    launch(1, 1024, [&]() { size_hist_compact_kernel(table.data(), &top, out_size.data(), out_count.data(), &n_out); });
    std::vector<std::pair<int64_t, uint64_t>> got_hist;
    for (uint32_t i = 0; i < n_out; ++i) got_hist.push_back({(int64_t)out_size[i], (uint64_t)out_count[i]});
    std::sort(big.begin(), big.begin() + n_big);
    for (uint32_t i = 0; i < n_big;) { uint32_t j = i + 1; while (j < n_big && big[j] == big[i]) ++j; got_hist.push_back({(int64_t)big[i], (uint64_t)(j - i)}); i = j; }
    if (got_hist.size() != want_hist.size()) return 1005;
    size_t at = 0;
    for (auto &kv : want_hist) { if (got_hist[at].first != kv.first || got_hist[at].second != kv.second) return 1006; ++at; }
    stats[0] = n; stats[1] = ns; stats[2] = n_kept; stats[3] = got_hist.size(); stats[4] = big_list[0]; stats[5] = g_k5_hashed_buckets; stats[6] = g_k5_sorted_buckets;
    return 0;
}
