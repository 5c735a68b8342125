// k1_emu.cpp -- TEST HARNESS ONLY (never loaded by the product).
//
// The per-record KERNELS themselves -- rnaseqc_amd/csrc/rsqc_k1.h (classify_ei_kernel), rsqc_k1s.h (classify_slow_kernel) and
// rsqc_kr.h (read_length_kernel) with the wave helpers of rsqc_wave.h, unmodified -- compiled for the host on top of the
// 64-lane fiber emulation of wavemu.h: per-wave LDS queues, ballot / mbcnt compaction, workgroup LDS tables, pair chunks, the
// overflow list and the general kernel that takes it, the Read-Length transfer function.  hostemu.cpp covers the per-record
// functions one record at a time; this covers what the wavefronts do with them.  `slow_kernel` = 0 sends the overflow list
// through the general per-record code on the host instead of classify_slow_kernel (the two must agree).
#include "wavemu.h"

#include <map>
#include <set>
#include <string>
#include <vector>

#include "../../rnaseqc_amd/csrc/rsqc_read.h"
#include "../../rnaseqc_amd/csrc/rsqc_device.h"
#include "../../rnaseqc_amd/csrc/rsqc_index.h"
#include "../../rnaseqc_amd/csrc/rsqc_wave.h"
#include "../../rnaseqc_amd/csrc/rsqc_k1.h"
#include "../../rnaseqc_amd/csrc/rsqc_k1s.h"
#include "../../rnaseqc_amd/csrc/rsqc_kr.h"
#define RSQC_FIN_STAMP(sec)
#define RSQC_FIN_SECT(base, sec)
#define RSQC_FIN_BEGIN
#include "../../rnaseqc_amd/csrc/rsqc_k4.h"

using namespace rsqc;

namespace {
struct SlowAccH {
    std::vector<uint64_t> *reads, *unique;
    std::vector<double> *exon_rows;
    std::vector<uint32_t> *cov;
    std::vector<std::set<std::pair<uint64_t, uint32_t>>> *names;          // a name = (64-bit hash, second hash)
    void gene_hit(uint32_t g, bool nd, uint64_t qh, uint32_t qh2) { (*reads)[g]++; if (nd) (*unique)[g]++; (*names)[g].insert({qh, qh2}); }
    void exon_add(uint32_t row, double f) { (*exon_rows)[row] += f; }
    void cov_range(uint32_t cidx, uint32_t len) { if (!len) return; (*cov)[cidx] += 1u; (*cov)[cidx + len] -= 1u; }
};
}  // namespace

extern "C" __attribute__((visibility("default"))) unsigned long long k1emu_uniform_calls() { return g_k1e_uniform_calls; }
extern "C" __attribute__((visibility("default"))) unsigned long long k1emu_ucache_hits() { return g_k1e_ucache_hits; }
extern "C" __attribute__((visibility("default"))) unsigned long long k1emu_uniform2_calls() { return g_k1e_uniform2_calls; }

// returns 0, an RSQC_ERR_* code, or 1000 + k for a failed internal check k
extern "C" __attribute__((visibility("default")))
int k1emu_run_bed(const rsqc_params *p, const rsqc_annotation *a, const rsqc_batch *b, const rsqc_bed *bed, int grid, int slow_kernel,
                  uint64_t *counters, uint64_t *gene_reads, uint64_t *gene_unique, uint64_t *gene_frag, double *exon_reads,
                  int32_t *read_length, uint32_t *cov_out, uint64_t *stats);
extern "C" __attribute__((visibility("default")))
int k1emu_run(const rsqc_params *p, const rsqc_annotation *a, const rsqc_batch *b, int grid, int slow_kernel,
              uint64_t *counters, uint64_t *gene_reads, uint64_t *gene_unique, uint64_t *gene_frag, double *exon_reads /*by exon id*/,
              int32_t *read_length, uint32_t *cov_out /*cov_entries or NULL*/, uint64_t *stats /*[4]: overflow, listed | deferred << 32, pairs, coarse-table hits*/) {
    return k1emu_run_bed(p, a, b, nullptr, grid, slow_kernel, counters, gene_reads, gene_unique, gene_frag, exon_reads, read_length, cov_out, stats);
}
// `bed` (may be NULL): with it the --bed instance classify_ei_kernel<true> runs, and the fragment-size candidates it leaves in the workgroups'
// regions must be EXACTLY the records that pass src/RNASeQC.cpp:372 and the block tests of bed_interval_of, computed here record by record
// without the kernel's wave-level cursor shortcut (that path had no emulation coverage); stats[3] then = candidates
extern "C" __attribute__((visibility("default")))
int k1emu_run_bed(const rsqc_params *p, const rsqc_annotation *a, const rsqc_batch *b, const rsqc_bed *bed, int grid, int slow_kernel,
              uint64_t *counters, uint64_t *gene_reads, uint64_t *gene_unique, uint64_t *gene_frag, double *exon_reads /*by exon id*/,
              int32_t *read_length, uint32_t *cov_out /*cov_entries or NULL*/, uint64_t *stats /*[4]: overflow, listed | deferred << 32, pairs, coarse-table hits*/) {
    HostIndex hx; std::string err;
    int rc = hx.build(a, nullptr, err);
    if (rc) return rc;
    DevAnnotation d{};
    d.n_ref = a->n_ref; d.n_contigs = a->n_contigs; d.n_genes = a->n_genes; d.n_listed = a->n_genes_listed; d.n_exons = a->n_exons;
    d.bin_shift = HostIndex::kBinShift;
    d.contig = hx.contig.data();
    if (hx.ex_rows.empty()) hx.ex_rows.push_back(ExonRow{0, 0, 0, 0});
    if (hx.gb.empty()) hx.gb.push_back(GeneBreak{0, 0});
    if (hx.ex_pmax.empty()) hx.ex_pmax.push_back(0);
    d.ex = hx.ex_rows.data(); d.gb = hx.gb.data(); d.ex_pmax = hx.ex_pmax.data();
    d.ex_binhi = hx.ex_binhi.data(); d.gb_bin = hx.gb_bin.data(); d.ex_cov = hx.ex_cov.data();
    std::vector<EiRank> rank; hx.build_rank(rank);
    d.ei = hx.ei.data(); d.ei_rank = rank.data(); d.ei_coarse = hx.ei_coarse.data();
    std::vector<uint32_t> ex_id(a->exon_row_id, a->exon_row_id + a->n_exons); if (ex_id.empty()) ex_id.push_back(0);
    d.ex_id = ex_id.data();
    std::vector<uint32_t> zero_range((size_t)a->n_contigs + 1, 0);
    d.bed_range = zero_range.data(); d.have_bed = 0;
    std::vector<int32_t> bed_pmax;
    if (bed) {                                                             // as rsqc_set_bed builds them (no bin table: bed_interval_of searches)
        for (int i = 0; i < bed->n_intervals; ++i) {
            if (bed->contig[i] < 0 || bed->contig[i] >= a->n_contigs) return RSQC_ERR_ARG;
            zero_range[(size_t)bed->contig[i] + 1]++;
        }
        for (int k = 0; k < a->n_contigs; ++k) zero_range[(size_t)k + 1] += zero_range[(size_t)k];
        bed_pmax.resize((size_t)std::max(bed->n_intervals, 1));
        for (int k = 0; k < a->n_contigs; ++k) {
            int32_t m = INT32_MIN;
            for (uint32_t i = zero_range[(size_t)k]; i < zero_range[(size_t)k + 1]; ++i) { m = std::max(m, bed->end[i]); bed_pmax[i] = m; }
        }
        d.bed_start = bed->start; d.bed_end = bed->end; d.bed_pmax = bed_pmax.data(); d.bed_binhi = nullptr; d.bed_bin_base = nullptr; d.have_bed = 1;
    }
    DevParams dp{p->mapq_threshold, p->base_mismatch, p->chimeric_distance, p->stranded, p->unpaired, p->exclude_chimeric, p->n_filter_tags, 0};

    // the batch with the slack the device buffers carry (kernels read a few entries past the end with ignored loads)
    const uint64_t n = b->n;
    std::vector<rsqc_rec_core> core((size_t)n + 2); std::vector<rsqc_rec_aux> aux((size_t)n + 2);
    std::vector<uint32_t> cigar((size_t)b->n_cigar_total + 8, 0u);
    if (n) { memcpy(core.data(), b->core, (size_t)n * sizeof(rsqc_rec_core)); memcpy(aux.data(), b->aux, (size_t)n * sizeof(rsqc_rec_aux)); }
    if (b->n_cigar_total) memcpy(cigar.data(), b->cigar, (size_t)b->n_cigar_total * 4);
    std::vector<uint32_t> qh2((size_t)n + 2, 0u);
    if (n && b->qhash2) memcpy(qh2.data(), b->qhash2, (size_t)n * 4);
    DevBatch db{};
    db.qhash2 = b->qhash2 ? qh2.data() : nullptr;
    db.n = n; db.record_base = b->file_index_base; db.core = core.data(); db.aux = aux.data(); db.cigar = cigar.data();
    db.n_seg = b->n_seg; db.seg_tid = b->seg_tid; db.seg_start = b->seg_start;
    db.n_wide = b->n_wide; db.wide_index = b->wide_index; db.wide_nm = b->wide_nm; db.wide_l_qseq = b->wide_l_qseq; db.wide_n_cigar = b->wide_n_cigar;

    const size_t G = (size_t)a->n_genes, E = (size_t)a->n_exons;
    std::vector<unsigned long long> u64(3 * G + RSQC_N_COUNTERS + 1, 0ull);
    std::vector<double> exon_acc(E + 1, 0.0);
    std::vector<uint32_t> cov((size_t)hx.cov_entries + 64, 0u);
    const uint64_t total_waves = (uint64_t)grid * K1E_WAVES;
    const uint64_t per_wave = (((n + total_waves - 1) / total_waves) + 63ull) & ~63ull;
    const uint32_t chunk_cap = (uint32_t)(per_wave * K1E_WAVES * FAST_SET);
    const uint32_t slow_cap = 1u << 16;
    const int lgrid = grid > 2 ? grid - 1 : grid;                          // workgroups of classify_long_kernel: each owns a chunk behind the K1 grid's
    const int n_chunks = grid + lgrid;
    std::vector<PairRec> pairs((size_t)chunk_cap * n_chunks + slow_cap + 8, PairRec{0xFFFFFFF0u, 0xFEEDu, 0ull});
    std::vector<uint32_t> chunk_count((size_t)n_chunks + 2, 0xDEADu);
    std::vector<uint32_t> ovf_count(4, 0u); std::vector<uint64_t> ovf_index(1u << 20);
    std::vector<uint32_t> defer_index((size_t)n + 64, 0xDEFE0000u), defer_list((size_t)n + 64 * (size_t)grid + 64, 0xDEFE0001u); uint32_t defer_total = 0;
    std::vector<uint32_t> tile_span((size_t)((n + 63) / 64) + 64 * (size_t)total_waves + 64, 0xABCDu);
    std::vector<uint32_t> rl_stats = {0u, 0xFFFFFFFFu, 0u};
    int32_t rl_state = 0; int error = 0;
    DevAccum acc{};
    acc.gene_reads = u64.data(); acc.gene_unique = acc.gene_reads + G; acc.gene_frag = acc.gene_unique + G; acc.counters = acc.gene_frag + G;
    acc.exon_acc = exon_acc.data(); acc.cov_diff = cov.data();
    acc.pairs = pairs.data();
    acc.pair_chunk_cap = chunk_cap; acc.pair_chunk_count = chunk_count.data();
    acc.pair_slow_base = chunk_cap * (uint32_t)n_chunks; acc.pair_slow_cap = slow_cap; acc.pair_slow_count = chunk_count.data() + n_chunks;
    acc.ovf_count = ovf_count.data(); acc.ovf_index = (uint64_t *)ovf_index.data(); acc.ovf_cap = (uint32_t)ovf_index.size();
    acc.defer_index = defer_index.data(); acc.defer_list = defer_list.data(); acc.defer_total = &defer_total;
    // --bed: the workgroups' candidate regions (a slot per record) and their counts
    std::vector<uint64_t> fr_file((size_t)n + 64), fr_qhash((size_t)n + 64);
    std::vector<int32_t> fr_name((size_t)n + 64), fr_end((size_t)n + 64);
    std::vector<uint32_t> fr_fs((size_t)n + 64), fr_h2((size_t)n + 64), fr_counts((size_t)grid + 1, 0xDEADu);
    acc.frag.file_index = fr_file.data(); acc.frag.qhash = fr_qhash.data(); acc.frag.name = fr_name.data(); acc.frag.endpos = fr_end.data();
    acc.frag.flag_size = fr_fs.data(); acc.frag.h2 = fr_h2.data(); acc.frag.chunk_count = fr_counts.data(); acc.frag.count = nullptr; acc.frag.cap = (uint32_t)n;
    acc.tile_span = tile_span.data();
    acc.rl_stats = rl_stats.data(); acc.read_length = &rl_state; acc.error = &error;

    const K1Args A{d, dp, db, acc};
    g_k1e_args = &A; g_k1e_coarse_hits = 0; g_k1e_uniform_calls = 0; g_k1e_ucache_hits = 0; g_k1e_uniform2_calls = 0;
    wavemu::grid_dim().x = (uint32_t)grid;
    for (int k = 0; k < grid; ++k) {
        wavemu::block_idx().x = (uint32_t)k;
        if (bed) wavemu::run_block(RSQC_K1_THREADS, [&]() { classify_ei_kernel<true>(A); });
        else wavemu::run_block(RSQC_K1_THREADS, [&]() { classify_ei_kernel<false>(A); });      // (the instance of runs without a BED)
    }
    uint64_t n_deferred = 0;
    for (uint32_t k = 0; k < defer_total; ++k) { ++n_deferred; if ((defer_list[k] & 0x7FFFFFFFu) >= n) return 1013; }
    {   // the records it deferred (more than eight operations / three blocks, the three-block ring's surplus) -> classify_long_kernel
        wavemu::grid_dim().x = (uint32_t)lgrid;
        for (int k = 0; k < lgrid; ++k) {
            wavemu::block_idx().x = (uint32_t)k;
            wavemu::run_block(RSQC_K1_THREADS, [&]() { classify_long_kernel(A, (uint32_t)grid); });
        }
        wavemu::grid_dim().x = (uint32_t)grid;
    }
    if (getenv("K1EMU_TRACE")) fprintf(stderr, "k1emu: per-record kernels done, error %d\n", error);
    uint64_t listed = 0;
    if (error) return error;

    // ---- the overflow list through the general per-record code ------------------------------------------------------
    std::vector<uint64_t> reads(G, 0), unique(G, 0);
    std::vector<double> exon_rows(E, 0.0);
    std::vector<std::set<std::pair<uint64_t, uint32_t>>> names(G);
    SlowAccH sacc{&reads, &unique, &exon_rows, &cov, &names};
    std::vector<int32_t> tid_of((size_t)n, -1);
    for (uint32_t s = 0; s < b->n_seg; ++s) for (uint64_t i = b->seg_start[s]; i < b->seg_start[s + 1]; ++i) tid_of[(size_t)i] = b->seg_tid[s];
    auto load = [&](uint64_t i, Record &r) -> bool {
        const rsqc_rec_core &co = b->core[i]; const rsqc_rec_aux &au = b->aux[i];
        r.tid = tid_of[(size_t)i]; r.pos = co.pos; r.mpos = co.mpos; r.isize = co.isize; r.flag = au.flag;
        r.mapq = au.mapq; r.tagbits = au.tagbits; r.l_qseq = au.l_qseq; r.nm = au.nm; r.n_cigar = au.n_cigar;
        if (au.l_qseq == RSQC_LQSEQ_ESCAPE || au.nm == RSQC_NM_ESCAPE || au.n_cigar == RSQC_NCIGAR_ESCAPE) {
            uint32_t w = 0;
            while (w < b->n_wide && b->wide_index[w] < i) ++w;
            if (w >= b->n_wide || b->wide_index[w] != i) return false;
            r.l_qseq = b->wide_l_qseq[w]; r.nm = b->wide_nm[w]; r.n_cigar = b->wide_n_cigar[w];
        }
        r.cigar = cigar.data() + co.cigar_off; r.qhash = au.qhash;
        return true;
    };
    std::set<uint64_t> seen_ovf;
    const uint32_t n_overflow = ovf_count[0];                  // (read_length_kernel, the batch's last kernel, zeroes the counter)
    if (slow_kernel) {
        // the general KERNEL on the list, sized as launch_classify_slow sizes it; it adds to the same accumulators and appends
        // its (gene, name) pairs to the slow-path region of the pair buffer
        for (uint32_t k = 0; k < ovf_count[0]; ++k) {
            const uint64_t i = ovf_index[k] & ~(1ull << 63);
            if (i >= n || !seen_ovf.insert(i).second) return 1002;
            if (ovf_index[k] >> 63) ++listed;
        }
        const uint32_t blocks = (uint32_t)std::max<uint64_t>(2, std::min<uint64_t>(8, n / 200 / RSQC_SLOW_THREADS + 1));
        wavemu::grid_dim().x = blocks;
        for (uint32_t k = 0; k < blocks; ++k) {
            wavemu::block_idx().x = k;
            wavemu::run_block(RSQC_SLOW_THREADS, [&]() { classify_slow_kernel<false>(d, dp, db, acc); });
        }
        if (error) return error;
        const uint32_t ns = chunk_count[(size_t)n_chunks];
        if (ns > slow_cap) return 1008;
        for (uint32_t j = 0; j < ns; ++j) {
            const PairRec &pr = pairs[(size_t)acc.pair_slow_base + j];
            if (pr.gene >= G) return 1005;
            names[pr.gene].insert({(uint64_t)pr.hash, pr.h2});
        }
    } else
    for (uint32_t k = 0; k < ovf_count[0]; ++k) {
        const uint64_t i = ovf_index[k] & ~(1ull << 63);
        const bool long_straggler = (ovf_index[k] >> 63) != 0;            // K1E_OVF_LONG: blocks and operations are this code's to count / check
        if (i >= n || !seen_ovf.insert(i).second) return 1002;            // every record at most once
        Record r;
        if (!load(i, r)) return RSQC_ERR_ARG;
        RecordCounters rc2; bool hq; uint32_t aligned; Blocks B;
        const bool go2 = gate_cascade(d, dp, r, rc2, hq, aligned, B);
        if (long_straggler) { if (rc2.error) return rc2.error; acc.counters[RSQC_C_ALIGNMENT_BLOCKS] += rc2.blocks; ++listed; }
        if (!go2) return 1003;                                            // only records that reach the feature stage are listed
        bool over = false;
        FeatureOut<SLOW_SET, SLOW_STAGE> so;
        exon_metrics<SLOW_SET>(d, dp, r, hq, aligned, sacc, so, over);
        if (over) return RSQC_ERR_CAPACITY;
        for (int k2 = 0; k2 < SLOW_STAGE; ++k2) {
            if (!((so.cmask >> k2) & 1u)) continue;
            const Commit &c = so.commit[k2];
            if (c.len > 0) sacc.exon_add(c.row, (double)c.len / (double)aligned);
            sacc.cov_range(c.cidx, c.len);
        }
        for (int k2 = 0; k2 < so.n_hit; ++k2) sacc.gene_hit(so.hit[k2], !(r.flag & RSQC_FDUP), r.qhash, b->qhash2 ? b->qhash2[i] : 0u);
        for (int c = 0; c < RSQC_N_COUNTERS; ++c) if ((so.bits >> c) & 1ull) acc.counters[c]++;
    }
    // ---- pairs -> distinct names per gene ------------------------------------------------------------------------------
    uint64_t n_pairs = 0;
    for (int k = 0; k < n_chunks; ++k) {
        const uint32_t cnt = chunk_count[(size_t)k];
        if (cnt > chunk_cap) return 1004;
        for (uint32_t j = 0; j < cnt; ++j) {
            const PairRec &pr = pairs[(size_t)k * chunk_cap + j];
            if (pr.gene >= G) return 1005;
            names[pr.gene].insert({(uint64_t)pr.hash, pr.h2}); ++n_pairs;
        }
    }
    // ---- Read-Length inputs: tile maxima and batch extremes against the per-record values; the state machine itself ---
    uint32_t rl = 0, smax = 0, lmin = 0xFFFFFFFFu, lmax = 0;
    {
        std::vector<uint32_t> want((size_t)((n + 63) / 64), 0u);
        for (uint64_t i = 0; i < n; ++i) {
            Record r;
            if (!load(i, r)) return RSQC_ERR_ARG;
            RecordCounters rc2; bool hq; uint32_t aligned; Blocks B;
            gate_cascade(d, dp, r, rc2, hq, aligned, B);
            if (rc2.error) return rc2.error;
            if (rc2.rl_eligible) {
                want[(size_t)(i >> 6)] = std::max(want[(size_t)(i >> 6)], rc2.rl_span);
                smax = std::max(smax, rc2.rl_span); lmin = std::min(lmin, (uint32_t)rc2.rl_lqseq); lmax = std::max(lmax, (uint32_t)rc2.rl_lqseq);
                if (rc2.rl_span > rl) rl = (uint32_t)rc2.rl_lqseq;
            }
        }
        // per-wave ranges are multiples of 64 records, so a wave's tile t is batch tile (wbeg / 64 + t)
        for (size_t t = 0; t < want.size(); ++t) if (tile_span[t] != want[t]) return 1006;
        if (rl_stats[0] != smax || rl_stats[1] != lmin || rl_stats[2] != lmax) return 1007;
        // the Read-Length KERNEL: the batch's transfer function applied to state 0 must leave the value of the walk above
        std::vector<uint32_t> summary(RSQC_RL_SUMMARY_WORDS + 8, 0u);
        wavemu::grid_dim().x = 1; wavemu::block_idx().x = 0;
        wavemu::run_block(64, [&]() { read_length_kernel(d, dp, db, acc, summary.data()); });
        if (error) return error;
        if ((uint32_t)rl_state != rl) return 1009;
    }
    // ---- the fragment-counting KERNELS (rsqc_k4.h) on the pair buffers as the kernels above left them: must equal the name sets
    if (slow_kernel && G > 0) {
        const uint64_t parts_bound = n_pairs / RSQC_K4_PART_READS + chunk_count[(size_t)n_chunks] / RSQC_K4_PART_READS + G + 2;
        const uint64_t all_pairs = n_pairs + chunk_count[(size_t)n_chunks];
        const uint64_t keys_bound = 2 * all_pairs + (uint64_t)RSQC_K4_SUB_CAP * std::min<uint64_t>(parts_bound, all_pairs / RSQC_K4_PART_READS + 1) + 16 * G + 16;
        const uint32_t lay_blocks = (uint32_t)((G + 1023) / 1024);
        std::vector<uint4> ginfo(G + 1), part_info(parts_bound);
            std::vector<uint32_t> part_first(G + 2), cursor(parts_bound, 0u), full_list(parts_bound), blk_parts(lay_blocks);
        std::vector<unsigned long long> blk_space(lay_blocks), frag(G, 0ull); std::vector<FragKey> list(keys_bound);
        uint32_t full_n = 0;
        wavemu::grid_dim().x = lay_blocks;
        for (uint32_t k = 0; k < lay_blocks; ++k) { wavemu::block_idx().x = k; wavemu::run_block(1024, [&]() { frag_layout_totals_kernel(acc.gene_reads, (uint32_t)G, blk_space.data(), blk_parts.data(), &error); }); }
        for (uint32_t k = 0; k < lay_blocks; ++k) { wavemu::block_idx().x = k; wavemu::run_block(1024, [&]() { frag_layout_kernel(acc.gene_reads, (uint32_t)G, blk_space.data(), blk_parts.data(), part_first.data(), ginfo.data(), cursor.data(), part_info.data(), &full_n); }); }
        if (error) return error;
        if (part_first[G] > parts_bound) return 1010;
        const uint32_t fgrid = frag_local_chunk_wgs((uint32_t)n_chunks) + 4u;
        wavemu::grid_dim().x = fgrid;
        for (uint32_t k = 0; k < fgrid; ++k) {
            wavemu::block_idx().x = k;
            wavemu::run_block(RSQC_K4L_THREADS, [&]() { frag_local_kernel(pairs.data(), chunk_cap, chunk_count.data(), (uint32_t)n_chunks, acc.pair_slow_base, slow_cap,
                                                                          ginfo.data(), cursor.data(), list.data(), &error); });
        }
        wavemu::grid_dim().x = 8;
        for (uint32_t k = 0; k < 8; ++k) { wavemu::block_idx().x = k; wavemu::run_block(RSQC_K4_COUNT_THREADS, [&]() { frag_count_kernel<RSQC_K4_PART_SLOTS / 2>(part_first.data() + G, cursor.data(), part_info.data(), list.data(), frag.data(), full_list.data(), &full_n, &error); }); }
        wavemu::grid_dim().x = 2;
        for (uint32_t k = 0; k < 2; ++k) { wavemu::block_idx().x = k; wavemu::run_block(RSQC_K4_COUNT_THREADS, [&]() { frag_count_kernel<RSQC_K4_PART_SLOTS>(part_first.data() + G, cursor.data(), part_info.data(), list.data(), frag.data(), full_list.data(), &full_n, &error); }); }
        if (error) return error;
        for (size_t g = 0; g < G; ++g) {
            // (a key of 0 is stored as a fixed non-zero constant by frag_local_kernel: a name hashing to 0 and one hashing to that
            //  constant would merge -- neither occurs in the tests' names)
            if (frag[g] != (unsigned long long)names[g].size()) return 1011;
        }
    }
    for (size_t g = 0; g < (size_t)a->n_genes_listed; ++g) {
        gene_reads[g] = acc.gene_reads[g] + reads[g]; gene_unique[g] = acc.gene_unique[g] + unique[g]; gene_frag[g] = names[g].size();
    }
    for (int c = 0; c < RSQC_N_COUNTERS; ++c) counters[c] = acc.counters[c];
    for (int e = 0; e < a->n_exons; ++e) exon_reads[a->exon_row_id[e]] = exon_rows[(size_t)e] + exon_acc[a->exon_row_id[e]];
    *read_length = (int32_t)rl;
    if (cov_out) memcpy(cov_out, cov.data(), hx.cov_entries * 4);
    uint64_t n_cand = 0;
    if (getenv("K1EMU_TRACE")) fprintf(stderr, "k1emu: reached the candidate check\n");
    if (bed) {                                                             // the kernel's candidates against the per-record rule, no shortcut
        std::map<uint64_t, size_t> got;                                    // record index -> slot
        for (int k = 0; k < grid; ++k) {
            uint32_t beg, end;
            k1e_wg_range((uint32_t)n, (uint32_t)grid, (uint32_t)k, beg, end);
            if (fr_counts[(size_t)k] > end - beg) return 1020;
            for (uint32_t j = 0; j < fr_counts[(size_t)k]; ++j) {
                const uint64_t idx = fr_file[(size_t)beg + j] - b->file_index_base;
                if (idx >= n || idx < beg || idx >= end || !got.emplace(idx, (size_t)beg + j).second) return 1021;
            }
        }
        for (uint64_t i = 0; i < n; ++i) {
            Record r;
            if (!load(i, r)) return RSQC_ERR_ARG;
            RecordCounters rc2; bool hq; uint32_t aligned; Blocks B;
            gate_cascade(d, dp, r, rc2, hq, aligned, B);
            const int32_t name = (rc2.frag_candidate && r.tid >= 0 && r.tid < a->n_contigs) ? bed_interval_of(d, r) : -1;
            const auto it = got.find(i);
            if ((name >= 0) != (it != got.end())) return name >= 0 ? 1022 : 1023;      // 1022: the kernel skipped a candidate (the cursor shortcut)
            if (name < 0) continue;
            ++n_cand;
            const size_t sl = it->second;
            const bool fok = !(r.flag & RSQC_FMREVERSE) && (r.flag & RSQC_FREVERSE) && r.pos != r.mpos;
            const uint32_t sz = (uint32_t)(r.isize < 0 ? -(int64_t)r.isize : (int64_t)r.isize);
            if (fr_name[sl] != name || fr_end[sl] != rc2.endpos || fr_qhash[sl] != r.qhash || fr_fs[sl] != ((sz & 0x7FFFFFFFu) | (fok ? 0x80000000u : 0u)) ||
                fr_h2[sl] != (b->qhash2 ? b->qhash2[i] : 0u)) return 1024;
        }
    }
    stats[0] = n_overflow; stats[1] = listed | (n_deferred << 32); stats[2] = n_pairs; stats[3] = bed ? n_cand : g_k1e_coarse_hits;
    return 0;
}
