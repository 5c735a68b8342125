// rsqc_index.h -- host-side construction of the device annotation index from the
// boundary struct (pure C++, no HIP): row ranges per contig, prefix-max-of-end columns,
// coarse position bins, and the per-base coverage layout.
#pragma once

#include <algorithm>
#include <cstdint>
#include <string>
#include <vector>

#include "../../include/rnaseqc_amd.h"
#include "rsqc_read.h"

namespace rsqc {

struct HostIndex {
    static constexpr int kBinShift = 11;
    int32_t n_ref = 0, n_contigs = 0, n_genes = 0, n_listed = 0, n_exons = 0;
    std::vector<uint32_t> ex_range, g_range, ex_binhi, gb_bin, ex_cov, gene_cov_off, gene_coding;
    std::vector<GeneBreak> gb;
    std::vector<int32_t> ex_pmax, g_pmax;
    std::vector<ExonRow> ex_rows;
    std::vector<EiEntry> ei;                          // elementary intervals (rsqc_read.h)
    std::vector<uint32_t> ei_range;                   // [n_contigs + 1]
    uint64_t rank_words = 0;                          // words of the rank table over all contigs
    std::string warning;                              // build(): what the annotation has that the reference only tolerates (empty: nothing)
    uint32_t n_exons_outside_gene = 0;                // ... exon rows outside the row of their gene (rsqc_results.exons_outside_gene_row)
    // Coarse table over the same positions, one word per 512 (= 8 rank words; every contig's part of the rank table starts at a
    // multiple of 8 words, so the two tables share ContigInfo::rk_base): 0, or 1 + the index of the interval that covers ALL of
    // the 1024 positions [b * 512, (b + 2) * 512) -- no breakpoint in there.  A read in the empty stretches of the genome
    // (intergenic, deep intronic: every sixth record of an RNA-seq file) then needs no rank word at all; each of them used to
    // pull a 64-byte sector of the 775 MB rank table from HBM for one look-up (2.5 GB per 102 M records, profiles/k1_traffic.json
    // of round 3), and every tile waited for the slowest of those misses.  24 MB for the human contig lengths.
    std::vector<uint32_t> ei_coarse;
    std::vector<GeneRow> gr_rows;                     // --legacy tables (LegacyTables)
    std::vector<uint32_t> ex_ord;
    std::vector<uint32_t> gr_binhi;                   // ... per bin of the exon bin table: first gene row with start >= (b + 1) << shift
    std::vector<ContigInfo> contig;
    std::vector<uint8_t> gene_flags, gene_owned;     // by listed gene id
    uint64_t cov_entries = 0;

    // returns 0 or an RSQC_ERR_* code with `err` set
    int build(const rsqc_annotation *a, const uint8_t *owned_contig, std::string &err) {
        const int nc = a->n_contigs, G = a->n_genes, L = a->n_genes_listed, E = a->n_exons;
        if (nc < a->n_ref || L > G || G < 0 || E < 0 || L < 0) { err = "inconsistent annotation sizes"; return RSQC_ERR_ARG; }
        n_ref = a->n_ref; n_contigs = nc; n_genes = G; n_listed = L; n_exons = E;
        auto build_rows = [&](int n, const int32_t *contig, const int32_t *start, const int32_t *end,
                              std::vector<uint32_t> &range, std::vector<int32_t> &pmax) -> bool {
            range.assign((size_t)nc + 1, 0);
            pmax.resize((size_t)n);
            for (int i = 0; i < n; ++i) {
                if (contig[i] < 0 || contig[i] >= nc || end[i] < start[i]) return false;
                if (i && (contig[i] < contig[i - 1] || (contig[i] == contig[i - 1] && start[i] < start[i - 1]))) return false;
                range[(size_t)contig[i] + 1]++;
            }
            for (int k = 0; k < nc; ++k) range[(size_t)k + 1] += range[(size_t)k];
            for (int k = 0; k < nc; ++k) {
                int32_t m = INT32_MIN;
                for (uint32_t i = range[(size_t)k]; i < range[(size_t)k + 1]; ++i) { m = std::max(m, end[i]); pmax[i] = m; }
            }
            return true;
        };
        if (!build_rows(E, a->exon_row_contig, a->exon_row_start, a->exon_row_end, ex_range, ex_pmax) ||
            !build_rows(L, a->gene_row_contig, a->gene_row_start, a->gene_row_end, g_range, g_pmax)) {
            err = "annotation rows must be sorted by (contig, start) with start <= end";
            return RSQC_ERR_ARG;
        }
        if (G > (int)ROW_GENE_MASK) { err = "more than 2^26 genes"; return RSQC_ERR_CAPACITY; }
        if (E >= (1 << 27)) { err = "more than 2^27 exons"; return RSQC_ERR_CAPACITY; }        // 16-byte rows, 32-bit offsets
        for (int i = 0; i < E; ++i)
            if (a->exon_row_gene[i] >= (uint32_t)G) { err = "exon_row_gene out of range"; return RSQC_ERR_ARG; }
        // gene breakpoints per contig: sweep over gene starts (+) and ends+1 (-) with per-class counters
        gb.clear();
        std::vector<uint32_t> gb_range((size_t)nc + 1, 0);
        for (int k = 0; k < nc; ++k) {
            std::vector<std::pair<int64_t, int>> ev;     // (position, +/-(1 + class + 3*ribo))
            for (uint32_t i = g_range[(size_t)k]; i < g_range[(size_t)k + 1]; ++i) {
                const int cls = a->gene_row_flags[i] & RSQC_FF_STRAND_MASK, ribo = (a->gene_row_flags[i] & RSQC_FF_RIBOSOMAL) ? 1 : 0;
                if (cls > 2) { err = "bad strand class"; return RSQC_ERR_ARG; }
                const int code = 1 + cls + 3 * ribo;
                ev.emplace_back((int64_t)a->gene_row_start[i], code);
                ev.emplace_back((int64_t)a->gene_row_end[i] + 1, -code);
            }
            std::sort(ev.begin(), ev.end());
            int cnt[6] = {0, 0, 0, 0, 0, 0};
            for (size_t e = 0; e < ev.size();) {
                const int64_t pos = ev[e].first;
                while (e < ev.size() && ev[e].first == pos) { const int c2 = ev[e].second; if (c2 > 0) cnt[c2 - 1]++; else cnt[-c2 - 1]--; ++e; }
                uint32_t mask = 0;
                for (int cls = 0; cls < 3; ++cls) {
                    if (cnt[cls] + cnt[3 + cls] > 0) mask |= 1u << cls;
                    if (cnt[3 + cls] > 0) mask |= 1u << (3 + cls);
                }
                if (pos > 0x7FFFFFFFll) break;
                if (gb.size() > gb_range[(size_t)k] && gb.back().mask == mask) continue;   // no change
                gb.push_back(GeneBreak{(int32_t)pos, mask});
            }
            gb_range[(size_t)k + 1] = (uint32_t)gb.size();
        }
        // per-contig info + bin tables
        contig.assign((size_t)nc, ContigInfo{0, 0, 0, 0, 0, 0, 0, 0});
        uint64_t total_bins = 0;
        for (int k = 0; k < nc; ++k) {
            int64_t ms = -1;
            if (ex_range[(size_t)k + 1] > ex_range[(size_t)k]) ms = std::max<int64_t>(ms, a->exon_row_start[ex_range[(size_t)k + 1] - 1]);
            if (gb_range[(size_t)k + 1] > gb_range[(size_t)k]) ms = std::max<int64_t>(ms, gb[gb_range[(size_t)k + 1] - 1].pos);
            const uint64_t nb = ms < 0 ? 0 : (uint64_t)(ms >> kBinShift) + 1;
            if (total_bins + nb > 0x3FFFFFF0ull) { err = "bin table too large"; return RSQC_ERR_CAPACITY; }
            contig[(size_t)k] = ContigInfo{ex_range[(size_t)k], ex_range[(size_t)k + 1], gb_range[(size_t)k], gb_range[(size_t)k + 1],
                                           (uint32_t)total_bins, (uint32_t)nb, 0, 0};
            total_bins += nb;
        }
        ex_binhi.assign((size_t)total_bins + 1, 0); gb_bin.assign((size_t)total_bins + 1, 0);
        for (int k = 0; k < nc; ++k) {
            const ContigInfo &ci = contig[(size_t)k];
            uint32_t row = ci.ex_lo, bp = ci.gb_lo;
            for (uint32_t b = 0; b < ci.n_bins; ++b) {
                const int64_t lim = ((int64_t)b + 1) << kBinShift, lo_pos = (int64_t)b << kBinShift;
                while (row < ci.ex_hi && (int64_t)a->exon_row_start[row] < lim) ++row;
                ex_binhi[(size_t)ci.bin_base + b] = row;                 // first exon row with start >= (b+1) << shift
                while (bp < ci.gb_hi && (int64_t)gb[bp].pos <= lo_pos) ++bp;
                gb_bin[(size_t)ci.bin_base + b] = bp;                    // first breakpoint with pos > b << shift
            }
        }
        // per-base coverage layout: exons of a gene contiguous, in exonsForGene order, + 1 pad slot per gene
        ex_cov.assign((size_t)E, 0);
        gene_cov_off.assign((size_t)std::max(L, 1), 0);
        gene_coding.assign((size_t)std::max(L, 1), 0);
        std::vector<uint8_t> seen((size_t)E, 0);
        uint64_t run = 0;
        for (int g = 0; g < G; ++g) {
            if (g < L) gene_cov_off[(size_t)g] = (uint32_t)run;
            uint64_t coding = 0;
            if (a->gene_exon_off[g] > a->gene_exon_off[g + 1] || a->gene_exon_off[g + 1] > (uint32_t)E) {
                err = "gene_exon_off is not monotone"; return RSQC_ERR_ARG;
            }
            for (uint32_t k = a->gene_exon_off[g]; k < a->gene_exon_off[g + 1]; ++k) {
                const uint32_t row = a->gene_exon_row[k];
                if (row >= (uint32_t)E || seen[row] || a->exon_row_gene[row] != (uint32_t)g) {
                    err = "gene_exon_row is not a partition of the exon rows by gene"; return RSQC_ERR_ARG;
                }
                seen[row] = 1;
                ex_cov[row] = (uint32_t)run;
                const uint64_t len = (uint64_t)(a->exon_row_end[row] - a->exon_row_start[row]) + 1;
                run += len; coding += len;
            }
            run += 1;        // pad slot: absorbs the -1 of a block that ends with the gene's last exon
            if (g < L) gene_coding[(size_t)g] = (uint32_t)std::min<uint64_t>(coding, 0xFFFFFFFFull);
            if (run >= 0xFFFFFFF0ull) { err = "annotation exceeds 2^32 exonic bases"; return RSQC_ERR_CAPACITY; }
        }
        if ((int)a->gene_exon_off[G] != E) { err = "gene_exon_off[n_genes] != n_exons"; return RSQC_ERR_ARG; }
        cov_entries = run;
        // packed 16-byte rows (after the coverage layout: a row carries its coverage offset)
        ex_rows.resize((size_t)E);
        for (int i = 0; i < E; ++i) {
            uint32_t fl = a->exon_row_flags[i] & 0x7u;   // strand + ribosomal
            if (a->gene_is_globin[a->exon_row_gene[i]]) fl |= ROWF_GLOBIN;
            if (ex_pmax[(size_t)i] != a->exon_row_end[i]) fl |= ROWF_PMAX_EXT;
            // closed to the left: first row of its contig, or every earlier row of the contig ends before this one starts
            const bool first_of_contig = i == 0 || a->exon_row_contig[i - 1] != a->exon_row_contig[i];
            if (first_of_contig || ex_pmax[(size_t)i - 1] < a->exon_row_start[i]) fl |= ROWF_LEFT_CLOSED;
            ex_rows[(size_t)i] = ExonRow{a->exon_row_start[i], a->exon_row_end[i], ex_cov[(size_t)i],
                                         a->exon_row_gene[i] | (fl << ROW_FLAG_SHIFT)};
        }
        // An exon normally lies inside its gene's row.  The reference retires a gene (coverage computed, fragment set dropped) when
        // its ROW reaches the front of the sorted feature list and lies behind the stream (src/Expression.cpp:84-93); an exon that
        // sticks out of the row keeps collecting reads for a gene that is gone: the reference prints "Gene encountered after
        // computing coverage" (src/Metrics.cpp:108-112), ignores the coverage and counts the read's name into a fresh fragment set.
        // Round 5: such an annotation is ACCEPTED (rounds 3-4 refused it).  The static index answers every record from the rows as
        // given -- counters, geneCounts, uniqueGeneCounts and exonCounts are then the reference's for any input (they depend on
        // overlaps only); geneFragmentCounts and the coverage / bias statistics of such a gene are what the reference would give if
        // the gene never retired early, and differ from its streamed result exactly when a record that starts behind the gene row's
        // end is counted to the gene.  The first such row is reported as a WARNING (`warning`, rsqc_last_error).
        warning.clear(); n_exons_outside_gene = 0;
        {
            std::vector<int32_t> gs((size_t)std::max(L, 1), 0), ge((size_t)std::max(L, 1), 0), gc((size_t)std::max(L, 1), -1);
            for (int i = 0; i < L; ++i) {
                const uint32_t id = a->gene_row_id[i];
                if (id >= (uint32_t)L) { err = "gene_row_id out of range"; return RSQC_ERR_ARG; }
                gs[id] = a->gene_row_start[i]; ge[id] = a->gene_row_end[i]; gc[id] = a->gene_row_contig[i];
            }
            size_t n_out = 0;
            for (int i = 0; i < E; ++i) {
                const uint32_t g = a->exon_row_gene[i];
                if (g >= (uint32_t)L) continue;                                   // (a gene id that only exon rows carry has no row to retire)
                if (a->exon_row_contig[i] != gc[g] || a->exon_row_start[i] < gs[g] || a->exon_row_end[i] > ge[g]) {
                    if (n_out++ == 0)
                        warning = "exon row " + std::to_string(i) + " (" + std::to_string(a->exon_row_start[i]) + "-" + std::to_string(a->exon_row_end[i]) +
                                  ") lies outside the row of its gene (" + std::to_string(gs[g]) + "-" + std::to_string(ge[g]) + ")";
                }
            }
            n_exons_outside_gene = (uint32_t)n_out;
            if (n_out) warning += (n_out > 1 ? " and " + std::to_string(n_out - 1) + " more" : std::string()) +
                                  ": the reference retires a gene when its row leaves the sorted stream (\"Gene encountered after computing coverage\"); "
                                  "fragment counts and coverage statistics of such genes are computed as if the gene stayed";
        }
        // elementary intervals per contig: sweep over the starts (+) and ends + 1 (-) of gene and exon rows
        ei.clear(); ei_range.assign((size_t)nc + 1, 0); rank_words = 0;
        for (int k = 0; k < nc; ++k) {
            struct Ev { int64_t pos; int32_t what; uint32_t idx; };      // what: +-(1 + class + 3 * ribo) gene, +-100 exon row idx
            std::vector<Ev> ev;
            for (uint32_t i = g_range[(size_t)k]; i < g_range[(size_t)k + 1]; ++i) {
                const int cls = a->gene_row_flags[i] & RSQC_FF_STRAND_MASK, ribo = (a->gene_row_flags[i] & RSQC_FF_RIBOSOMAL) ? 1 : 0;
                const int code = 1 + cls + 3 * ribo;
                ev.push_back(Ev{std::max<int64_t>(0, a->gene_row_start[i]), code, 0u});
                ev.push_back(Ev{(int64_t)a->gene_row_end[i] + 1, -code, 0u});
            }
            for (uint32_t i = ex_range[(size_t)k]; i < ex_range[(size_t)k + 1]; ++i) {
                ev.push_back(Ev{std::max<int64_t>(0, a->exon_row_start[i]), 100, i});
                ev.push_back(Ev{(int64_t)a->exon_row_end[i] + 1, -100, i});
            }
            rank_words = (rank_words + 7ull) & ~7ull;               // (ei_coarse: 8 rank words per entry, see above)
            contig[(size_t)k].rk_base = (uint32_t)rank_words; contig[(size_t)k].rk_words = 0;
            if (ev.empty()) { ei_range[(size_t)k + 1] = (uint32_t)ei.size(); continue; }
            std::stable_sort(ev.begin(), ev.end(), [](const Ev &x, const Ev &y) { return x.pos < y.pos; });
            int cnt[6] = {0, 0, 0, 0, 0, 0};
            std::vector<uint32_t> active;                      // exon rows covering the current interval, ascending
            auto emit = [&](int64_t pos) {
                EiEntry e{(int32_t)pos, 0u, EI_NONE, EI_NONE, 0u, 0u, 0u, 0u};
                for (int cls = 0; cls < 3; ++cls) {
                    if (cnt[cls] + cnt[3 + cls] > 0) e.mask |= 1u << cls;
                    if (cnt[3 + cls] > 0) e.mask |= 1u << (3 + cls);
                }
                for (uint32_t row : active) {
                    const uint32_t fl = a->exon_row_flags[row];
                    e.mask |= 1u << (EIM_EXON_SHIFT + (fl & RSQC_FF_STRAND_MASK));
                    if (fl & RSQC_FF_RIBOSOMAL) e.mask |= 1u << (EIM_EXON_SHIFT + 3 + (fl & RSQC_FF_STRAND_MASK));
                }
                if (active.size() > 2) e.mask |= EIM_DEEP;
                if (!active.empty()) {
                    const uint32_t r = active.back();
                    e.eidA = a->exon_row_id[r]; e.gfA = ex_rows[r].gf; e.cdA = ex_rows[r].cov - (uint32_t)ex_rows[r].start;
                }
                if (active.size() > 1) {
                    const uint32_t r = active[active.size() - 2];
                    e.eidB = a->exon_row_id[r]; e.gfB = ex_rows[r].gf; e.cdB = ex_rows[r].cov - (uint32_t)ex_rows[r].start;
                }
                const bool first = ei.size() == ei_range[(size_t)k];
                if (!first) {
                    const EiEntry &p = ei.back();
                    if (p.mask == e.mask && p.eidA == e.eidA && p.eidB == e.eidB) return;     // nothing changed
                }
                ei.push_back(e);
            };
            size_t e = 0;
            if (ev[0].pos > 0) emit(0);                        // the sentinel interval
            while (e < ev.size()) {
                const int64_t pos = ev[e].pos;
                if (pos > 0x7FFFFFFFll) break;
                while (e < ev.size() && ev[e].pos == pos) {
                    const Ev &x = ev[e++];
                    if (x.what == 100) active.insert(std::upper_bound(active.begin(), active.end(), x.idx), x.idx);
                    else if (x.what == -100) active.erase(std::find(active.begin(), active.end(), x.idx));
                    else if (x.what > 0) cnt[x.what - 1]++;
                    else cnt[-x.what - 1]--;
                }
                emit(pos);
            }
            ei_range[(size_t)k + 1] = (uint32_t)ei.size();
            const uint64_t words = ((uint64_t)(uint32_t)ei.back().pos >> 6) + 1;
            contig[(size_t)k].rk_words = (uint32_t)words;
            rank_words += words;
            if (rank_words >= (1ull << 28) || ei.size() >= (1ull << 27)) { err = "annotation too large for the interval index"; return RSQC_ERR_CAPACITY; }
        }
        ei_coarse.assign((size_t)((rank_words + 7ull) >> 3) + 2, 0u);
        for (int k = 0; k < nc; ++k) {
            const ContigInfo &ci = contig[(size_t)k];
            if (ci.rk_words == 0) continue;
            const uint32_t nb = (ci.rk_words + 7u) >> 3;                       // blocks of 512 positions that hold the contig's breakpoints
            std::vector<uint32_t> first((size_t)nb + 1);                       // index of the first breakpoint at or after b * 512
            uint32_t j = ei_range[(size_t)k];
            for (uint32_t b = 0; b <= nb; ++b) {
                while (j < ei_range[(size_t)k + 1] && ((uint64_t)(uint32_t)ei[j].pos >> 9) < b) ++j;
                first[b] = j;
            }
            // (block 0 holds the contig's first interval at position 0 and the last block its last breakpoint: neither is empty)
            for (uint32_t b = 0; b + 2 <= nb; ++b)
                if (first[b + 2] == first[b] && first[b] > ei_range[(size_t)k]) ei_coarse[(size_t)(ci.rk_base >> 3) + b] = first[b];   // = 1 + (first[b] - 1)
        }
        if (ei.empty()) ei.push_back(EiEntry{0, 0u, EI_NONE, EI_NONE, 0u, 0u, 0u, 0u});    // lanes without a look-up read entry 0
        // a terminator behind the last contig's last interval: the scalar look-up of the per-record kernel (rsqc_k1.h, k1e_interval_of)
        // reads the start of entry j + 1 with entry j; position 0 reads as "entry j is the last interval of its contig"
        ei.push_back(EiEntry{0, 0u, EI_NONE, EI_NONE, 0u, 0u, 0u, 0u});
        gene_flags.assign((size_t)std::max(L, 1), 0);
        gene_owned.assign((size_t)std::max(L, 1), 0);
        for (int i = 0; i < L; ++i) {
            const uint32_t id = a->gene_row_id[i];
            if (id >= (uint32_t)L) { err = "gene_row_id out of range"; return RSQC_ERR_ARG; }
            gene_flags[id] = a->gene_row_flags[i];
            gene_owned[id] = owned_contig ? (owned_contig[a->gene_row_contig[i]] ? 1 : 0) : 1;
        }
        for (int i = 0; i < E; ++i)
            if (a->exon_row_id[i] >= (uint32_t)E) { err = "exon_row_id out of range"; return RSQC_ERR_ARG; }
        // --legacy: gene rows as rows, and the rank of every row in the contig's one start-sorted list of genes and
        // exons (std::list::sort by start, stable: ties keep GTF order, src/RNASeQC.cpp:150-152).  Both row sets are
        // already in that order among themselves, so the list is their merge; without GTF positions a gene row goes
        // before the exon rows of the same start.
        gr_rows.resize((size_t)L); ex_ord.assign((size_t)E, 0);
        // (round 6: the gene rows under the contig's bins, like the exon rows -- legacy_metrics found "the first gene row that starts behind
        //  the read's span" by a binary search over the contig's rows: eleven dependent loads in front of every record)
        gr_binhi.assign(ex_binhi.size(), 0);
        for (int k = 0; k < nc; ++k) {
            const ContigInfo &ci = contig[(size_t)k];
            uint32_t row = g_range[(size_t)k];
            for (uint32_t b = 0; b < ci.n_bins; ++b) {
                const int64_t lim = ((int64_t)b + 1) << kBinShift;
                while (row < g_range[(size_t)k + 1] && (int64_t)a->gene_row_start[row] < lim) ++row;
                gr_binhi[(size_t)ci.bin_base + b] = row;
            }
        }
        const bool have_order = a->gene_row_order && a->exon_row_order;
        for (int k = 0; k < nc; ++k) {
            uint32_t gi = g_range[(size_t)k], ei = ex_range[(size_t)k], rank = 0;
            const uint32_t gN = g_range[(size_t)k + 1], eN = ex_range[(size_t)k + 1];
            while (gi < gN || ei < eN) {
                bool take_gene;
                if (gi == gN) take_gene = false;
                else if (ei == eN) take_gene = true;
                else if (a->gene_row_start[gi] != a->exon_row_start[ei]) take_gene = a->gene_row_start[gi] < a->exon_row_start[ei];
                else take_gene = have_order ? a->gene_row_order[gi] < a->exon_row_order[ei] : true;
                if (take_gene) {
                    gr_rows[gi] = GeneRow{a->gene_row_start[gi], a->gene_row_end[gi],
                                          a->gene_row_id[gi] | ((uint32_t)(a->gene_row_flags[gi] & 0x7u) << ROW_FLAG_SHIFT), rank};
                    ++gi;
                } else ex_ord[ei++] = rank;
                ++rank;
            }
        }
        return 0;
    }
    // the rank table over all contigs, on the host (tests; the product fills it on the device: ei_rank_kernel)
    void build_rank(std::vector<EiRank> &out) const {
        out.assign((size_t)rank_words + 1, EiRank{0u, 0u, 0u, 0u});
        for (int k = 0; k < n_contigs; ++k) {
            const ContigInfo &ci = contig[(size_t)k];
            uint32_t j = ei_range[(size_t)k];
            for (uint32_t w = 0; w < ci.rk_words; ++w) {
                EiRank &r = out[(size_t)ci.rk_base + w];
                r.rank = j;
                while (j < ei_range[(size_t)k + 1] && ((uint32_t)ei[j].pos >> 6) == w) {
                    const uint32_t b = (uint32_t)ei[j].pos & 63u;
                    if (b < 32) r.lo |= 1u << b; else r.hi |= 1u << (b - 32);
                    ++j;
                }
            }
        }
    }
};

}  // namespace rsqc
