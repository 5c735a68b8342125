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
    std::vector<uint32_t> ex_range, g_range, ex_binhi, g_binhi, ex_cov, gene_cov_off, gene_coding;
    std::vector<int32_t> ex_pmax, g_pmax;
    std::vector<ExonRow> ex_rows;
    std::vector<GeneRow> g_rows;
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
        // packed 16-byte rows
        ex_rows.resize((size_t)E); g_rows.resize((size_t)L);
        for (int i = 0; i < E; ++i) {
            if (a->exon_row_gene[i] >= (uint32_t)G) { err = "exon_row_gene out of range"; return RSQC_ERR_ARG; }
            uint32_t fl = a->exon_row_flags[i] & 0x7u;
            if (a->gene_is_globin[a->exon_row_gene[i]]) fl |= ROWF_GLOBIN;
            ex_rows[(size_t)i] = ExonRow{a->exon_row_start[i], a->exon_row_end[i], ex_pmax[(size_t)i],
                                         a->exon_row_gene[i] | (fl << ROW_FLAG_SHIFT)};
        }
        for (int i = 0; i < L; ++i)
            g_rows[(size_t)i] = GeneRow{a->gene_row_start[i], a->gene_row_end[i], g_pmax[(size_t)i],
                                        (uint32_t)(a->gene_row_flags[i] & 0x7u) << ROW_FLAG_SHIFT};
        // per-contig info + bin tables: binhi[b] = first row with start >= (b+1) << shift
        contig.assign((size_t)nc, ContigInfo{0, 0, 0, 0});
        auto max_start = [&](const std::vector<uint32_t> &range, const int32_t *start, int k) -> int64_t {
            return range[(size_t)k + 1] > range[(size_t)k] ? (int64_t)start[range[(size_t)k + 1] - 1] : -1;
        };
        uint64_t total_bins = 0;
        for (int k = 0; k < nc; ++k) {
            const int64_t ms = std::max(max_start(ex_range, a->exon_row_start, k), max_start(g_range, a->gene_row_start, k));
            const uint64_t nb = ms < 0 ? 0 : (uint64_t)(ms >> kBinShift) + 1;
            if (total_bins + nb > 0xFFFFFFF0ull) { err = "bin table too large"; return RSQC_ERR_CAPACITY; }
            contig[(size_t)k] = ContigInfo{ex_range[(size_t)k], g_range[(size_t)k], (uint32_t)total_bins, (uint32_t)nb};
            total_bins += nb;
        }
        auto build_bins = [&](const std::vector<uint32_t> &range, const int32_t *start, std::vector<uint32_t> &bins) {
            bins.assign((size_t)total_bins + 1, 0);
            for (int k = 0; k < nc; ++k) {
                const ContigInfo &ci = contig[(size_t)k];
                uint32_t row = range[(size_t)k];
                const uint32_t hi = range[(size_t)k + 1];
                for (uint32_t b = 0; b < ci.n_bins; ++b) {
                    const int64_t lim = ((int64_t)b + 1) << kBinShift;
                    while (row < hi && (int64_t)start[row] < lim) ++row;
                    bins[(size_t)ci.bin_base + b] = row;
                }
            }
        };
        build_bins(ex_range, a->exon_row_start, ex_binhi);
        build_bins(g_range, a->gene_row_start, g_binhi);
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
        return 0;
    }
};

}  // namespace rsqc
