// k3_emu.cpp -- TEST HARNESS ONLY (never loaded by the product).
//
// The end-of-file coverage KERNEL -- rnaseqc_amd/csrc/rsqc_k3.h (gene_coverage_kernel in its four instances), unmodified --
// compiled for the host on top of the 64-lane fiber emulation of wavemu.h.  Input: the per-base difference array and the gene
// counts a pass over the records leaves (the caller takes them from hostemu_run, the per-record code on the host); output: what
// the kernel writes -- per-gene mean / std / CV (+ validity), per-exon CV, the bias accumulators -- for comparison with the oracle.
#include "wavemu.h"

#include <algorithm>
#include <string>
#include <vector>

#include "../../rnaseqc_amd/csrc/rsqc_read.h"
#include "../../rnaseqc_amd/csrc/rsqc_device.h"
#include "../../rnaseqc_amd/csrc/rsqc_index.h"
#include "../../rnaseqc_amd/csrc/rsqc_wave.h"
#define RSQC_FIN_STAMP(sec)
#define RSQC_FIN_SECT(base, sec)
#define RSQC_FIN_BEGIN
#include "../../rnaseqc_amd/csrc/rsqc_k3.h"

using namespace rsqc;

// force: 0 = the classes the library would choose, 1 = every gene through the 1024-thread / 146 KB instance, 2 = through the
// 1024-thread / 64 KB one (longer genes scan in place in memory), 3 = 256 threads, 4 = one wave (genes beyond the instance's LDS
// capacity run its in-memory mode)
extern "C" __attribute__((visibility("default")))
int k3emu_run(const rsqc_params *p, const rsqc_annotation *a, const uint32_t *cov_diff, const uint64_t *gene_reads, int force,
              double *g_mean, double *g_std, double *g_cv, uint8_t *g_valid, double *e_cv, uint8_t *e_cv_valid,
              uint64_t *bias3, uint64_t *bias5, uint64_t *stats /*[4]: genes per class*/) {
    HostIndex hx; std::string err;
    int rc = hx.build(a, nullptr, err);
    if (rc) return rc;
    const int L = a->n_genes_listed;
    if (hx.ex_rows.empty()) hx.ex_rows.push_back(ExonRow{0, 0, 0, 0});
    std::vector<uint32_t> cov(cov_diff, cov_diff + hx.cov_entries);           // (the in-memory mode scans in place)
    std::vector<uint32_t> ex_id(a->exon_row_id, a->exon_row_id + a->n_exons); if (ex_id.empty()) ex_id.push_back(0);
    std::vector<uint32_t> order((size_t)std::max(L, 1), 0);
    for (int g = 0; g < L; ++g) order[(size_t)g] = (uint32_t)g;
    std::stable_sort(order.begin(), order.begin() + L, [&](uint32_t x, uint32_t y) { return hx.gene_coding[x] > hx.gene_coding[y]; });
    std::vector<unsigned long long> reads(gene_reads, gene_reads + std::max(L, 1)), b3((size_t)std::max(L, 1), 0ull), b5((size_t)std::max(L, 1), 0ull);
    int error = 0;
    GeneCovArgs A{};
    A.ge_off = a->gene_exon_off; A.ge_row = a->gene_exon_row; A.ex = hx.ex_rows.data(); A.ex_cov = hx.ex_cov.data(); A.ex_id = ex_id.data();
    A.gene_cov_off = hx.gene_cov_off.data(); A.gene_coding = hx.gene_coding.data(); A.gene_flags = hx.gene_flags.data(); A.gene_owned = hx.gene_owned.data();
    A.gene_order = order.data(); A.gene_reads = reads.data(); A.cov = cov.data(); A.n_listed = L;
    A.mask = p->coverage_mask; A.bias_offset = p->bias_offset; A.bias_window = p->bias_window; A.bias_gene_length = p->bias_gene_length;
    A.g_mean = g_mean; A.g_std = g_std; A.g_cv = g_cv; A.g_valid = g_valid; A.e_cv = e_cv; A.e_cv_valid = e_cv_valid;
    A.bias3 = b3.data(); A.bias5 = b5.data(); A.error = &error;
    if (A.bias_window > 128) return RSQC_ERR_ARG;                              // (the harness instantiates the 128-entry window only)
    uint32_t nl = 0, nm = 0, nx = 0;
    for (int k = 0; k < L; ++k) {
        const uint32_t len = hx.gene_coding[order[(size_t)k]];
        if (len > (uint32_t)RSQC_K3_MEDIUM_MAX) nl++; else if (len > (uint32_t)RSQC_K3_SMALL_MAX) nm++;
        if (len > (uint32_t)RSQC_K3_LARGE2_LDS16) nx++;
    }
    if (force == 1) { nl = (uint32_t)L; nm = 0; nx = (uint32_t)L; } else if (force == 2) { nl = (uint32_t)L; nm = 0; nx = 0; }
    else if (force == 3) { nl = 0; nm = (uint32_t)L; nx = 0; } else if (force == 4) { nl = 0; nm = 0; nx = 0; }
    const uint32_t n = (uint32_t)L, ns = n - nl - nm;
    auto launch = [&](auto kernel, uint32_t threads, uint32_t count, uint32_t first) {
        wavemu::grid_dim().x = count;
        for (uint32_t b = 0; b < count; ++b) { wavemu::block_idx().x = b; wavemu::run_block((int)threads, [&]() { kernel(A, first); }); }
    };
    launch(gene_coverage_kernel<1024, 128, uint16_t, RSQC_K3_LARGE_LDS16>, 1024, nx, 0u);
    launch(gene_coverage_kernel<1024, 128, uint16_t, RSQC_K3_LARGE2_LDS16>, 1024, nl - nx, nx);
    launch(gene_coverage_kernel<256, 128, uint32_t, RSQC_K3_MEDIUM_MAX>, 256, nm, nl);
    launch(gene_coverage_kernel<64, 128, uint32_t, RSQC_K3_SMALL_MAX>, 64, ns, nl + nm);
    for (int g = 0; g < L; ++g) { bias3[g] = b3[(size_t)g]; bias5[g] = b5[(size_t)g]; }
    if (stats) { stats[0] = nx; stats[1] = nl - nx; stats[2] = nm; stats[3] = ns; }
    return error;
}
