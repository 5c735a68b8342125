// hostemu.cpp -- TEST HARNESS ONLY (never loaded by the product).
//
// Compiles the product's per-record core (rnaseqc_amd/csrc/rsqc_read.h, which is
// __host__ __device__) and index builder (rsqc_index.h) with g++ and runs them one record
// at a time, so that the semantics the HIP kernels execute can be diffed against the
// oracle in the GPU-less build container.  Wave-level code (reductions, de-dup, coverage
// statistics, read-length scan) is NOT covered here; the -m gpu tests cover it.
#include <cstdlib>
#include <cstring>
#include <map>
#include <set>
#include <string>
#include <vector>

#include "../../rnaseqc_amd/csrc/rsqc_read.h"
#include "../../rnaseqc_amd/csrc/rsqc_index.h"

using namespace rsqc;

namespace {
struct Acc {
    std::vector<uint64_t> *reads, *unique;
    std::vector<double> *exon_rows;
    std::vector<uint32_t> *cov;
    const uint32_t *ex_cov;
    std::vector<std::set<uint64_t>> *names;
    void gene_hit(uint32_t g, bool nd, uint64_t qh) { (*reads)[g]++; if (nd) (*unique)[g]++; (*names)[g].insert(qh); }
    void exon_add(uint32_t row, double f) { (*exon_rows)[row] += f; }
    void cov_range(uint32_t cidx, uint32_t len) {
        if (!len) return;
        (*cov)[cidx] += 1u;
        (*cov)[cidx + len] -= 1u;
    }
};
template <int K, int NST>
void apply(Acc &acc, const DevAnnotation &d, const FeatureOut<K, NST> &fo, const Record &r, uint32_t aligned) {
    for (int k = 0; k < NST; ++k) {
        if (!((fo.cmask >> k) & 1u)) continue;
        const Commit &c = fo.commit[k];
        if (c.len > 0) acc.exon_add(c.row, (double)c.len / (double)aligned);
        acc.cov_range(c.cidx, c.len);
    }
    for (int k = 0; k < fo.n_hit; ++k) acc.gene_hit(fo.hit[k], !(r.flag & RSQC_FDUP), r.qhash);
}
}  // namespace

// 0: the row-table feature stage (exon_metrics_fast), 1: the elementary-interval one (exon_metrics_ei) for records of
// 1-4 blocks -- what the per-record kernel runs since round 3
static int g_mode = 1;
extern "C" __attribute__((visibility("default"))) void hostemu_set_mode(int m) { g_mode = m; }

template <int NB>
static void run_ei(const DevAnnotation &d, const DevParams &dp, const Record &r, const Blocks &B, bool hq, EiOut &eo, bool &over, BitSink &sink) {
    int32_t bs[NB]; uint32_t len[NB];
    for (int k = 0; k < NB; ++k) { bs[k] = B.bs[k]; len[k] = B.len[k]; }
    exon_metrics_ei<NB, BitSink>(d, dp, d.contig[r.tid], r.flag, bs, len, hq, eo, over, sink);
}

extern "C" __attribute__((visibility("default")))
int hostemu_run(const rsqc_params *p, const rsqc_annotation *a, const rsqc_batch *b,
                uint64_t *counters /*N_COUNTERS*/, uint64_t *gene_reads, uint64_t *gene_unique, uint64_t *gene_frag,
                double *exon_reads /*by exon id*/, int32_t *read_length, uint32_t *cov_out /*cov_entries or NULL*/,
                uint64_t *n_overflow) {
    HostIndex hx; std::string err;
    int rc = hx.build(a, nullptr, err);
    if (rc) return rc;
    DevAnnotation d{};
    d.n_ref = a->n_ref; d.n_contigs = a->n_contigs; d.n_genes = a->n_genes; d.n_listed = a->n_genes_listed; d.n_exons = a->n_exons;
    d.bin_shift = HostIndex::kBinShift;
    d.contig = hx.contig.data();
    // the kernels' unconditional loads read entry 0 of a table for lanes that have nothing to look up
    if (hx.ex_rows.empty()) hx.ex_rows.push_back(ExonRow{0, 0, 0, 0});
    if (hx.gb.empty()) hx.gb.push_back(GeneBreak{0, 0});
    if (hx.ex_pmax.empty()) hx.ex_pmax.push_back(0);
    d.ex = hx.ex_rows.data(); d.gb = hx.gb.data(); d.ex_pmax = hx.ex_pmax.data();
    d.ex_binhi = hx.ex_binhi.data(); d.gb_bin = hx.gb_bin.data(); d.ex_cov = hx.ex_cov.data();
    std::vector<EiRank> rank;
    hx.build_rank(rank);
    d.ei = hx.ei.data(); d.ei_rank = rank.data(); d.ei_coarse = hx.ei_coarse.data();
    std::vector<double> exon_ids((size_t)a->n_exons, 0.0);       // by exon id (the elementary-interval stage commits by id)
    DevParams dp{p->mapq_threshold, p->base_mismatch, p->chimeric_distance, p->stranded, p->unpaired, p->exclude_chimeric, p->n_filter_tags,
                 p->legacy ? 1 : 0};
    if (hx.gr_rows.empty()) hx.gr_rows.push_back(GeneRow{0, 0, 0, 0});
    if (hx.g_pmax.empty()) hx.g_pmax.push_back(0);
    if (hx.ex_ord.empty()) hx.ex_ord.push_back(0);
    if (hx.gr_binhi.empty()) hx.gr_binhi.push_back(0);
    const LegacyTables lt{hx.gr_rows.data(), hx.g_pmax.data(), hx.g_range.data(), hx.ex_ord.data(), hx.gr_binhi.data()};
    d.legacy = &lt;
    std::vector<uint64_t> reads((size_t)a->n_genes, 0), unique((size_t)a->n_genes, 0);
    std::vector<double> exon_rows((size_t)a->n_exons, 0.0);
    std::vector<uint32_t> cov((size_t)hx.cov_entries + 1, 0);
    std::vector<std::set<uint64_t>> names((size_t)a->n_genes);
    Acc acc{&reads, &unique, &exon_rows, &cov, hx.ex_cov.data(), &names};
    memset(counters, 0, sizeof(uint64_t) * RSQC_N_COUNTERS);
    uint32_t rl = 0; uint32_t w = 0; *n_overflow = 0;
    for (uint32_t s = 0; s < b->n_seg; ++s) for (uint64_t i = b->seg_start[s]; i < b->seg_start[s + 1]; ++i) {
        Record r;
        const rsqc_rec_core &co = b->core[i]; const rsqc_rec_aux &au = b->aux[i];
        r.tid = b->seg_tid[s]; r.pos = co.pos; r.mpos = co.mpos; r.isize = co.isize; r.flag = au.flag;
        r.mapq = au.mapq; r.tagbits = au.tagbits; r.l_qseq = au.l_qseq; r.nm = au.nm; r.n_cigar = au.n_cigar;
        if (au.l_qseq == RSQC_LQSEQ_ESCAPE || au.nm == RSQC_NM_ESCAPE || au.n_cigar == RSQC_NCIGAR_ESCAPE) {
            while (w < b->n_wide && b->wide_index[w] < i) ++w;
            if (w >= b->n_wide || b->wide_index[w] != i) return RSQC_ERR_ARG;
            r.l_qseq = b->wide_l_qseq[w]; r.nm = b->wide_nm[w]; r.n_cigar = b->wide_n_cigar[w];
        }
        r.cigar = b->cigar + co.cigar_off; r.qhash = au.qhash;
        RecordCounters rc2; bool hq; uint32_t aligned; Blocks B;
        const bool go = gate_cascade(d, dp, r, rc2, hq, aligned, B);
        uint64_t bits = rc2.bits;
        if (rc2.error) return rc2.error;
        if (go && dp.legacy) {
            LegacyOut<MID_SET> lo;
            legacy_metrics<MID_SET>(d, dp, r, hq, acc, lo);
            bits |= lo.bits;
            for (int k = 0; k < lo.n_hit; ++k) acc.gene_hit(lo.hit[k], !(r.flag & RSQC_FDUP), r.qhash);
        } else if (go && g_mode == 1 && B.nb >= 1 && B.nb <= (uint32_t)FAST_BLOCKS) {
            bool over = false;
            EiOut eo; BitSink fsink;
            if (B.nb == 1) run_ei<1>(d, dp, r, B, hq, eo, over, fsink);
            else if (B.nb == 2) run_ei<2>(d, dp, r, B, hq, eo, over, fsink);
            else if (B.nb == 3) run_ei<3>(d, dp, r, B, hq, eo, over, fsink);
            else run_ei<4>(d, dp, r, B, hq, eo, over, fsink);
            if (!over) {
                bits |= fsink.bits;
                for (int k = 0; k < NSLOT; ++k) {
                    if (!((eo.cmask >> k) & 1u)) continue;
                    const uint32_t len = B.len[k >> 1];
                    if (len > 0) exon_ids[eo.eid[k]] += (double)len / (double)aligned;
                    acc.cov_range(eo.cidx[k], len);
                }
                for (int k = 0; k < eo.n_hit; ++k) acc.gene_hit(eo.hit[k], !(r.flag & RSQC_FDUP), r.qhash);
            } else {
                ++*n_overflow;
                FeatureOut<SLOW_SET, SLOW_STAGE> so;
                exon_metrics<SLOW_SET>(d, dp, r, hq, aligned, acc, so, over);
                if (over) return RSQC_ERR_CAPACITY;
                bits |= so.bits; apply(acc, d, so, r, aligned);
            }
        } else if (go) {
            bool over = false;
            FastOut fo;
            BitSink fsink;
            exon_metrics_fast<2, BitSink>(d, dp, d.contig[r.tid], r.flag, B, hq, aligned, fo, over, fsink);
            if (!over) {
                bits |= fsink.bits;
                for (int k = 0; k < NSLOT; ++k) {
                    if (!((fo.cmask >> k) & 1u)) continue;
                    const uint32_t len = B.len[k >> 1];
                    if (len > 0) acc.exon_add(fo.row[k], (double)len / (double)aligned);
                    acc.cov_range(fo.cidx[k], len);
                }
                for (int k = 0; k < fo.n_hit; ++k) acc.gene_hit(fo.hit[k], !(r.flag & RSQC_FDUP), r.qhash);
            }
            else {
                ++*n_overflow;
                FeatureOut<SLOW_SET, SLOW_STAGE> so;
                exon_metrics<SLOW_SET>(d, dp, r, hq, aligned, acc, so, over);
                if (over) return RSQC_ERR_CAPACITY;
                bits |= so.bits; apply(acc, d, so, r, aligned);
            }
        }
        for (int c = 0; c < RSQC_N_COUNTERS; ++c) if ((bits >> c) & 1ull) counters[c]++;
        counters[RSQC_C_END1_MISMATCHES] += rc2.e1_mm; counters[RSQC_C_END1_BASES] += rc2.e1_bases;
        counters[RSQC_C_END2_MISMATCHES] += rc2.e2_mm; counters[RSQC_C_END2_BASES] += rc2.e2_bases;
        counters[RSQC_C_MISMATCHED_BASES] += rc2.mm; counters[RSQC_C_TOTAL_BASES] += rc2.bases;
        counters[RSQC_C_ALIGNMENT_BLOCKS] += rc2.blocks;
        if (rc2.rl_eligible && rc2.rl_span > rl) rl = (uint32_t)rc2.rl_lqseq;
    }
    for (int g = 0; g < a->n_genes_listed; ++g) { gene_reads[g] = reads[(size_t)g]; gene_unique[g] = unique[(size_t)g]; gene_frag[g] = names[(size_t)g].size(); }
    for (int e = 0; e < a->n_exons; ++e) exon_reads[a->exon_row_id[e]] = exon_rows[(size_t)e] + exon_ids[a->exon_row_id[e]];
    *read_length = (int32_t)rl;
    if (cov_out) memcpy(cov_out, cov.data(), hx.cov_entries * 4);
    return 0;
}
