/*
 * ref_metrics_harness.cpp -- TEST INFRASTRUCTURE.
 *
 * Drives the REFERENCE's own src/Metrics.cpp (compiled unmodified, in place,
 * by oracle/Makefile into oracle/_ref/libref_metrics.so) so that our plain-C
 * restatement of BaseCoverage / computeCoverage / BiasCounter / computeMedian /
 * getStatistics / Metrics printing can be pinned against the real code.
 *
 * What is and is not the reference here: Metrics.cpp and the headers it
 * includes (Metrics.h, GTF.h, Fasta.h, bioio.hpp) are the reference's, as they
 * lie under /root/reference.  Metrics.cpp reads four symbols that live in
 * translation units which need boost (GTF.cpp, Fasta.cpp) and therefore
 * cannot be built in this image:
 *   - rnaseqc::exonsForGene, rnaseqc::exonLengths : the annotation tables.  They
 *     are this harness's INPUT, filled below from the caller's arrays.
 *   - Fasta::hasContig / Fasta::getSeq / gc : only reached with --fasta (out of
 *     scope); defined here as "no FASTA was given".
 * No SeqLib / htslib / boost stand-in exists anywhere in this repository; the
 * per-read part of the reference (RNASeQC.cpp, Expression.cpp) stays unbuilt.
 */
#include "Metrics.h"

#include <cstdint>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <list>
#include <string>
#include <vector>

namespace rnaseqc {
    // inputs of Metrics.cpp that GTF.cpp would have filled (src/GTF.cpp:22-27)
    std::map<std::string, FeatureSpan> exonLengths;
    std::map<std::string, std::vector<std::string>> exonsForGene;
    // "no --fasta": src/Fasta.cpp answers false when no index was loaded
    bool Fasta::hasContig(chrom) const { return false; }
    std::string Fasta::getSeq(chrom, coord, coord) { return std::string(); }
    double gc(std::string &) { return -1; }      // src/Fasta.cpp:67-74 on an empty sequence; FASTA-only branch
}

using namespace rnaseqc;

extern "C" {

// Collector (src/Metrics.h:43-59, src/Metrics.cpp:48-93) driven with a recorded call sequence: kind 0 = add(gene, exon,
// frac), 1 = queryGene(gene), 2 = collect(gene); a new Collector is constructed whenever `read` changes, like the one
// exonAlignmentMetrics constructs per alignment (src/Expression.cpp:320).  The target map plays exonCounts.
// Outputs: the map's value per exon id and whether the exon has a map ENTRY (src/RNASeQC.cpp:513 counts entries), the
// answer of every queryGene, and Collector::sum() after the last call of each read that collected something.
__attribute__((visibility("default")))
int ref_collector_replay(uint64_t n, const uint8_t *kind, const uint32_t *read, const uint32_t *gene, const uint32_t *exon, const double *frac,
                         uint32_t n_exons, double *exon_value, uint8_t *exon_entry, uint8_t *query_out, double *max_sum) {
    std::map<std::string, double> target;
    Collector *col = nullptr; uint32_t cur = 0; bool have = false;
    *max_sum = 0.0;
    auto name = [](char p, uint32_t v) { char b[16]; snprintf(b, sizeof b, "%c%09u", p, v); return std::string(b); };
    for (uint64_t i = 0; i < n; ++i) {
        if (!have || read[i] != cur) { if (col) { if (col->sum() > *max_sum) *max_sum = col->sum(); delete col; } col = new Collector(&target); cur = read[i]; have = true; }
        if (kind[i] == 0) col->add(name('G', gene[i]), name('E', exon[i]), frac[i]);
        else if (kind[i] == 1) query_out[i] = col->queryGene(name('G', gene[i])) ? 1 : 0;
        else col->collect(name('G', gene[i]));
    }
    if (col) { if (col->sum() > *max_sum) *max_sum = col->sum(); delete col; }
    for (uint32_t e = 0; e < n_exons; ++e) {
        auto it = target.find(name('E', e));
        exon_entry[e] = it != target.end(); exon_value[e] = it != target.end() ? it->second : 0.0;
    }
    return 0;
}

// computeMedian on an already ordered list (src/Metrics.h:147-160)
__attribute__((visibility("default")))
int ref_median(const double *v, uint64_t n, double *out) {
    try {
        std::vector<double> d(v, v + n);
        *out = computeMedian(d.size(), d.begin());
        return 0;
    } catch (std::range_error &) { return -6; }
}

// getAdvancedStatistics (src/Metrics.h:188-206) over a list of unsigned values: avg, skewness, std, kurtosis
__attribute__((visibility("default")))
void ref_advanced_statistics(const unsigned int *v, uint64_t n, double out[4]) {
    std::list<unsigned int> d(v, v + n);
    statsTuple t = getAdvancedStatistics(d);
    out[0] = std::get<StatIdx::avg>(t); out[1] = std::get<StatIdx::skew>(t);
    out[2] = std::get<StatIdx::std>(t); out[3] = std::get<StatIdx::kurt>(t);
}

// getStatistics (src/Metrics.h:166-186): avg, median, std, MAD
__attribute__((visibility("default")))
void ref_statistics(const double *v, uint64_t n, double out[4]) {
    std::vector<double> d(v, v + n);
    statsTuple t = getStatistics(d);
    out[0] = std::get<StatIdx::avg>(t); out[1] = std::get<StatIdx::med>(t);
    out[2] = std::get<StatIdx::std>(t); out[3] = std::get<StatIdx::mad>(t);
}

// operator<<(ofstream&, Metrics&) (src/Metrics.cpp:342-412): the counter block of metrics.tsv
__attribute__((visibility("default")))
int ref_metrics_print(int n, const char *const *names, const uint64_t *values, const char *path) {
    Metrics m;
    for (int i = 0; i < n; ++i) {
        uint64_t v = values[i];
        while (v > 0) { int step = v > 1000000000ull ? 1000000000 : (int)v; m.increment(names[i], step); v -= (uint64_t)step; }
    }
    std::ofstream out(path);
    if (!out.is_open()) return -1;
    out << m;
    out.close();
    return 0;
}

// Metrics::frac (src/Metrics.cpp:43-46)
__attribute__((visibility("default")))
double ref_frac(uint64_t a, uint64_t b) {
    Metrics m;
    for (uint64_t v = a; v > 0;) { int s = v > 1000000000ull ? 1000000000 : (int)v; m.increment("a", s); v -= (uint64_t)s; }
    for (uint64_t v = b; v > 0;) { int s = v > 1000000000ull ? 1000000000 : (int)v; m.increment("b", s); v -= (uint64_t)s; }
    return m.frac("a", "b");
}

/*
 * Whole coverage path: BaseCoverage::add/commit/reset per committed block, then
 * BaseCoverage::compute per gene (-> computeCoverage -> BiasCounter::computeBias),
 * then BiasCounter::getBias per gene.  Genes are "g<i>", exons "e<j>" (j = row).
 *   gene_exon_off[G+1], exon_len[E] : exonsForGene order (rows contiguous per gene)
 *   gene_strand[G] : 0 '+', 1 '-', 2 '.'
 *   commits: (exon row, offset, length) in file order; commit_read[k] groups blocks of one read
 * outputs (size G / E): gene_valid, gene_mean/std/cv, exon_cv_valid, exon_cv, bias_ratio (-1 = none)
 * returns number of genes counted by getBias (countGenes), or <0 on exception.
 */
__attribute__((visibility("default")))
int ref_coverage_run(int G, const uint32_t *gene_exon_off, const int64_t *exon_len, const int32_t *gene_strand,
                     uint64_t n_commits, const uint32_t *commit_exon, const int64_t *commit_off,
                     const uint32_t *commit_len, const uint32_t *commit_read,
                     unsigned mask, int bias_offset, int bias_window, unsigned long bias_gene_length,
                     uint8_t *gene_valid, double *gene_mean, double *gene_std, double *gene_cv,
                     uint8_t *exon_cv_valid, double *exon_cv, double *bias_ratio,
                     const char *coverage_tsv_path) {
    try {
        exonLengths.clear(); exonsForGene.clear();
        const uint32_t E = gene_exon_off[G];
        std::vector<uint32_t> gene_of(E);
        for (int g = 0; g < G; ++g) {
            std::string gid = "g" + std::to_string(g);
            exonsForGene[gid];
            for (uint32_t j = gene_exon_off[g]; j < gene_exon_off[g + 1]; ++j) {
                std::string eid = "e" + std::to_string(j);
                exonsForGene[gid].push_back(eid);
                exonLengths[eid] = {1, 1000, exon_len[j]};
                gene_of[j] = (uint32_t)g;
            }
        }
        alignas(Fasta) static char fasta_storage[sizeof(Fasta)];      // never constructed: no FASTA
        Fasta &fasta = *reinterpret_cast<Fasta *>(fasta_storage);
        BiasCounter bias(bias_offset, bias_window, bias_gene_length, 5u);
        BaseCoverage cov(fasta, coverage_tsv_path ? coverage_tsv_path : "", mask, coverage_tsv_path != nullptr, bias);
        uint64_t k = 0;
        while (k < n_commits) {
            uint64_t k1 = k;
            std::vector<std::string> genes;
            while (k1 < n_commits && commit_read[k1] == commit_read[k]) {
                Feature exon;
                exon.start = 1000; exon.end = 1000 + exon_len[commit_exon[k1]] - 1;
                exon.feature_id = "e" + std::to_string(commit_exon[k1]);
                exon.gene_id = "g" + std::to_string(gene_of[commit_exon[k1]]);
                exon.type = FeatureType::Exon;
                cov.add(exon, 1000 + commit_off[k1], 1000 + commit_off[k1] + commit_len[k1]);
                bool have = false;
                for (auto &s : genes) if (s == exon.gene_id) have = true;
                if (!have) genes.push_back(exon.gene_id);
                ++k1;
            }
            for (auto &s : genes) cov.commit(s);
            cov.reset();
            k = k1;
        }
        for (int g = 0; g < G; ++g) {
            Feature gene;
            gene.feature_id = "g" + std::to_string(g); gene.gene_id = gene.feature_id;
            gene.type = FeatureType::Gene;
            gene.strand = gene_strand[g] == 0 ? Strand::Forward : (gene_strand[g] == 1 ? Strand::Reverse : Strand::Unknown);
            size_t before = cov.getGeneMeans().size();
            cov.compute(gene);
            if (cov.getGeneMeans().size() > before) {
                gene_valid[g] = 1;
                gene_mean[g] = cov.getGeneMeans().back(); gene_std[g] = cov.getGeneStds().back(); gene_cv[g] = cov.getGeneCVs().back();
            } else gene_valid[g] = 0;
        }
        cov.close();
        for (uint32_t j = 0; j < E; ++j) exon_cv_valid[j] = 0;
        for (auto &kv : cov.getExonCoverage()) {
            uint32_t j = (uint32_t)std::stoul(kv.first.substr(1));
            exon_cv_valid[j] = 1; exon_cv[j] = kv.second.cv;
        }
        for (int g = 0; g < G; ++g) bias_ratio[g] = bias.getBias("g" + std::to_string(g));
        return (int)bias.countGenes();
    } catch (std::range_error &) {
        return -6;
    } catch (...) {
        return -100;
    }
}

}  // extern "C"
