#include "report.hpp"

#include <algorithm>
#include <climits>
#include <cmath>
#include <fstream>
#include <iostream>
#include <list>
#include <map>
#include <stdexcept>

namespace rsqc_host {

static const double MAD_FACTOR = 1.4826;

double compute_median(const std::vector<double> &v) {
    const unsigned long size = v.size();
    if (size == 0) throw std::range_error("Cannot compute median of an empty list");
    if (size == 1) return v[0];
    const unsigned long mid = (size - 1) / 2;
    if (size % 2) return (v[mid] + v[mid + 1]) / 2.0;
    return v[mid];
}

void get_statistics(std::vector<double> &data, double &avg, double &med, double &sd, double &mad) {
    if (data.empty()) { avg = med = sd = mad = NAN; return; }
    std::sort(data.begin(), data.end());
    const double size = (double)data.size();
    med = compute_median(data);
    avg = 0.0; sd = 0.0;
    std::vector<double> dev;
    for (double e : data) { avg += e / size; dev.push_back(std::fabs(e - med)); }
    std::sort(dev.begin(), dev.end());
    mad = compute_median(dev) * MAD_FACTOR;
    for (double e : data) sd += std::pow(e - avg, 2.0) / size;
    sd = std::pow(sd, 0.5);
}

// The reference walks x = unique, unique+1, ... < 1e9 and keeps the FIRST x that minimises
// (unsigned)|x(1-exp(-n/x)) - unique| (src/RNASeQC.cpp:405-414).  The estimate grows with x from below
// `unique` towards n > unique, so the truncated error is non-increasing down to its minimum: the first
// minimiser is the first x whose error is below (minimum + 1).  Bisect for it, then confirm with a short
// literal scan (guards against last-bit wiggles of exp).
unsigned library_complexity(double duplicates, double unique, double limit) {
    const double numReads = duplicates + unique;
    if (!(duplicates > 0) || !(unique < limit)) return 0u;
    typedef unsigned long long ull;
    auto x_of = [&](ull k) { return unique + (double)k; };
    auto gap = [&](ull k) { const double x = x_of(k); return unique - x * (1.0 - std::exp(-1.0 * numReads / x)); };   // > 0 before the crossing
    auto err = [&](ull k) { return (unsigned)std::fabs(gap(k)); };
    const ull steps = (ull)std::ceil(limit - unique);                 // iterations of `for (x = unique; x < limit; ++x)`
    if (steps == 0) return 0u;
    const double last_gap = gap(steps - 1);
    const double target = last_gap < 1.0 ? 1.0 : (double)(unsigned)last_gap + 1.0;    // first x with gap < target
    ull lo = 0, hi = steps - 1;
    while (lo < hi) { const ull mid = lo + (hi - lo) / 2; if (gap(mid) < target) hi = mid; else lo = mid + 1; }
    const ull from = lo > 256 ? lo - 256 : 0, to = std::min(steps, lo + 256);
    unsigned minError = UINT_MAX, minReads = 0u;
    for (ull k = from; k < to; ++k) {
        const unsigned e = err(k);
        if (e < minError) { minError = e; minReads = (unsigned)x_of(k); }
    }
    return minReads;
}

static std::string gene_name_of(Annotation &ann, const std::string &id) { return ann.gene_name(id); }

void write_reports(const ReportConfig &cfg, Annotation &ann, const rsqc_results &r,
                   const std::vector<int> &contig_visit_order) {
    // '\n' instead of std::endl: the same bytes without a flush (a write syscall) per line; the files are flushed on close
    const char endl = '\n';
    const std::string base = cfg.output_dir + "/" + cfg.sample_name;
    auto cnt = [&](int c) { return (unsigned long)r.counters[c]; };
    auto frac = [&](int a, int b) { return static_cast<double>(cnt(a)) / cnt(b); };       // Metrics::frac

    // ---- coverage.tsv (BaseCoverage writer; rows in gene exit order) --------------------------------
    if (cfg.write_coverage) {
        std::ofstream cov(base + ".coverage.tsv");
        cov << "gene_id\tcoverage_mean\tcoverage_std\tcoverage_CV" << endl;
        std::vector<char> seen(ann.contig_names.size(), 0);
        std::vector<int> order;
        for (int c : contig_visit_order) if (c >= 0 && c < (int)seen.size() && !seen[(size_t)c]) { seen[(size_t)c] = 1; order.push_back(c); }
        // remaining contigs are flushed at EOF in chromosomeMap id order (src/RNASeQC.cpp:385-386)
        std::vector<std::pair<int, int>> rest;
        for (size_t c = 0; c < seen.size(); ++c) if (!seen[c]) rest.emplace_back(ann.chrom_of_contig[c], (int)c);
        std::sort(rest.begin(), rest.end());
        for (auto &p : rest) order.push_back(p.second);
        for (int c : order) for (uint32_t g : ann.genes_by_contig[(size_t)c]) {
            cov << ann.gene_list[g] << "\t";
            if (r.gene_cov_valid[g]) cov << r.gene_cov_mean[g] << "\t" << r.gene_cov_std[g] << "\t" << r.gene_cov_cv[g] << endl;
            else cov << "0\t0\tnan" << endl;
        }
    }

    // ---- library complexity (src/RNASeQC.cpp:398-415) -------------------------------------------------
    const double duplicates = (double)cnt(RSQC_C_DUPLICATE_PAIRS), unique = (double)cnt(RSQC_C_UNIQUE_FRAGMENTS);
    const unsigned minReads = library_complexity(duplicates, unique);

    // ---- gene tables (:419-475) ---------------------------------------------------------------------------
    unsigned genesDetected = 0; unsigned biasGenes = 0;
    double fragmentMed = 0.0;
    std::vector<double> ratios;
    {
        std::ofstream geneReport(base + ".gene_reads.gct");
        std::ofstream geneRPKM(base + ".gene_" + (cfg.use_rpkm ? "rpkm" : "tpm") + ".gct");
        std::ofstream fragmentReport(base + ".gene_fragments.gct");
        geneReport << "#1.2" << endl; geneRPKM << "#1.2" << endl; fragmentReport << "#1.2" << endl;
        geneReport << ann.gene_list.size() << "\t1" << endl;
        geneRPKM << ann.gene_list.size() << "\t1" << endl;
        fragmentReport << ann.gene_list.size() << "\t1" << endl;
        geneReport << "Name\tDescription\t" << (cfg.sample_given ? cfg.sample_name : std::string("Counts")) << endl;
        geneRPKM << "Name\tDescription\t" << (cfg.sample_given ? cfg.sample_name : std::string(cfg.use_rpkm ? "RPKM" : "TPM")) << endl;
        geneRPKM << std::fixed;
        fragmentReport << "Name\tDescription\t" << (cfg.sample_given ? cfg.sample_name : std::string("Fragments")) << endl;
        const double scaleRPKM = static_cast<double>(cnt(RSQC_C_EXONIC_READS)) / 1000000.0;
        double scaleTPM = 0.0;
        std::vector<double> tpms(ann.gene_list.size(), 0.0);
        for (size_t g = 0; g < ann.gene_list.size(); ++g) {
            const std::string &gene = ann.gene_list[g];
            const double geneCount = (double)r.gene_reads[g];
            const double codingLength = static_cast<double>(ann.coding_length(gene));
            geneReport << gene << "\t" << gene_name_of(ann, gene) << "\t" << static_cast<long>(geneCount) << endl;
            fragmentReport << gene << "\t" << gene_name_of(ann, gene) << "\t" << static_cast<long>((double)r.gene_fragments[g]) << endl;
            if (cfg.use_rpkm) {
                const double RPKM = (1000.0 * geneCount / scaleRPKM) / codingLength;
                geneRPKM << gene << "\t" << gene_name_of(ann, gene) << "\t" << RPKM << endl;
            } else {
                const double TPM = (1000.0 * geneCount) / codingLength;
                tpms[g] = TPM; scaleTPM += TPM;
            }
            if ((double)r.gene_unique[g] >= cfg.detection_threshold) ++genesDetected;
            // BiasCounter::getBias (src/Metrics.cpp:239-249)
            const double cov5 = (double)r.bias_five[g], cov3 = (double)r.bias_three[g];
            if (cov5 + cov3 > 0.0) { ++biasGenes; ratios.push_back(cov3 / (cov5 + cov3)); }
        }
        if (!cfg.use_rpkm) {
            scaleTPM /= 1000000.0;
            for (size_t g = 0; g < ann.gene_list.size(); ++g)
                geneRPKM << ann.gene_list[g] << "\t" << gene_name_of(ann, ann.gene_list[g]) << "\t" << tpms[g] / scaleTPM << endl;
        }
    }
    // ---- 3'/5' bias summary (:477-508) -----------------------------------------------------------------------
    double ratioAvg = 0.0, ratioMedDev = 0.0, ratioMedian = 0.0, ratioStd = 0.0, ratio75 = 0.0, ratio25 = 0.0;
    if (ratios.size() > 1) {
        get_statistics(ratios, ratioAvg, ratioMedian, ratioStd, ratioMedDev);
        // the reference indexes past the end for 2..4 ratios (undefined behaviour); we clamp
        auto at = [&](double i) { size_t k = (size_t)static_cast<int>(i); return ratios[std::min(k, ratios.size() - 1)]; };
        double index = .25 * ratios.size();
        if (index > std::floor(index)) { index = std::ceil(index); ratio25 = at(index); }
        else { index = std::ceil(index); ratio25 = (at(index) + at(index + 1)) / 2.0; }
        index = .75 * ratios.size();
        if (index > std::floor(index)) { index = std::ceil(index); ratio75 = at(index); }
        else { index = std::ceil(index); ratio75 = (at(index) + at(index + 1)) / 2.0; }
    }
    // ---- exon table (:509-521) -----------------------------------------------------------------------------------
    {
        std::ofstream exonReport(base + ".exon_reads.gct");
        size_t hit = 0;
        for (int e = 0; e < r.n_exons; ++e) hit += r.exon_hit[e] ? 1 : 0;
        exonReport << "#1.2" << endl;
        exonReport << hit << "\t1" << endl;                      // exonCounts.size() before the lookups below (Q7)
        exonReport << "Name\tDescription\t" << (cfg.sample_given ? cfg.sample_name : std::string("Counts")) << endl;
        exonReport << std::fixed;
        for (size_t e = 0; e < ann.exon_list.size(); ++e)
            exonReport << ann.exon_list[e] << "\t" << gene_name_of(ann, ann.exon_list[e]) << "\t" << r.exon_reads[e] << endl;
    }
    // ---- metrics.tsv (:523-658) ---------------------------------------------------------------------------------------
    std::ofstream output(base + ".metrics.tsv");
    output << "Sample\t" << cfg.sample_name << endl;
    output << "Mapping Rate\t" << frac(RSQC_C_MAPPED_READS, RSQC_C_UNIQUE_VENDOR_PASSED) << endl;
    output << "Unique Rate of Mapped\t" << frac(RSQC_C_MAPPED_UNIQUE_READS, RSQC_C_MAPPED_READS) << endl;
    output << "Duplicate Rate of Mapped\t" << frac(RSQC_C_MAPPED_DUPLICATE_READS, RSQC_C_MAPPED_READS) << endl;
    output << "Duplicate Rate of Mapped, excluding Globins\t" << frac(RSQC_C_NON_GLOBIN_DUPLICATE_READS, RSQC_C_NON_GLOBIN_READS) << endl;
    output << "Base Mismatch\t" << frac(RSQC_C_MISMATCHED_BASES, RSQC_C_TOTAL_BASES) << endl;
    output << "End 1 Mapping Rate\t" << 2.0 * frac(RSQC_C_END1_MAPPED_READS, RSQC_C_UNIQUE_VENDOR_PASSED) << endl;
    output << "End 2 Mapping Rate\t" << 2.0 * frac(RSQC_C_END2_MAPPED_READS, RSQC_C_UNIQUE_VENDOR_PASSED) << endl;
    output << "End 1 Mismatch Rate\t" << frac(RSQC_C_END1_MISMATCHES, RSQC_C_END1_BASES) << endl;
    output << "End 2 Mismatch Rate\t" << frac(RSQC_C_END2_MISMATCHES, RSQC_C_END2_BASES) << endl;
    output << "Expression Profiling Efficiency\t" << frac(RSQC_C_EXONIC_READS, RSQC_C_UNIQUE_VENDOR_PASSED) << endl;
    output << "High Quality Rate\t" << frac(RSQC_C_HIGH_QUALITY_READS, RSQC_C_MAPPED_READS) << endl;
    output << "Exonic Rate\t" << frac(RSQC_C_EXONIC_READS, RSQC_C_MAPPED_READS) << endl;
    output << "Intronic Rate\t" << frac(RSQC_C_INTRONIC_READS, RSQC_C_MAPPED_READS) << endl;
    output << "Intergenic Rate\t" << frac(RSQC_C_INTERGENIC_READS, RSQC_C_MAPPED_READS) << endl;
    output << "Intragenic Rate\t" << frac(RSQC_C_INTRAGENIC_READS, RSQC_C_MAPPED_READS) << endl;
    output << "Ambiguous Alignment Rate\t" << frac(RSQC_C_AMBIGUOUS_READS, RSQC_C_MAPPED_READS) << endl;
    output << "High Quality Exonic Rate\t" << frac(RSQC_C_HQ_EXONIC_READS, RSQC_C_HIGH_QUALITY_READS) << endl;
    output << "High Quality Intronic Rate\t" << frac(RSQC_C_HQ_INTRONIC_READS, RSQC_C_HIGH_QUALITY_READS) << endl;
    output << "High Quality Intergenic Rate\t" << frac(RSQC_C_HQ_INTERGENIC_READS, RSQC_C_HIGH_QUALITY_READS) << endl;
    output << "High Quality Intragenic Rate\t" << frac(RSQC_C_HQ_INTRAGENIC_READS, RSQC_C_HIGH_QUALITY_READS) << endl;
    output << "High Quality Ambiguous Alignment Rate\t" << frac(RSQC_C_HQ_AMBIGUOUS_READS, RSQC_C_HIGH_QUALITY_READS) << endl;
    output << "Discard Rate\t" << static_cast<double>(cnt(RSQC_C_MAPPED_READS) - cnt(RSQC_C_READS_USED)) / cnt(RSQC_C_MAPPED_READS) << endl;
    output << "rRNA Rate\t" << frac(RSQC_C_RRNA_READS, RSQC_C_MAPPED_READS) << endl;
    output << "End 1 Sense Rate\t" << static_cast<double>(cnt(RSQC_C_END1_SENSE)) / (cnt(RSQC_C_END1_SENSE) + cnt(RSQC_C_END1_ANTISENSE)) << endl;
    output << "End 2 Sense Rate\t" << static_cast<double>(cnt(RSQC_C_END2_SENSE)) / (cnt(RSQC_C_END2_SENSE) + cnt(RSQC_C_END2_ANTISENSE)) << endl;
    output << "Avg. Splits per Read\t" << frac(RSQC_C_ALIGNMENT_BLOCKS, RSQC_C_MAPPED_READS) - 1.0 << endl;
    // operator<<(ofstream&, Metrics&), src/Metrics.cpp:342-412
    {
        static const int keys[] = {RSQC_C_END1_ANTISENSE, RSQC_C_END2_ANTISENSE, RSQC_C_END1_BASES, RSQC_C_END2_BASES,
                                   RSQC_C_END1_MAPPED_READS, RSQC_C_END2_MAPPED_READS, RSQC_C_END1_MISMATCHES, RSQC_C_END2_MISMATCHES,
                                   RSQC_C_END1_SENSE, RSQC_C_END2_SENSE, RSQC_C_EXONIC_READS, RSQC_C_FAILED_VENDOR_QC,
                                   RSQC_C_HIGH_QUALITY_READS, RSQC_C_INTERGENIC_READS, RSQC_C_INTRAGENIC_READS, RSQC_C_AMBIGUOUS_READS,
                                   RSQC_C_INTRONIC_READS, RSQC_C_LOW_MAPPING_QUALITY, RSQC_C_LOW_QUALITY_READS,
                                   RSQC_C_MAPPED_DUPLICATE_READS, RSQC_C_MAPPED_READS, RSQC_C_MAPPED_UNIQUE_READS,
                                   RSQC_C_MISMATCHED_BASES, RSQC_C_NON_GLOBIN_READS, RSQC_C_NON_GLOBIN_DUPLICATE_READS,
                                   RSQC_C_READS_USED, RSQC_C_RRNA_READS, RSQC_C_SPLIT_READS /* printed only when non-zero, src/Metrics.cpp:398 */,
                                   RSQC_C_TOTAL_BASES, RSQC_C_TOTAL_MAPPED_PAIRS, RSQC_C_UNIQUE_VENDOR_PASSED, RSQC_C_UNPAIRED_READS};
        output << "Total Alignments\t" << cnt(RSQC_C_TOTAL_ALIGNMENTS) << endl;
        output << "Alternative Alignments\t" << cnt(RSQC_C_ALTERNATIVE_ALIGNMENTS) << endl;
        output << "Supplementary Alignments\t" << cnt(RSQC_C_SUPPLEMENTARY_ALIGNMENTS) << endl;
        output << "Total Reads\t" << cnt(RSQC_C_TOTAL_ALIGNMENTS) - cnt(RSQC_C_ALTERNATIVE_ALIGNMENTS) - cnt(RSQC_C_SUPPLEMENTARY_ALIGNMENTS) << endl;
        output << "Chimeric Fragments\t";
        if (cnt(RSQC_C_CHIMERIC_TAG)) {
            output << cnt(RSQC_C_CHIMERIC_TAG) << endl;
            output << "Chimeric Alignment Rate\t" << frac(RSQC_C_CHIMERIC_TAG, RSQC_C_TOTAL_MAPPED_PAIRS) << endl;
        } else {
            output << cnt(RSQC_C_CHIMERIC_AUTO) << endl;
            output << "Chimeric Alignment Rate\t" << frac(RSQC_C_CHIMERIC_AUTO, RSQC_C_TOTAL_MAPPED_PAIRS) << endl;
        }
        for (int k : keys) if (k != RSQC_C_SPLIT_READS || cnt(k)) output << rsqc_counter_name(k) << "\t" << cnt(k) << endl;
        // "Filtered by tag: X" entries exist only for tags that fired, in std::map (string) order
        std::map<std::string, unsigned long> filtered;
        for (size_t t = 0; t < cfg.filter_tags.size() && t < RSQC_MAX_FILTER_TAGS; ++t)
            if (cnt(RSQC_C_FILTERED_TAG0 + (int)t)) filtered["Filtered by tag: " + cfg.filter_tags[t]] += cnt(RSQC_C_FILTERED_TAG0 + (int)t);
        for (auto &kv : filtered) output << kv.first << "\t" << kv.second << endl;
    }
    output << "Read Length\t" << r.read_length << endl;
    output << "Genes Detected\t" << genesDetected << endl;
    output << "Estimated Library Complexity\t" << minReads << endl;
    output << "Genes used in 3' bias\t" << biasGenes << endl;
    output << "Mean 3' bias\t" << ratioAvg << endl;
    output << "Median 3' bias\t" << ratioMedian << endl;
    output << "3' bias Std\t" << ratioStd << endl;
    output << "3' bias MAD_Std\t" << ratioMedDev << endl;
    output << "3' Bias, 25th Percentile\t" << ratio25 << endl;
    output << "3' Bias, 75th Percentile\t" << ratio75 << endl;
    if (r.n_fragment_sizes) {                                                   // :570-607
        double fragmentAvg = 0.0, fragmentStd = 0.0, fragmentMedDev = 0.0;
        std::vector<double> expansion;
        for (uint32_t i = 0; i < r.n_fragment_sizes; ++i)
            for (unsigned long k = 0; k < r.fragment_count[i]; ++k) expansion.push_back((double)r.fragment_size[i]);
        std::sort(expansion.begin(), expansion.end());
        const double size = static_cast<double>(expansion.size());
        fragmentMed = compute_median(expansion);
        std::ofstream fragmentList(base + ".fragmentSizes.txt");
        fragmentList << "Fragment Size\tCount" << endl;
        std::vector<double> deviations;
        for (uint32_t i = 0; i < r.n_fragment_sizes; ++i) {
            fragmentList << r.fragment_size[i] << "\t" << r.fragment_count[i] << endl;
            fragmentAvg += static_cast<double>(r.fragment_size[i] * (long long)r.fragment_count[i]) / size;
            const double deviation = std::fabs(static_cast<double>(r.fragment_size[i]) - fragmentMed);
            for (unsigned long k = 0; k < r.fragment_count[i]; ++k) deviations.push_back(deviation);
        }
        std::sort(deviations.begin(), deviations.end());
        fragmentMedDev = compute_median(deviations) * MAD_FACTOR;
        for (uint32_t i = 0; i < r.n_fragment_sizes; ++i)
            for (unsigned long k = 0; k < r.fragment_count[i]; ++k)
                fragmentStd += std::pow(static_cast<double>(r.fragment_size[i]) - fragmentAvg, 2.0) / size;
        fragmentStd = std::pow(fragmentStd, 0.5);
        output << "Average Fragment Length\t" << fragmentAvg << endl;
        output << "Fragment Length Median\t" << fragmentMed << endl;
        output << "Fragment Length Std\t" << fragmentStd << endl;
        output << "Fragment Length MAD_Std\t" << fragmentMedDev << endl;
    }
    {                                                                           // :609-658
        std::vector<double> means, stdDevs, cvs;
        for (size_t g = 0; g < ann.gene_list.size(); ++g) if (r.gene_cov_valid[g]) {
            means.push_back(r.gene_cov_mean[g]); stdDevs.push_back(r.gene_cov_std[g]);
            const double cv = r.gene_cov_cv[g];
            if (!(std::isnan(cv) || std::isinf(cv))) cvs.push_back(cv);
        }
        std::sort(means.begin(), means.end()); std::sort(stdDevs.begin(), stdDevs.end()); std::sort(cvs.begin(), cvs.end());
        output << "Median of Avg Transcript Coverage\t" << compute_median(means) << endl;      // throws when no gene survives the mask (Q15)
        output << "Median of Transcript Coverage Std\t" << compute_median(stdDevs) << endl;
        output << "Median of Transcript Coverage CV\t" << (cvs.size() ? compute_median(cvs) : 0.0) << endl;
        std::map<std::string, double> exonCoverage;                               // std::map<string, ExonCoverage>: id order
        for (size_t e = 0; e < ann.exon_list.size(); ++e) if (r.exon_cv_valid[e]) exonCoverage[ann.exon_list[e]] = r.exon_cv[e];
        std::ofstream cvReport(base + ".exon_cv.tsv");
        cvReport << "Exon ID\tExon CV";
        if (r.have_reference) cvReport << "\tGC Content";                          // :633-634
        cvReport << endl;
        std::vector<double> totalExonCV;
        if (r.have_reference) {                                                   // :636-640 (ExonCoverage{cv, gc})
            std::map<std::string, double> exonGC;
            for (size_t e = 0; e < ann.exon_list.size(); ++e) if (r.exon_cv_valid[e]) exonGC[ann.exon_list[e]] = r.exon_gc[e];
            for (auto &kv : exonCoverage) { cvReport << kv.first << "\t" << kv.second << "\t" << exonGC[kv.first] << endl; totalExonCV.push_back(kv.second); }
        } else
        for (auto &kv : exonCoverage) { cvReport << kv.first << "\t" << kv.second << endl; totalExonCV.push_back(kv.second); }
        double a, m, s, d;
        get_statistics(totalExonCV, a, m, s, d);
        output << "Median Exon CV\t" << m << endl;
        output << "Exon CV MAD\t" << d << endl;
    }
    if (r.have_reference) {                                                       // :660-674
        std::ofstream gcReport(base + ".gc_content.tsv");
        gcReport << "Content Bin\tCount" << endl;
        // getAdvancedStatistics (src/Metrics.h:188-206) over the bin index repeated gcBins[i] times, ascending
        double avg = 0.0, m2 = 0.0, m3 = 0.0, m4 = 0.0, count = 0.0;
        bool any = false;
        for (unsigned int i = 0; i < RSQC_GC_BINS; ++i) {
            gcReport << (double)i / 100.0 << "\t" << r.gc_bins[i] << endl;
            for (uint64_t j = 0; j < r.gc_bins[i]; ++j) {
                any = true;
                const double prev_count = count++;
                const double delta = static_cast<double>(i) - avg;
                const double delta_n = delta / count;
                const double delta_n2 = delta_n * delta_n;
                const double t = delta * delta_n * prev_count;
                avg += delta_n;
                m4 += t * delta_n2 * (count * count - 3 * count + 3) + 6 * delta_n2 * m2 - 4 * delta_n * m3;
                m3 += t * delta_n * (count - 2) - 3 * delta_n * m2;
                m2 += t;
            }
        }
        double sd = NAN, skew = NAN, kurt = NAN;
        if (any) { sd = pow(m2 / count, 0.5); skew = m3 / count / pow(sd, 3.0); kurt = (count * m4) / (m2 * m2) - 3; }
        else avg = NAN;
        output << "Fragment GC Content Mean\t" << (double)avg / 100.0 << endl;
        output << "Fragment GC Content Std\t" << (double)sd / 100.0 << endl;
        output << "Fragment GC Content Skewness\t" << skew << endl;
        output << "Fragment GC Content Kurtosis\t" << kurt << endl;
    }
    output.close();
}

}  // namespace rsqc_host
