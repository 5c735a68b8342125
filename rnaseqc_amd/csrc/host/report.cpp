// report.cpp -- the report tail of the CLI on top of rsqc_results.  What must be reproduced byte for byte is the reference's
// OUTPUT (src/RNASeQC.cpp:397-676 and operator<<(ofstream&, Metrics&), src/Metrics.cpp:342-412): labels, their order, the
// stream formatting of every number and the order of the floating-point operations behind it.  How it is produced is this
// file's own: the rows of metrics.tsv are DATA (a label, a formula kind, counter operands), the four GCT tables go through one
// emitter, the summaries are small value types, and the big tables are written by their own threads beside metrics.tsv.
#include "report.hpp"

#include <algorithm>
#include <climits>
#include <cmath>
#include <fstream>
#include <functional>
#include <iostream>
#include <map>
#include <stdexcept>
#include <thread>

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

namespace {

// '\n', not std::endl: the same bytes without a flush (a write syscall) per line; the files are flushed on close
constexpr char NL = '\n';

// ---------------------------------------------------------------------------------------------------------------------------
// metrics.tsv, part 1: the rate rows (src/RNASeQC.cpp:526-552) as data.  `kind` says how the two counters combine.
enum class Rate { Share /* a / b */, TwiceShare /* 2 a / b */, Lost /* (a - b) / a */, OfPair /* a / (a + b) */, ShareLessOne /* a / b - 1 */ };
struct RateRow { const char *label; Rate kind; int a, b; };
const RateRow RATE_ROWS[] = {
    {"Mapping Rate", Rate::Share, RSQC_C_MAPPED_READS, RSQC_C_UNIQUE_VENDOR_PASSED},
    {"Unique Rate of Mapped", Rate::Share, RSQC_C_MAPPED_UNIQUE_READS, RSQC_C_MAPPED_READS},
    {"Duplicate Rate of Mapped", Rate::Share, RSQC_C_MAPPED_DUPLICATE_READS, RSQC_C_MAPPED_READS},
    {"Duplicate Rate of Mapped, excluding Globins", Rate::Share, RSQC_C_NON_GLOBIN_DUPLICATE_READS, RSQC_C_NON_GLOBIN_READS},
    {"Base Mismatch", Rate::Share, RSQC_C_MISMATCHED_BASES, RSQC_C_TOTAL_BASES},
    {"End 1 Mapping Rate", Rate::TwiceShare, RSQC_C_END1_MAPPED_READS, RSQC_C_UNIQUE_VENDOR_PASSED},
    {"End 2 Mapping Rate", Rate::TwiceShare, RSQC_C_END2_MAPPED_READS, RSQC_C_UNIQUE_VENDOR_PASSED},
    {"End 1 Mismatch Rate", Rate::Share, RSQC_C_END1_MISMATCHES, RSQC_C_END1_BASES},
    {"End 2 Mismatch Rate", Rate::Share, RSQC_C_END2_MISMATCHES, RSQC_C_END2_BASES},
    {"Expression Profiling Efficiency", Rate::Share, RSQC_C_EXONIC_READS, RSQC_C_UNIQUE_VENDOR_PASSED},
    {"High Quality Rate", Rate::Share, RSQC_C_HIGH_QUALITY_READS, RSQC_C_MAPPED_READS},
    {"Exonic Rate", Rate::Share, RSQC_C_EXONIC_READS, RSQC_C_MAPPED_READS},
    {"Intronic Rate", Rate::Share, RSQC_C_INTRONIC_READS, RSQC_C_MAPPED_READS},
    {"Intergenic Rate", Rate::Share, RSQC_C_INTERGENIC_READS, RSQC_C_MAPPED_READS},
    {"Intragenic Rate", Rate::Share, RSQC_C_INTRAGENIC_READS, RSQC_C_MAPPED_READS},
    {"Ambiguous Alignment Rate", Rate::Share, RSQC_C_AMBIGUOUS_READS, RSQC_C_MAPPED_READS},
    {"High Quality Exonic Rate", Rate::Share, RSQC_C_HQ_EXONIC_READS, RSQC_C_HIGH_QUALITY_READS},
    {"High Quality Intronic Rate", Rate::Share, RSQC_C_HQ_INTRONIC_READS, RSQC_C_HIGH_QUALITY_READS},
    {"High Quality Intergenic Rate", Rate::Share, RSQC_C_HQ_INTERGENIC_READS, RSQC_C_HIGH_QUALITY_READS},
    {"High Quality Intragenic Rate", Rate::Share, RSQC_C_HQ_INTRAGENIC_READS, RSQC_C_HIGH_QUALITY_READS},
    {"High Quality Ambiguous Alignment Rate", Rate::Share, RSQC_C_HQ_AMBIGUOUS_READS, RSQC_C_HIGH_QUALITY_READS},
    {"Discard Rate", Rate::Lost, RSQC_C_MAPPED_READS, RSQC_C_READS_USED},
    {"rRNA Rate", Rate::Share, RSQC_C_RRNA_READS, RSQC_C_MAPPED_READS},
    {"End 1 Sense Rate", Rate::OfPair, RSQC_C_END1_SENSE, RSQC_C_END1_ANTISENSE},
    {"End 2 Sense Rate", Rate::OfPair, RSQC_C_END2_SENSE, RSQC_C_END2_ANTISENSE},
    {"Avg. Splits per Read", Rate::ShareLessOne, RSQC_C_ALIGNMENT_BLOCKS, RSQC_C_MAPPED_READS},
};
// part 2: the counters that the reference prints from its std::map, i.e. in the string order of their names (src/Metrics.cpp:
// 342-412); rsqc_counter_name supplies the label.  A row flagged `only_if_set` is skipped while its counter is 0 (:398).
struct CountRow { int counter; bool only_if_set; };
const CountRow COUNT_ROWS[] = {
    {RSQC_C_END1_ANTISENSE, false}, {RSQC_C_END2_ANTISENSE, false}, {RSQC_C_END1_BASES, false}, {RSQC_C_END2_BASES, false},
    {RSQC_C_END1_MAPPED_READS, false}, {RSQC_C_END2_MAPPED_READS, false}, {RSQC_C_END1_MISMATCHES, false}, {RSQC_C_END2_MISMATCHES, false},
    {RSQC_C_END1_SENSE, false}, {RSQC_C_END2_SENSE, false}, {RSQC_C_EXONIC_READS, false}, {RSQC_C_FAILED_VENDOR_QC, false},
    {RSQC_C_HIGH_QUALITY_READS, false}, {RSQC_C_INTERGENIC_READS, false}, {RSQC_C_INTRAGENIC_READS, false}, {RSQC_C_AMBIGUOUS_READS, false},
    {RSQC_C_INTRONIC_READS, false}, {RSQC_C_LOW_MAPPING_QUALITY, false}, {RSQC_C_LOW_QUALITY_READS, false},
    {RSQC_C_MAPPED_DUPLICATE_READS, false}, {RSQC_C_MAPPED_READS, false}, {RSQC_C_MAPPED_UNIQUE_READS, false},
    {RSQC_C_MISMATCHED_BASES, false}, {RSQC_C_NON_GLOBIN_READS, false}, {RSQC_C_NON_GLOBIN_DUPLICATE_READS, false},
    {RSQC_C_READS_USED, false}, {RSQC_C_RRNA_READS, false}, {RSQC_C_SPLIT_READS, true},
    {RSQC_C_TOTAL_BASES, false}, {RSQC_C_TOTAL_MAPPED_PAIRS, false}, {RSQC_C_UNIQUE_VENDOR_PASSED, false}, {RSQC_C_UNPAIRED_READS, false},
};

struct Tally {                                    // the run's scalar counters
    const rsqc_results &r;
    unsigned long operator[](int c) const { return (unsigned long)r.counters[c]; }
    double share(int a, int b) const { return static_cast<double>((*this)[a]) / (*this)[b]; }       // Metrics::frac
    double rate(const RateRow &row) const {
        switch (row.kind) {
            case Rate::Share: return share(row.a, row.b);
            case Rate::TwiceShare: return 2.0 * share(row.a, row.b);
            case Rate::Lost: return static_cast<double>((*this)[row.a] - (*this)[row.b]) / (*this)[row.a];
            case Rate::OfPair: return static_cast<double>((*this)[row.a]) / ((*this)[row.a] + (*this)[row.b]);
            case Rate::ShareLessOne: return share(row.a, row.b) - 1.0;
        }
        return NAN;
    }
};

// ---------------------------------------------------------------------------------------------------------------------------
// One GCT table: "#1.2", "<rows>\t1", "Name\tDescription\t<column>", then id, description and the value of every row.  The
// declared row count is a parameter because exon_reads.gct declares the entries of the reference's exonCounts map, not its rows
// (src/RNASeQC.cpp:513).  Four of these are written per run; each runs on its own thread.
struct GctSpec {
    std::string path, column;
    size_t declared_rows;
    bool fixed_notation;
    const std::vector<std::string> *ids;
    std::function<void(std::ostream &, size_t)> put_value;
};
void emit_gct(const GctSpec &t, const Annotation &ann) {
    std::ofstream f(t.path);
    f << "#1.2" << NL << t.declared_rows << "\t1" << NL << "Name\tDescription\t" << t.column << NL;
    if (t.fixed_notation) f << std::fixed;
    for (size_t i = 0; i < t.ids->size(); ++i) {
        const std::string &id = (*t.ids)[i];
        f << id << '\t' << ann.gene_name(id) << '\t';
        t.put_value(f, i);
        f << NL;
    }
}

// threads that are always joined, an exception of the caller included (the reference has written its tables by the time
// computeMedian throws: the files must be complete then, too)
struct Crew {
    std::vector<std::thread> t;
    template <class F> void go(F &&f) { t.emplace_back(std::forward<F>(f)); }
    void wait() { for (auto &x : t) if (x.joinable()) x.join(); }
    ~Crew() { wait(); }
};

// ---------------------------------------------------------------------------------------------------------------------------
// 3'/5' bias over the genes with any bias coverage (BiasCounter::getBias, src/Metrics.cpp:239-249; summary :477-508)
struct BiasSummary { unsigned genes = 0; double mean = 0.0, median = 0.0, sd = 0.0, mad = 0.0, q25 = 0.0, q75 = 0.0; };
double quartile_like_reference(const std::vector<double> &sorted, double fraction) {
    // the reference indexes past the end for 2..4 values (undefined behaviour there): clamped here
    auto at = [&](double i) { const size_t k = (size_t)static_cast<int>(i); return sorted[std::min(k, sorted.size() - 1)]; };
    double index = fraction * sorted.size();
    const bool between = !(index > std::floor(index));
    index = std::ceil(index);
    return between ? (at(index) + at(index + 1)) / 2.0 : at(index);
}
BiasSummary summarize_bias(const rsqc_results &r, size_t n_genes) {
    BiasSummary s;
    std::vector<double> share3;
    for (size_t g = 0; g < n_genes; ++g) {
        const double five = (double)r.bias_five[g], three = (double)r.bias_three[g];
        if (five + three > 0.0) { ++s.genes; share3.push_back(three / (five + three)); }
    }
    if (share3.size() > 1) {
        get_statistics(share3, s.mean, s.median, s.sd, s.mad);          // (sorts)
        s.q25 = quartile_like_reference(share3, .25);
        s.q75 = quartile_like_reference(share3, .75);
    }
    return s;
}

// fragment sizes (src/RNASeQC.cpp:570-607): statistics over the histogram, every size taken `count` times
struct FragmentSummary { double mean = 0.0, median = 0.0, sd = 0.0, mad = 0.0; };
FragmentSummary summarize_fragments(const rsqc_results &r) {
    FragmentSummary s;
    std::vector<double> all;
    for (uint32_t i = 0; i < r.n_fragment_sizes; ++i) all.insert(all.end(), (size_t)r.fragment_count[i], (double)r.fragment_size[i]);
    std::sort(all.begin(), all.end());
    const double n = static_cast<double>(all.size());
    s.median = compute_median(all);
    std::vector<double> away;
    for (uint32_t i = 0; i < r.n_fragment_sizes; ++i) {
        s.mean += static_cast<double>(r.fragment_size[i] * (long long)r.fragment_count[i]) / n;
        away.insert(away.end(), (size_t)r.fragment_count[i], std::fabs(static_cast<double>(r.fragment_size[i]) - s.median));
    }
    std::sort(away.begin(), away.end());
    s.mad = compute_median(away) * MAD_FACTOR;
    for (uint32_t i = 0; i < r.n_fragment_sizes; ++i)
        for (unsigned long k = 0; k < r.fragment_count[i]; ++k) s.sd += std::pow(static_cast<double>(r.fragment_size[i]) - s.mean, 2.0) / n;
    s.sd = std::pow(s.sd, 0.5);
    return s;
}

// one-pass central moments in the update order of the reference's getAdvancedStatistics (src/Metrics.h:188-206): the printed
// digits depend on that order
struct Moments {
    double n = 0.0, mean = 0.0, m2 = 0.0, m3 = 0.0, m4 = 0.0;
    void push(double x) {
        const double before = n++;
        const double d = x - mean, dn = d / n, dn2 = dn * dn, t = d * dn * before;
        mean += dn;
        m4 += t * dn2 * (n * n - 3 * n + 3) + 6 * dn2 * m2 - 4 * dn * m3;
        m3 += t * dn * (n - 2) - 3 * dn * m2;
        m2 += t;
    }
    double sd() const { return pow(m2 / n, 0.5); }
    double skewness() const { return m3 / n / pow(sd(), 3.0); }
    double kurtosis() const { return (n * m4) / (m2 * m2) - 3; }
};

// coverage.tsv: one row per gene in the order the reference retires genes -- contigs as the BAM visited them, the rest at end
// of file in chromosomeMap order (src/RNASeQC.cpp:385-386)
void emit_coverage_table(const std::string &path, const Annotation &ann, const rsqc_results &r, const std::vector<int> &visited) {
    std::ofstream f(path);
    f << "gene_id\tcoverage_mean\tcoverage_std\tcoverage_CV" << NL;
    std::vector<char> seen(ann.contig_names.size(), 0);
    std::vector<int> order;
    for (int c : visited) if (c >= 0 && c < (int)seen.size() && !seen[(size_t)c]) { seen[(size_t)c] = 1; order.push_back(c); }
    std::vector<std::pair<int, int>> unvisited;
    for (size_t c = 0; c < seen.size(); ++c) if (!seen[c]) unvisited.emplace_back(ann.chrom_of_contig[c], (int)c);
    std::sort(unvisited.begin(), unvisited.end());
    for (auto &p : unvisited) order.push_back(p.second);
    for (int c : order) for (uint32_t g : ann.genes_by_contig[(size_t)c]) {
        f << ann.gene_list[g] << '\t';
        if (r.gene_cov_valid[g]) f << r.gene_cov_mean[g] << '\t' << r.gene_cov_std[g] << '\t' << r.gene_cov_cv[g] << NL;
        else f << "0\t0\tnan" << NL;
    }
}

}  // namespace

void write_reports(const ReportConfig &cfg, Annotation &ann, const rsqc_results &r,
                   const std::vector<int> &contig_visit_order) {
    const std::string stem = cfg.output_dir + "/" + cfg.sample_name;
    const Tally tally{r};
    const size_t n_genes = ann.gene_list.size();
    auto column = [&](const char *fallback) { return cfg.sample_given ? cfg.sample_name : std::string(fallback); };

    // ---- the three gene tables and the exon table (src/RNASeQC.cpp:419-475, 509-521): values first, then one thread per file --
    std::vector<double> abundance(n_genes, 0.0);                  // RPKM, or TPM after scaling
    {
        const double per_million_exonic = static_cast<double>(tally[RSQC_C_EXONIC_READS]) / 1000000.0;
        double tpm_total = 0.0;
        for (size_t g = 0; g < n_genes; ++g) {
            const double reads = (double)r.gene_reads[g], coding = static_cast<double>(ann.coding_length(ann.gene_list[g]));
            if (cfg.use_rpkm) abundance[g] = (1000.0 * reads / per_million_exonic) / coding;
            else { abundance[g] = (1000.0 * reads) / coding; tpm_total += abundance[g]; }
        }
        if (!cfg.use_rpkm) { tpm_total /= 1000000.0; for (double &v : abundance) v = v / tpm_total; }
    }
    unsigned genes_detected = 0;
    for (size_t g = 0; g < n_genes; ++g) if ((double)r.gene_unique[g] >= cfg.detection_threshold) ++genes_detected;
    size_t exon_entries = 0;                                      // exonCounts.size() before the writer's look-ups add entries (Q7)
    for (int e = 0; e < r.n_exons; ++e) exon_entries += r.exon_hit[e] ? 1 : 0;
    const GctSpec tables[] = {
        {stem + ".gene_reads.gct", column("Counts"), n_genes, false, &ann.gene_list,
         [&](std::ostream &o, size_t g) { o << static_cast<long>((double)r.gene_reads[g]); }},
        {stem + ".gene_" + (cfg.use_rpkm ? "rpkm" : "tpm") + ".gct", column(cfg.use_rpkm ? "RPKM" : "TPM"), n_genes, true, &ann.gene_list,
         [&](std::ostream &o, size_t g) { o << abundance[g]; }},
        {stem + ".gene_fragments.gct", column("Fragments"), n_genes, false, &ann.gene_list,
         [&](std::ostream &o, size_t g) { o << static_cast<long>((double)r.gene_fragments[g]); }},
        {stem + ".exon_reads.gct", column("Counts"), exon_entries, true, &ann.exon_list,
         [&](std::ostream &o, size_t e) { o << r.exon_reads[e]; }},
    };
    // declared AFTER everything its threads read (`abundance`, `tables`): when a row below throws (Q15: no gene survives the
    // coverage mask), the unwinding joins the writers (~Crew) before those are destroyed, and the files they write are complete
    Crew crew;
    if (cfg.write_coverage) crew.go([&] { emit_coverage_table(stem + ".coverage.tsv", ann, r, contig_visit_order); });
    for (const GctSpec &t : tables) crew.go([&t, &ann] { emit_gct(t, ann); });

    // ---- metrics.tsv, in the reference's row order; a throw below leaves the rows written so far, as it does there --------
    const unsigned complexity = library_complexity((double)tally[RSQC_C_DUPLICATE_PAIRS], (double)tally[RSQC_C_UNIQUE_FRAGMENTS]);
    const BiasSummary bias = summarize_bias(r, n_genes);
    std::ofstream m(stem + ".metrics.tsv");
    auto row = [&m](const char *label, auto value) { m << label << '\t' << value << NL; };
    row("Sample", cfg.sample_name);
    for (const RateRow &rr : RATE_ROWS) row(rr.label, tally.rate(rr));
    row("Total Alignments", tally[RSQC_C_TOTAL_ALIGNMENTS]);
    row("Alternative Alignments", tally[RSQC_C_ALTERNATIVE_ALIGNMENTS]);
    row("Supplementary Alignments", tally[RSQC_C_SUPPLEMENTARY_ALIGNMENTS]);
    row("Total Reads", tally[RSQC_C_TOTAL_ALIGNMENTS] - tally[RSQC_C_ALTERNATIVE_ALIGNMENTS] - tally[RSQC_C_SUPPLEMENTARY_ALIGNMENTS]);
    {   // chimeric pairs: from the aligner's tag when any record carried one, else from mate contig / distance
        const int which = tally[RSQC_C_CHIMERIC_TAG] ? RSQC_C_CHIMERIC_TAG : RSQC_C_CHIMERIC_AUTO;
        row("Chimeric Fragments", tally[which]);
        row("Chimeric Alignment Rate", tally.share(which, RSQC_C_TOTAL_MAPPED_PAIRS));
    }
    for (const CountRow &cr : COUNT_ROWS) if (!cr.only_if_set || tally[cr.counter]) row(rsqc_counter_name(cr.counter), tally[cr.counter]);
    {   // "Filtered by tag: X" exists only for tags that fired; the reference's map prints them in string order
        std::map<std::string, unsigned long> fired;
        for (size_t t = 0; t < cfg.filter_tags.size() && t < RSQC_MAX_FILTER_TAGS; ++t)
            if (tally[RSQC_C_FILTERED_TAG0 + (int)t]) fired["Filtered by tag: " + cfg.filter_tags[t]] += tally[RSQC_C_FILTERED_TAG0 + (int)t];
        for (auto &kv : fired) row(kv.first.c_str(), kv.second);
    }
    row("Read Length", r.read_length);
    row("Genes Detected", genes_detected);
    row("Estimated Library Complexity", complexity);
    row("Genes used in 3' bias", bias.genes);
    row("Mean 3' bias", bias.mean);
    row("Median 3' bias", bias.median);
    row("3' bias Std", bias.sd);
    row("3' bias MAD_Std", bias.mad);
    row("3' Bias, 25th Percentile", bias.q25);
    row("3' Bias, 75th Percentile", bias.q75);

    if (r.n_fragment_sizes) {                                                     // src/RNASeQC.cpp:570-607
        const FragmentSummary fs = summarize_fragments(r);
        std::ofstream list(stem + ".fragmentSizes.txt");
        list << "Fragment Size\tCount" << NL;
        for (uint32_t i = 0; i < r.n_fragment_sizes; ++i) list << r.fragment_size[i] << '\t' << r.fragment_count[i] << NL;
        row("Average Fragment Length", fs.mean);
        row("Fragment Length Median", fs.median);
        row("Fragment Length Std", fs.sd);
        row("Fragment Length MAD_Std", fs.mad);
    }

    {                                                                             // coverage summary, src/RNASeQC.cpp:609-658
        std::vector<double> mean, sd, cv;
        for (size_t g = 0; g < n_genes; ++g) if (r.gene_cov_valid[g]) {
            mean.push_back(r.gene_cov_mean[g]); sd.push_back(r.gene_cov_std[g]);
            if (std::isfinite(r.gene_cov_cv[g])) cv.push_back(r.gene_cov_cv[g]);
        }
        std::sort(mean.begin(), mean.end()); std::sort(sd.begin(), sd.end()); std::sort(cv.begin(), cv.end());
        row("Median of Avg Transcript Coverage", compute_median(mean));           // throws when no gene survived the mask (Q15)
        row("Median of Transcript Coverage Std", compute_median(sd));
        row("Median of Transcript Coverage CV", cv.size() ? compute_median(cv) : 0.0);
        // exon_cv.tsv in exon-id (string) order, with the exon's GC content when a FASTA was given (:633-640)
        std::map<std::string, size_t> by_id;
        for (size_t e = 0; e < ann.exon_list.size(); ++e) if (r.exon_cv_valid[e]) by_id[ann.exon_list[e]] = e;
        std::ofstream table(stem + ".exon_cv.tsv");
        table << "Exon ID\tExon CV" << (r.have_reference ? "\tGC Content" : "") << NL;
        std::vector<double> all_cv;
        for (auto &kv : by_id) {
            table << kv.first << '\t' << r.exon_cv[kv.second];
            if (r.have_reference) table << '\t' << r.exon_gc[kv.second];
            table << NL;
            all_cv.push_back(r.exon_cv[kv.second]);
        }
        double avg, med, dev, mad;
        get_statistics(all_cv, avg, med, dev, mad);
        row("Median Exon CV", med);
        row("Exon CV MAD", mad);
    }

    if (r.have_reference) {                                                       // fragment GC content, src/RNASeQC.cpp:660-674
        std::ofstream bins(stem + ".gc_content.tsv");
        bins << "Content Bin\tCount" << NL;
        Moments mo;
        for (unsigned int i = 0; i < RSQC_GC_BINS; ++i) {
            bins << (double)i / 100.0 << '\t' << r.gc_bins[i] << NL;
            for (uint64_t j = 0; j < r.gc_bins[i]; ++j) mo.push(static_cast<double>(i));
        }
        const bool any = mo.n > 0.0;
        row("Fragment GC Content Mean", (any ? mo.mean : NAN) / 100.0);
        row("Fragment GC Content Std", (any ? mo.sd() : NAN) / 100.0);
        row("Fragment GC Content Skewness", any ? mo.skewness() : NAN);
        row("Fragment GC Content Kurtosis", any ? mo.kurtosis() : NAN);
    }
    m.close();
    crew.wait();
}

}  // namespace rsqc_host
