// report.hpp -- the report tail of the CLI: restates src/RNASeQC.cpp:397-676 and
// operator<<(ofstream&, Metrics&) (src/Metrics.cpp:342-412) on top of rsqc_results.
#pragma once

#include <string>
#include <vector>

#include "gtf.hpp"

namespace rsqc_host {

struct ReportConfig {
    std::string output_dir, sample_name;     // SAMPLENAME (src/RNASeQC.cpp:99)
    bool sample_given = false;               // --sample given: GCT value column is the sample name
    bool use_rpkm = false, write_coverage = false;
    unsigned detection_threshold = 5;
    std::vector<std::string> filter_tags;    // names for "Filtered by tag: X"
};

// quirky computeMedian (src/Metrics.h:147-160) on an ordered sequence; throws std::range_error when empty
double compute_median(const std::vector<double> &sorted);
// getStatistics (src/Metrics.h:166-186): avg, median, std, MAD (sorts its argument)
void get_statistics(std::vector<double> &data, double &avg, double &med, double &sd, double &mad);
// Lander-Waterman search of src/RNASeQC.cpp:398-415 without the 1e9-step loop (same result)
unsigned library_complexity(double duplicates, double unique, double limit = 1e9);

// Writes every output file.  contig_visit_order: boundary contig ids in the order the BAM visited them
// (coverage.tsv row order).  Throws std::range_error exactly where the reference does.
void write_reports(const ReportConfig &cfg, Annotation &ann, const rsqc_results &r,
                   const std::vector<int> &contig_visit_order);

}  // namespace rsqc_host
