/*
 * ref_bed_harness.cpp -- TEST INFRASTRUCTURE.
 *
 * Drives the REFERENCE's own src/BED.cpp (compiled unmodified, in place, by oracle/Makefile into
 * oracle/_ref/libref_metrics.so) so that the product's BED loader (rnaseqc_amd/csrc/host/gtf.cpp, Annotation::load_bed)
 * can be compared with `extractBED` line for line, hostile lines included (short lines, signs, trailing junk, comments,
 * numbers out of range).
 *
 * What is and is not the reference here: BED.cpp and the headers it includes (BED.h, GTF.h, Fasta.h, bioio.hpp) are the
 * reference's, as they lie under /root/reference.  BED.cpp calls ONE symbol of a translation unit that cannot be built in
 * this image (src/Fasta.cpp needs boost::filesystem): rnaseqc::chromosomeMap, the name -> small-integer table
 * (src/Fasta.cpp:17-25).  Its ids are opaque keys -- nothing but their identity is ever looked at -- so the table is this
 * harness's INPUT in the same sense as exonsForGene is ref_metrics_harness.cpp's: ids are handed out in first-sight order
 * and reported back as names.  The loop around extractBED is src/RNASeQC.cpp:185.
 */
#include "BED.h"

#include <cstdint>
#include <cstring>
#include <fstream>
#include <map>
#include <string>
#include <vector>

namespace rnaseqc {
    static std::map<std::string, chrom> g_names;            // first-sight order, 1-based (src/Fasta.cpp:17-25)
    static std::vector<std::string> g_name_of;
    chrom chromosomeMap(std::string chr) {
        auto it = g_names.find(chr);
        if (it != g_names.end()) return it->second;
        const chrom id = (chrom)(g_names.size() + 1u);
        g_names[chr] = id; g_name_of.push_back(chr);
        return id;
    }
}

using namespace rnaseqc;

extern "C" {

// Reads `path` the way src/RNASeQC.cpp:178-186 does.  Returns the number of features (<= cap are written), -1 when the file
// cannot be opened, -2 when extractBED threw bedException (message -> err, features read before it are still written).
// chrom_index[i] = index into the name list (first-sight order); names are returned as one '\n'-joined string.
__attribute__((visibility("default")))
long long ref_bed_read(const char *path, long long cap, int32_t *chrom_index, long long *start, long long *end,
                       char *names, long long names_cap, char *err, long long err_cap) {
    g_names.clear(); g_name_of.clear();
    std::ifstream reader(path);
    if (!reader.is_open()) return -1;
    long long n = 0; bool threw = false;
    try {
        Feature line;
        while (extractBED(reader, line)) {                                  // src/RNASeQC.cpp:185
            if (n < cap) { chrom_index[n] = (int32_t)line.chromosome - 1; start[n] = (long long)line.start; end[n] = (long long)line.end; }
            ++n;
        }
    } catch (bedException &e) {
        threw = true;
        if (err && err_cap > 0) { strncpy(err, e.error.c_str(), (size_t)err_cap - 1); err[err_cap - 1] = 0; }
    }
    std::string joined;
    for (auto &s : g_name_of) { joined += s; joined += '\n'; }
    if (names && names_cap > 0) { strncpy(names, joined.c_str(), (size_t)names_cap - 1); names[names_cap - 1] = 0; }
    return threw ? -2 - n : n;                                              // (-2 - n: threw after n features)
}

}  // extern "C"
