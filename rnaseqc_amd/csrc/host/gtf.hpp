// gtf.hpp -- GTF / BED ingest of the CLI: reproduces the reference's bookkeeping (id assignment,
// ordering, lengths, name table, parse quirks) and flattens it into the boundary's rsqc_annotation.
//   reference: src/GTF.cpp:30-148 (operator>>, parseAttributes), src/RNASeQC.cpp:104-164 (load loop,
//   per-contig stable sort, exonsForGene), src/BED.cpp:18-45, src/Fasta.cpp:17-25 (chromosomeMap)
#pragma once

#include <cstdint>
#include <map>
#include <stdexcept>
#include <string>
#include <unordered_map>
#include <vector>

#include "../../../include/rnaseqc_amd.h"

namespace rsqc_host {

struct GtfError : std::runtime_error { using std::runtime_error::runtime_error; };     // exit code 11
struct BedError : std::runtime_error { using std::runtime_error::runtime_error; };     // exit code 11
struct FileError : std::runtime_error { using std::runtime_error::runtime_error; };    // exit code 10

struct Annotation {
    // chromosomeMap: name -> id in first-sight order (GTF, then BED, then BAM header), 1-based like the reference
    std::map<std::string, int> chrom_id;
    std::vector<std::string> chrom_name;           // id-1 -> name
    int chromosome(const std::string &name);

    // GTF state
    std::vector<std::string> gene_list, exon_list;                  // geneList / exonList (GTF order)
    // ids (gene, transcript and exon ids share one table) are interned while parsing: a row carries two small
    // integers, and the per-id tables of the reference (geneNames, geneCodingLengths, uniqueness sets) are vectors
    // indexed by them.  The table is open addressing over one character arena: no allocation per line.
    std::vector<uint32_t> id_slots;                 // hash slots -> key + 1 (0 = empty)
    std::vector<uint64_t> id_hash;                  // per key
    std::vector<uint32_t> id_off;                   // per key: [id_off[k], id_off[k + 1]) in id_chars
    std::vector<char> id_chars;
    uint32_t intern(const char *p, size_t n);
    uint32_t intern(const std::string &s) { return intern(s.data(), s.size()); }
    bool lookup(const std::string &s, uint32_t &key) const;
    std::string id_text(uint32_t key) const { return std::string(id_chars.data() + id_off[key], id_chars.data() + id_off[key + 1]); }
    size_t n_ids() const { return id_hash.size(); }
    std::vector<std::string> name_of_key;           // geneNames[feature_id] by key ("" + has flag below)
    std::vector<uint8_t> has_name, seen_gene, seen_exon;
    std::vector<long long> coding_of_key;           // geneCodingLengths by key
    std::string gene_name(const std::string &feature_id) const;          // geneNames[feature_id] ("" when absent)
    long long coding_length(const std::string &gene_id) const;          // geneCodingLengths[gene_id] (0 when absent)
    // excluded: --legacy leaves 1-base features out of the feature lists (src/RNASeQC.cpp:129-135) but they are already
    // in geneList / exonList; flatten() parks such rows on a contig no record can name, so that the boundary's id
    // spaces (and the report rows) stay those of the lists
    struct Row { int chrom; long long start, end; int strand; bool is_gene; bool ribosomal; uint32_t feature_key, gene_key; size_t order; bool excluded; };
    bool legacy = false;                                            // --legacy (set before load_gtf)
    std::vector<Row> rows;                                          // kept gene/exon rows, GTF order
    void load_gtf(const std::string &path);

    // BED rows (+1/+1), file order
    struct BedRow { int chrom; long long start, end; };
    std::vector<BedRow> bed_rows;
    void load_bed(const std::string &path);

    // ---- flattened form for the boundary (filled by flatten) -----------------------------------------
    // contig ids of the boundary: BAM header order first, then contigs only the GTF/BED name
    std::vector<std::string> contig_names; int n_ref = 0;
    std::vector<int32_t> g_contig, g_start, g_end, e_contig, e_start, e_end, b_contig, b_start, b_end;
    std::vector<uint8_t> g_flags, e_flags, globin;
    std::vector<uint32_t> g_id, e_id, e_gene, ge_off, ge_row, g_order, e_order;
    std::vector<std::string> gene_id_of;            // gene id (boundary) -> gene_id string (listed first, then phantom)
    std::vector<int> chrom_of_contig;               // boundary contig id -> chromosomeMap id
    int n_genes = 0;
    rsqc_annotation ann{};
    rsqc_bed bed{};
    void flatten(const std::vector<std::string> &bam_contigs);
    // listed gene ids of a boundary contig in start order (coverage.tsv row order)
    std::vector<std::vector<uint32_t>> genes_by_contig;
};

}  // namespace rsqc_host
