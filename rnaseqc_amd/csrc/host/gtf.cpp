#include "gtf.hpp"

#include <algorithm>
#include <fstream>
#include <iostream>
#include <set>
#include <sstream>
#include <unordered_set>

namespace rsqc_host {

static const std::set<std::string> kGlobins = {"HBA1", "HBA2", "HBB", "HBD", "HBG1", "HBG2", "HBE1", "HBM", "HBQ1", "HBZ", "HBBP1", "HBZP1"};

int Annotation::chromosome(const std::string &name) {                 // chromosomeMap, src/Fasta.cpp:17-25
    auto it = chrom_id.find(name);
    if (it != chrom_id.end()) return it->second;
    const int id = (int)chrom_id.size() + 1;
    chrom_id[name] = id;
    chrom_name.push_back(name);
    return id;
}

// parseAttributes, src/GTF.cpp:133-148: split on ';' first, then take the text between the first two quotes
static void parse_attributes(const std::string &intake, std::map<std::string, std::string> &attributes) {
    std::istringstream tokenizer(intake);
    std::string buffer;
    while (std::getline(tokenizer, buffer, ';')) {
        std::istringstream splitter(buffer);
        std::string current;
        std::getline(splitter, current, '"');
        std::string key = current.substr(0, current.length() ? current.length() - 1 : 0);
        while (!key.empty() && (key[0] == ' ' || key[0] == '\t')) key = key.substr(1);
        current.clear();
        std::getline(splitter, current, '"');
        attributes[key] = current;
    }
}

void Annotation::load_gtf(const std::string &path) {
    std::ifstream in(path);
    if (!in.is_open()) throw FileError("Unable to open GTF file: " + path);
    std::unordered_set<std::string> gene_ids, exon_ids;
    std::map<std::string, unsigned> exon_names;
    // one Feature object is reused for the whole file (src/RNASeQC.cpp:109,127): fields the line does not
    // set keep the previous line's value (Q16)
    std::string feature_id, gene_id, transcript_type;
    std::string line;
    size_t order = 0;
    try {
        while (std::getline(in, line)) {
            if (line[0] == '#') continue;                               // note: a blank line fails below, like the reference
            std::istringstream tokenizer(line);
            std::string buffer;
            if (!std::getline(tokenizer, buffer, '\t')) throw GtfError("Unable to parse chromosome. Invalid GTF line: " + line);
            const int chrom = chromosome(buffer);
            if (!std::getline(tokenizer, buffer, '\t')) throw GtfError("Unable to parse track. Invalid GTF line: " + line);
            if (!std::getline(tokenizer, buffer, '\t')) throw GtfError("Unable to parse feature type. Invalid GTF line: " + line);
            int type = 3;                                               // Gene 0, Transcript 1, Exon 2, Other 3
            if (buffer == "exon") type = 2; else if (buffer == "gene") type = 0; else if (buffer == "transcript") type = 1;
            if (!std::getline(tokenizer, buffer, '\t')) throw GtfError("Unable to parse start. Invalid GTF line: " + line);
            const long long start = (long long)std::stoull(buffer);
            if (!std::getline(tokenizer, buffer, '\t')) throw GtfError("Unable to parse end. Invalid GTF line: " + line);
            const long long end = (long long)std::stoull(buffer);
            if (!std::getline(tokenizer, buffer, '\t')) throw GtfError("Unable to parse score. Invalid GTF line: " + line);
            if (!std::getline(tokenizer, buffer, '\t')) throw GtfError("Unable to parse strand. Invalid GTF line: " + line);
            int strand = RSQC_STRAND_UNKNOWN;
            if (!buffer.empty() && buffer[0] == '+') strand = RSQC_STRAND_FORWARD; else if (!buffer.empty() && buffer[0] == '-') strand = RSQC_STRAND_REVERSE;
            if (!std::getline(tokenizer, buffer, '\t')) throw GtfError("Unable to parse frame. Invalid GTF line: " + line);
            if (!std::getline(tokenizer, buffer)) throw GtfError("Unable to parse attributes. Invalid GTF line: " + line);
            std::map<std::string, std::string> attributes;
            parse_attributes(buffer, attributes);
            if (end < start) std::cerr << "Bad feature range:" << start << " - " << end << std::endl;
            const bool has_gene_id = attributes.count("gene_id") != 0;
            if (type == 0 && has_gene_id) {
                feature_id = attributes["gene_id"];
                if (gene_ids.count(feature_id)) throw GtfError("Detected non-unique Gene ID: " + feature_id);
                gene_ids.insert(feature_id);
                gene_list.push_back(feature_id);
            }
            if (type == 1 && attributes.count("transcript_id")) feature_id = attributes["transcript_id"];
            if (has_gene_id) gene_id = attributes["gene_id"];
            if (type == 2) {
                if (attributes.count("exon_id")) feature_id = attributes["exon_id"];
                else if (has_gene_id) {
                    feature_id = attributes["gene_id"] + "_" + std::to_string(++exon_names[attributes["gene_id"]]);
                    std::cerr << "Unnamed exon: Gene: " << attributes["gene_id"] << " Position: [" << start << ", " << end
                              << "] Inferred Exon Name: " << feature_id << std::endl;
                } else throw GtfError("Exon missing exon_id and gene_id fields: " + line);
                if (exon_ids.count(feature_id)) throw GtfError("Detected non-unique Exon ID: " + feature_id);
                exon_ids.insert(feature_id);
                exon_list.push_back(feature_id);
                gene_coding_length[gene_id] += 1 + (end - start);
            }
            if (attributes.count("transcript_type")) transcript_type = attributes["transcript_type"];
            if (attributes.count("gene_name")) gene_names[feature_id] = attributes["gene_name"];
            else if (has_gene_id) gene_names[feature_id] = attributes["gene_id"];
            const bool ribosomal = transcript_type.find("rRNA") != std::string::npos;     // regex_search "rRNA"
            if (type == 0 || type == 2) {                                                // src/RNASeQC.cpp:137-139
                if (end < start) throw GtfError("feature with end < start is not supported: " + line);
                rows.push_back(Row{chrom, start, end, strand, type == 0, ribosomal, feature_id, gene_id, order});
            }
            ++order;
        }
    } catch (GtfError &) {
        throw;
    } catch (std::invalid_argument &e) {
        throw GtfError(std::string("GTF is in an invalid format: ") + e.what());
    } catch (std::exception &e) {
        throw GtfError(std::string("Uncountered an unknown error while parsing GTF: ") + e.what());
    }
}

void Annotation::load_bed(const std::string &path) {                   // extractBED, src/BED.cpp:18-45
    std::ifstream in(path);
    if (!in.is_open()) throw FileError("Unable to open BED file: " + path);
    std::string line;
    try {
        while (std::getline(in, line)) {
            if (line[0] == '#') continue;
            std::istringstream tokenizer(line);
            std::string buffer;
            tokenizer >> buffer;
            const int chrom = chromosome(buffer);
            tokenizer >> buffer;
            const long long start = (long long)std::stoull(buffer) + 1;
            tokenizer >> buffer;
            const long long end = (long long)std::stoull(buffer) + 1;
            bed_rows.push_back(BedRow{chrom, start, end});
        }
    } catch (std::exception &e) {
        throw BedError(std::string("Encountered an unknown error while parsing the BED: ") + e.what());
    }
}

void Annotation::flatten(const std::vector<std::string> &bam_contigs) {
    // the BAM header names join chromosomeMap after GTF and BED (src/RNASeQC.cpp:224-226)
    for (auto &n : bam_contigs) chromosome(n);
    n_ref = (int)bam_contigs.size();
    contig_names = bam_contigs;
    std::map<int, int> contig_of_chrom;                 // chromosomeMap id -> boundary contig id
    for (int i = 0; i < n_ref; ++i) if (!contig_of_chrom.count(chrom_id[bam_contigs[(size_t)i]])) contig_of_chrom[chrom_id[bam_contigs[(size_t)i]]] = i;
    for (size_t k = 0; k < chrom_name.size(); ++k) {
        const int id = (int)k + 1;
        if (!contig_of_chrom.count(id)) { contig_of_chrom[id] = (int)contig_names.size(); contig_names.push_back(chrom_name[k]); }
    }
    chrom_of_contig.assign(contig_names.size(), 0);
    for (auto &kv : contig_of_chrom) chrom_of_contig[(size_t)kv.second] = kv.first;
    // gene ids: listed genes in geneList order, then gene_ids only exon rows name
    std::map<std::string, uint32_t> gene_index;
    for (size_t g = 0; g < gene_list.size(); ++g) gene_index[gene_list[g]] = (uint32_t)g;
    gene_id_of = gene_list;
    for (auto &r : rows) if (!r.is_gene && !gene_index.count(r.gene_id)) { gene_index[r.gene_id] = (uint32_t)gene_id_of.size(); gene_id_of.push_back(r.gene_id); }
    n_genes = (int)gene_id_of.size();
    std::map<std::string, uint32_t> exon_index;
    for (size_t e = 0; e < exon_list.size(); ++e) exon_index[exon_list[e]] = (uint32_t)e;
    // stable sort by (contig, start): std::list::sort(compIntervalStart) per contig (src/RNASeQC.cpp:150-152)
    std::vector<const Row *> gr, er;
    for (auto &r : rows) (r.is_gene ? gr : er).push_back(&r);
    auto cmp = [&](const Row *a, const Row *b) {
        const int ca = contig_of_chrom[a->chrom], cb = contig_of_chrom[b->chrom];
        if (ca != cb) return ca < cb;
        return a->start < b->start;
    };
    std::stable_sort(gr.begin(), gr.end(), cmp);
    std::stable_sort(er.begin(), er.end(), cmp);
    auto flags_of = [](const Row *r) { return (uint8_t)(r->strand | (r->ribosomal ? RSQC_FF_RIBOSOMAL : 0)); };
    g_contig.clear(); g_start.clear(); g_end.clear(); g_flags.clear(); g_id.clear();
    genes_by_contig.assign(contig_names.size(), {});
    for (auto *r : gr) {
        g_contig.push_back(contig_of_chrom[r->chrom]); g_start.push_back((int32_t)r->start); g_end.push_back((int32_t)r->end);
        g_flags.push_back(flags_of(r)); g_id.push_back(gene_index[r->feature_id]);
        genes_by_contig[(size_t)contig_of_chrom[r->chrom]].push_back(gene_index[r->feature_id]);
    }
    e_contig.clear(); e_start.clear(); e_end.clear(); e_flags.clear(); e_id.clear(); e_gene.clear();
    std::vector<std::vector<uint32_t>> per_gene((size_t)n_genes);
    for (size_t i = 0; i < er.size(); ++i) {
        const Row *r = er[i];
        e_contig.push_back(contig_of_chrom[r->chrom]); e_start.push_back((int32_t)r->start); e_end.push_back((int32_t)r->end);
        e_flags.push_back(flags_of(r)); e_id.push_back(exon_index[r->feature_id]);
        const uint32_t g = gene_index[r->gene_id];
        e_gene.push_back(g);
        per_gene[g].push_back((uint32_t)i);                 // exonsForGene: sorted order (src/RNASeQC.cpp:153-154)
    }
    ge_off.assign((size_t)n_genes + 1, 0); ge_row.clear();
    for (int g = 0; g < n_genes; ++g) { ge_off[(size_t)g + 1] = ge_off[(size_t)g] + (uint32_t)per_gene[(size_t)g].size(); ge_row.insert(ge_row.end(), per_gene[(size_t)g].begin(), per_gene[(size_t)g].end()); }
    globin.assign((size_t)n_genes, 0);
    for (int g = 0; g < n_genes; ++g) { auto it = gene_names.find(gene_id_of[(size_t)g]); if (it != gene_names.end() && kGlobins.count(it->second)) globin[(size_t)g] = 1; }
    ann = rsqc_annotation{n_ref, (int32_t)contig_names.size(), n_genes, (int32_t)gene_list.size(), (int32_t)exon_list.size(),
                          g_contig.data(), g_start.data(), g_end.data(), g_flags.data(), g_id.data(),
                          e_contig.data(), e_start.data(), e_end.data(), e_flags.data(), e_id.data(), e_gene.data(),
                          globin.data(), ge_off.data(), ge_row.data()};
    // BED: grouped by contig, file order inside (must ascend by start)
    std::vector<const BedRow *> br;
    for (auto &b : bed_rows) br.push_back(&b);
    std::stable_sort(br.begin(), br.end(), [&](const BedRow *a, const BedRow *b) { return contig_of_chrom[a->chrom] < contig_of_chrom[b->chrom]; });
    b_contig.clear(); b_start.clear(); b_end.clear();
    for (auto *b : br) { b_contig.push_back(contig_of_chrom[b->chrom]); b_start.push_back((int32_t)b->start); b_end.push_back((int32_t)b->end); }
    bed = rsqc_bed{(int32_t)b_contig.size(), b_contig.data(), b_start.data(), b_end.data()};
}

}  // namespace rsqc_host
