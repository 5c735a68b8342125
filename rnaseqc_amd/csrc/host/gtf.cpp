#include "gtf.hpp"

#include <algorithm>
#include <cstring>
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

static inline uint64_t hash_bytes(const char *p, size_t n) {            // FNV-1a + a finaliser
    uint64_t h = 0xCBF29CE484222325ull;
    for (size_t i = 0; i < n; ++i) { h ^= (unsigned char)p[i]; h *= 0x100000001B3ull; }
    h ^= h >> 32; h *= 0xD6E8FEB86659FD93ull; h ^= h >> 32;
    return h;
}
uint32_t Annotation::intern(const char *p, size_t n) {
    if (id_slots.empty()) { id_slots.assign(1u << 16, 0u); id_off.assign(1, 0u); }
    const uint64_t h = hash_bytes(p, n);
    size_t mask = id_slots.size() - 1, s = (size_t)h & mask;
    for (;; s = (s + 1) & mask) {
        const uint32_t v = id_slots[s];
        if (!v) break;
        const uint32_t k = v - 1;
        if (id_hash[k] == h && id_off[k + 1] - id_off[k] == n && memcmp(id_chars.data() + id_off[k], p, n) == 0) return k;
    }
    const uint32_t k = (uint32_t)id_hash.size();
    id_hash.push_back(h);
    id_chars.insert(id_chars.end(), p, p + n);
    id_off.push_back((uint32_t)id_chars.size());
    name_of_key.emplace_back(); has_name.push_back(0); seen_gene.push_back(0); seen_exon.push_back(0); coding_of_key.push_back(0);
    id_slots[s] = k + 1;
    if ((size_t)k * 2 > id_slots.size()) {                                // grow and re-seat
        std::vector<uint32_t> bigger(id_slots.size() * 4, 0u);
        mask = bigger.size() - 1;
        for (uint32_t q = 0; q <= k; ++q) { size_t t = (size_t)id_hash[q] & mask; while (bigger[t]) t = (t + 1) & mask; bigger[t] = q + 1; }
        id_slots.swap(bigger);
    }
    return k;
}
bool Annotation::lookup(const std::string &str, uint32_t &key) const {
    if (id_slots.empty()) return false;
    const uint64_t h = hash_bytes(str.data(), str.size());
    const size_t mask = id_slots.size() - 1;
    for (size_t s = (size_t)h & mask;; s = (s + 1) & mask) {
        const uint32_t v = id_slots[s];
        if (!v) return false;
        const uint32_t k = v - 1;
        if (id_hash[k] == h && id_off[k + 1] - id_off[k] == str.size() && memcmp(id_chars.data() + id_off[k], str.data(), str.size()) == 0) { key = k; return true; }
    }
}
std::string Annotation::gene_name(const std::string &feature_id) const {
    uint32_t k;
    return lookup(feature_id, k) && has_name[k] ? name_of_key[k] : std::string();
}
long long Annotation::coding_length(const std::string &gene_id) const {
    uint32_t k;
    return lookup(gene_id, k) ? coding_of_key[k] : 0;
}

// parseAttributes, src/GTF.cpp:133-148, without the per-line std::map: the text is split on ';' first (also inside
// quotes, like the reference), a piece's key is what precedes its first '"' minus one character, stripped of leading
// blanks, its value what lies between the first two quotes; a later piece with the same key overwrites an earlier
// one.  Only the five keys the loader reads are kept.
namespace {
struct Span { const char *p = nullptr; size_t n = 0; bool has = false; void set(const char *a, const char *b) { p = a; n = (size_t)(b - a); has = true; } };
struct Attrs { Span gene_id, transcript_id, exon_id, transcript_type, gene_name; };    // views into the current line
void parse_attributes(const char *p, const char *end, Attrs &a) {
    while (p < end) {
        const char *semi = (const char *)memchr(p, ';', (size_t)(end - p));
        const char *pe = semi ? semi : end;                      // piece [p, pe)
        const char *q1 = (const char *)memchr(p, '"', (size_t)(pe - p));
        const char *kend = q1 ? q1 : pe;                         // "current" of the first getline
        const char *ks = p, *ke = kend > p ? kend - 1 : p;       // key = current minus its last character ...
        while (ks < ke && (*ks == ' ' || *ks == '\t')) ++ks;     // ... stripped of leading blanks
        const char *vs = pe, *ve = pe;
        if (q1) { vs = q1 + 1; const char *q2 = (const char *)memchr(vs, '"', (size_t)(pe - vs)); ve = q2 ? q2 : pe; }
        const size_t kl = (size_t)(ke - ks);
        auto is = [&](const char *name, size_t n) { return kl == n && memcmp(ks, name, n) == 0; };
        if (is("gene_id", 7)) a.gene_id.set(vs, ve);
        else if (is("transcript_id", 13)) a.transcript_id.set(vs, ve);
        else if (is("exon_id", 7)) a.exon_id.set(vs, ve);
        else if (is("transcript_type", 15)) a.transcript_type.set(vs, ve);
        else if (is("gene_name", 9)) a.gene_name.set(vs, ve);
        if (!semi) break;
        p = semi + 1;
    }
}
}  // namespace

void Annotation::load_gtf(const std::string &path) {
    std::ifstream in(path);
    if (!in.is_open()) throw FileError("Unable to open GTF file: " + path);
    std::unordered_map<std::string, unsigned> exon_names;
    // one Feature object is reused for the whole file (src/RNASeQC.cpp:109,127): fields the line does not
    // set keep the previous line's value (Q16).  feature_id / gene_id are carried as interned keys.
    const uint32_t NOKEY = 0xFFFFFFFFu;
    uint32_t feature_key = intern("", 0), gene_key = feature_key;       // the empty initial ids of a default Feature
    (void)NOKEY;
    std::string transcript_type;
    std::string line, num;
    size_t order = 0;
    static const char *const what[9] = {"chromosome", "track", "feature type", "start", "end", "score", "strand", "frame", "attributes"};
    try {
        while (std::getline(in, line)) {
            if (line[0] == '#') continue;                               // note: a blank line fails below, like the reference
            // nine tab-separated fields, the last one being the rest of the line.  std::getline fails exactly when
            // nothing is left behind the previous separator (an empty field FOLLOWED by a tab is a valid empty field).
            const char *f[9], *fe[9];
            const char *p = line.data(), *end = p + line.size();
            for (int k = 0; k < 9; ++k) {
                if (p == end) throw GtfError(std::string("Unable to parse ") + what[k] + ". Invalid GTF line: " + line);
                const char *t = k < 8 ? (const char *)memchr(p, '\t', (size_t)(end - p)) : nullptr;
                f[k] = p; fe[k] = t ? t : end;
                p = t ? t + 1 : end;
            }
            const int chrom = chromosome(std::string(f[0], fe[0]));
            int type = 3;                                               // Gene 0, Transcript 1, Exon 2, Other 3
            const size_t tl = (size_t)(fe[2] - f[2]);
            if (tl == 4 && memcmp(f[2], "exon", 4) == 0) type = 2;
            else if (tl == 4 && memcmp(f[2], "gene", 4) == 0) type = 0;
            else if (tl == 10 && memcmp(f[2], "transcript", 10) == 0) type = 1;
            num.assign(f[3], fe[3]);
            const long long start = (long long)std::stoull(num);
            num.assign(f[4], fe[4]);
            const long long endp = (long long)std::stoull(num);
            int strand = RSQC_STRAND_UNKNOWN;
            if (fe[6] > f[6] && f[6][0] == '+') strand = RSQC_STRAND_FORWARD; else if (fe[6] > f[6] && f[6][0] == '-') strand = RSQC_STRAND_REVERSE;
            Attrs at;
            parse_attributes(f[8], fe[8], at);
            if (endp < start) std::cerr << "Bad feature range:" << start << " - " << endp << std::endl;
            const uint32_t gid_key = at.gene_id.has ? intern(at.gene_id.p, at.gene_id.n) : 0u;
            if (type == 0 && at.gene_id.has) {
                feature_key = gid_key;
                if (seen_gene[feature_key]) throw GtfError("Detected non-unique Gene ID: " + id_text(feature_key));
                seen_gene[feature_key] = 1;
                gene_list.push_back(id_text(feature_key));
            }
            if (type == 1 && at.transcript_id.has) feature_key = intern(at.transcript_id.p, at.transcript_id.n);
            if (at.gene_id.has) gene_key = gid_key;
            if (type == 2) {
                if (at.exon_id.has) feature_key = intern(at.exon_id.p, at.exon_id.n);
                else if (at.gene_id.has) {
                    const std::string g(at.gene_id.p, at.gene_id.n);
                    const std::string inferred = g + "_" + std::to_string(++exon_names[g]);
                    feature_key = intern(inferred);
                    std::cerr << "Unnamed exon: Gene: " << g << " Position: [" << start << ", " << endp
                              << "] Inferred Exon Name: " << inferred << std::endl;
                } else throw GtfError("Exon missing exon_id and gene_id fields: " + line);
                if (seen_exon[feature_key]) throw GtfError("Detected non-unique Exon ID: " + id_text(feature_key));
                seen_exon[feature_key] = 1;
                exon_list.push_back(id_text(feature_key));
                coding_of_key[gene_key] += 1 + (endp - start);
            }
            if (at.transcript_type.has) transcript_type.assign(at.transcript_type.p, at.transcript_type.n);
            if (at.gene_name.has) { name_of_key[feature_key].assign(at.gene_name.p, at.gene_name.n); has_name[feature_key] = 1; }
            else if (at.gene_id.has) { name_of_key[feature_key].assign(at.gene_id.p, at.gene_id.n); has_name[feature_key] = 1; }
            const bool ribosomal = transcript_type.find("rRNA") != std::string::npos;     // regex_search "rRNA"
            if (type == 0 || type == 2) {                                                // src/RNASeQC.cpp:137-139
                if (endp < start) throw GtfError("feature with end < start is not supported: " + line);
                const bool excluded = legacy && endp == start;                           // src/RNASeQC.cpp:129-135
                if (excluded && type == 2) coding_of_key[gene_key] -= 1;
                rows.push_back(Row{chrom, start, endp, strand, type == 0, ribosomal, feature_key, gene_key, order, excluded});
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

// One BED line = one interval, `extractBED` (src/BED.cpp:18-45) as the loop of src/RNASeQC.cpp:185 drives it.  The reference
// pulls three whitespace-separated tokens through ONE string, so what a malformed line does follows from that and is kept:
//   * a line whose first character is '#' is skipped; an empty line is NOT (its chromosome is the empty name and its start
//     fails to parse: the run ends with "Failed to parse the BED");
//   * a missing token leaves the previous one in place: "chr1 100" ends at 101 like it starts, "chr1" alone fails on "chr1";
//   * numbers are std::stoull's: optional sign (a '-' wraps around), leading digits only ("100abc" is 100), nothing
//     parsable or a value beyond 64 bits is an exception; both bounds are stored + 1.
// pinned against the reference's own BED.cpp: tests/test_reference_metrics.py::test_bed_loader_vs_reference_extractBED
namespace {
struct BedTokens { const char *tok[3]; size_t len[3]; int n; };
inline BedTokens bed_split(const std::string &line) {
    BedTokens t{{nullptr, nullptr, nullptr}, {0, 0, 0}, 0};
    const char *p = line.data(), *const e = p + line.size();
    auto space = [](char c) { return c == ' ' || (c >= '\t' && c <= '\r'); };         // the "C" locale's isspace, what operator>> skips
    while (t.n < 3) {
        while (p < e && space(*p)) ++p;
        if (p == e) break;
        const char *q = p;
        while (q < e && !space(*q)) ++q;
        t.tok[t.n] = p; t.len[t.n] = (size_t)(q - p); ++t.n;
        p = q;
    }
    return t;
}
}  // namespace

void Annotation::load_bed(const std::string &path) {
    std::ifstream in(path);
    if (!in.is_open()) throw FileError("Unable to open BED file: " + path);
    std::string line, field;
    try {
        while (std::getline(in, line)) {
            if (!line.empty() && line[0] == '#') continue;
            const BedTokens t = bed_split(line);
            // token k that is missing repeats token k - 1 (the reference's buffer keeps its value when extraction fails)
            auto text = [&](int k) { while (k >= t.n && k > 0) --k; return t.n ? std::string(t.tok[k], t.len[k]) : std::string(); };
            const int chrom = chromosome(text(0));
            field = text(1);
            const unsigned long long lo = std::stoull(field);
            field = text(2);
            const unsigned long long hi = std::stoull(field);
            bed_rows.push_back(BedRow{chrom, (long long)(lo + 1ull), (long long)(hi + 1ull)});
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
    // rows --legacy excludes live on one extra contig behind all others (chromosomeMap id 0 = none)
    bool any_excluded = false;
    for (auto &r : rows) any_excluded = any_excluded || r.excluded;
    const int parked = any_excluded ? (int)contig_names.size() : -1;
    if (any_excluded) contig_names.push_back("(legacy: excluded 1-base features)");
    chrom_of_contig.assign(contig_names.size(), 0);
    for (auto &kv : contig_of_chrom) chrom_of_contig[(size_t)kv.second] = kv.first;
    auto contig_of = [&](const Row *r) { return r->excluded ? parked : contig_of_chrom[r->chrom]; };
    // gene ids: listed genes in geneList order, then gene_ids only exon rows name
    const uint32_t NONE = 0xFFFFFFFFu;
    std::vector<uint32_t> gene_index(n_ids() + 1, NONE), exon_index(n_ids() + 1, NONE);   // by interned key (every listed id is interned)
    for (size_t g = 0; g < gene_list.size(); ++g) gene_index[intern(gene_list[g])] = (uint32_t)g;        // (already interned)
    gene_id_of = gene_list;
    for (auto &r : rows) if (!r.is_gene && gene_index[r.gene_key] == NONE) { gene_index[r.gene_key] = (uint32_t)gene_id_of.size(); gene_id_of.push_back(id_text(r.gene_key)); }
    // excluded exon rows belong to one extra unlisted gene: they must not lengthen the transcript of their own gene
    // (exonsForGene never sees them, src/RNASeQC.cpp:150-154)
    uint32_t parked_gene = NONE;
    for (auto &r : rows) if (r.excluded && !r.is_gene && parked_gene == NONE) { parked_gene = (uint32_t)gene_id_of.size(); gene_id_of.push_back("(legacy: excluded)"); }
    n_genes = (int)gene_id_of.size();
    for (size_t e = 0; e < exon_list.size(); ++e) exon_index[intern(exon_list[e])] = (uint32_t)e;
    // stable sort by (contig, start): std::list::sort(compIntervalStart) per contig (src/RNASeQC.cpp:150-152)
    std::vector<const Row *> gr, er;
    for (auto &r : rows) (r.is_gene ? gr : er).push_back(&r);
    auto cmp = [&](const Row *a, const Row *b) {
        const int ca = contig_of(a), cb = contig_of(b);
        if (ca != cb) return ca < cb;
        return a->start < b->start;
    };
    std::stable_sort(gr.begin(), gr.end(), cmp);
    std::stable_sort(er.begin(), er.end(), cmp);
    auto flags_of = [](const Row *r) { return (uint8_t)(r->strand | (r->ribosomal ? RSQC_FF_RIBOSOMAL : 0)); };
    g_contig.clear(); g_start.clear(); g_end.clear(); g_flags.clear(); g_id.clear(); g_order.clear(); e_order.clear();
    genes_by_contig.assign(contig_names.size(), {});
    for (auto *r : gr) {
        g_contig.push_back(contig_of(r)); g_start.push_back((int32_t)r->start); g_end.push_back((int32_t)r->end);
        g_flags.push_back(flags_of(r)); g_id.push_back(gene_index[r->feature_key]); g_order.push_back((uint32_t)r->order);
        if (!r->excluded) genes_by_contig[(size_t)contig_of(r)].push_back(gene_index[r->feature_key]);   // (an excluded gene never reaches BaseCoverage::compute: no coverage.tsv row)
    }
    e_contig.clear(); e_start.clear(); e_end.clear(); e_flags.clear(); e_id.clear(); e_gene.clear();
    std::vector<std::vector<uint32_t>> per_gene((size_t)n_genes);
    for (size_t i = 0; i < er.size(); ++i) {
        const Row *r = er[i];
        e_contig.push_back(contig_of(r)); e_start.push_back((int32_t)r->start); e_end.push_back((int32_t)r->end);
        e_flags.push_back(flags_of(r)); e_id.push_back(exon_index[r->feature_key]); e_order.push_back((uint32_t)r->order);
        const uint32_t g = r->excluded ? parked_gene : gene_index[r->gene_key];
        e_gene.push_back(g);
        per_gene[g].push_back((uint32_t)i);                 // exonsForGene: sorted order (src/RNASeQC.cpp:153-154)
    }
    ge_off.assign((size_t)n_genes + 1, 0); ge_row.clear();
    for (int g = 0; g < n_genes; ++g) { ge_off[(size_t)g + 1] = ge_off[(size_t)g] + (uint32_t)per_gene[(size_t)g].size(); ge_row.insert(ge_row.end(), per_gene[(size_t)g].begin(), per_gene[(size_t)g].end()); }
    globin.assign((size_t)n_genes, 0);
    for (int g = 0; g < n_genes; ++g) if ((uint32_t)g != parked_gene && kGlobins.count(gene_name(gene_id_of[(size_t)g]))) globin[(size_t)g] = 1;
    ann = rsqc_annotation{n_ref, (int32_t)contig_names.size(), n_genes, (int32_t)gene_list.size(), (int32_t)exon_list.size(),
                          g_contig.data(), g_start.data(), g_end.data(), g_flags.data(), g_id.data(),
                          e_contig.data(), e_start.data(), e_end.data(), e_flags.data(), e_id.data(), e_gene.data(),
                          globin.data(), ge_off.data(), ge_row.data(), g_order.data(), e_order.data()};
    // BED: grouped by contig, file order inside (must ascend by start)
    std::vector<const BedRow *> br;
    for (auto &b : bed_rows) br.push_back(&b);
    std::stable_sort(br.begin(), br.end(), [&](const BedRow *a, const BedRow *b) { return contig_of_chrom[a->chrom] < contig_of_chrom[b->chrom]; });
    b_contig.clear(); b_start.clear(); b_end.clear();
    for (auto *b : br) { b_contig.push_back(contig_of_chrom[b->chrom]); b_start.push_back((int32_t)b->start); b_end.push_back((int32_t)b->end); }
    bed = rsqc_bed{(int32_t)b_contig.size(), b_contig.data(), b_start.data(), b_end.data()};
}

}  // namespace rsqc_host
