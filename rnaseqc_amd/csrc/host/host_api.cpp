// host_api.cpp -- thin C entry points over the CLI's host-side pieces (GTF/BED ingest, BAM decode,
// report writers) so that the CPU tests can exercise them without a GPU.  Not part of the product ABI.
#include <zlib.h>

#include <cstdio>
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>

#include "bam.hpp"
#include "bgzf_feed.hpp"
#include "gtf.hpp"
#include "report.hpp"

using namespace rsqc_host;
#define HAPI extern "C" __attribute__((visibility("default")))

HAPI void *host_annotation_load_ex(const char *gtf, const char *bed, const char *const *bam_contigs, int n_contigs, int legacy, int *err) {
    Annotation *a = new Annotation();
    a->legacy = legacy != 0;
    *err = 0;
    try {
        a->load_gtf(gtf);
        if (bed && *bed) a->load_bed(bed);
        std::vector<std::string> c;
        for (int i = 0; i < n_contigs; ++i) c.emplace_back(bam_contigs[i]);
        a->flatten(c);
    } catch (FileError &) { *err = 10; } catch (GtfError &) { *err = 11; } catch (BedError &) { *err = 11; } catch (...) { *err = -1; }
    if (*err) { delete a; return nullptr; }
    return a;
}
HAPI void *host_annotation_load(const char *gtf, const char *bed, const char *const *bam_contigs, int n_contigs, int *err) {
    return host_annotation_load_ex(gtf, bed, bam_contigs, n_contigs, 0, err);
}
HAPI const rsqc_annotation *host_annotation_struct(void *h) { return &((Annotation *)h)->ann; }
HAPI const rsqc_bed *host_annotation_bed(void *h) { return &((Annotation *)h)->bed; }
// the BED loader alone, rows as read (file order, before flattening): returns the number of rows, -1 = cannot open,
// -2 - n = BedError after n rows (message -> err).  chrom_index = index into the '\n'-joined names (first-sight order).
HAPI long long host_bed_read(const char *path, long long cap, int32_t *chrom_index, long long *start, long long *end,
                             char *names, long long names_cap, char *err, long long err_cap) {
    Annotation a;
    bool threw = false;
    try { a.load_bed(path); }
    catch (FileError &) { return -1; }
    catch (BedError &e) { threw = true; if (err && err_cap > 0) { strncpy(err, e.what(), (size_t)err_cap - 1); err[err_cap - 1] = 0; } }
    long long n = 0;
    for (auto &r : a.bed_rows) { if (n < cap) { chrom_index[n] = r.chrom - 1; start[n] = r.start; end[n] = r.end; } ++n; }
    std::string joined;
    for (auto &s : a.chrom_name) { joined += s; joined += '\n'; }
    if (names && names_cap > 0) { strncpy(names, joined.c_str(), (size_t)names_cap - 1); names[names_cap - 1] = 0; }
    return threw ? -2 - n : n;
}
HAPI const char *host_annotation_gene_name(void *h, int listed_gene) {
    Annotation *a = (Annotation *)h; static thread_local std::string tmp; tmp = a->gene_name(a->gene_list[(size_t)listed_gene]); return tmp.c_str();
}
HAPI const char *host_annotation_gene_id(void *h, int listed_gene) { return ((Annotation *)h)->gene_list[(size_t)listed_gene].c_str(); }
HAPI const char *host_annotation_exon_id(void *h, int exon) { return ((Annotation *)h)->exon_list[(size_t)exon].c_str(); }
HAPI long long host_annotation_coding_length(void *h, int listed_gene) {
    Annotation *a = (Annotation *)h; return a->coding_length(a->gene_list[(size_t)listed_gene]);
}
HAPI void host_annotation_free(void *h) { delete (Annotation *)h; }

HAPI int host_write_reports(void *h, const rsqc_results *r, const char *out_dir, const char *sample, int sample_given,
                            int use_rpkm, int write_coverage, unsigned detection, const char *const *tags, int n_tags,
                            const int *visit, int n_visit) {
    ReportConfig cfg;
    cfg.output_dir = out_dir; cfg.sample_name = sample; cfg.sample_given = sample_given != 0; cfg.use_rpkm = use_rpkm != 0;
    cfg.write_coverage = write_coverage != 0; cfg.detection_threshold = detection;
    for (int i = 0; i < n_tags; ++i) cfg.filter_tags.emplace_back(tags[i]);
    try { write_reports(cfg, *(Annotation *)h, *r, std::vector<int>(visit, visit + n_visit)); }
    catch (std::range_error &) { return 2; } catch (...) { return -1; }
    return 0;
}

HAPI unsigned host_library_complexity(double dup, double unique, double limit) { return library_complexity(dup, unique, limit); }

struct BamHandle { BamReader reader; HostBatch batch; rsqc_batch view; std::vector<std::string> names; };
HAPI void *host_bam_read_all(const char *path, const char *ch_tag, const char *const *tags, int n_tags) {
    BamHandle *b = new BamHandle();
    if (!b->reader.open(path)) { delete b; return nullptr; }
    std::vector<std::string> t;
    for (int i = 0; i < n_tags; ++i) t.emplace_back(tags[i]);
    b->reader.set_tags(ch_tag, t);
    try { while (b->reader.read_batch(b->batch, 1u << 20)) {} } catch (...) { delete b; return nullptr; }
    b->view = b->batch.view();
    b->names = b->reader.contigs();
    return b;
}
// same with an explicit decode-thread count and read_batch granularity (the tests sweep both)
HAPI void *host_bam_read_all_ex(const char *path, const char *ch_tag, const char *const *tags, int n_tags, int threads,
                                unsigned long long batch_records) {
    BamHandle *b = new BamHandle();
    b->reader.set_threads(threads);
    if (!b->reader.open(path)) { delete b; return nullptr; }
    std::vector<std::string> t;
    for (int i = 0; i < n_tags; ++i) t.emplace_back(tags[i]);
    b->reader.set_tags(ch_tag, t);
    try { while (b->reader.read_batch(b->batch, (size_t)batch_records)) {} } catch (...) { delete b; return nullptr; }
    b->view = b->batch.view();
    b->names = b->reader.contigs();
    return b;
}
// the records of ONE reference sequence through the BAM index (<path>.bai): BamReader::load_index + seek
HAPI void *host_bam_read_contig(const char *path, const char *ch_tag, const char *const *tags, int n_tags, int threads, int contig) {
    BamHandle *b = new BamHandle();
    b->reader.set_threads(threads);
    if (!b->reader.open(path) || !b->reader.load_index(std::string(path) + ".bai")) { delete b; return nullptr; }
    const auto &idx = b->reader.index();
    if (contig < 0 || (size_t)contig >= idx.size()) { delete b; return nullptr; }
    std::vector<std::string> t;
    for (int i = 0; i < n_tags; ++i) t.emplace_back(tags[i]);
    b->reader.set_tags(ch_tag, t);
    b->names = b->reader.contigs();
    if (idx[(size_t)contig].present) {
        if (!b->reader.seek(idx[(size_t)contig].beg, idx[(size_t)contig].end)) { delete b; return nullptr; }
        uint64_t left = idx[(size_t)contig].n_records;
        try { while (left) { const size_t got = b->reader.read_batch(b->batch, (size_t)std::min<uint64_t>(left, 1u << 20)); if (!got) break; left -= got; } }
        catch (...) { delete b; return nullptr; }
    }
    b->view = b->batch.view();
    return b;
}
HAPI int host_bam_unsorted(void *h) { return ((BamHandle *)h)->batch.unsorted ? 1 : 0; }
HAPI int host_bam_bad_refid_count(void *h) { return (int)((BamHandle *)h)->batch.bad_refid.size(); }
HAPI const char *host_bam_bad_refid(void *h, int i) { return ((BamHandle *)h)->batch.bad_refid[(size_t)i].c_str(); }
HAPI const rsqc_batch *host_bam_batch(void *h) { return &((BamHandle *)h)->view; }
HAPI int host_bam_n_contigs(void *h) { return (int)((BamHandle *)h)->names.size(); }
HAPI const char *host_bam_contig(void *h, int i) { return ((BamHandle *)h)->names[(size_t)i].c_str(); }
HAPI void host_bam_free(void *h) { delete (BamHandle *)h; }

// (the synthetic BAM writer lives in bam_write.cpp)

// ---- the device decode's host side (bgzf_feed.hpp) for the tests: chunks of framed BGZF blocks
namespace { struct FeedHandle { rsqc_host::BgzfFeeder feed; std::string error; }; }
HAPI void *host_feed_open(const char *path) {
    FeedHandle *h = new FeedHandle();
    if (!h->feed.open(path)) { delete h; return nullptr; }
    return h;
}
HAPI unsigned long long host_feed_first_voffset(void *hv) {
    FeedHandle *h = (FeedHandle *)hv;
    try { return h->feed.first_record_voffset(); } catch (std::exception &e) { h->error = e.what(); return ~0ull; }
}
HAPI void host_feed_cpu_share(void *hv, int threads, double initial_share, double max_share) { ((FeedHandle *)hv)->feed.set_cpu_share(threads, initial_share, max_share); }
// page-locks (or, without a device, allocates) the chunk buffers now; max_out != 0: sized for this file's compression, never grown
HAPI int host_feed_reserve(void *hv, unsigned long long chunk_bytes, unsigned long long max_out) {
    FeedHandle *h = (FeedHandle *)hv;
    try { h->feed.reserve((size_t)chunk_bytes, max_out); return 0; }
    catch (std::exception &e) { h->error = e.what(); return -1; }
}
HAPI int host_feed_start(void *hv, unsigned long long voff_beg, unsigned long long voff_end, unsigned long long chunk_bytes, unsigned long long max_out, int read_threads) {
    FeedHandle *h = (FeedHandle *)hv;
    try { h->feed.read_threads = read_threads; h->feed.start(voff_beg, voff_end, (size_t)chunk_bytes, max_out); return 0; }
    catch (std::exception &e) { h->error = e.what(); return -1; }
}
// 1 = a chunk, 0 = end of the range, -1 = error (host_feed_error)
HAPI int host_feed_next(void *hv, const uint8_t **data, unsigned long long *bytes, const rsqc_bgzf_block **blocks, uint32_t *n_blocks,
                        uint32_t *skip, unsigned long long *limit, int *last) {
    FeedHandle *h = (FeedHandle *)hv;
    try {
        rsqc_host::BgzfFeeder::Chunk *c = h->feed.next();
        if (!c) return 0;
        *data = c->data; *bytes = c->total_bytes; *blocks = c->blocks.data(); *n_blocks = (uint32_t)c->blocks.size(); *skip = c->skip; *limit = c->limit; *last = c->last ? 1 : 0;
        return 1;
    } catch (std::exception &e) { h->error = e.what(); return -1; }
}
HAPI const char *host_feed_error(void *hv) { return ((FeedHandle *)hv)->error.c_str(); }
HAPI void host_feed_free(void *hv) { delete (FeedHandle *)hv; }
