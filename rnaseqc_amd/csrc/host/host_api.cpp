// host_api.cpp -- thin C entry points over the CLI's host-side pieces (GTF/BED ingest, BAM decode,
// report writers) so that the CPU tests can exercise them without a GPU.  Not part of the product ABI.
#include <zlib.h>

#include <cstdio>
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>

#include "bam.hpp"
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
HAPI const rsqc_batch *host_bam_batch(void *h) { return &((BamHandle *)h)->view; }
HAPI int host_bam_n_contigs(void *h) { return (int)((BamHandle *)h)->names.size(); }
HAPI const char *host_bam_contig(void *h, int i) { return ((BamHandle *)h)->names[(size_t)i].c_str(); }
HAPI void host_bam_free(void *h) { delete (BamHandle *)h; }

// ---- synthetic-input tool: writes an rsqc_batch as a BAM file (what rnaseqc_amd/bamio.py does, at C speed, for the
// CLI benchmark).  SEQ is all 'A', QUAL 0xff; QNAME = the batch's names, or 16 hex digits of qhash when it has none.
namespace {
void put32(std::vector<uint8_t> &o, uint32_t v) { o.push_back(v & 255); o.push_back((v >> 8) & 255); o.push_back((v >> 16) & 255); o.push_back(v >> 24); }
void put16(std::vector<uint8_t> &o, uint16_t v) { o.push_back(v & 255); o.push_back(v >> 8); }
}
HAPI int host_bam_write(const char *path, const char *const *contig_names, const unsigned *contig_len, int n_contigs,
                        const rsqc_batch *b, const char *ch_tag, const char *filter_tag, int threads) {
    FILE *fp = fopen(path, "wb");
    if (!fp) return 10;
    WorkPool pool(threads < 1 ? 1 : threads);
    std::vector<uint8_t> raw;
    std::string text = "@HD\tVN:1.6\tSO:coordinate\n";
    for (int i = 0; i < n_contigs; ++i) text += std::string("@SQ\tSN:") + contig_names[i] + "\tLN:" + std::to_string(contig_len[i]) + "\n";
    raw.insert(raw.end(), {'B', 'A', 'M', 1});
    put32(raw, (uint32_t)text.size()); raw.insert(raw.end(), text.begin(), text.end());
    put32(raw, (uint32_t)n_contigs);
    for (int i = 0; i < n_contigs; ++i) {
        const size_t l = strlen(contig_names[i]) + 1;
        put32(raw, (uint32_t)l); raw.insert(raw.end(), contig_names[i], contig_names[i] + l); put32(raw, contig_len[i]);
    }
    auto flush = [&](bool final) {
        const size_t BS = 65280;
        const size_t nb = final ? (raw.size() + BS - 1) / BS : raw.size() / BS;
        std::vector<std::vector<uint8_t>> comp(nb);
        pool.run(nb, [&](size_t k) {
            const size_t o = k * BS, len = std::min(BS, raw.size() - o);
            std::vector<uint8_t> &c = comp[k];
            c.resize(18 + compressBound((uLong)len) + 8);
            z_stream zs{};
            deflateInit2(&zs, 1, Z_DEFLATED, -15, 8, Z_DEFAULT_STRATEGY);
            zs.next_in = raw.data() + o; zs.avail_in = (uInt)len; zs.next_out = c.data() + 18; zs.avail_out = (uInt)(c.size() - 26);
            deflate(&zs, Z_FINISH);
            const size_t clen = zs.total_out;
            deflateEnd(&zs);
            static const uint8_t hdr[16] = {0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 0xff, 6, 0, 'B', 'C', 2, 0};
            memcpy(c.data(), hdr, 16);
            const uint16_t bsize = (uint16_t)(clen + 25);
            c[16] = bsize & 255; c[17] = bsize >> 8;
            const uint32_t crc = (uint32_t)crc32(crc32(0L, Z_NULL, 0), raw.data() + o, (uInt)len);
            uint8_t *t = c.data() + 18 + clen;
            t[0] = crc & 255; t[1] = (crc >> 8) & 255; t[2] = (crc >> 16) & 255; t[3] = crc >> 24;
            t[4] = len & 255; t[5] = (len >> 8) & 255; t[6] = (len >> 16) & 255; t[7] = (uint8_t)(len >> 24);
            c.resize(18 + clen + 8);
        });
        for (auto &c : comp) fwrite(c.data(), 1, c.size(), fp);
        raw.erase(raw.begin(), raw.begin() + (long)std::min(raw.size(), nb * BS));
    };
    uint32_t seg = 0, w = 0;
    for (uint64_t i = 0; i < b->n; ++i) {
        while (seg + 1 < b->n_seg && b->seg_start[seg + 1] <= i) ++seg;
        const int32_t tid = b->n_seg ? b->seg_tid[seg] : -1;
        const rsqc_rec_core &co = b->core[i]; const rsqc_rec_aux &au = b->aux[i];
        int32_t lq = au.l_qseq, nm = au.nm; uint32_t nc = au.n_cigar;
        if (au.l_qseq == RSQC_LQSEQ_ESCAPE || au.nm == RSQC_NM_ESCAPE || au.n_cigar == RSQC_NCIGAR_ESCAPE) {
            while (w < b->n_wide && b->wide_index[w] < i) ++w;
            if (w < b->n_wide && b->wide_index[w] == i) { lq = b->wide_l_qseq[w]; nm = b->wide_nm[w]; nc = b->wide_n_cigar[w]; }
        }
        char hexname[17];
        const char *name; size_t nlen;
        if (b->qname) { name = (const char *)b->qname + b->qname_off[i]; nlen = b->qname_off[i + 1] - b->qname_off[i]; }
        else { snprintf(hexname, sizeof hexname, "%016llx", (unsigned long long)au.qhash); name = hexname; nlen = 16; }
        const int32_t mtid = (au.tagbits & RSQC_TB_MTID_SAME) ? tid : (tid + 1 < n_contigs ? tid + 1 : (tid != 0 ? 0 : -1));
        std::vector<uint8_t> tags;
        if (au.tagbits & RSQC_TB_HAS_NM) {
            tags.push_back('N'); tags.push_back('M');
            if (nm >= 0 && nm < 256) { tags.push_back('C'); tags.push_back((uint8_t)nm); } else { tags.push_back('i'); put32(tags, (uint32_t)nm); }
        }
        if (au.tagbits & RSQC_TB_HAS_CH) { tags.push_back(ch_tag[0]); tags.push_back(ch_tag[1]); tags.push_back('Z'); tags.push_back('1'); tags.push_back(0); }
        if (au.tagbits & RSQC_TB_FILTER0) { tags.push_back(filter_tag[0]); tags.push_back(filter_tag[1]); tags.push_back('i'); put32(tags, 1); }
        const size_t l_seq = lq < 0 ? 0 : (size_t)lq;
        const uint32_t block_size = (uint32_t)(32 + nlen + 1 + 4 * (size_t)nc + (l_seq + 1) / 2 + l_seq + tags.size());
        put32(raw, block_size);
        put32(raw, (uint32_t)tid); put32(raw, (uint32_t)co.pos);
        raw.push_back((uint8_t)(nlen + 1)); raw.push_back(au.mapq); put16(raw, 4680); put16(raw, (uint16_t)nc); put16(raw, au.flag);
        put32(raw, (uint32_t)lq); put32(raw, (uint32_t)mtid); put32(raw, (uint32_t)co.mpos); put32(raw, (uint32_t)co.isize);
        raw.insert(raw.end(), name, name + nlen); raw.push_back(0);
        for (uint32_t c = 0; c < nc; ++c) put32(raw, b->cigar[co.cigar_off + c]);
        raw.insert(raw.end(), (l_seq + 1) / 2, 0x11); raw.insert(raw.end(), l_seq, 0xff);
        raw.insert(raw.end(), tags.begin(), tags.end());
        if (raw.size() >= ((size_t)64 << 20)) flush(false);
    }
    flush(true);
    static const uint8_t eof[28] = {0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 0xff, 6, 0, 'B', 'C', 2, 0, 0x1b, 0, 3, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    fwrite(eof, 1, 28, fp);
    fclose(fp);
    return 0;
}
