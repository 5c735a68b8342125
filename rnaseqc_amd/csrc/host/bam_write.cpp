// bam_write.cpp -- synthetic-input tool: writes an rsqc_batch as a coordinate-sorted BAM (+ a minimal .bai) at the
// speed the 100 M-record end-to-end benchmark needs.  Not part of the product path: tests and bench.py use it to make
// the file the CLI then decodes.  Records are serialised and deflated (BGZF, level 1) by a pool of threads, a group
// of records per task; groups are written in order.
//
// seq_mode 0: SEQ all 'A', QUAL 0xff (the contract's BAM).
// seq_mode 1: random bases and binned Phred-like qualities with runs (the entropy of a real file: inflate costs what it
//             costs on real data).
#include <zlib.h>

#include <algorithm>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "bam.hpp"

using namespace rsqc_host;
#define HAPI extern "C" __attribute__((visibility("default")))

namespace {

struct Out {
    std::vector<uint8_t> v;
    void u8(uint8_t x) { v.push_back(x); }
    void u16(uint16_t x) { const size_t n = v.size(); v.resize(n + 2); memcpy(v.data() + n, &x, 2); }
    void u32(uint32_t x) { const size_t n = v.size(); v.resize(n + 4); memcpy(v.data() + n, &x, 4); }
    void bytes(const void *p, size_t k) { const size_t n = v.size(); v.resize(n + k); memcpy(v.data() + n, p, k); }
    void fill(uint8_t x, size_t k) { v.insert(v.end(), k, x); }
};

constexpr size_t kBlock = 65280;     // raw bytes per BGZF block

// raw -> BGZF blocks appended to `comp`; block_off gets the offset of every block inside `comp`
void deflate_blocks(const uint8_t *raw, size_t n, std::vector<uint8_t> &comp, std::vector<uint64_t> &block_off) {
    static const uint8_t hdr[16] = {0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 0xff, 6, 0, 'B', 'C', 2, 0};
    z_stream zs{};
    deflateInit2(&zs, 1, Z_DEFLATED, -15, 8, Z_DEFAULT_STRATEGY);
    for (size_t o = 0; o < n; o += kBlock) {
        const size_t len = std::min(kBlock, n - o);
        const size_t at = comp.size();
        block_off.push_back(at);
        comp.resize(at + 18 + compressBound((uLong)len) + 8);
        deflateReset(&zs);
        zs.next_in = const_cast<uint8_t *>(raw + o); zs.avail_in = (uInt)len;
        zs.next_out = comp.data() + at + 18; zs.avail_out = (uInt)(comp.size() - at - 26);
        deflate(&zs, Z_FINISH);
        size_t clen = zs.total_out;
        if (clen + 26 > 65536) {                     // incompressible: stored block (cannot happen at 65280 raw bytes, kept for safety)
            deflateEnd(&zs); deflateInit2(&zs, 0, Z_DEFLATED, -15, 8, Z_DEFAULT_STRATEGY);
            zs.next_in = const_cast<uint8_t *>(raw + o); zs.avail_in = (uInt)len;
            zs.next_out = comp.data() + at + 18; zs.avail_out = (uInt)(comp.size() - at - 26);
            deflate(&zs, Z_FINISH); clen = zs.total_out;
            deflateEnd(&zs); deflateInit2(&zs, 1, Z_DEFLATED, -15, 8, Z_DEFAULT_STRATEGY);
        }
        uint8_t *c = comp.data() + at;
        memcpy(c, hdr, 16);
        const uint16_t bsize = (uint16_t)(clen + 25);
        memcpy(c + 16, &bsize, 2);
        const uint32_t crc = (uint32_t)crc32(crc32(0L, Z_NULL, 0), raw + o, (uInt)len), isize = (uint32_t)len;
        memcpy(c + 18 + clen, &crc, 4); memcpy(c + 18 + clen + 4, &isize, 4);
        comp.resize(at + 18 + clen + 8);
    }
    deflateEnd(&zs);
}

inline uint64_t xs(uint64_t &s) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return s; }

struct Group {
    std::vector<uint8_t> comp;
    std::vector<uint64_t> block_off;
    // segment starts that fall into this group: (segment, raw offset of the record inside the group)
    std::vector<std::pair<uint32_t, uint64_t>> seg_at;
    uint64_t raw_bytes = 0;
};

}  // namespace

// Returns 0, or 10 when the file cannot be written.  voff_out (may be NULL): [n_seg + 1] BGZF virtual offsets of the
// first record of every segment of the batch and of the end of the records.
HAPI int host_bam_write_ex(const char *path, const char *const *contig_names, const unsigned *contig_len, int n_contigs,
                           const rsqc_batch *b, const char *ch_tag, const char *filter_tag, int threads, int seq_mode,
                           int write_bai, unsigned long long *voff_out) {
    FILE *fp = fopen(path, "wb");
    if (!fp) return 10;
    if (threads < 1) threads = 1;
    WorkPool pool(threads);
    uint64_t file_off = 0;
    {   // header
        Out h;
        std::string text = "@HD\tVN:1.6\tSO:coordinate\n";
        for (int i = 0; i < n_contigs; ++i) text += std::string("@SQ\tSN:") + contig_names[i] + "\tLN:" + std::to_string(contig_len[i]) + "\n";
        h.bytes("BAM\1", 4); h.u32((uint32_t)text.size()); h.bytes(text.data(), text.size()); h.u32((uint32_t)n_contigs);
        for (int i = 0; i < n_contigs; ++i) {
            const size_t l = strlen(contig_names[i]) + 1;
            h.u32((uint32_t)l); h.bytes(contig_names[i], l); h.u32(contig_len[i]);
        }
        std::vector<uint8_t> comp; std::vector<uint64_t> bo;
        deflate_blocks(h.v.data(), h.v.size(), comp, bo);
        if (fwrite(comp.data(), 1, comp.size(), fp) != comp.size()) { fclose(fp); return 10; }
        file_off += comp.size();
    }
    const uint64_t GROUP = 1u << 15;
    const uint64_t n_groups = (b->n + GROUP - 1) / GROUP;
    std::vector<uint64_t> voff((size_t)b->n_seg + 1, 0);
    const size_t wave = (size_t)threads * 2;
    std::vector<Group> groups(wave);
    for (uint64_t g0 = 0; g0 < n_groups; g0 += wave) {
        const size_t ng = (size_t)std::min<uint64_t>(wave, n_groups - g0);
        pool.run(ng, [&](size_t k) {
            Group &G = groups[k];
            G.comp.clear(); G.block_off.clear(); G.seg_at.clear();
            const uint64_t r0 = (g0 + k) * GROUP, r1 = std::min<uint64_t>(r0 + GROUP, b->n);
            Out raw; raw.v.reserve((size_t)(r1 - r0) * 300);
            uint32_t seg = 0;
            { uint32_t lo = 0, hi = b->n_seg; while (hi - lo > 1) { const uint32_t m = (lo + hi) >> 1; if (b->seg_start[m] <= r0) lo = m; else hi = m; } seg = lo; }
            uint32_t w = 0;
            { uint32_t lo = 0, hi = b->n_wide; while (lo < hi) { const uint32_t m = (lo + hi) >> 1; if (b->wide_index[m] < r0) lo = m + 1; else hi = m; } w = lo; }
            for (uint64_t i = r0; i < r1; ++i) {
                while (seg + 1 < b->n_seg && b->seg_start[seg + 1] <= i) ++seg;
                if (b->n_seg && b->seg_start[seg] == i) G.seg_at.emplace_back(seg, raw.v.size());
                const int32_t tid = b->n_seg ? b->seg_tid[seg] : -1;
                const rsqc_rec_core &co = b->core[i]; const rsqc_rec_aux &au = b->aux[i];
                int32_t lq = au.l_qseq, nm = au.nm; uint32_t nc = au.n_cigar;
                if (au.l_qseq == RSQC_LQSEQ_ESCAPE || au.nm == RSQC_NM_ESCAPE || au.n_cigar == RSQC_NCIGAR_ESCAPE) {
                    while (w < b->n_wide && b->wide_index[w] < i) ++w;
                    if (w < b->n_wide && b->wide_index[w] == i) { lq = b->wide_l_qseq[w]; nm = b->wide_nm[w]; nc = b->wide_n_cigar[w]; }
                }
                char hexname[17];
                const char *name; size_t nlen;
                if (b->qname) { name = b->qname + b->qname_off[i]; nlen = b->qname_off[i + 1] - b->qname_off[i]; }
                else { snprintf(hexname, sizeof hexname, "%016llx", (unsigned long long)au.qhash); name = hexname; nlen = 16; }
                const int32_t mtid = (au.tagbits & RSQC_TB_MTID_SAME) ? tid : (tid + 1 < n_contigs ? tid + 1 : (tid != 0 ? 0 : -1));
                uint8_t tags[24]; size_t nt = 0;
                if (au.tagbits & RSQC_TB_HAS_NM) {
                    tags[nt++] = 'N'; tags[nt++] = 'M';
                    if (nm >= 0 && nm < 256) { tags[nt++] = 'C'; tags[nt++] = (uint8_t)nm; } else { tags[nt++] = 'i'; memcpy(tags + nt, &nm, 4); nt += 4; }
                }
                if (au.tagbits & RSQC_TB_HAS_CH) { tags[nt++] = (uint8_t)ch_tag[0]; tags[nt++] = (uint8_t)ch_tag[1]; tags[nt++] = 'Z'; tags[nt++] = '1'; tags[nt++] = 0; }
                if (au.tagbits & RSQC_TB_FILTER0) { tags[nt++] = (uint8_t)filter_tag[0]; tags[nt++] = (uint8_t)filter_tag[1]; tags[nt++] = 'i'; const uint32_t one = 1; memcpy(tags + nt, &one, 4); nt += 4; }
                const size_t l_seq = lq < 0 ? 0 : (size_t)lq;
                raw.u32((uint32_t)(32 + nlen + 1 + 4 * (size_t)nc + (l_seq + 1) / 2 + l_seq + nt));
                raw.u32((uint32_t)tid); raw.u32((uint32_t)co.pos);
                raw.u8((uint8_t)(nlen + 1)); raw.u8(au.mapq); raw.u16(4680); raw.u16((uint16_t)nc); raw.u16(au.flag);
                raw.u32((uint32_t)lq); raw.u32((uint32_t)mtid); raw.u32((uint32_t)co.mpos); raw.u32((uint32_t)co.isize);
                raw.bytes(name, nlen); raw.u8(0);
                raw.bytes(b->cigar + co.cigar_off, 4 * (size_t)nc);
                if (seq_mode == 0) { raw.fill(0x11, (l_seq + 1) / 2); raw.fill(0xff, l_seq); }
                else {
                    uint64_t s = au.qhash ^ ((uint64_t)au.flag << 48) ^ 0x9E3779B97F4A7C15ull; if (!s) s = 1;
                    static const uint8_t base4[4] = {1, 2, 4, 8};
                    const size_t at = raw.v.size(); raw.v.resize(at + (l_seq + 1) / 2 + l_seq);
                    uint8_t *sq = raw.v.data() + at, *ql = sq + (l_seq + 1) / 2;
                    for (size_t k2 = 0; k2 < (l_seq + 1) / 2; k2 += 16) {
                        uint64_t r = xs(s);
                        for (size_t j = k2; j < std::min(k2 + 16, (l_seq + 1) / 2); ++j, r >>= 4) sq[j] = (uint8_t)(base4[r & 3] << 4 | base4[(r >> 2) & 3]);
                    }
                    // binned qualities with runs: mostly 37, excursions to 25 / 11 / 2 that last a few bases
                    static const uint8_t qbin[8] = {37, 37, 37, 37, 37, 25, 11, 2};
                    uint8_t q = 37; uint64_t r = xs(s); int left = 0;
                    for (size_t j = 0; j < l_seq; ++j) {
                        if (left == 0) { r = xs(s); left = 12; }
                        if ((r & 31) < 3) q = qbin[(r >> 5) & 7];
                        r >>= 5; --left;
                        ql[j] = q;
                    }
                }
                raw.bytes(tags, nt);
            }
            G.raw_bytes = raw.v.size();
            deflate_blocks(raw.v.data(), raw.v.size(), G.comp, G.block_off);
        });
        for (size_t k = 0; k < ng; ++k) {
            Group &G = groups[k];
            for (auto &sa : G.seg_at) {
                const size_t blk = (size_t)(sa.second / kBlock);
                voff[sa.first] = ((file_off + G.block_off[blk]) << 16) | (sa.second % kBlock);
            }
            if (fwrite(G.comp.data(), 1, G.comp.size(), fp) != G.comp.size()) { fclose(fp); return 10; }
            file_off += G.comp.size();
        }
    }
    voff[b->n_seg] = file_off << 16;                   // the EOF block starts here
    static const uint8_t eof[28] = {0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 0xff, 6, 0, 'B', 'C', 2, 0, 0x1b, 0, 3, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    fwrite(eof, 1, 28, fp);
    if (fclose(fp) != 0) return 10;
    if (voff_out) memcpy(voff_out, voff.data(), voff.size() * 8);
    if (write_bai) {
        // Minimal BAI: per reference only the metadata pseudo-bin 37450 {ref_beg, ref_end, n_mapped, n_unmapped} that samtools
        // writes too -- what a by-contig reader needs (rsqc_host::BamReader::seek_contig); no binning / linear index.
        Out o;
        o.bytes("BAI\1", 4); o.u32((uint32_t)n_contigs);
        std::vector<int> seg_of((size_t)n_contigs, -1);
        for (uint32_t s = 0; s < b->n_seg; ++s) if (b->seg_tid[s] >= 0 && b->seg_tid[s] < n_contigs && seg_of[(size_t)b->seg_tid[s]] < 0) seg_of[(size_t)b->seg_tid[s]] = (int)s;
        uint64_t no_coor = 0;
        for (uint32_t s = 0; s < b->n_seg; ++s) if (b->seg_tid[s] < 0) no_coor += b->seg_start[s + 1] - b->seg_start[s];
        for (int c = 0; c < n_contigs; ++c) {
            const int s = seg_of[(size_t)c];
            if (s < 0) { o.u32(0); o.u32(0); continue; }
            o.u32(1); o.u32(37450); o.u32(2);
            const uint64_t beg = voff[(size_t)s], end = voff[(size_t)s + 1], n_rec = b->seg_start[s + 1] - b->seg_start[s], zero = 0;
            o.bytes(&beg, 8); o.bytes(&end, 8); o.bytes(&n_rec, 8); o.bytes(&zero, 8);
            o.u32(0);                                  // n_intv
        }
        o.bytes(&no_coor, 8);
        FILE *fi = fopen((std::string(path) + ".bai").c_str(), "wb");
        if (!fi) return 10;
        fwrite(o.v.data(), 1, o.v.size(), fi);
        fclose(fi);
    }
    return 0;
}

HAPI int host_bam_write(const char *path, const char *const *contig_names, const unsigned *contig_len, int n_contigs,
                        const rsqc_batch *b, const char *ch_tag, const char *filter_tag, int threads) {
    return host_bam_write_ex(path, contig_names, contig_len, n_contigs, b, ch_tag, filter_tag, threads, 0, 0, nullptr);
}
