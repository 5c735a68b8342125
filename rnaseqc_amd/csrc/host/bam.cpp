#include "bam.hpp"

#include <zlib.h>

#include <algorithm>
#include <cstring>
#include <stdexcept>

namespace rsqc_host {

void HostBatch::clear() {
    core.clear(); aux.clear(); cigar.clear(); seg_tid.clear(); seg_start.clear();
    wide_index.clear(); wide_nm.clear(); wide_lq.clear(); wide_ncig.clear();
}

rsqc_batch HostBatch::view() {
    if (seg_start.size() == seg_tid.size()) seg_start.push_back(core.size());
    else seg_start.back() = core.size();
    rsqc_batch b{};
    b.n = core.size(); b.file_index_base = file_index_base;
    b.core = core.data(); b.aux = aux.data(); b.cigar = cigar.data(); b.n_cigar_total = cigar.size();
    b.n_seg = (uint32_t)seg_tid.size(); b.seg_tid = seg_tid.data(); b.seg_start = seg_start.data();
    b.n_wide = (uint32_t)wide_index.size(); b.wide_index = wide_index.data(); b.wide_nm = wide_nm.data();
    b.wide_l_qseq = wide_lq.data(); b.wide_n_cigar = wide_ncig.data();
    return b;
}

BamReader::~BamReader() { if (fp_) fclose(fp_); }

void BamReader::set_tags(const std::string &chimeric, const std::vector<std::string> &filters) {
    ch_tag_ = chimeric; filter_tags_ = filters;
}

static inline uint32_t le32(const uint8_t *p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24); }
static inline uint16_t le16(const uint8_t *p) { return (uint16_t)(p[0] | (p[1] << 8)); }

bool BamReader::inflate_block() {
    uint8_t hdr[18];
    const size_t got = fread(hdr, 1, 18, fp_);
    if (got == 0) { eof_ = true; return false; }
    if (got != 18 || hdr[0] != 0x1f || hdr[1] != 0x8b || hdr[2] != 8 || !(hdr[3] & 4)) throw std::runtime_error("not a BGZF block");
    const uint16_t xlen = le16(hdr + 10);
    // the BC subfield is the first (and normally only) extra field
    uint32_t bsize = 0;
    std::vector<uint8_t> extra(xlen);
    memcpy(extra.data(), hdr + 12, std::min<size_t>(6, xlen));
    if (xlen > 6 && fread(extra.data() + 6, 1, xlen - 6, fp_) != (size_t)(xlen - 6)) throw std::runtime_error("truncated BGZF block");
    for (size_t o = 0; o + 4 <= xlen;) {
        const uint16_t slen = le16(extra.data() + o + 2);
        if (extra[o] == 'B' && extra[o + 1] == 'C' && slen == 2) bsize = (uint32_t)le16(extra.data() + o + 4) + 1;
        o += 4 + slen;
    }
    if (!bsize) throw std::runtime_error("BGZF block without BC field");
    const size_t clen = bsize - xlen - 12 - 8;
    cbuf_.resize(clen + 8);
    if (fread(cbuf_.data(), 1, clen + 8, fp_) != clen + 8) throw std::runtime_error("truncated BGZF block");
    const uint32_t isize = le32(cbuf_.data() + clen + 4);
    const size_t old = buf_.size();
    buf_.resize(old + isize);
    if (isize) {
        z_stream zs{};
        if (inflateInit2(&zs, -15) != Z_OK) throw std::runtime_error("zlib init failed");
        zs.next_in = cbuf_.data(); zs.avail_in = (uInt)clen;
        zs.next_out = buf_.data() + old; zs.avail_out = isize;
        const int rc = inflate(&zs, Z_FINISH);
        inflateEnd(&zs);
        if (rc != Z_STREAM_END) throw std::runtime_error("BGZF inflate failed");
    }
    return true;
}

bool BamReader::fill(size_t need) {
    if (buf_.size() - pos_ >= need) return true;
    if (pos_ > (1u << 22)) { buf_.erase(buf_.begin(), buf_.begin() + (long)pos_); pos_ = 0; }
    while (buf_.size() - pos_ < need) { if (eof_ || !inflate_block()) break; }
    return buf_.size() - pos_ >= need;
}

bool BamReader::open(const std::string &path) {
    fp_ = fopen(path.c_str(), "rb");
    if (!fp_) return false;
    try {
        if (!fill(12) || memcmp(buf_.data() + pos_, "BAM\1", 4) != 0) return false;
        const uint32_t l_text = le32(buf_.data() + pos_ + 4);
        if (!fill(12 + (size_t)l_text)) return false;
        const uint32_t n_ref = le32(buf_.data() + pos_ + 8 + l_text);
        pos_ += 12 + l_text;
        for (uint32_t i = 0; i < n_ref; ++i) {
            if (!fill(4)) return false;
            const uint32_t l_name = le32(buf_.data() + pos_);
            if (!fill(8 + (size_t)l_name)) return false;
            names_.emplace_back((const char *)buf_.data() + pos_ + 4, l_name ? l_name - 1 : 0);
            pos_ += 8 + l_name;
        }
    } catch (std::exception &) { return false; }
    return true;
}

// SeqLib::BamRecord::GetIntTag: an integer-typed aux field (htslib bam_aux2i)
static bool aux_int(const uint8_t *v, char type, int32_t &out) {
    switch (type) {
    case 'c': out = (int8_t)v[0]; return true;
    case 'C': out = v[0]; return true;
    case 's': out = (int16_t)le16(v); return true;
    case 'S': out = le16(v); return true;
    case 'i': out = (int32_t)le32(v); return true;
    case 'I': out = (int32_t)le32(v); return true;
    default: return false;
    }
}

size_t BamReader::read_batch(HostBatch &out, size_t max_records) {
    size_t n = 0;
    while (n < max_records) {
        if (!fill(4)) break;
        const uint32_t block_size = le32(buf_.data() + pos_);
        if (!fill(4 + (size_t)block_size)) throw std::runtime_error("truncated BAM record");
        const uint8_t *r = buf_.data() + pos_ + 4;
        const int32_t tid = (int32_t)le32(r), pos = (int32_t)le32(r + 4);
        const uint8_t l_read_name = r[8], mapq = r[9];
        const uint16_t n_cigar = le16(r + 12), flag = le16(r + 14);
        const int32_t l_seq = (int32_t)le32(r + 16), mtid = (int32_t)le32(r + 20), mpos = (int32_t)le32(r + 24), isize = (int32_t)le32(r + 28);
        const char *qname = (const char *)r + 32;
        const uint8_t *cig = r + 32 + l_read_name;
        const uint8_t *auxp = cig + 4 * (size_t)n_cigar + (size_t)((l_seq + 1) / 2) + (size_t)l_seq;
        const uint8_t *end = r + block_size;
        if (out.seg_tid.empty() || out.seg_tid.back() != tid) { out.seg_tid.push_back(tid); out.seg_start.push_back(out.core.size()); }
        rsqc_rec_core co{pos, mpos, isize, (uint32_t)out.cigar.size()};
        rsqc_rec_aux au{};
        const size_t qlen = l_read_name ? strnlen(qname, l_read_name) : 0;
        au.qhash = rsqc_qname_hash(qname, qlen);
        au.flag = flag; au.mapq = mapq;
        uint8_t tagbits = (tid == mtid) ? RSQC_TB_MTID_SAME : 0;
        int32_t nm = 0;
        // aux fields
        for (const uint8_t *p = auxp; p + 3 <= end;) {
            const char t0 = (char)p[0], t1 = (char)p[1], type = (char)p[2];
            const uint8_t *v = p + 3;
            size_t vlen = 0;
            switch (type) {
            case 'A': case 'c': case 'C': vlen = 1; break;
            case 's': case 'S': vlen = 2; break;
            case 'i': case 'I': case 'f': vlen = 4; break;
            case 'Z': case 'H': vlen = strnlen((const char *)v, (size_t)(end - v)) + 1; break;
            case 'B': { const char st = (char)v[0]; const uint32_t cnt = le32(v + 1);
                        const size_t es = (st == 'c' || st == 'C') ? 1 : (st == 's' || st == 'S') ? 2 : 4; vlen = 5 + es * (size_t)cnt; break; }
            default: vlen = (size_t)(end - v); break;
            }
            if (t0 == 'N' && t1 == 'M') { int32_t x; if (aux_int(v, type, x)) { nm = x; tagbits |= RSQC_TB_HAS_NM; } }
            if (ch_tag_.size() == 2 && t0 == ch_tag_[0] && t1 == ch_tag_[1]) {         // readStringTag, src/RNASeQC.cpp:780-800
                if (type == 'Z' || (type == 'A' && v[0] != 0)) tagbits |= RSQC_TB_HAS_CH;
            }
            for (size_t k = 0; k < filter_tags_.size() && k < RSQC_MAX_FILTER_TAGS; ++k)   // GetTag: Z, integer or float
                if (filter_tags_[k].size() == 2 && t0 == filter_tags_[k][0] && t1 == filter_tags_[k][1]) {
                    int32_t x;
                    if (type == 'Z' || type == 'f' || aux_int(v, type, x)) tagbits |= (uint8_t)(RSQC_TB_FILTER0 << k);
                }
            p = v + vlen;
        }
        const bool wide = l_seq >= RSQC_LQSEQ_ESCAPE || l_seq < 0 || nm >= RSQC_NM_ESCAPE || nm < 0 || n_cigar >= RSQC_NCIGAR_ESCAPE;
        au.l_qseq = wide && (l_seq >= RSQC_LQSEQ_ESCAPE || l_seq < 0) ? RSQC_LQSEQ_ESCAPE : (uint16_t)l_seq;
        au.nm = wide && (nm >= RSQC_NM_ESCAPE || nm < 0) ? RSQC_NM_ESCAPE : (uint8_t)nm;
        au.n_cigar = n_cigar >= RSQC_NCIGAR_ESCAPE ? RSQC_NCIGAR_ESCAPE : (uint8_t)n_cigar;
        au.tagbits = tagbits;
        if (wide) { out.wide_index.push_back(out.core.size()); out.wide_nm.push_back(nm); out.wide_lq.push_back(l_seq); out.wide_ncig.push_back(n_cigar); }
        for (uint16_t k = 0; k < n_cigar; ++k) out.cigar.push_back(le32(cig + 4 * (size_t)k));
        out.core.push_back(co); out.aux.push_back(au);
        pos_ += 4 + (size_t)block_size;
        ++n; ++n_read_;
    }
    return n;
}

}  // namespace rsqc_host
