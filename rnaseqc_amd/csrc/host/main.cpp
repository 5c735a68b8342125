// main.cpp -- `rnaseqc [OPTIONS] gtf bam output`: the reference's command line (flag table
// src/RNASeQC.cpp:39-65, defaults :87-100, exit codes :678-766) in front of the HIP hot path.
// Host side only: GTF/BED ingest, BAM decode into SoA batches, report writers.  Every per-record
// computation happens on the GPU through the C ABI (include/rnaseqc_amd.h); without a GPU the
// program exits with code 10.
#include <sys/stat.h>

#include <memory>
#include <sys/types.h>

#include <algorithm>
#include <chrono>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <future>
#include <iostream>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>

#include "bam.hpp"
#include "bgzf_feed.hpp"
#include "fasta.hpp"
#include "gtf.hpp"
#include "report.hpp"

using namespace rsqc_host;

namespace {

const char *VERSION = "RNASeQC 2.4.3";

struct ParseError : std::runtime_error { using std::runtime_error::runtime_error; };          // exit 5
struct ValidationError : std::runtime_error { using std::runtime_error::runtime_error; };    // exit 6
struct Help {};

struct Options {
    std::vector<std::string> positional;
    std::string sample, bed, fasta, stranded, chimeric_tag = "ch";
    bool has_sample = false, has_bed = false, has_fasta = false, has_stranded = false;
    bool legacy = false, exclude_chimeric = false, unpaired = false, rpkm = false, coverage = false, version = false;
    int verbosity = 0;
    long chimeric_distance = 2000000; unsigned long fragment_samples = 1000000, mapq = 255, base_mismatch = 6;
    bool has_mapq = false;
    long bias_offset = 0, bias_window = 100; unsigned long bias_gene_length = 200, coverage_mask = 500, detection = 5;
    std::vector<std::string> tags;
    int gpus = 0;                      // --gpus (extension): GPUs to shard the BAM over by contig; 0 = RSQC_GPUS or 1
};

void usage(std::ostream &o) {
    o << "  rnaseqc {OPTIONS} [gtf] [bam] [output]\n\n    " << VERSION << "\n\n  OPTIONS:\n\n"
         "      -h, --help                        Display this message and quit\n"
         "      --version                         Display the version and quit\n"
         "      gtf                               The input GTF file containing features to check the bam against\n"
         "      bam                               The input SAM/BAM file containing reads to process\n"
         "      output                            Output directory\n"
         "      -s[sample], --sample=[sample]     The name of the current sample.  Default: The bam's filename\n"
         "      --bed=[BEDFILE]                   Optional input BED file containing non-overlapping exons used for fragment size calculations\n"
         "      --fasta=[fasta]                   Optional input FASTA (with .fai index): enables the GC-content statistics (CRAM input is not supported)\n"
         "      --chimeric-distance=[DISTANCE]    Maximum accepted distance between read mates. Default: 2000000 [bp]\n"
         "      --fragment-samples=[SAMPLES]      Number of fragment size samples. Default: 1000000\n"
         "      -q[QUALITY], --mapping-quality=[QUALITY]  Lower bound on read quality for exon coverage counting. Default: 255\n"
         "      --base-mismatch=[MISMATCHES]      Maximum number of allowed mismatches. Default: 6\n"
         "      --offset=[OFFSET]                 Offset into the gene for the 3' and 5' windows. Default: 0 [bp]\n"
         "      --window-size=[SIZE]              Size of the 3' and 5' windows. Default: 100 [bp]\n"
         "      --gene-length=[LENGTH]            Minimum size of a gene for bias calculation. Default: 200 [bp]\n"
         "      --legacy                          Use legacy counting rules.  Gene and exon counts match output of RNA-SeQC 1.1.9\n"
         "      --stranded=[stranded]             'RF', 'rf', 'FR', or 'fr'\n"
         "      -v, --verbose                     Give some feedback; twice for progress updates\n"
         "      -t[TAG...], --tag=[TAG...]        Filter out reads with the specified tag\n"
         "      --chimeric-tag=[TAG]              Reads marked with this tag are chimeric. Default: ch\n"
         "      --exclude-chimeric                Exclude chimeric reads from the read counts\n"
         "      -u, --unpaired                    Allow unpaired reads to be quantified\n"
         "      --rpkm                            Output gene RPKM values instead of TPMs\n"
         "      --coverage                        Write per-transcript coverage statistics to a table\n"
         "      --coverage-mask=[SIZE]            Bases masked at both transcript ends. Default: 500bp\n"
         "      -d[threshold], --detection-threshold=[threshold]  Counts to call a gene detected. Default: 5 reads\n"
         "      --gpus=[N]                        (extension) Shard the BAM by contig over N GPUs of this node; needs [bam].bai. Default: 1\n";
}

long to_long(const std::string &flag, const std::string &v) {
    char *e = nullptr;
    const long x = strtol(v.c_str(), &e, 10);
    if (v.empty() || (e && *e)) throw ParseError("Argument '" + flag + "' received invalid value type '" + v + "'");
    return x;
}
unsigned long to_ulong(const std::string &flag, const std::string &v) {
    char *e = nullptr;
    if (v.empty() || v[0] == '-') throw ParseError("Argument '" + flag + "' received invalid value type '" + v + "'");
    const unsigned long x = strtoul(v.c_str(), &e, 10);
    if (e && *e) throw ParseError("Argument '" + flag + "' received invalid value type '" + v + "'");
    return x;
}

Options parse(int argc, char **argv) {
    Options o;
    bool only_positional = false;
    for (int i = 1; i < argc; ++i) {
        std::string a = argv[i];
        if (only_positional || a.size() < 2 || a[0] != '-') { o.positional.push_back(a); continue; }
        if (a == "--") { only_positional = true; continue; }
        std::string name, value; bool has_value = false;
        if (a[1] == '-') {
            const size_t eq = a.find('=');
            name = a.substr(2, eq == std::string::npos ? std::string::npos : eq - 2);
            if (eq != std::string::npos) { value = a.substr(eq + 1); has_value = true; }
        } else {
            // short flags: -v, -vv, -u, -h may be bundled; -s/-q/-t/-d take the rest of the token or the next one
            size_t k = 1;
            bool consumed = false;
            while (k < a.size() && !consumed) {
                const char c = a[k];
                if (c == 'v') { ++o.verbosity; ++k; }
                else if (c == 'u') { o.unpaired = true; ++k; }
                else if (c == 'h') throw Help();
                else if (c == 's' || c == 'q' || c == 't' || c == 'd') {
                    std::string v = a.substr(k + 1);
                    if (v.empty()) { if (i + 1 >= argc) throw ParseError(std::string("Flag '") + c + "' requires an argument but received none"); v = argv[++i]; }
                    const std::string f(1, c);
                    if (c == 's') { o.sample = v; o.has_sample = true; }
                    else if (c == 'q') { o.mapq = to_ulong(f, v); o.has_mapq = true; }
                    else if (c == 't') o.tags.push_back(v);
                    else o.detection = to_ulong(f, v);
                    consumed = true;
                } else throw ParseError(std::string("Flag could not be matched: '") + c + "'");
            }
            continue;
        }
        auto need = [&]() -> std::string {
            if (has_value) return value;
            if (i + 1 >= argc) throw ParseError("Flag '" + name + "' requires an argument but received none");
            return std::string(argv[++i]);
        };
        if (name == "help") throw Help();
        else if (name == "version") o.version = true;
        else if (name == "sample") { o.sample = need(); o.has_sample = true; }
        else if (name == "bed") { o.bed = need(); o.has_bed = true; }
        else if (name == "fasta") { o.fasta = need(); o.has_fasta = true; }
        else if (name == "chimeric-distance") o.chimeric_distance = to_long(name, need());
        else if (name == "fragment-samples") o.fragment_samples = to_ulong(name, need());
        else if (name == "mapping-quality") { o.mapq = to_ulong(name, need()); o.has_mapq = true; }
        else if (name == "base-mismatch") o.base_mismatch = to_ulong(name, need());
        else if (name == "offset") o.bias_offset = to_long(name, need());
        else if (name == "window-size") o.bias_window = to_long(name, need());
        else if (name == "gene-length") o.bias_gene_length = to_ulong(name, need());
        else if (name == "legacy") o.legacy = true;
        else if (name == "stranded") { o.stranded = need(); o.has_stranded = true; }
        else if (name == "verbose") ++o.verbosity;
        else if (name == "tag") o.tags.push_back(need());
        else if (name == "chimeric-tag") o.chimeric_tag = need();
        else if (name == "exclude-chimeric") o.exclude_chimeric = true;
        else if (name == "unpaired") o.unpaired = true;
        else if (name == "rpkm") o.rpkm = true;
        else if (name == "coverage") o.coverage = true;
        else if (name == "gpus") o.gpus = (int)to_ulong(name, need());
        else if (name == "coverage-mask") o.coverage_mask = to_ulong(name, need());
        else if (name == "detection-threshold") o.detection = to_ulong(name, need());
        else throw ParseError("Flag could not be matched: " + name);
    }
    return o;
}

bool make_dirs(const std::string &path) {          // boost::filesystem::create_directories
    std::string cur;
    for (size_t i = 0; i <= path.size(); ++i) {
        if (i == path.size() || path[i] == '/') {
            if (!cur.empty() && cur != "/") {
                struct stat st;
                if (stat(cur.c_str(), &st) != 0) { if (mkdir(cur.c_str(), 0777) != 0) return false; }
                else if (!S_ISDIR(st.st_mode)) return false;
            }
        }
        if (i < path.size()) cur += path[i];
    }
    return true;
}

std::string basename_of(const std::string &p) {
    const size_t s = p.find_last_of('/');
    return s == std::string::npos ? p : p.substr(s + 1);
}

}  // namespace

// ---- one process, several GPUs: the BAM sharded by contig ------------------------------------------
// Every GPU holds the whole annotation and owns a set of contigs (longest-processing-time packing on the index's record
// counts); a host thread per GPU reads ITS contigs through the BAM index (BamReader::seek) and feeds its context; at
// end of file the contexts' result ranges are summed onto the first GPU (rsqc_reduce_peer: peer copies over xGMI), and the
// two order-dependent outputs are composed on the host from the shards' summaries (rsqc_shard_summary).
struct Shard {
    rsqc_ctx *gpu = nullptr; int device = 0;
    std::vector<int> contigs; bool tail = false;
    uint64_t load = 0, n_records = 0;
    int rc = RSQC_OK; std::string error; bool unsorted = false; std::vector<std::string> bad_refid;
};

constexpr int kFileIndexShift = 36;      // virtual file index of a batch: (contig << 36) + records of the contig before it

// ---- device decode (rsqc_decode_*): the default.  RSQC_DECODE=host keeps inflate + record parsing on the CPU threads.
// The device path reads the file with pread through the block feeder, which needs a regular file: a FIFO, /dev/stdin or a
// process substitution is streamed by the host reader instead (as every input was before the device decode existed).
bool g_input_is_stream = false;
bool device_decode_wanted() {
    if (g_input_is_stream) return false;
    const char *e = getenv("RSQC_DECODE");
    return !(e && (!strcmp(e, "host") || !strcmp(e, "cpu")));
}
rsqc_decode_params decode_params(const Options &o, int n_ref, uint64_t file_index_base) {
    rsqc_decode_params dp{};
    dp.n_ref = n_ref;
    if (o.chimeric_tag.size() == 2) { dp.has_chimeric_tag = 1; dp.chimeric_tag[0] = o.chimeric_tag[0]; dp.chimeric_tag[1] = o.chimeric_tag[1]; }
    for (size_t k = 0; k < o.tags.size() && k < RSQC_MAX_FILTER_TAGS; ++k)
        if (o.tags[k].size() == 2) { dp.filter_tag[k][0] = o.tags[k][0]; dp.filter_tag[k][1] = o.tags[k][1]; }   // (other lengths never match)
    dp.file_index_base = file_index_base;
    return dp;
}
// inflated bytes a call of decode_range can hold: what the device's window buffers are sized for, once (rsqc_decode_begin)
uint64_t decode_reserve_bytes(uint64_t file_size) {
    const uint64_t max_out = getenv("RSQC_DECODE_MAX_OUT") ? (uint64_t)atoll(getenv("RSQC_DECODE_MAX_OUT")) : (uint64_t)1024 << 20;
    return std::min<uint64_t>(max_out + (1u << 20), file_size * 16 + (1u << 20));
}
// One stream of BGZF blocks [voff_beg, voff_end) through the GPU: the feeder reads and frames the blocks, every chunk is one
// rsqc_decode_submit.  on_window sees what each call decoded.  Returns an RSQC_* code; info describes the whole stream.
template <class F>
int decode_range(rsqc_ctx *gpu, BgzfFeeder &feed, const rsqc_decode_params &dp, uint64_t voff_beg, uint64_t voff_end, rsqc_decode_info &info, F &&on_window) {
    // Calls are large on purpose: the inflate kernel runs one wave per BGZF block, twenty waves per CU = 5 120 on the chip, and a
    // call's time is that of its LAST block: a call of 5 300 blocks (128 MB of a file compressed 3 x, round 4's chunk) runs 180 of
    // them on an empty chip.  Up to 1 GB of inflated data = 16 000 blocks per call whatever the file's compression (the feeder reads
    // what that takes, up to 512 MB of file), the device buffers sized for that once (profiles/r5_decode_chunk_sweep.txt).
    // (RSQC_DECODE_CHUNK / RSQC_DECODE_MAX_OUT: compressed bytes read per call at most / inflated bytes per call -- the tests use
    //  small values so that records straddle many calls)
    const size_t chunk = getenv("RSQC_DECODE_CHUNK") ? (size_t)atoll(getenv("RSQC_DECODE_CHUNK")) : (size_t)512 << 20;
    const uint64_t max_out = getenv("RSQC_DECODE_MAX_OUT") ? (uint64_t)atoll(getenv("RSQC_DECODE_MAX_OUT")) : (uint64_t)1024 << 20;
    rsqc_decode_params dpr = dp;
    dpr.pipelined = 1;                          // a call's records are reported by the call after it (the last by rsqc_decode_end)
    dpr.reserve_inflated_bytes = decode_reserve_bytes(feed.file_size());
    int rc = rsqc_decode_begin(gpu, &dpr);
    if (rc != RSQC_OK) return rc;
    const bool prof = getenv("RSQC_DECODE_PROFILE") != nullptr;
    double t_feed = 0;
    auto now = [] { return std::chrono::steady_clock::now(); };
    auto t0 = now();
    feed.start(voff_beg, voff_end, chunk, max_out);
    for (;;) {
        auto tf = now();
        BgzfFeeder::Chunk *ch = feed.next();
        t_feed += std::chrono::duration<double, std::milli>(now() - tf).count();
        if (!ch) break;
        if (ch->blocks.empty()) continue;
        rsqc_decode_window w{};
        rc = rsqc_decode_submit(gpu, ch->data, ch->total_bytes, ch->blocks.data(), (uint32_t)ch->blocks.size(), ch->skip, ch->limit, &w);
        if (rc != RSQC_OK) { rsqc_decode_info dropped{}; (void)rsqc_decode_end(gpu, &dropped); return rc; }
        if (w.n_records) on_window(w);
    }
    if (prof) fprintf(stderr, "[decode] host: CPU share of the inflate work at the end %.2f; %.1f ms waiting for file chunks of %.1f ms in the range\n", feed.cpu_share(), t_feed, std::chrono::duration<double, std::milli>(now() - t0).count());
    rc = rsqc_decode_end(gpu, &info);
    if (rc == RSQC_OK && info.last.n_records) on_window(info.last);
    return rc;
}

void shard_worker(Shard &sh, const std::string &bam_path, const Options &o, int threads, const std::vector<BamReader::ContigRange> &index,
                  int n_ref, uint64_t tail_voff, size_t BATCH, BgzfFeeder *device_feed) {
    try {
        if (device_feed) {                                             // device decode: the shard's feeder, opened and page-locked by the caller
            BgzfFeeder &feed = *device_feed;
            feed.read_threads = std::max(1, threads / 4);
            std::vector<int> ranges = sh.contigs;
            if (sh.tail) ranges.push_back(n_ref);
            for (int c : ranges) {
                const bool is_tail = c == n_ref;
                rsqc_decode_info di{};
                bool stale = false;
                sh.rc = decode_range(sh.gpu, feed, decode_params(o, n_ref, (uint64_t)c << kFileIndexShift), is_tail ? tail_voff : index[(size_t)c].beg,
                                     is_tail ? 0 : index[(size_t)c].end, di, [&](const rsqc_decode_window &w) {
                                         sh.n_records += w.n_records;
                                         for (uint32_t k = 0; k < w.n_runs; ++k) if (is_tail ? w.run_tid[k] >= 0 : w.run_tid[k] != c) stale = true;
                                     });
                if (sh.rc != RSQC_OK) { sh.error = rsqc_last_error(sh.gpu); return; }
                if (stale) { sh.rc = RSQC_ERR_ARG; sh.error = is_tail ? "placed records behind the last indexed contig: stale index?" : "records of another contig inside an indexed range: stale index?"; return; }
                if (di.unsorted) sh.unsorted = true;
                for (int k = 0; k < di.n_bad_refid && k < 64; ++k) if (sh.bad_refid.size() < 64) sh.bad_refid.push_back(di.bad_refid[k]);
            }
            sh.rc = rsqc_finalize_device(sh.gpu);
            if (sh.rc != RSQC_OK) sh.error = rsqc_last_error(sh.gpu);
            return;
        }
        HostBatch bufs[2];
        for (auto &hb : bufs) { hb.core.use_pinned(true); hb.aux.use_pinned(true); hb.qh2.use_pinned(true); hb.cigar.use_pinned(true); hb.core.reserve(BATCH); hb.aux.reserve(BATCH); hb.qh2.reserve(BATCH); hb.cigar.reserve(BATCH * 2); }
        int cur = 0; bool in_flight = false;
        std::vector<int> ranges = sh.contigs;
        if (sh.tail) ranges.push_back(n_ref);                          // the unplaced records behind the last contig
        for (int c : ranges) {
            BamReader bam;
            bam.set_threads(threads);
            if (!bam.open(bam_path)) { sh.rc = RSQC_ERR_ARG; sh.error = "Unable to open BAM file: " + bam_path; return; }
            bam.set_tags(o.chimeric_tag, o.tags);
            const bool is_tail = c == n_ref;
            if (!bam.seek(is_tail ? tail_voff : index[(size_t)c].beg, is_tail ? 0 : index[(size_t)c].end)) { sh.rc = RSQC_ERR_ARG; sh.error = "cannot seek in " + bam_path; return; }
            uint64_t left = is_tail ? ~0ull : (index[(size_t)c].n_records ? index[(size_t)c].n_records : ~0ull), done = 0;
            while (left) {
                HostBatch &hb = bufs[cur];
                hb.clear();
                hb.file_index_base = ((uint64_t)c << kFileIndexShift) + done;
                size_t n = bam.read_batch(hb, (size_t)std::min<uint64_t>(left, BATCH));
                bool last = false;
                if (n && !is_tail && hb.keep_leading_contig(c)) { n = hb.size(); last = true; }   // ran into the next contig
                if (is_tail && n) {                                    // (the tail holds unplaced records only: tid -1)
                    for (int32_t t : hb.seg_tid) if (t >= 0) { sh.rc = RSQC_ERR_ARG; sh.error = "placed records behind the last indexed contig: stale index?"; return; }
                }
                if (in_flight) { if ((sh.rc = rsqc_wait(sh.gpu)) != RSQC_OK) { sh.error = rsqc_last_error(sh.gpu); return; } in_flight = false; }
                if (n == 0) break;
                if (hb.unsorted) sh.unsorted = true;
                for (auto &nm : hb.bad_refid) if (sh.bad_refid.size() < 64) sh.bad_refid.push_back(nm);
                rsqc_batch view = hb.view();
                if ((sh.rc = rsqc_submit(sh.gpu, &view)) != RSQC_OK) { sh.error = rsqc_last_error(sh.gpu); return; }
                in_flight = true; cur ^= 1;
                done += n; sh.n_records += n;
                if (left != ~0ull) left -= std::min<uint64_t>(left, n);
                if (last) break;
            }
            if (in_flight) { if ((sh.rc = rsqc_wait(sh.gpu)) != RSQC_OK) { sh.error = rsqc_last_error(sh.gpu); return; } in_flight = false; }
        }
        sh.rc = rsqc_finalize_device(sh.gpu);
        if (sh.rc != RSQC_OK) sh.error = rsqc_last_error(sh.gpu);
    } catch (std::exception &e) { sh.rc = RSQC_ERR_ARG; sh.error = e.what(); }
}

// merged outputs that do not come out of the reduction
struct ShardMerge { int32_t read_length = 0; std::vector<int64_t> fsize; std::vector<uint64_t> fcount; uint32_t remaining = 0; };

int merge_shards(std::vector<Shard> &shards, uint32_t fragment_samples, ShardMerge &m, std::string &err) {
    struct Item { uint64_t file; const uint32_t *span; const int32_t *state; uint32_t n; };
    std::vector<Item> items;
    std::vector<std::pair<uint64_t, uint32_t>> samples;
    for (auto &sh : shards) {
        rsqc_shard_info si{};
        const int rc = rsqc_shard_summary(sh.gpu, &si);
        if (rc != RSQC_OK) { err = rsqc_last_error(sh.gpu); return rc; }
        for (uint32_t b = 0; b < si.n_batches; ++b)
            items.push_back(Item{si.batch_file_index[b], si.rl_span + si.rl_offset[b], si.rl_state + si.rl_offset[b], si.rl_offset[b + 1] - si.rl_offset[b]});
        for (uint32_t k = 0; k < si.n_samples; ++k) samples.emplace_back(si.sample_file_index[k], si.sample_size[k]);
    }
    // Read Length (src/RNASeQC.cpp:275-278): the batches of all shards in file order, each applied as its transfer function
    std::sort(items.begin(), items.end(), [](const Item &a, const Item &b) { return a.file < b.file; });
    uint32_t r = 0;
    for (auto &it : items) for (uint32_t k = 0; k < it.n; ++k) if (it.span[k] > r) { r = (uint32_t)it.state[k]; break; }
    m.read_length = (int32_t)r;
    // fragment sizes (src/Expression.cpp:482-540): the --fragment-samples first samples of the union, in file order
    const size_t keep = std::min<size_t>(samples.size(), fragment_samples);
    if (keep < samples.size()) std::nth_element(samples.begin(), samples.begin() + (long)keep, samples.end());
    std::vector<uint32_t> sz(keep);
    for (size_t k = 0; k < keep; ++k) sz[k] = samples[k].second;
    std::sort(sz.begin(), sz.end());
    for (size_t i = 0; i < keep;) { size_t j = i + 1; while (j < keep && sz[j] == sz[i]) ++j; m.fsize.push_back((int64_t)sz[i]); m.fcount.push_back((uint64_t)(j - i)); i = j; }
    m.remaining = fragment_samples - (uint32_t)keep;
    return RSQC_OK;
}

int main(int argc, char **argv) {
    using std::cerr; using std::cout; using std::endl;
    try {
        Options o = parse(argc, argv);
        if (o.version) { cout << VERSION << endl; return 0; }
        if (o.positional.size() < 1) throw ValidationError("No GTF file provided");
        if (o.positional.size() < 2) throw ValidationError("No BAM file provided");
        if (o.positional.size() < 3) throw ValidationError("No output directory provided");
        if (o.positional.size() > 3) throw ParseError("Passed in argument, but no positional arguments were ready to receive it: " + o.positional[3]);
        const std::string gtf_path = o.positional[0], bam_path = o.positional[1], out_dir = o.positional[2];
        int strand = RSQC_STRAND_UNKNOWN;
        if (o.has_stranded) {
            if (o.stranded == "RF" || o.stranded == "rf") strand = RSQC_STRAND_REVERSE;
            else if (o.stranded == "FR" || o.stranded == "fr") strand = RSQC_STRAND_FORWARD;
            else throw ValidationError("--stranded argument must be in {'RF', 'rf', 'FR', 'fr'}");
        }
        if (o.tags.size() > RSQC_MAX_FILTER_TAGS) { cerr << "at most " << RSQC_MAX_FILTER_TAGS << " --tag filters are supported" << endl; return 7; }

        rsqc_params P{};
        P.abi_version = RSQC_ABI_VERSION;
        P.device = getenv("RSQC_DEVICE") ? atoi(getenv("RSQC_DEVICE")) : 0;
        if (const char *e = getenv("RSQC_GPU_LIST")) if (*e) P.device = atoi(e);       // the first shard's context runs on the list's first device
        P.mapq_threshold = o.has_mapq ? (uint32_t)o.mapq : (o.legacy ? 4u : 255u);     // src/RNASeQC.cpp:90
        P.legacy = o.legacy ? 1 : 0;
        P.base_mismatch = (uint32_t)o.base_mismatch;
        P.chimeric_distance = (int32_t)o.chimeric_distance; P.fragment_samples = (uint32_t)o.fragment_samples;
        P.bias_offset = (int32_t)o.bias_offset; P.bias_window = (int32_t)o.bias_window; P.bias_gene_length = o.bias_gene_length;
        P.coverage_mask = (uint32_t)o.coverage_mask; P.stranded = strand; P.unpaired = o.unpaired; P.exclude_chimeric = o.exclude_chimeric;
        P.n_filter_tags = (int32_t)o.tags.size();
        const std::string SAMPLENAME = o.has_sample ? o.sample : basename_of(bam_path);

        // the HIP context comes up (~0.3 s) while the GTF is being parsed; its status is looked at where the
        // reference would first need it, so input errors keep their precedence and exit codes
        rsqc_ctx *gpu = nullptr;
        std::future<int> gpu_ready = std::async(std::launch::async, [&P, &gpu] { return rsqc_create(&P, &gpu); });
        // ... and so do the page-locked chunk buffers of the device decode's feeder (one GPU: ~1.3 GB, a few hundred ms of
        // page-locking that would otherwise sit between the GTF and the BAM loop).  A file that cannot be opened is reported
        // later, where the reference reports it.
        const auto t_start = std::chrono::steady_clock::now();
        { struct stat st_in; g_input_is_stream = stat(bam_path.c_str(), &st_in) == 0 && !S_ISREG(st_in.st_mode); }
        auto feeder_cpu_threads = [] {
            const int spare = effective_cpus() - 4;
            return getenv("RSQC_DECODE_CPU_THREADS") ? atoi(getenv("RSQC_DECODE_CPU_THREADS")) : (spare >= 4 ? spare : 0);
        };
        const size_t feeder_chunk = getenv("RSQC_DECODE_CHUNK") ? (size_t)atoll(getenv("RSQC_DECODE_CHUNK")) : (size_t)512 << 20;
        const bool feeder_prepin = !(getenv("RSQC_FEED_PREPIN") && !atoi(getenv("RSQC_FEED_PREPIN")));
        std::unique_ptr<BgzfFeeder> early_feed;
        std::future<bool> early_feed_ready;
        if (device_decode_wanted() && feeder_prepin && o.gpus <= 1 && !getenv("RSQC_GPUS") && !getenv("RSQC_GPU_LIST")) {
            early_feed.reset(new BgzfFeeder());
            BgzfFeeder *ef = early_feed.get();
            const int ct = feeder_cpu_threads();
            const uint64_t feeder_max_out = getenv("RSQC_DECODE_MAX_OUT") ? (uint64_t)atoll(getenv("RSQC_DECODE_MAX_OUT")) : (uint64_t)1024 << 20;
            early_feed_ready = std::async(std::launch::async, [ef, bam_path, ct, feeder_chunk, feeder_max_out] {
                try {
                    if (!ef->open(bam_path)) return false;
                    // (the room behind a chunk's file bytes is page-locked for the largest share the CPUs can reach: 12 threads have
                    //  settled at 0.10-0.25 of a call on every file measured; 0.5 only where there are the threads for it)
                    if (ct > 0) ef->set_cpu_share(ct, 0.15, std::min(0.5, std::max(0.1, 0.025 * ct)), feeder_max_out);
                    ef->reserve(feeder_chunk, feeder_max_out);
                    return true;
                } catch (std::exception &) { return false; }
            });
        }
        const auto t0 = std::chrono::steady_clock::now();
        Annotation ann;
        ann.legacy = o.legacy;
        FastaFile fasta;
        std::vector<std::vector<uint8_t>> fasta_seq;
        std::future<void> fasta_loaded;                                       // (declared last: its destructor joins the reader before the buffers go)
        if (o.has_fasta) {                                                    // src/RNASeQC.cpp:111-121: GTF openable, then Fasta::open
            { std::ifstream probe(gtf_path); if (!probe.is_open()) { cerr << "Unable to open GTF file: " << gtf_path << endl; return 10; } }
            fasta.open(o.fasta);                                              // FileError -> 10
            for (auto &e : fasta.index) ann.chromosome(e.name);               // the index names join chromosomeMap first (src/Fasta.cpp:93-94)
            if (o.verbosity > 1) cout << "A FASTA has been provided. This will enable GC-content statistics but adds additional runtime and memory costs" << endl;
            fasta_loaded = std::async(std::launch::async, [&fasta, &fasta_seq] { fasta.load(fasta_seq); });   // read beside the GTF parse
        }
        if (o.verbosity) cout << "Reading GTF Features..." << endl;
        ann.load_gtf(gtf_path);                                               // FileError -> 10, GtfError -> 11
        if (!(ann.gene_list.size() && ann.exon_list.size())) {
            cerr << "There were either no genes or no exons in the GTF" << endl;
            cerr << ann.gene_list.size() << " genes parsed" << endl << ann.exon_list.size() << " exons parsed" << endl;
            return 11;
        }
        const auto t1 = std::chrono::steady_clock::now();
        if (o.verbosity) cout << "Finished processing GTF in " << std::chrono::duration<double>(t1 - t0).count() << " seconds" << endl;
        std::vector<char> gtf_chrom(ann.chrom_name.size() + 1, 0);
        for (auto &r : ann.rows) if (!r.excluded) gtf_chrom[(size_t)r.chrom] = 1;
        if (o.has_bed) {
            if (o.verbosity) cout << "Parsing BED intervals for fragment size computations..." << endl;
            ann.load_bed(o.bed);
        }
        if (!make_dirs(out_dir)) { cerr << "Filesystem error:  cannot create " << out_dir << endl; return 8; }
        // The header: with the device decode (the default) the host only inflates the header's own blocks; the multi-threaded
        // CPU reader (whose pools start inflating the file as soon as it is opened) comes up only for RSQC_DECODE=host.
        BamReader bam;
        std::vector<std::string> bam_contigs;
        uint64_t first_voff = 0;
        bool bam_open = false;
        auto open_host_reader = [&]() -> bool {
            if (bam_open) return true;
            if (!bam.open(bam_path)) return false;
            bam.set_tags(o.chimeric_tag, o.tags);
            bam_open = true;
            return true;
        };
        if (device_decode_wanted()) {
            BgzfFeeder probe;
            if (!probe.open(bam_path)) { cerr << "Unable to open BAM file: " << bam_path << endl; return 10; }
            try { first_voff = probe.first_record_voffset(&bam_contigs); }
            catch (std::exception &) { cerr << "Unable to open BAM file: " << bam_path << endl; return 10; }
        } else {
            if (!open_host_reader()) { cerr << "Unable to open BAM file: " << bam_path << endl; return 10; }
            bam_contigs = bam.contigs();
        }
        // header check: at least one BAM contig must carry GTF features (src/RNASeQC.cpp:216-238)
        if (o.verbosity > 1) cout << "Checking bam header..." << endl;
        bool overlap = false;
        for (auto &n : bam_contigs) { auto it = ann.chrom_id.find(n); if (it != ann.chrom_id.end() && (size_t)it->second < gtf_chrom.size() && gtf_chrom[(size_t)it->second]) overlap = true; }
        if (!overlap) { cerr << "BAM file shares no contigs with GTF" << endl; return 11; }
        ann.flatten(bam_contigs);

        // ---- GPUs: one by default; --gpus N (or RSQC_GPUS) shards the file by contig, which needs the BAM index
        std::vector<int> devices;
        {
            int want = o.gpus > 0 ? o.gpus : (getenv("RSQC_GPUS") ? atoi(getenv("RSQC_GPUS")) : 1);
            if (const char *e = getenv("RSQC_GPU_LIST")) { for (const char *q = e; *q;) { devices.push_back(atoi(q)); while (*q && *q != ',') ++q; if (*q) ++q; } }
            else for (int k = 0; k < std::max(1, want); ++k) devices.push_back(P.device + k);
            if (devices.size() > 1 && !bam.load_index(bam_path + ".bai")) {
                cerr << "Warning: sharding over " << devices.size() << " GPUs needs the BAM index " << bam_path << ".bai; running on one GPU" << endl;
                devices.resize(1);
            }
        }
        const int n_ref_bam = (int)bam_contigs.size();
        std::vector<Shard> shards(devices.size());
        uint64_t tail_voff = 0;
        if (devices.size() > 1) {
            // longest-processing-time packing of the contigs on the index's record counts (compressed bytes when an index
            // carries no counts); the unplaced tail goes to the lightest shard
            const auto &idx = bam.index();
            std::vector<std::pair<uint64_t, int>> byload;
            for (int c2 = 0; c2 < n_ref_bam && (size_t)c2 < idx.size(); ++c2) if (idx[(size_t)c2].present) {
                const uint64_t load = idx[(size_t)c2].n_records ? idx[(size_t)c2].n_records : ((idx[(size_t)c2].end >> 16) - (idx[(size_t)c2].beg >> 16)) / 24 + 1;
                byload.emplace_back(load, c2);
                tail_voff = std::max(tail_voff, idx[(size_t)c2].end);
            }
            std::sort(byload.begin(), byload.end(), [](const std::pair<uint64_t, int> &x, const std::pair<uint64_t, int> &y) { return x.first != y.first ? x.first > y.first : x.second < y.second; });
            for (auto &bl : byload) {
                size_t best = 0;
                for (size_t g = 1; g < shards.size(); ++g) if (shards[g].load < shards[best].load) best = g;
                shards[best].contigs.push_back(bl.second); shards[best].load += bl.first;
            }
            size_t lightest = 0;
            for (size_t g = 1; g < shards.size(); ++g) if (shards[g].load < shards[lightest].load) lightest = g;
            shards[lightest].tail = true;
            for (auto &sh : shards) std::sort(sh.contigs.begin(), sh.contigs.end());
        }

        const auto t_gpu0 = std::chrono::steady_clock::now();
        int rc = gpu_ready.get();
        const auto t_gpu1 = std::chrono::steady_clock::now();
        if (rc != RSQC_OK) { cerr << "Unable to initialise the GPU hot path: " << rsqc_strerror(rc) << endl; return 10; }
        std::vector<std::vector<uint8_t>> owned_masks(shards.size());
        for (size_t g = 0; g < shards.size(); ++g) {
            Shard &sh = shards[g];
            sh.device = devices[g];
            if (g == 0) sh.gpu = gpu;
            else { rsqc_params Pg = P; Pg.device = sh.device; if ((rc = rsqc_create(&Pg, &sh.gpu)) != RSQC_OK) { cerr << "Unable to initialise GPU " << sh.device << ": " << rsqc_strerror(rc) << endl; return 10; } }
        }
        // the exchange group of a sharded run: the RCCL communicators come up HERE, on a thread beside the annotation upload and
        // the feeders' set-up -- not inside the `Average Reads/Sec` window that the end-of-file reduction belongs to
        rsqc_group *xgroup = nullptr;
        std::future<int> group_ready;
        if (shards.size() > 1) {
            std::vector<rsqc_ctx *> members;
            for (auto &sh : shards) members.push_back(sh.gpu);
            group_ready = std::async(std::launch::async, [members, &xgroup]() mutable { return rsqc_group_create(members.data(), (int)members.size(), &xgroup); });
        }
        for (size_t g = 0; g < shards.size(); ++g) {
            Shard &sh = shards[g];
            const uint8_t *owned = nullptr;
            if (shards.size() > 1) {
                owned_masks[g].assign(ann.contig_names.size(), 0);
                for (int c2 : sh.contigs) owned_masks[g][(size_t)c2] = 1;
                owned = owned_masks[g].data();
            }
            if ((rc = rsqc_set_annotation(sh.gpu, &ann.ann, owned)) != RSQC_OK) {
                // (the GTF parsed: this is the device index refusing the annotation -- e.g. no memory for the interval tables -- and
                //  its own message says which)
                cerr << "Unable to build the annotation index on the GPU: " << rsqc_last_error(sh.gpu) << endl; return 11;
            }
            // accepted with a warning (an exon outside its gene's row): the reference's counterpart is its per-read
            // "Gene encountered after computing coverage" (src/Metrics.cpp:108-112); said once, before the BAM loop
            if (g == 0 && rsqc_last_error(sh.gpu)[0]) cerr << "Warning: " << rsqc_last_error(sh.gpu) << endl;
            if (o.has_bed && (rc = rsqc_set_bed(sh.gpu, &ann.bed)) != RSQC_OK) { cerr << "Failed to parse the BED: " << rsqc_last_error(sh.gpu) << endl; return 11; }
        }
        std::vector<char> in_fasta;
        if (o.has_fasta) {
            fasta_loaded.get();                                               // FileError -> 10
            in_fasta.assign(ann.contig_names.size(), 0);
            std::vector<int32_t> r_contig; std::vector<uint64_t> r_len; std::vector<const uint8_t *> r_seq;
            for (size_t i = 0; i < fasta.index.size(); ++i) {
                int cid = -1;
                for (size_t k = 0; k < ann.contig_names.size(); ++k) if (ann.contig_names[k] == fasta.index[i].name) { cid = (int)k; break; }
                if (cid < 0) continue;                                        // a contig neither the BAM nor the GTF/BED names
                r_contig.push_back(cid); r_len.push_back(fasta_seq[i].size()); r_seq.push_back(fasta_seq[i].data());
                in_fasta[(size_t)cid] = 1;
            }
            rsqc_reference ref{(int32_t)r_contig.size(), r_contig.data(), r_len.data(), r_seq.data()};
            for (auto &sh : shards)
                if ((rc = rsqc_set_reference(sh.gpu, &ref)) != RSQC_OK) { cerr << "Failed to load the reference: " << rsqc_last_error(sh.gpu) << endl; return 10; }
            std::vector<std::vector<uint8_t>>().swap(fasta_seq);              // the bases live on the device now
        }

        // device decode needs exact range ends from the index when the file is sharded
        bool device_decode = device_decode_wanted();
        if (device_decode && shards.size() > 1)
            for (auto &r : bam.index()) if (r.present && !r.end) device_decode = false;
        if (!device_decode && shards.size() == 1 && !open_host_reader()) { cerr << "Unable to open BAM file: " << bam_path << endl; return 10; }
        if (device_decode && !(getenv("RSQC_DECODE_PRERESERVE") && !atoi(getenv("RSQC_DECODE_PRERESERVE")))) {
            // the device's window buffers are set up before the loop, like the host path's page-locked batches below
            struct stat st{};
            const uint64_t fsz = stat(bam_path.c_str(), &st) == 0 ? (uint64_t)st.st_size : 0;
            for (auto &sh : shards) {
                rsqc_decode_params dp = decode_params(o, n_ref_bam, 0);
                dp.reserve_inflated_bytes = decode_reserve_bytes(fsz);
                rsqc_decode_info none{};
                if ((rc = rsqc_decode_begin(sh.gpu, &dp)) != RSQC_OK || (rc = rsqc_decode_end(sh.gpu, &none)) != RSQC_OK) {
                    cerr << "Unable to set up the device decode: " << rsqc_last_error(sh.gpu) << endl; return 10;
                }
            }
        }
        // the feeders' chunk buffers are page-locked here, before the loop: page-locking takes the HIP runtime's lock, and done
        // by the read-ahead thread during the loop it stalled the thread that feeds the GPU (193 -> 237 M reads/s)
        std::vector<std::unique_ptr<BgzfFeeder>> feeders;
        if (device_decode) {
            // RSQC_DECODE_CPU_THREADS=n: n spare CPU threads inflate the tail of every chunk beside the GPU (BgzfFeeder::set_cpu_share)
            // (default: the CPUs the process may use minus four for the file reads and the thread that feeds the GPU; measured on the
            //  16-CPU box, 12 threads: 258 -> 279 M reads/s, 85 -> 97 M on the realistic-entropy file, profiles/r2_decode_cpu_share_ab.txt)
            const int cpu_share_threads = feeder_cpu_threads();
            const bool early_ok = early_feed && early_feed_ready.valid() && early_feed_ready.get();
            for (size_t g = 0; g < shards.size(); ++g) {
                if (g == 0 && early_ok && shards.size() == 1) { feeders.push_back(std::move(early_feed)); continue; }
                feeders.emplace_back(new BgzfFeeder());
                if (!feeders.back()->open(bam_path)) { cerr << "Unable to open BAM file: " << bam_path << endl; return 10; }
                if (cpu_share_threads > 0)             // (a shard's calls are a contig at a time: a fraction of the room)
                    feeders.back()->set_cpu_share(std::max(1, cpu_share_threads / (int)shards.size()), 0.15, 0.5, ((uint64_t)1 << 30) / shards.size());
                if (feeder_prepin) feeders.back()->reserve(shards.size() == 1 ? feeder_chunk : std::max<size_t>(feeder_chunk / shards.size(), (size_t)16 << 20));
            }
            // (threads that read one chunk's slices side by side; RSQC_FEED_READ_THREADS overrides)
            feeders[0]->read_threads = getenv("RSQC_FEED_READ_THREADS") ? std::max(1, atoi(getenv("RSQC_FEED_READ_THREADS"))) : std::max(1, std::min(8, effective_cpus() / 2));
        }
        if (o.verbosity) cout << "Parsing bam..." << endl;
        const size_t BATCH = getenv("RSQC_BATCH") ? (size_t)atol(getenv("RSQC_BATCH")) : (size_t)1 << 21;
        HostBatch bufs[2];
        for (auto &hb : bufs) {                                  // page-locked staging, sized once
            hb.core.use_pinned(true); hb.aux.use_pinned(true); hb.qh2.use_pinned(true); hb.cigar.use_pinned(true);
            hb.core.reserve(BATCH); hb.aux.reserve(BATCH); hb.qh2.reserve(BATCH); hb.cigar.reserve(BATCH * 2);
        }
        std::vector<int> visit;
        unsigned long long alignmentCount = 0;
        int cur = 0; bool in_flight = false, warned_unsorted = false;
        ShardMerge merged;
        double group_init_ms = 0.0;
        if (group_ready.valid()) {                               // (before the window opens)
            if ((rc = group_ready.get()) != RSQC_OK) { cerr << "Unable to set up the multi-GPU exchange: " << rsqc_strerror(rc) << endl; return 10; }
            int uses = 0; const char *note = "";
            rsqc_group_info(xgroup, &uses, &group_init_ms, nullptr, &note);
            if (o.verbosity > 1) cout << "Exchange group of " << shards.size() << " GPUs ready in " << group_init_ms << " ms ("
                                      << (uses ? "RCCL communicators" : (std::string("peer copies: ") + note).c_str()) << "), before the BAM loop" << endl;
        }
        const auto tb0 = std::chrono::steady_clock::now();
        if (shards.size() > 1) {
            // one reader thread per GPU; the decode threads of the process are shared out between them
            int budget = 2 * effective_cpus();
            if (const char *e = getenv("RSQC_HOST_THREADS")) budget = atoi(e);
            const int per = std::max(2, budget / (int)shards.size());
            std::vector<std::thread> th;
            for (auto &sh : shards) th.emplace_back(shard_worker, std::ref(sh), std::cref(bam_path), std::cref(o), per, std::cref(bam.index()), n_ref_bam, tail_voff, BATCH, device_decode ? feeders[(size_t)(&sh - shards.data())].get() : nullptr);
            for (auto &t : th) t.join();
            rc = RSQC_OK;
            for (auto &sh : shards) {
                alignmentCount += sh.n_records;
                if (sh.rc != RSQC_OK && rc == RSQC_OK) { rc = sh.rc; if (rc != RSQC_ERR_BAD_CIGAR && rc != RSQC_ERR_EMPTY_MEDIAN) cerr << "GPU " << sh.device << ": " << sh.error << endl; }
                if (o.verbosity) for (auto &nm : sh.bad_refid) cerr << "Unrecognized RefID on alignment: " << nm << endl;
                if (sh.unsorted && !warned_unsorted) { cerr << "Warning: The input bam does not appear to be sorted. An unsorted bam will yield incorrect results" << endl; warned_unsorted = true; }
            }
            for (int c2 = 0; c2 < n_ref_bam && (size_t)c2 < bam.index().size(); ++c2) if (bam.index()[(size_t)c2].present) {
                visit.push_back(c2);
                if (o.has_fasta && (size_t)c2 < in_fasta.size() && !in_fasta[(size_t)c2])
                    cerr << "Warning: Provided Fasta does not contain chromosome " << ann.contig_names[(size_t)c2]
                         << ". No GC statistics will be collected for this chromosome" << endl;
            }
            // the exchange step: result ranges summed onto the first GPU, order-dependent outputs composed from the summaries
            int used_rccl = 0;
            if (rc == RSQC_OK) {
                rc = rsqc_group_reduce(xgroup, &used_rccl);
                if (rc != RSQC_OK) cerr << rsqc_last_error(shards[0].gpu) << endl;
            }
            if (rc == RSQC_OK) { std::string merr; rc = merge_shards(shards, P.fragment_samples, merged, merr); if (rc != RSQC_OK) cerr << merr << endl; }
            if (o.verbosity > 1) {
                cout << "Alignments processed: " << alignmentCount << " on " << shards.size() << " GPUs (";
                for (size_t g = 0; g < shards.size(); ++g) cout << (g ? ", " : "") << shards[g].n_records;
                double reduce_ms = 0.0; rsqc_group_info(xgroup, nullptr, nullptr, &reduce_ms, nullptr);
                cout << " records); shards summed by " << (used_rccl ? "RCCL ncclReduce" : "peer copies") << " in " << reduce_ms << " ms (group set-up " << group_init_ms << " ms, outside the window)" << endl;
            }
        } else if (device_decode) {
            // ---- one GPU, device decode: the host reads the file and frames the BGZF blocks, nothing else
            rsqc_decode_info di{};
            BgzfFeeder &feed = *feeders[0];
            rc = decode_range(gpu, feed, decode_params(o, n_ref_bam, 0), first_voff, 0, di, [&](const rsqc_decode_window &w) {
                bool revisit = false;
                for (uint32_t k = 0; k < w.n_runs; ++k) {
                    const int32_t t = w.run_tid[k];
                    if (t < 0 || (!visit.empty() && visit.back() == t)) continue;
                    if (std::find(visit.begin(), visit.end(), t) != visit.end()) revisit = true;     // a contig that comes back
                    visit.push_back(t);
                    if (o.has_fasta && (size_t)t < in_fasta.size() && !in_fasta[(size_t)t])      // src/RNASeQC.cpp:350-352
                        cerr << "Warning: Provided Fasta does not contain chromosome " << ann.contig_names[(size_t)t]
                             << ". No GC statistics will be collected for this chromosome" << endl;
                }
                if (revisit && !warned_unsorted) {
                    cerr << "Warning: The input bam does not appear to be sorted. An unsorted bam will yield incorrect results" << endl;
                    warned_unsorted = true;
                }
                alignmentCount += w.n_records;
                if (o.verbosity > 1) cout << "Alignments processed: " << alignmentCount << endl;
            });
            if (rc == RSQC_ERR_INPUT) throw std::runtime_error(rsqc_last_error(gpu));
            if (rc == RSQC_OK) {
                if (o.verbosity) for (int k = 0; k < di.n_bad_refid && k < 64; ++k) cerr << "Unrecognized RefID on alignment: " << di.bad_refid[k] << endl;
                if (di.unsorted && !warned_unsorted) {
                    cerr << "Warning: The input bam does not appear to be sorted. An unsorted bam will yield incorrect results" << endl;
                    warned_unsorted = true;
                }
            }
        } else
        for (;;) {
            HostBatch &hb = bufs[cur];
            hb.clear();
            hb.file_index_base = alignmentCount;
            const size_t n = bam.read_batch(hb, BATCH);              // decode overlaps the previous batch on the GPU
            if (in_flight) { if ((rc = rsqc_wait(gpu)) != RSQC_OK) break; in_flight = false; }
            if (n == 0) break;
            // stderr of the reference's loop: the names of records with a RefID the header lacks (src/RNASeQC.cpp:333-337, under -v)
            // and the sort warning (:354-355).  The reference repeats the warning for every offending record; it is given
            // once here -- an unsorted file voids the results either way (the static index does not reproduce what the
            // reference's destructively trimmed window would count).
            if (o.verbosity) for (auto &nm : hb.bad_refid) cerr << "Unrecognized RefID on alignment: " << nm << endl;
            bool revisit = false;
            for (int32_t t : hb.seg_tid) if (t >= 0 && (visit.empty() || visit.back() != t)) {
                if (std::find(visit.begin(), visit.end(), t) != visit.end()) revisit = true;     // a contig that comes back
                visit.push_back(t);
                if (o.has_fasta && (size_t)t < in_fasta.size() && !in_fasta[(size_t)t])      // src/RNASeQC.cpp:350-352
                    cerr << "Warning: Provided Fasta does not contain chromosome " << ann.contig_names[(size_t)t]
                         << ". No GC statistics will be collected for this chromosome" << endl;
            }
            if ((hb.unsorted || revisit) && !warned_unsorted) {
                cerr << "Warning: The input bam does not appear to be sorted. An unsorted bam will yield incorrect results" << endl;
                warned_unsorted = true;
            }
            alignmentCount += n;
            rsqc_batch view = hb.view();
            if ((rc = rsqc_submit(gpu, &view)) != RSQC_OK) break;
            in_flight = true;
            cur ^= 1;
            if (o.verbosity > 1) cout << "Alignments processed: " << alignmentCount << endl;
        }
        rsqc_results res{};
        if (rc == RSQC_OK && shards.size() > 1) {
            rc = rsqc_refresh_results(gpu, &res);
            res.read_length = merged.read_length;
            res.n_fragment_sizes = (uint32_t)merged.fsize.size(); res.fragment_size = merged.fsize.data(); res.fragment_count = merged.fcount.data();
            res.fragment_samples_remaining = merged.remaining;
        } else
        if (rc == RSQC_OK) rc = rsqc_finalize(gpu, &res);
        const auto tb1 = std::chrono::steady_clock::now();
        if (rc == RSQC_ERR_BAD_CIGAR) throw std::invalid_argument("Unrecognized Cigar Op ");
        if (rc == RSQC_ERR_EMPTY_MEDIAN) throw std::range_error("Cannot compute median of an empty list");
        if (rc != RSQC_OK) { cerr << rsqc_strerror(rc) << ": " << rsqc_last_error(gpu) << endl; rsqc_destroy(gpu); return 10; }
        if (o.verbosity) {
            const double secs = std::chrono::duration<double>(tb1 - tb0).count();
            cout << "Time Elapsed: " << secs << "; Alignments processed: " << alignmentCount << endl;
            if (o.verbosity > 1) cout << "Average Reads/Sec: " << (double)alignmentCount / secs << endl;
            if (o.verbosity > 1 && device_decode) cout << "(decode: BGZF inflate and record parsing on the GPU)" << endl;
            else if (o.verbosity > 1 && shards.size() == 1) cout << "(decode threads: " << bam.inflate_threads() << " inflate + " << bam.parse_threads() << " parse)" << endl;
            cout << "Estimating library complexity..." << endl;
            cout << "Generating report" << endl;
        }
        ReportConfig cfg;
        cfg.output_dir = out_dir; cfg.sample_name = SAMPLENAME; cfg.sample_given = o.has_sample;
        cfg.use_rpkm = o.rpkm; cfg.write_coverage = o.coverage; cfg.detection_threshold = (unsigned)o.detection;
        cfg.filter_tags = o.tags;
        const auto tr0 = std::chrono::steady_clock::now();
        write_reports(cfg, ann, res, visit);
        const auto tr1 = std::chrono::steady_clock::now();
        rsqc_group_destroy(xgroup);
        for (auto &sh : shards) rsqc_destroy(sh.gpu);
        if (o.verbosity > 1) {
            // where the wall time outside the reference's `Average Reads/Sec` window goes (extension; the window itself is above)
            auto sec = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return std::chrono::duration<double>(b - a).count(); };
            cout << "Wall time: " << sec(t_start, std::chrono::steady_clock::now()) << " s = GTF " << sec(t0, t1)
                 << " + waiting for the GPU context " << sec(t_gpu0, t_gpu1) << " + index / annotation upload / buffers " << sec(t_gpu1, tb0) - 0.0
                 << " + BAM loop " << sec(tb0, tb1) << " + reports " << sec(tr0, tr1) << " + release " << sec(tr1, std::chrono::steady_clock::now())
                 << " (the GPU context and the page-locked feed buffers come up beside the GTF parse)" << endl;
        }
    } catch (Help &) {
        usage(cout);
        return 4;
    } catch (ParseError &e) {
        usage(cerr); cerr << endl << "Argument parsing error: " << e.what() << endl;
        return 5;
    } catch (ValidationError &e) {
        usage(cerr); cerr << endl << "Argument validation error: " << e.what() << endl;
        return 6;
    } catch (std::invalid_argument &e) {
        cerr << "Invalid argument type provided: " << e.what() << endl;
        return 7;
    } catch (FileError &e) {
        cerr << e.what() << endl;
        return 10;
    } catch (GtfError &e) {
        cerr << "Failed to parse the GTF: " << e.what() << endl;
        return 11;
    } catch (BedError &e) {
        cerr << "Failed to parse the BED: " << e.what() << endl;
        return 11;
    } catch (std::length_error &e) {
        cerr << "Unable to parse the GFT lines" << endl << e.what() << endl;
        return 1;
    } catch (std::range_error &e) {
        cerr << "Invalid range" << endl << e.what() << endl;
        return 2;
    } catch (std::domain_error &e) {
        cerr << "Unable to perform string conversion" << endl << e.what() << endl;
        return 3;
    } catch (std::bad_alloc &e) {
        cerr << "Memory allocation failure. Out of memory" << endl << e.what() << endl;
        return 10;
    } catch (std::exception &e) {
        cerr << "Encountered an IO failure" << endl << e.what() << endl;
        return 10;
    } catch (...) {
        cerr << "Unknown error" << endl;
        return -1;
    }
    return 0;
}
