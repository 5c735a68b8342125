// gtf_fuzz.cpp -- TEST HARNESS ONLY (never loaded by the product).
//
// The GTF / BED ingest (rnaseqc_amd/csrc/host/gtf.cpp; the reference's src/GTF.cpp:30-148, src/BED.cpp) on damaged text, built
// with -fsanitize=address,undefined: a malformed annotation must end in the errors the reference has for it (GtfError /
// BedError / FileError -> exit codes 11 / 10) or load, never in an access outside a buffer, an endless loop, or an exception of
// another kind escaping.
//
//   gtf_fuzz <cases> <seed> <tmp dir>      exit 0 = nothing found
#include <signal.h>
#include <unistd.h>

#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <string>
#include <vector>

#include "../../rnaseqc_amd/csrc/host/gtf.hpp"

static uint64_t g_state = 1;
static uint32_t rnd() { g_state = g_state * 6364136223846793005ull + 1442695040888963407ull; return (uint32_t)(g_state >> 33); }
static uint32_t rnd(uint32_t n) { return n ? rnd() % n : 0; }

static std::string gtf_line(int gene, int exon, long start, long end, const char *type) {
    char b[512];
    const char *tt = gene % 7 == 0 ? "rRNA" : "protein_coding";
    snprintf(b, sizeof b, "chr%d\tsrc\t%s\t%ld\t%ld\t.\t%c\t.\tgene_id \"G%d\"; transcript_id \"G%d\"; gene_type \"%s\"; gene_name \"N%d\"; transcript_type \"%s\";%s\n",
             gene % 3 + 1, type, start, end, gene % 2 ? '+' : '-', gene, gene, tt, gene, tt,
             exon >= 0 ? (" exon_id \"G" + std::to_string(gene) + "_" + std::to_string(exon) + "\";").c_str() : "");
    return b;
}

static void on_alarm(int) { const char m[] = "gtf_fuzz: a case did not end\n"; if (write(2, m, sizeof m - 1) < 0) {} _exit(3); }

int main(int argc, char **argv) {
    const long cases = argc > 1 ? atol(argv[1]) : 300;
    g_state = argc > 2 ? strtoull(argv[2], nullptr, 10) * 2 + 1 : 1;
    const std::string dir = argc > 3 ? argv[3] : "/tmp";
    const std::string gtf = dir + "/fuzz.gtf", bed = dir + "/fuzz.bed";
    signal(SIGALRM, on_alarm);
    long loaded = 0, refused = 0;
    for (long c = 0; c < cases; ++c) {
        std::vector<std::string> lines;
        lines.push_back("##description: fuzz\n");
        const int n_genes = 1 + (int)rnd(40);
        long pos = 100;
        for (int g = 0; g < n_genes; ++g) {
            const int ne = 1 + (int)rnd(5); const long gs = pos; std::vector<std::pair<long, long>> ex;
            for (int e = 0; e < ne; ++e) { const long len = 20 + rnd(400); ex.push_back({pos, pos + len}); pos += len + 30 + rnd(2000); }
            lines.push_back(gtf_line(g, -1, gs, ex.back().second, "gene"));
            lines.push_back(gtf_line(g, -1, gs, ex.back().second, "transcript"));
            for (int e = 0; e < ne; ++e) lines.push_back(gtf_line(g, e, ex[e].first, ex[e].second, "exon"));
            if (rnd(3) == 0) pos -= rnd(500);                                     // overlapping genes
        }
        const bool damaged = rnd(6) != 0;
        if (damaged) for (uint32_t k = 0, n = 1 + rnd(4); k < n; ++k) {
            std::string &l = lines[rnd((uint32_t)lines.size())];
            switch (rnd(12)) {
            case 0: if (!l.empty()) l.resize(rnd((uint32_t)l.size())); break;                                       // cut (no newline)
            case 1: if (!l.empty()) l[rnd((uint32_t)l.size())] = (char)rnd(256); break;
            case 2: { size_t t = l.find('\t'); if (t != std::string::npos) l.erase(t, 1); break; }                  // a field less
            case 3: { size_t t = l.find('\t', 10); if (t != std::string::npos) l.insert(t, "\t"); break; }          // an empty field
            case 4: { size_t t = l.find("exon\t"); if (t != std::string::npos) l.replace(t + 5, 3, "x1y"); break; } // a coordinate that is no number
            case 5: { size_t t = l.find("gene_id"); if (t != std::string::npos) l.erase(t, 7); break; }
            case 6: { size_t t = l.find('"'); if (t != std::string::npos) l.erase(t, 1); break; }                   // an unbalanced quote
            case 7: l = "\n"; break;
            case 8: l = std::string(1 + rnd(5000), (char)('A' + rnd(26))) + "\n"; break;
            case 9: { size_t t = l.find("exon\t"); if (t != std::string::npos) l.replace(t + 5, 1, "99999999999999999999999"); break; }   // beyond 64 bits
            case 10: { size_t t = l.find("\t+\t"); if (t != std::string::npos) l[t + 1] = '?'; break; }
            default: { size_t t = l.rfind('\n'); if (t != std::string::npos) l.replace(t, 1, "\r\n"); break; }
            }
        }
        { std::ofstream o(gtf, std::ios::binary); for (auto &l : lines) o << l; }
        std::vector<std::string> bl;
        for (int k = 0, n = (int)rnd(30); k < n; ++k) { const long s = rnd(100000); bl.push_back("chr" + std::to_string(1 + rnd(3)) + "\t" + std::to_string(s) + "\t" + std::to_string(s + 1 + rnd(3000)) + "\n"); }
        if (damaged && !bl.empty() && rnd(2)) {
            std::string &l = bl[rnd((uint32_t)bl.size())];
            switch (rnd(5)) { case 0: l = "chr1\n"; break; case 1: l = "chr1\tabc\tdef\n"; break; case 2: l.resize(rnd((uint32_t)l.size())); break; case 3: l = "#c\n"; break; default: l = "chr2\t-5\t99999999999999999999999999\n"; break; }
        }
        { std::ofstream o(bed, std::ios::binary); for (auto &l : bl) o << l; }
        alarm(60);
        bool ok = false;
        try {
            rsqc_host::Annotation a;
            a.legacy = rnd(4) == 0;
            a.load_gtf(gtf);
            if (rnd(2)) a.load_bed(bed);
            a.flatten({"chr1", "chr2", "chrX"});
            ok = true;
        } catch (rsqc_host::FileError &) {} catch (rsqc_host::GtfError &) {} catch (rsqc_host::BedError &) {}
        catch (const std::exception &e) { fprintf(stderr, "gtf_fuzz: case %ld: an exception the CLI has no exit code for: %s\n", c, e.what()); return 1; }
        alarm(0);
        if (!damaged && !ok) { fprintf(stderr, "gtf_fuzz: case %ld: an undamaged annotation was refused\n", c); return 1; }
        ok ? ++loaded : ++refused;
    }
    unlink(gtf.c_str()); unlink(bed.c_str());
    printf("gtf_fuzz: %ld cases: %ld annotations loaded, %ld refused with the reference's errors\n", cases, loaded, refused);
    return 0;
}
