// decode_fuzz.cpp -- TEST HARNESS ONLY (never loaded by the product).
//
// The device decode's framing and parsing (rnaseqc_amd/csrc/rsqc_bamrec.h, rsqc_decode.h: what the kernels of
// rsqc_decode.hip run per segment and per record) on windows of DAMAGED inflated data, built with
// -fsanitize=address,undefined.  A BGZF block's CRC-32 only says that the bytes are the ones the writer compressed; a
// writer's bug, or a file that is not a BAM behind its header, reaches these functions as it is.  Whatever the bytes:
// every read stays inside [0, end) of the window, every write inside buffers sized the way rsqc_api.cpp's decode_reserve
// sizes them, every loop ends, and the parallel chain repair leaves what the plain sequential walk leaves (checked inside
// emu_decode_window).  On the GPU the first three are a dead device, not a wrong answer.
//
//   decode_fuzz <cases> <seed>      exit 0 = nothing found; the sanitizers abort the process on a finding
#include <signal.h>
#include <unistd.h>

#include <cstdio>

#include "decode_emu.cpp"

#include "fuzz_records.h"

static const char *g_what = "";
static long g_case = -1;
static void on_alarm(int) {
    char msg[160];
    const int k = snprintf(msg, sizeof msg, "decode_fuzz: case %ld (%s) did not end\n", g_case, g_what);
    if (write(2, msg, (size_t)k) < 0) {}
    _exit(3);
}

int main(int argc, char **argv) {
    const long cases = argc > 1 ? atol(argv[1]) : 500;
    g_state = argc > 2 ? strtoull(argv[2], nullptr, 10) * 2 + 1 : 1;
    signal(SIGALRM, on_alarm);
    long clean = 0, refused = 0, decoded = 0;
    std::vector<uint8_t> base;
    int32_t n_ref = 3;
    for (long c = 0; c < cases; ++c) {
        g_case = c;
        if (c % 6 == 0) {
            n_ref = 1 + (int32_t)rnd(rnd(3) ? 4 : 3000);
            base.clear();
            const uint32_t want = 2000 + rnd(rnd(5) ? 120000 : 900000);
            while (base.size() < want) put_record(base, n_ref, rnd(400) == 0);
        }
        std::vector<uint8_t> w = base;
        auto at = [&]() { return rnd((uint32_t)w.size()); };
        switch (rnd(12)) {
        case 0: g_what = "clean"; break;
        case 1: g_what = "bit flip"; w[at()] ^= (uint8_t)(1u << rnd(8)); break;
        case 2: g_what = "bit flips"; for (uint32_t k = 0, n = 2 + rnd(40); k < n; ++k) w[at()] ^= (uint8_t)(1u << rnd(8)); break;
        case 3: g_what = "bytes overwritten"; for (uint32_t k = 0, p = at(), n = 1 + rnd(64); k < n && p + k < w.size(); ++k) w[p + k] = (uint8_t)rnd(256); break;
        case 4: g_what = "cut short"; w.resize(rnd((uint32_t)w.size())); break;
        case 5: g_what = "garbage"; for (auto &b : w) b = (uint8_t)rnd(256); break;
        case 6: g_what = "zeros"; for (uint32_t k = 0, p = at(), n = 1 + rnd(20000); k < n && p + k < w.size(); ++k) w[p + k] = 0; break;
        case 7: g_what = "0xff"; for (uint32_t k = 0, p = at(), n = 1 + rnd(20000); k < n && p + k < w.size(); ++k) w[p + k] = 0xff; break;
        case 8: {   g_what = "a block_size field damaged";                               // walk to a record and set its length to something else
            uint64_t p = 0; for (uint32_t hops = rnd(200); hops && p + 4 <= w.size(); --hops) { uint32_t bs; memcpy(&bs, &w[p], 4); if (p + 4 + bs + 4 > w.size()) break; p += 4 + (uint64_t)bs; }
            if (p + 4 <= w.size()) { const uint32_t v = rnd(4) == 0 ? rnd(40) : rnd(3) == 0 ? 0xFFFFFFF0u + rnd(16) : rnd(1u << (1 + rnd(27))); memcpy(&w[p], &v, 4); }
            break; }
        case 9: {   g_what = "l_name / n_cigar / l_seq damaged";
            uint64_t p = 0; for (uint32_t hops = rnd(200); hops && p + 4 <= w.size(); --hops) { uint32_t bs; memcpy(&bs, &w[p], 4); if (p + 4 + bs + 4 > w.size()) break; p += 4 + (uint64_t)bs; }
            if (p + 36 <= w.size()) { const uint32_t f = rnd(3); if (f == 0) w[p + 12] = (uint8_t)rnd(256); else if (f == 1) { w[p + 16] = (uint8_t)rnd(256); w[p + 17] = (uint8_t)rnd(256); } else { const uint32_t v = rnd(2) ? rnd() : 0x80000000u + rnd(100); memcpy(&w[p + 20], &v, 4); } }
            break; }
        case 10: g_what = "an aux type damaged"; for (uint32_t k = 0; k < 50; ++k) { const uint32_t p = at(); if (w[p] == 'N' || w[p] == 'C' || w[p] == 'c') { if (p + 2 < w.size()) w[p + 2] = (uint8_t)"ZBHd?"[rnd(5)]; } } break;
        default: g_what = "a slice of another place"; { const uint32_t n = 1 + rnd(3000), from = at(), to = at(); for (uint32_t k = 0; k < n && from + k < w.size() && to + k < w.size(); ++k) w[to + k] = w[from + k]; } break;
        }
        // exact-size heap buffers: the window's bytes; columns sized like decode_reserve sizes them for a window of this many bytes
        const uint32_t start = rnd(3) ? 0u : std::min<uint32_t>((uint32_t)w.size(), rnd(5000)), end = (uint32_t)w.size();
        const size_t bytes = end - start, n_rec = bytes / 36 + 4;
        uint8_t *buf = (uint8_t *)malloc(end ? end : 1);
        if (end) memcpy(buf, w.data(), end);
        rsqc_rec_core *core = (rsqc_rec_core *)malloc(n_rec * sizeof(rsqc_rec_core));
        rsqc_rec_aux *aux = (rsqc_rec_aux *)malloc(n_rec * sizeof(rsqc_rec_aux));
        uint32_t *qh2 = (uint32_t *)malloc(n_rec * 4);
        uint32_t *cigar = (uint32_t *)malloc(bytes + 256);
        int32_t *seg_tid = (int32_t *)malloc(n_rec * 4); uint64_t *seg_start = (uint64_t *)malloc((n_rec + 1) * 8);
        uint64_t *wide_index = (uint64_t *)malloc(n_rec * 8); int32_t *wide_nm = (int32_t *)malloc(n_rec * 4), *wide_lq = (int32_t *)malloc(n_rec * 4);
        uint32_t *wide_nc = (uint32_t *)malloc(n_rec * 4);
        uint32_t summary[8 + 64] = {0};
        int32_t carry[3] = {0, 0, 0};
        BamTagSpec tags{};
        tags.n_ref = n_ref; tags.have_ch = 1; tags.ch0 = 'c'; tags.ch1 = 'h'; tags.n_filter = 1; tags.f0[0] = 'X'; tags.f1[0] = 'F';
        alarm(60);
        const int rc = emu_decode_window(buf, start, end, &tags, 1 + (int)rnd(64), carry, 0, core, aux, qh2, cigar, seg_tid, seg_start, wide_index, wide_nm, wide_lq, wide_nc, summary);
        alarm(0);
        if (rc != 0) { fprintf(stderr, "decode_fuzz: case %ld (%s): the listed repair and the sequential walk disagree (%d)\n", c, g_what, rc); return 1; }
        const uint32_t n = summary[0], n_ops = summary[1], consumed = summary[6], status = summary[7];
        if (status) ++refused;
        else {
            ++decoded;
            if (n > bytes / 36 || (uint64_t)n_ops * 4 > bytes || consumed > end || consumed < start) { fprintf(stderr, "decode_fuzz: case %ld (%s): counts beyond what the window can hold\n", c, g_what); return 1; }
            if (g_what[0] == 'c' && g_what[1] == 'l') { if (start == 0 && consumed != end) { fprintf(stderr, "decode_fuzz: case %ld: clean window not consumed\n", c); return 1; } ++clean; }
        }
        free(buf); free(core); free(aux); free(qh2); free(cigar); free(seg_tid); free(seg_start); free(wide_index); free(wide_nm); free(wide_lq); free(wide_nc);
    }
    printf("decode_fuzz: %ld cases: %ld windows decoded (%ld of them undamaged), %ld refused as bad records\n", cases, decoded, clean, refused);
    return 0;
}
