// feeder_bench.cpp -- how fast can the HOST side of the device decode deliver a BAM file?  ("Host feed for 8 GPUs")
//
// `rnaseqc --gpus N` is one process with one BgzfFeeder per GPU (host/bgzf_feed.cpp: pread into page-locked chunks on a
// read-ahead thread + one hop per BGZF block header); everything after that runs on the GPUs.  This tool runs N feeders side
// by side WITHOUT any GPU work -- every feeder streams the whole file, chunks are taken and dropped -- and reports the file
// bytes per second they sustain together, i.e. the ceiling the host puts on a sharded run.  No GPU is needed (without a
// device the chunk buffers are ordinary memory; with one they are page-locked as in the CLI).
//
//   build:  g++ -O2 -std=c++17 tools/feeder_bench.cpp rnaseqc_amd/csrc/host/bgzf_feed.cpp -Lrnaseqc_amd/lib -lrnaseqc_amd -lrsqc_host ...
//           (tools/feeder_bench.sh does it)
//   run:    feeder_bench <file.bam> <feeders> [read threads per feeder] [chunk MiB]
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <memory>
#include <string>
#include <thread>
#include <vector>

#include "../rnaseqc_amd/csrc/host/bgzf_feed.hpp"

using rsqc_host::BgzfFeeder;

int main(int argc, char **argv) {
    if (argc < 3) { fprintf(stderr, "usage: %s file.bam feeders [read_threads] [chunk_MiB]\n", argv[0]); return 2; }
    const std::string path = argv[1];
    const int n = atoi(argv[2]);
    const int rt = argc > 3 ? atoi(argv[3]) : 0;
    const size_t chunk = (size_t)(argc > 4 ? atoi(argv[4]) : 128) << 20;
    std::vector<std::unique_ptr<BgzfFeeder>> feeders;
    uint64_t first = 0;
    for (int k = 0; k < n; ++k) {
        feeders.emplace_back(new BgzfFeeder());
        if (!feeders.back()->open(path)) { fprintf(stderr, "cannot open %s\n", path.c_str()); return 1; }
        if (k == 0) first = feeders[0]->first_record_voffset();
        if (rt > 0) feeders.back()->read_threads = rt;
        feeders.back()->reserve(chunk);                       // page-locked before the clock starts, as in the CLI
    }
    std::vector<uint64_t> bytes((size_t)n, 0), blocks((size_t)n, 0), inflated((size_t)n, 0);
    const auto t0 = std::chrono::steady_clock::now();
    std::vector<std::thread> th;
    for (int k = 0; k < n; ++k)
        th.emplace_back([&, k] {
            BgzfFeeder &f = *feeders[(size_t)k];
            f.start(first, 0, chunk);
            while (BgzfFeeder::Chunk *c = f.next()) {
                bytes[(size_t)k] += c->bytes; blocks[(size_t)k] += c->blocks.size();
                for (auto &b : c->blocks) inflated[(size_t)k] += b.out_bytes;
            }
        });
    for (auto &t : th) t.join();
    const double s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    uint64_t tb = 0, ti = 0, tk = 0;
    for (int k = 0; k < n; ++k) { tb += bytes[(size_t)k]; ti += inflated[(size_t)k]; tk += blocks[(size_t)k]; }
    printf("feeders %d  read_threads %d  chunk %zu MiB  file %.2f GB: %.3f s  %.2f GB/s of file bytes (%.2f GB/s per feeder), %.1f GB/s inflated-equivalent, %.2f M blocks/s\n",
           n, feeders[0]->read_threads, chunk >> 20, feeders[0]->file_size() / 1e9, s, tb / s / 1e9, tb / s / 1e9 / n, ti / s / 1e9, tk / s / 1e6);
    return 0;
}
