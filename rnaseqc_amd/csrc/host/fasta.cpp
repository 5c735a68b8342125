#include "fasta.hpp"

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <cstring>
#include <fstream>
#include <thread>

namespace rsqc_host {

static bool file_exists(const std::string &p) { struct stat st; return stat(p.c_str(), &st) == 0; }

void FastaFile::open(const std::string &fasta_path) {
    path = fasta_path;
    { std::ifstream probe(fasta_path); if (!probe.is_open()) throw FileError("Unable to open reference fasta: " + fasta_path); }
    // boost::filesystem::path(filename).replace_extension(".fai") first, then filename + ".fai"
    std::string stem = fasta_path;
    const size_t slash = stem.find_last_of('/'), dot = stem.find_last_of('.');
    if (dot != std::string::npos && (slash == std::string::npos || dot > slash) && dot != slash + 1) stem.erase(dot);
    std::string index_path = fasta_path + ".fai";
    if (file_exists(stem + ".fai")) index_path = stem + ".fai";
    else if (!file_exists(index_path)) throw FileError("Unable to locate fasta index: " + fasta_path);
    std::ifstream in(index_path, std::ios::binary);
    std::string line;
    while (std::getline(in, line)) {                // name \t length \t offset \t line bases \t line bytes (bioio FastaContigIndex)
        Entry e{};
        size_t p = 0; std::string f[5];
        for (int k = 0; k < 5; ++k) {
            const size_t t = line.find('\t', p);
            f[k] = line.substr(p, t == std::string::npos ? std::string::npos : t - p);
            if (t == std::string::npos) { p = line.size(); } else p = t + 1;
        }
        if (f[0].empty() && line.empty()) continue;
        try {
            e.name = f[0]; e.length = std::stoull(f[1]); e.offset = std::stoull(f[2]); e.line_bases = std::stoull(f[3]); e.line_bytes = std::stoull(f[4]);
        } catch (std::exception &) { throw FileError("Malformed fasta index line: " + line); }
        index.push_back(e);
    }
    if (index.empty()) throw FileError("No contigs found in fasta index: " + index_path);
}

void FastaFile::load(std::vector<std::vector<uint8_t>> &sequences, int threads) const {
    const int fd = ::open(path.c_str(), O_RDONLY);
    if (fd < 0) throw FileError("Unable to open reference fasta: " + path);
    struct stat st; fstat(fd, &st);
    const size_t size = (size_t)st.st_size;
    const uint8_t *map = size ? (const uint8_t *)mmap(nullptr, size, PROT_READ, MAP_PRIVATE, fd, 0) : nullptr;
    if (size && map == MAP_FAILED) { ::close(fd); throw FileError("Unable to map reference fasta: " + path); }
    sequences.assign(index.size(), {});
    struct Task { size_t contig; uint64_t line0, line1; };
    std::vector<Task> tasks;
    const uint64_t kLines = 1 << 16;
    for (size_t i = 0; i < index.size(); ++i) {
        sequences[i].resize(index[i].length);
        if (!index[i].length || !index[i].line_bases) continue;
        const uint64_t n_lines = (index[i].length + index[i].line_bases - 1) / index[i].line_bases;
        for (uint64_t l = 0; l < n_lines; l += kLines) tasks.push_back(Task{i, l, std::min(n_lines, l + kLines)});
    }
    std::atomic<size_t> next{0};
    std::atomic<bool> short_file{false};
    auto work = [&]() {
        for (size_t t = next++; t < tasks.size(); t = next++) {
            const Entry &e = index[tasks[t].contig];
            uint8_t *dst = sequences[tasks[t].contig].data();
            for (uint64_t l = tasks[t].line0; l < tasks[t].line1; ++l) {
                const uint64_t b0 = l * e.line_bases, nb = std::min(e.line_bases, e.length - b0);
                const uint64_t off = e.offset + l * e.line_bytes;
                if (off + nb > size) { short_file = true; return; }
                memcpy(dst + b0, map + off, nb);
            }
        }
    };
    std::vector<std::thread> pool;
    const int n_thr = std::max(1, std::min<int>(threads, (int)tasks.size()));
    for (int k = 1; k < n_thr; ++k) pool.emplace_back(work);
    work();
    for (auto &th : pool) th.join();
    if (map) munmap((void *)map, size);
    ::close(fd);
    if (short_file) throw FileError("Reference fasta is shorter than its index says: " + path);
}

}  // namespace rsqc_host
