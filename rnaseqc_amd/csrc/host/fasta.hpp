// fasta.hpp -- the reference sequence of --fasta runs, read through its .fai index into per-contig base strings
// (what rsqc_set_reference takes).  Replaces Fasta::open / getSeq / the bioio page reads (src/Fasta.cpp:77-140,
// bioio.hpp:236-331): the whole sequence goes to the device once, there is no page cache.
#pragma once

#include <cstdint>
#include <string>
#include <vector>

#include "gtf.hpp"

namespace rsqc_host {

struct FastaFile {
    struct Entry { std::string name; uint64_t length, offset, line_bases, line_bytes; };
    std::vector<Entry> index;                       // .fai order
    std::string path;
    // Throws FileError like the reference: unopenable FASTA, missing index (looked for at <stem>.fai, then
    // <path>.fai, src/Fasta.cpp:86-91), empty index.
    void open(const std::string &fasta_path);
    // base strings of every indexed contig (FASTA text without line ends), read with `threads` threads
    void load(std::vector<std::vector<uint8_t>> &sequences, int threads = 16) const;
};

}  // namespace rsqc_host
