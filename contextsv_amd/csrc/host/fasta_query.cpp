#include "fasta_query.h"

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <cstring>
#include <iostream>
#include <stdexcept>

#include "log.h"

namespace {
// read-only mapping of a whole file; empty files map to (nullptr, 0)
struct FileMap {
    const char *p = nullptr;
    size_t n = 0;
    explicit FileMap(const std::string &path)
    {
        int fd = ::open(path.c_str(), O_RDONLY);
        if (fd < 0) throw std::runtime_error("Could not open FASTA file " + path);
        struct stat st;
        if (fstat(fd, &st) != 0 || !S_ISREG(st.st_mode)) { ::close(fd); throw std::runtime_error("Could not open FASTA file " + path); }
        n = (size_t)st.st_size;
        if (n) {
            void *m = mmap(nullptr, n, PROT_READ, MAP_PRIVATE, fd, 0);
            if (m == MAP_FAILED) { ::close(fd); throw std::runtime_error("Could not map FASTA file " + path); }
            (void)madvise(m, n, MADV_SEQUENTIAL);
            p = (const char *)m;
        }
        ::close(fd);
    }
    ~FileMap() { if (p) munmap((void *)p, n); }
    FileMap(const FileMap &) = delete;
    FileMap &operator=(const FileMap &) = delete;
};
}  // namespace

void ReferenceGenome::addContig(const std::string &name, std::string sequence)
{
    chromosomes.push_back(name);
    chr_to_length[name] = (uint32_t)sequence.length();
    chr_to_seq[name] = std::move(sequence);
    std::sort(chromosomes.begin(), chromosomes.end());
}

int ReferenceGenome::setFilepath(std::string path)
{
    if (path == "") {
        std::cout << "No FASTA filepath provided" << std::endl;
        return 1;
    }
    fasta_filepath = path;
    FileMap file(path);

    // Line semantics of std::getline (:40): split on '\n' only, a final unterminated line counts, '\r' stays in the data.
    // The pending sequence is cleared only when a named contig is stored (:46-53), so residues in front of the first
    // header or under an empty-named header run into the next named contig, as they do in the reference.
    std::string current_chr, sequence;
    auto store = [&] {
        chromosomes.push_back(current_chr);
        chr_to_length[current_chr] = (uint32_t)sequence.length();
        chr_to_seq[current_chr] = std::move(sequence);
        sequence.clear();
    };
    const char *cur = file.p, *end = file.p + file.n;
    while (cur < end) {
        const char *nl = (const char *)memchr(cur, '\n', (size_t)(end - cur));
        const char *eol = nl ? nl : end;
        if (eol > cur && *cur == '>') {
            if (current_chr != "") store();
            const char *sp = (const char *)memchr(cur + 1, ' ', (size_t)(eol - cur - 1));   // description starts at the first blank (:58-62)
            current_chr.assign(cur + 1, sp ? sp : eol);
        } else {
            sequence.append(cur, eol);
        }
        cur = nl ? nl + 1 : end;
    }
    if (current_chr != "") store();
    std::sort(chromosomes.begin(), chromosomes.end());
    return 0;
}

std::string_view ReferenceGenome::query(const std::string &chr, uint32_t pos_start, uint32_t pos_end) const
{
    pos_start--;                                  // uint32 wrap for 0 is part of the contract (:91-92)
    pos_end--;
    const std::string &sequence = chr_to_seq.at(chr);
    if (pos_end >= sequence.length() || pos_start > pos_end) return {};
    return std::string_view(sequence).substr(pos_start, (size_t)(pos_end - pos_start) + 1);
}

bool ReferenceGenome::compare(const std::string &chr, uint32_t pos_start, uint32_t pos_end, const std::string &compare_seq, float match_threshold) const
{
    pos_start--;
    pos_end--;
    const std::string &sequence = chr_to_seq.at(chr);
    if (pos_end >= sequence.length() || pos_start >= pos_end) return false;
    std::string_view sub = std::string_view(sequence).substr(pos_start, (size_t)(pos_end - pos_start) + 1);
    if (sub.length() != compare_seq.length()) {
        printError("ERROR: Sequence lengths do not match for comparison");
        return false;
    }
    size_t matches = 0;
    for (size_t i = 0; i < sub.length(); i++) matches += sub[i] == compare_seq[i];
    return (float)matches / (float)sub.length() >= match_threshold;
}

std::string ReferenceGenome::getContigHeader() const
{
    std::vector<std::string> names;
    names.reserve(chr_to_seq.size());
    for (const auto &kv : chr_to_seq) names.push_back(kv.first);
    std::sort(names.begin(), names.end());
    std::string out;
    for (const std::string &name : names) out += "##contig=<ID=" + name + ",length=" + std::to_string(chr_to_seq.at(name).length()) + ">\n";
    if (!out.empty()) out.pop_back();             // the reference pops unconditionally (UB on an empty genome, :159)
    return out;
}

uint32_t ReferenceGenome::getChromosomeLength(std::string chr) const
{
    auto it = chr_to_length.find(chr);
    if (it == chr_to_length.end()) {
        printError("Length for chromosome " + chr + " not found in reference genome");
        return 0;
    }
    return it->second;
}
