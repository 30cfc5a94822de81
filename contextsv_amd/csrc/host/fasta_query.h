// fasta_query.h — reference-genome lookup with the reference's interface (include/fasta_query.h:16-44,
// src/fasta_query.cpp:18-185): whole FASTA in memory, 1-based inclusive queries returning views.
// The file is read in one block and split in place instead of getline + string append per line.
#pragma once
#include <cstdint>
#include <string>
#include <string_view>
#include <unordered_map>
#include <vector>

class ReferenceGenome {
public:
    // 0 on success, 1 for an empty path; a file that cannot be opened is fatal in the reference (exit(1), :31-35):
    // here it throws std::runtime_error so that a library caller decides.
    int setFilepath(std::string fasta_filepath);
    std::string getFilepath() const { return fasta_filepath; }

    // [pos_start, pos_end], 1-based inclusive; empty view when pos_end is past the contig or pos_start > pos_end (:88-102).
    // Unknown contig: std::out_of_range, as unordered_map::at.
    std::string_view query(const std::string &chr, uint32_t pos_start, uint32_t pos_end) const;

    // fraction of equal characters against compare_seq >= match_threshold (:105-136)
    bool compare(const std::string &chr, uint32_t pos_start, uint32_t pos_end, const std::string &compare_seq, float match_threshold) const;

    // "##contig=<ID=…,length=…>" lines, contigs in std::sort order, no trailing newline (:139-162)
    std::string getContigHeader() const;
    std::vector<std::string> getChromosomes() const { return chromosomes; }   // sorted; duplicates kept (:50, :73, :78)
    uint32_t getChromosomeLength(std::string chr) const;                       // 0 + printError when unknown (:169-180)

    // in-memory construction for tests / synthetic runs: same post-conditions as setFilepath
    void addContig(const std::string &name, std::string sequence);

private:
    std::string fasta_filepath;
    std::vector<std::string> chromosomes;
    std::unordered_map<std::string, std::string> chr_to_seq;
    std::unordered_map<std::string, uint32_t> chr_to_length;
};
