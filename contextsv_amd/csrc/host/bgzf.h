// bgzf.h — BGZF block layer on zlib (SAMv1 §4.1), replacing the part of htslib the reference uses underneath
// sam_open / sam_itr_next (src/sv_caller.cpp:48-181, src/cnv_caller.cpp:419-554). Blocks are independent raw-deflate
// streams of at most 64 KiB, so a file is inflated by a pool of threads, one block per task.
#pragma once
#include <cstddef>
#include <cstdint>
#include <cstdio>
#include <string>
#include <vector>

namespace bgzf {

constexpr uint32_t kMaxBlock = 0x10000;      // both compressed and uncompressed block sizes are <= 64 KiB
constexpr uint32_t kWriteBlock = 0xff00;     // uncompressed payload per written block (htslib's BGZF_BLOCK_SIZE)

inline uint64_t voffset(uint64_t coffset, uint32_t uoffset) { return (coffset << 16) | uoffset; }

struct Block {
    uint64_t coffset;        // file offset of the block header
    uint32_t csize;          // whole block: header + deflate data + 8-byte trailer
    uint32_t isize;          // uncompressed length
    uint16_t data_off;       // offset of the deflate stream inside the block (12 + XLEN)
};

// Parse one block header at p[0..avail). Returns false (with *err) when it is not a BGZF block or is truncated.
bool parse_block(const uint8_t *p, size_t avail, uint64_t coffset, Block &out, std::string *err);

// Block table of a whole mapped file starting at coffset `from`; stops at the end of the data.
bool scan_blocks(const uint8_t *file, size_t size, uint64_t from, std::vector<Block> &out, std::string *err);

// Inflate one block into dst[0..isize); verifies length and CRC32.
bool inflate_block(const uint8_t *file, const Block &b, uint8_t *dst, std::string *err);

// Inflate blocks[first..last) into dst (back to back) with up to `threads` threads.
bool inflate_range(const uint8_t *file, const std::vector<Block> &blocks, size_t first, size_t last, uint8_t *dst, int threads, std::string *err);

// Read-only memory map of a file.
class MappedFile {
public:
    MappedFile() = default;
    ~MappedFile();
    MappedFile(const MappedFile &) = delete;
    MappedFile &operator=(const MappedFile &) = delete;
    bool open(const std::string &path, std::string *err);
    const uint8_t *data() const { return p; }
    size_t size() const { return n; }
private:
    const uint8_t *p = nullptr;
    size_t n = 0;
};

// Block writer: append() bytes, blocks of kWriteBlock are deflated `threads` at a time and written in order.
// tell() is the virtual offset of the next byte — final, because a block's file offset is only assigned when it is
// written, tell() flushes nothing but is exact only through block_of(): callers that need virtual offsets (the BAI builder)
// record (block ordinal, in-block offset) pairs and resolve them with block_coffset() after close().
class Writer {
public:
    Writer() = default;
    ~Writer();
    bool open(const std::string &path, int level, int threads, std::string *err);
    void append(const void *data, size_t n);
    // position of the next byte as (block ordinal, offset inside that block)
    void where(uint64_t &block, uint32_t &uoffset) const { block = n_sealed + pending.size() / kWriteBlock; uoffset = (uint32_t)(pending.size() % kWriteBlock); }
    bool close(std::string *err);                      // flushes, writes the EOF marker block
    uint64_t block_coffset(uint64_t block) const { return block < coffsets.size() ? coffsets[block] : end_coffset; }
    uint64_t bytes_written() const { return end_coffset; }
private:
    bool flush_full(bool all);
    FILE *f = nullptr;
    int level = 1, threads = 1;
    std::vector<uint8_t> pending;            // not yet compressed; whole blocks are cut from its front
    uint64_t n_sealed = 0;                   // blocks already written
    std::vector<uint64_t> coffsets;          // file offset of every written block
    uint64_t end_coffset = 0;
    std::string werr;
};

// One-shot helpers
bool deflate_block(const uint8_t *src, uint32_t n, int level, std::vector<uint8_t> &out);   // appends one complete BGZF block to out

}  // namespace bgzf
