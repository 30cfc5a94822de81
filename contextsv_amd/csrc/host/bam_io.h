// bam_io.h — BAM / BAI reader and writer without htslib (SAMv1 §4.2, §5.2), feeding the struct-of-arrays shard the
// device path takes. It covers what the reference does through htslib on this path:
//   sam_open + sam_hdr_read + sam_index_load            (src/sv_caller.cpp:72-100, src/cnv_caller.cpp:419-445)
//   sam_itr_querys(idx, hdr, chr) + sam_itr_next loop    (src/sv_caller.cpp:114, :506-537; src/cnv_caller.cpp:466-530)
//   sam_itr_queryi(idx, HTS_IDX_START, 0, 0)             (src/sv_caller.cpp:121)
//   bam1_t core fields, bam_get_cigar / bam_get_qname / bam_get_seq, CG-tag long CIGARs (restored on read, as htslib does)
// One decode of a contig replaces the reference's three iterator passes over it: the records land in arrays, the
// arrays are uploaded once, and every pass runs on the device.
#pragma once
#include <cstdint>
#include <functional>
#include <memory>
#include <string>
#include <vector>

#include "../../../include/csvgpu.h"
#include "bgzf.h"
#include "hugevec.h"

struct BamHeader {
    std::string text;
    std::vector<std::string> names;
    std::vector<uint32_t> lens;
    int tid(const std::string &name) const;          // -1 when absent (bam_name2id)
};

// One contig's records in file order.
struct BamShard {
    int32_t tid = -1;
    std::string name;
    uint32_t target_len = 0;
    std::vector<int32_t>  pos;                        // 0-based, bam1_core_t::pos
    std::vector<uint16_t> flag;
    std::vector<uint8_t>  mapq;
    std::vector<uint64_t> cigar_off;                  // [n + 1]
    HugeVec<uint32_t> cigar;                          // BAM words (len << 4 | op), CG-tag CIGARs already restored
    std::vector<uint64_t> seq_off;                    // [n + 1] byte offsets into seq (want_seq)
    HugeVec<uint8_t> seq;                             // 4-bit packed, high nibble first, as bam_get_seq
    std::vector<std::string> qnames;                  // (want_qnames)
    uint64_t n_reads() const { return pos.size(); }
    csv_reads view() const;                           // host pointers into this shard (tid = nullptr)
    void clear();
};

struct BamReadOptions {
    bool want_seq = false;
    bool want_qnames = false;
    int threads = 8;                                  // inflate threads (hts_set_threads)
    uint32_t window_blocks = 2048;                    // BGZF blocks inflated per batch (<= 128 MiB of records)
};

class BamReader {
public:
    BamReader();
    ~BamReader();
    // Opens and reads the header; false + error() on failure (missing file, not BGZF/BAM, truncated header).
    bool open(const std::string &path);
    // Loads <path>.bai or, failing that, <path minus .bam>.bai (or an explicit path). Required by readContig, as the
    // reference requires an index for its iterators.
    bool loadIndex(const std::string &index_path = "");
    const BamHeader &header() const { return hdr; }
    const std::string &error() const { return err; }

    // sam_itr_querys(idx, hdr, chr): every record placed on the contig, in file order. `chr` is a contig name (the only form
    // the reference passes). False when the contig is unknown, the index is missing, or the data are corrupt.
    bool readContig(const std::string &chr, const BamReadOptions &opt, BamShard &out);

    // sam_itr_queryi(idx, HTS_IDX_START, 0, 0): one pass over the whole file; `sink` receives each contig's shard when its
    // last record has been seen (coordinate-sorted input: once per contig, ascending tid). Records without a contig are counted.
    bool readAll(const BamReadOptions &opt, const std::function<void(BamShard &&)> &sink, uint64_t *n_unplaced = nullptr);

    uint64_t bytes() const { return file.size(); }

private:
    struct Index;
    struct RecRef { const uint8_t *p; uint32_t len; };        // one record behind its block_size field
    // Hands the records of each inflated batch to on_batch, which returns how many of them it consumed: fewer than n ends the stream.
    bool stream(uint64_t start_voffset, const BamReadOptions &opt, const std::function<size_t(const RecRef *, size_t)> &on_batch);
    bool append(const RecRef *recs, size_t n, const BamReadOptions &opt, BamShard &out);
    bgzf::MappedFile file;
    BamHeader hdr;
    uint64_t first_record_voffset = 0;
    std::unique_ptr<Index> index;
    std::string path, err;
};

// BAM + BAI writer (synthetic inputs and tests). Records must be added in coordinate order for the index to be valid.
class BamWriter {
public:
    BamWriter();
    ~BamWriter();
    bool open(const std::string &path, const BamHeader &header, int level = 1, int threads = 8);
    // seq4: 4-bit packed bases ((l_seq + 1) / 2 bytes) or nullptr with l_seq == 0; qual: l_seq bytes or nullptr (0xff = missing).
    // CIGARs of more than 65535 operations go to a CG:B,I tag behind the <l_seq>S<ref_len>N placeholder (SAMv1 §4.2.2).
    void add(int32_t tid, int32_t pos, uint8_t mapq, uint16_t flag, const std::string &qname, const uint32_t *cigar, uint32_t n_cigar,
             const uint8_t *seq4, int32_t l_seq, const uint8_t *qual);
    // Finishes the BAM (EOF marker) and writes <path>.bai. False + error() on failure.
    bool close();
    const std::string &error() const { return err; }
    uint64_t records() const { return n_records; }

private:
    struct IndexEntry { int32_t tid; int32_t beg, end; uint64_t blk0; uint32_t uo0; uint64_t blk1; uint32_t uo1; uint16_t flag; };
    bool writeIndex();
    bgzf::Writer out;
    std::vector<IndexEntry> entries;
    std::vector<uint8_t> rec;
    BamHeader hdr;
    std::string path, err;
    uint64_t n_records = 0;
    bool opened = false;
};

// reference length and end of an alignment (bam_cigar2rlen / bam_endpos: pos + 1 when the CIGAR consumes no reference)
int32_t bam_ref_len(const uint32_t *cigar, uint32_t n_cigar);
inline int32_t bam_end_pos(int32_t pos, uint16_t flag, const uint32_t *cigar, uint32_t n_cigar)
{
    const int32_t rl = ((flag & 4) || n_cigar == 0) ? 0 : bam_ref_len(cigar, n_cigar);
    return pos + (rl > 0 ? rl : 1);
}
int bam_reg2bin(int64_t beg, int64_t end);          // SAMv1 §5.3
