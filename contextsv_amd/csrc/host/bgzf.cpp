#include "bgzf.h"
#include "fast_inflate.h"

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <zlib.h>

#include <algorithm>
#include <atomic>
#include <cstdlib>
#include <cstring>
#include <thread>

namespace bgzf {

namespace {
inline uint16_t le16(const uint8_t *p) { return (uint16_t)(p[0] | (p[1] << 8)); }
inline uint32_t le32(const uint8_t *p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24); }
inline void put16(uint8_t *p, uint16_t v) { p[0] = (uint8_t)v; p[1] = (uint8_t)(v >> 8); }
inline void put32(uint8_t *p, uint32_t v) { p[0] = (uint8_t)v; p[1] = (uint8_t)(v >> 8); p[2] = (uint8_t)(v >> 16); p[3] = (uint8_t)(v >> 24); }
void set(std::string *err, const std::string &m) { if (err) *err = m; }

const uint8_t kEof[28] = {0x1f, 0x8b, 0x08, 0x04, 0, 0, 0, 0, 0, 0xff, 0x06, 0, 0x42, 0x43, 0x02, 0, 0x1b, 0, 0x03, 0, 0, 0, 0, 0, 0, 0, 0, 0};
}  // namespace

bool parse_block(const uint8_t *p, size_t avail, uint64_t coffset, Block &out, std::string *err)
{
    if (avail < 18) { set(err, "BGZF: truncated block header at offset " + std::to_string(coffset)); return false; }
    if (p[0] != 0x1f || p[1] != 0x8b || p[2] != 8 || !(p[3] & 4)) { set(err, "BGZF: not a BGZF block at offset " + std::to_string(coffset)); return false; }
    const uint32_t xlen = le16(p + 10);
    if (avail < 12 + (size_t)xlen) { set(err, "BGZF: truncated extra field at offset " + std::to_string(coffset)); return false; }
    // the BC subfield carries BSIZE = block size - 1; other subfields may sit beside it
    int64_t bsize = -1;
    for (uint32_t o = 0; o + 4 <= xlen;) {
        const uint8_t *sf = p + 12 + o;
        const uint32_t slen = le16(sf + 2);
        if (sf[0] == 'B' && sf[1] == 'C' && slen == 2 && o + 6 <= xlen) bsize = le16(sf + 4);
        o += 4 + slen;
    }
    if (bsize < 0) { set(err, "BGZF: block without BC subfield at offset " + std::to_string(coffset)); return false; }
    out.coffset = coffset;
    out.csize = (uint32_t)bsize + 1;
    out.data_off = (uint16_t)(12 + xlen);
    if (out.csize < (uint32_t)out.data_off + 8 || avail < out.csize) { set(err, "BGZF: truncated block at offset " + std::to_string(coffset)); return false; }
    out.isize = le32(p + out.csize - 4);
    if (out.isize > kMaxBlock) { set(err, "BGZF: block longer than 64 KiB at offset " + std::to_string(coffset)); return false; }
    return true;
}

bool scan_blocks(const uint8_t *file, size_t size, uint64_t from, std::vector<Block> &out, std::string *err)
{
    uint64_t off = from;
    while (off < size) {
        Block b;
        if (!parse_block(file + off, size - off, off, b, err)) return false;
        out.push_back(b);
        off += b.csize;
    }
    return true;
}

namespace {
// one inflate state per worker, reset per block (inflateInit2 allocates ~40 KiB; doing that per 64 KiB block is measurable
// and serialises threads in the allocator)
struct Inflater {
    z_stream zs;
    bool ready = false;
    Inflater() { memset(&zs, 0, sizeof zs); }
    ~Inflater() { if (ready) inflateEnd(&zs); }
    bool run(const uint8_t *file, const Block &b, uint8_t *dst, std::string *err)
    {
        if (b.isize == 0) return true;
        const uint8_t *src = file + b.coffset;
        const uint32_t want_crc = le32(src + b.csize - 8);
        // the fast decoder first; whatever it declines (or gets wrong: the CRC decides) goes through zlib
        static const bool zlib_only = [] { const char *e = getenv("CSV_ZLIB_ONLY"); return e && *e == '1'; }();
        if (!zlib_only && fastz::inflate(src + b.data_off, b.csize - b.data_off - 8, dst, b.isize) && fastz::crc32(dst, b.isize) == want_crc) return true;
        if (!ready) {
            if (inflateInit2(&zs, -15) != Z_OK) { set(err, "BGZF: inflateInit2 failed"); return false; }
            ready = true;
        } else if (inflateReset(&zs) != Z_OK) { set(err, "BGZF: inflateReset failed"); return false; }
        zs.next_in = const_cast<Bytef *>(src + b.data_off);
        zs.avail_in = b.csize - b.data_off - 8;
        zs.next_out = dst;
        zs.avail_out = b.isize;
        const int rc = inflate(&zs, Z_FINISH);
        if (rc != Z_STREAM_END || zs.total_out != b.isize) { set(err, "BGZF: inflate failed at offset " + std::to_string(b.coffset)); return false; }
        if (fastz::crc32(dst, b.isize) != want_crc) {
            set(err, "BGZF: CRC mismatch at offset " + std::to_string(b.coffset));
            return false;
        }
        return true;
    }
};
}  // namespace

bool inflate_block(const uint8_t *file, const Block &b, uint8_t *dst, std::string *err)
{
    Inflater z;
    return z.run(file, b, dst, err);
}

bool inflate_range(const uint8_t *file, const std::vector<Block> &blocks, size_t first, size_t last, uint8_t *dst, int threads, std::string *err)
{
    if (first >= last) return true;
    std::vector<uint64_t> uoff(last - first);
    uint64_t acc = 0;
    for (size_t i = first; i < last; i++) { uoff[i - first] = acc; acc += blocks[i].isize; }
    const int nt = (int)std::max<size_t>(1, std::min<size_t>((size_t)std::max(threads, 1), last - first));
    std::atomic<size_t> next{first};
    std::atomic<bool> failed{false};
    std::string first_err;
    std::atomic_flag err_lock = ATOMIC_FLAG_INIT;
    auto work = [&] {
        Inflater z;
        for (;;) {
            const size_t i = next.fetch_add(1);
            if (i >= last || failed.load(std::memory_order_relaxed)) return;
            std::string e;
            if (!z.run(file, blocks[i], dst + uoff[i - first], &e)) {
                failed = true;
                if (!err_lock.test_and_set()) first_err = e;
                return;
            }
        }
    };
    if (nt == 1) work();
    else {
        std::vector<std::thread> pool;
        for (int t = 0; t < nt; t++) pool.emplace_back(work);
        for (auto &t : pool) t.join();
    }
    if (failed) { set(err, first_err); return false; }
    return true;
}

MappedFile::~MappedFile() { if (p) munmap((void *)p, n); }

bool MappedFile::open(const std::string &path, std::string *err)
{
    const int fd = ::open(path.c_str(), O_RDONLY);
    if (fd < 0) { set(err, "cannot open " + path); return false; }
    struct stat st;
    if (fstat(fd, &st) != 0 || !S_ISREG(st.st_mode)) { ::close(fd); set(err, "cannot stat " + path); return false; }
    n = (size_t)st.st_size;
    if (n) {
        void *m = mmap(nullptr, n, PROT_READ, MAP_PRIVATE, fd, 0);
        if (m == MAP_FAILED) { ::close(fd); n = 0; set(err, "cannot map " + path); return false; }
        p = (const uint8_t *)m;
    }
    ::close(fd);
    return true;
}

bool deflate_block(const uint8_t *src, uint32_t n, int level, std::vector<uint8_t> &out)
{
    if (n > kMaxBlock) return false;
    const size_t base = out.size();
    // deflate output is bounded for n <= 0xff00 so that the block fits 64 KiB; larger payloads at level 0 could overflow, callers keep to kWriteBlock
    out.resize(base + 18 + compressBound(n) + 8);
    uint8_t *p = out.data() + base;
    const uint8_t head[16] = {0x1f, 0x8b, 0x08, 0x04, 0, 0, 0, 0, 0, 0xff, 0x06, 0, 'B', 'C', 2, 0};
    memcpy(p, head, 16);
    z_stream zs;
    memset(&zs, 0, sizeof zs);
    if (deflateInit2(&zs, level, Z_DEFLATED, -15, 8, Z_DEFAULT_STRATEGY) != Z_OK) return false;
    zs.next_in = const_cast<Bytef *>(src);
    zs.avail_in = n;
    zs.next_out = p + 18;
    zs.avail_out = (uInt)(out.size() - base - 18 - 8);
    const int rc = deflate(&zs, Z_FINISH);
    const size_t clen = zs.total_out;
    deflateEnd(&zs);
    if (rc != Z_STREAM_END) return false;
    const size_t total = 18 + clen + 8;
    if (total > kMaxBlock) return false;
    put16(p + 16, (uint16_t)(total - 1));
    put32(p + 18 + clen, (uint32_t)crc32(crc32(0L, Z_NULL, 0), src, n));
    put32(p + 18 + clen + 4, n);
    out.resize(base + total);
    return true;
}

Writer::~Writer() { if (f) fclose(f); }

bool Writer::open(const std::string &path, int lvl, int nthreads, std::string *err)
{
    f = fopen(path.c_str(), "wb");
    if (!f) { set(err, "cannot create " + path); return false; }
    level = lvl;
    threads = std::max(1, nthreads);
    pending.reserve((size_t)threads * 8 * kWriteBlock + kWriteBlock);
    return true;
}

void Writer::append(const void *data, size_t n)
{
    const uint8_t *p = (const uint8_t *)data;
    pending.insert(pending.end(), p, p + n);
    if (pending.size() >= (size_t)threads * 8 * kWriteBlock) flush_full(false);
}

// compress every whole block at the front of `pending` (and the partial tail when all == true), write them in order
bool Writer::flush_full(bool all)
{
    size_t nblk = pending.size() / kWriteBlock;
    const size_t tail = pending.size() - nblk * kWriteBlock;
    const size_t total = nblk + ((all && tail) ? 1 : 0);
    if (total == 0) return true;
    std::vector<std::vector<uint8_t>> comp(total);
    std::atomic<size_t> next{0};
    std::atomic<bool> bad{false};
    auto work = [&] {
        for (;;) {
            const size_t i = next.fetch_add(1);
            if (i >= total) return;
            const uint32_t len = i < nblk ? kWriteBlock : (uint32_t)tail;
            if (!deflate_block(pending.data() + i * kWriteBlock, len, level, comp[i])) bad = true;
        }
    };
    const int nt = (int)std::min<size_t>((size_t)threads, total);
    if (nt <= 1) work();
    else {
        std::vector<std::thread> pool;
        for (int t = 0; t < nt; t++) pool.emplace_back(work);
        for (auto &t : pool) t.join();
    }
    if (bad) { werr = "BGZF: deflate failed"; return false; }
    for (size_t i = 0; i < total; i++) {
        coffsets.push_back(end_coffset);
        if (fwrite(comp[i].data(), 1, comp[i].size(), f) != comp[i].size()) { werr = "BGZF: short write"; return false; }
        end_coffset += comp[i].size();
    }
    n_sealed += total;
    pending.erase(pending.begin(), pending.begin() + (all ? pending.size() : nblk * kWriteBlock));
    return true;
}

bool Writer::close(std::string *err)
{
    if (!f) return true;
    bool ok = flush_full(true) && werr.empty();
    if (ok) {
        coffsets.push_back(end_coffset);             // a position at the very end resolves to the EOF marker block
        ok = fwrite(kEof, 1, sizeof kEof, f) == sizeof kEof;
        end_coffset += sizeof kEof;
        if (!ok) werr = "BGZF: short write";
    }
    if (fclose(f) != 0 && ok) { ok = false; werr = "BGZF: close failed"; }
    f = nullptr;
    if (!ok) set(err, werr);
    return ok;
}

}  // namespace bgzf
