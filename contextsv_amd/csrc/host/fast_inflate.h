// fast_inflate.h — raw-DEFLATE decoder for BGZF blocks (RFC 1951), written for the from-file path: one whole block in, one whole
// block out, 64-bit bit buffer refilled eight bytes at a time, one table look-up per symbol (11-bit primary tables with
// sub-tables for longer codes), matches copied eight bytes at a time. zlib 1.2.11's inflate decodes ~0.2 GB/s per core, which
// makes BGZF the bound of everything that starts from a BAM; this decoder runs several times faster.
// It is deliberately conservative about what it accepts: anything unusual (malformed or incomplete code sets, a stream that does
// not end exactly at the announced sizes) makes it return false, and the caller (bgzf::inflate_block) then decodes the block
// with zlib instead — so the result never depends on this file being right about a corner of the format, and every block is
// CRC-checked afterwards in either case.
#pragma once
#include <cstddef>
#include <cstdint>

namespace fastz {

// Decodes exactly out_len bytes from in[0..in_len) into out. Returns false when the stream is not a well-formed DEFLATE stream
// of exactly that size, or uses a feature this decoder leaves to zlib. Never reads outside in[0..in_len) or writes outside
// out[0..out_len).
bool inflate(const uint8_t *in, size_t in_len, uint8_t *out, size_t out_len);

// CRC-32 (IEEE 802.3, the gzip polynomial) of buf[0..len): carry-less-multiply folding where the CPU has PCLMULQDQ (checked once
// at start-up against the table-driven result), table-driven otherwise.
uint32_t crc32(const uint8_t *buf, size_t len);

}  // namespace fastz
