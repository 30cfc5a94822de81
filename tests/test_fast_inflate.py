"""The from-file path's own DEFLATE decoder and CRC-32 (host/fast_inflate.cpp) against zlib: every block type, strategy and
level on literal-heavy, repetitive and CIGAR-like data; and garbage in must never crash or be accepted silently (whatever the
decoder accepts is CRC-checked by its caller, whatever it declines goes to zlib). CPU only."""
import ctypes as C
import zlib

import numpy as np
import pytest

from contextsv_amd import host


@pytest.fixture(scope="module")
def lib():
    l = host.load()
    l.csvhost_fast_inflate.argtypes = [C.c_char_p, C.c_uint64, C.c_void_p, C.c_uint64]
    l.csvhost_fast_crc32.argtypes = [C.c_char_p, C.c_uint64]
    l.csvhost_fast_crc32.restype = C.c_uint32
    return l


def deflate(data, level, strategy=zlib.Z_DEFAULT_STRATEGY):
    c = zlib.compressobj(level, zlib.DEFLATED, -15, 8, strategy)
    return c.compress(data) + c.flush()


def datasets():
    rng = np.random.default_rng(1)
    out = []
    for n in (0, 1, 2, 5, 17, 100, 1000, 20_000, 65_280):
        out.append(bytes(rng.integers(0, 256, n, dtype=np.uint8)))            # incompressible: stored / literal-only blocks
        out.append(bytes(rng.integers(0, 4, n, dtype=np.uint8)))              # short codes
        out.append(b"A" * n)                                                   # distance-1 runs
        out.append((b"abcdefgh" * (n // 8 + 1))[:n])                           # distance-8 matches
        out.append((b"xyz" * (n // 3 + 1))[:n])                                # overlapping copies, distance < 8
        w = rng.integers(1, 400, max(n // 4, 1)).astype(np.uint32) << 4 | rng.choice([0, 1, 2, 4], max(n // 4, 1)).astype(np.uint32)
        out.append(w.tobytes()[:n])                                            # CIGAR words
    return out


def test_matches_zlib_on_every_block_kind(lib):
    n_decoded = n_total = 0
    for data in datasets():
        assert lib.csvhost_fast_crc32(data, len(data)) == zlib.crc32(data) & 0xFFFFFFFF
        for level in (0, 1, 3, 6, 9):
            for strategy in (zlib.Z_DEFAULT_STRATEGY, zlib.Z_FIXED, zlib.Z_HUFFMAN_ONLY, zlib.Z_RLE, zlib.Z_FILTERED):
                comp = deflate(data, level, strategy)
                out = np.full(len(data) + 16, 0xEE, np.uint8)                  # guard bytes behind the block
                ok = lib.csvhost_fast_inflate(comp, len(comp), out.ctypes.data, len(data))
                n_total += 1
                if ok:
                    n_decoded += 1
                    assert out[: len(data)].tobytes() == data, (len(data), level, strategy)
                assert (out[len(data):] == 0xEE).all(), "wrote past the end of the block"
                # the announced size must be exact: one byte less or more is declined, not truncated or padded
                if len(data) > 0:
                    assert not lib.csvhost_fast_inflate(comp, len(comp), out.ctypes.data, len(data) - 1)
                assert not lib.csvhost_fast_inflate(comp, len(comp), out.ctypes.data, len(data) + 1)
    assert n_decoded == n_total                                                # zlib's output never needs the fallback


def test_crc_all_lengths_and_alignments(lib):
    rng = np.random.default_rng(3)
    buf = bytes(rng.integers(0, 256, 5000, dtype=np.uint8))
    for n in list(range(0, 300)) + [511, 512, 513, 1000, 4095, 4096, 4999]:
        for off in (0, 1, 3, 7):
            piece = buf[off: off + n]
            assert lib.csvhost_fast_crc32(piece, len(piece)) == zlib.crc32(piece) & 0xFFFFFFFF, (n, off)


def test_garbage_is_declined_or_caught(lib):
    rng = np.random.default_rng(5)
    data = datasets()[-1]
    comp = deflate(data, 6)
    want_crc = zlib.crc32(data) & 0xFFFFFFFF
    accepted_wrong = 0
    for _ in range(4000):
        b = bytearray(comp)
        for _ in range(int(rng.integers(1, 4))):
            b[int(rng.integers(0, len(b)))] ^= int(rng.integers(1, 256))
        if rng.random() < 0.3:
            b = b[: int(rng.integers(0, len(b)))]
        out = np.full(len(data) + 16, 0xEE, np.uint8)
        if lib.csvhost_fast_inflate(bytes(b), len(b), out.ctypes.data, len(data)):
            got = out[: len(data)].tobytes()
            if got != data:
                accepted_wrong += 1
                assert lib.csvhost_fast_crc32(got, len(got)) != want_crc        # the caller's CRC check rejects it
        assert (out[len(data):] == 0xEE).all()
    assert accepted_wrong > 0                                                   # the CRC check is what stands behind the decoder
    for junk in (b"", b"\x00", b"\xff" * 40, bytes(rng.integers(0, 256, 300, dtype=np.uint8))):
        out = np.zeros(64, np.uint8)
        lib.csvhost_fast_inflate(junk, len(junk), out.ctypes.data, 48)
