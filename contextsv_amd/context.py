"""Thin object wrapper over the csvgpu C-ABI: one `Context` per GPU (host-pointer entry points)
and `Shard` for inputs that stay resident in HBM. All compute happens behind the C-ABI."""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass

import numpy as np

from . import _lib
from ._lib import SIG_DTYPE, CsvError, csv_chr_result, csv_hmm, csv_reads, ptr


@dataclass
class Reads:
    """Struct-of-arrays view of one shard's alignment records (include/csvgpu.h csv_reads)."""
    pos: np.ndarray        # int32   [n]
    flag: np.ndarray       # uint16  [n]
    mapq: np.ndarray       # uint8   [n]
    cigar_off: np.ndarray  # uint64  [n+1]
    cigar: np.ndarray      # uint32  [m] packed len<<4|op
    tid: np.ndarray | None = None

    def __post_init__(self):
        self.pos = np.ascontiguousarray(self.pos, dtype=np.int32)
        self.flag = np.ascontiguousarray(self.flag, dtype=np.uint16)
        self.mapq = np.ascontiguousarray(self.mapq, dtype=np.uint8)
        self.cigar_off = np.ascontiguousarray(self.cigar_off, dtype=np.uint64)
        self.cigar = np.ascontiguousarray(self.cigar, dtype=np.uint32)
        n = len(self.pos)
        if not (len(self.flag) == n and len(self.mapq) == n and len(self.cigar_off) == n + 1):
            raise ValueError("Reads: array lengths disagree")
        if int(self.cigar_off[-1]) != len(self.cigar):
            raise ValueError("Reads: cigar_off[-1] != len(cigar)")

    @property
    def n_reads(self) -> int:
        return len(self.pos)

    @property
    def n_cigar(self) -> int:
        return len(self.cigar)

    def c_struct(self) -> csv_reads:
        r = csv_reads()
        r.n_reads, r.n_cigar = self.n_reads, self.n_cigar
        r.pos, r.flag, r.mapq = ptr(self.pos), ptr(self.flag), ptr(self.mapq)
        r.tid = ptr(self.tid) if self.tid is not None else None
        r.cigar_off, r.cigar = ptr(self.cigar_off), ptr(self.cigar)
        return r

    @staticmethod
    def from_cigar_lists(pos, flag, mapq, cigars) -> "Reads":
        """cigars: list of lists of (op, length) with BAM op codes (M0 I1 D2 N3 S4 H5 P6 =7 X8)."""
        off = np.zeros(len(cigars) + 1, dtype=np.uint64)
        words = []
        for i, c in enumerate(cigars):
            for op, ln in c:
                words.append((int(ln) << 4) | int(op))
            off[i + 1] = len(words)
        return Reads(np.asarray(pos), np.asarray(flag), np.asarray(mapq), off, np.asarray(words, dtype=np.uint32))


class Gate:
    """csv_gate: several Contexts on one GPU queue their scan + depth pairs back to back on the gate's one stream (csvgpu_gate_*)."""

    def __init__(self, device: int | None = None):
        """device given: the gate's stream is created now (csvgpu_gate_open) — do that BEFORE creating the lanes' Contexts, see csvgpu.h"""
        self.lib = _lib.load()
        self.h = self.lib.csvgpu_gate_create()
        if not self.h:
            raise CsvError(_lib.CSV_ENOMEM, "csvgpu_gate_create failed")
        if device is not None:
            rc = self.lib.csvgpu_gate_open(self.h, device)
            if rc:
                raise CsvError(rc, "csvgpu_gate_open failed")

    def close(self):
        if getattr(self, "h", None):
            self.lib.csvgpu_gate_destroy(self.h)
            self.h = None


class Context:
    """One csv_ctx (= one GPU). `stream` may be an existing hipStream_t handle (e.g. torch's)."""

    def __init__(self, device: int = 0, stream: int | None = None, background: bool = False):
        self.lib = _lib.load()
        self.h = self.lib.csvgpu_create_background(device) if background else self.lib.csvgpu_create(device, stream)
        if not self.h:
            msg = self.lib.csvgpu_last_error(None)
            raise CsvError(_lib.CSV_ENODEV, (msg or b"csvgpu_create failed").decode())

    def close(self):
        if getattr(self, "h", None):
            self.lib.csvgpu_destroy(self.h)
            self.h = None

    def set_gate(self, gate: "Gate | None"):
        self._check(self.lib.csvgpu_set_gate(self.h, gate.h if gate is not None else None))

    __del__ = close

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def _check(self, rc: int):
        if rc != 0:
            raise CsvError(rc, (self.lib.csvgpu_last_error(self.h) or b"").decode())

    # -------------------------------------------------------------------------------- host-pointer seams
    def cigar_scan(self, reads: Reads, depth_len: int, min_oplen: int = 50, min_mapq: int = 20, capacity: int | None = None):
        """-> structured array of signatures in the reference's chr_sv_calls order."""
        cap = reads.n_cigar if capacity is None else capacity
        out = np.zeros(max(cap, 1), dtype=SIG_DTYPE)
        n = C.c_uint64(cap)
        rs = reads.c_struct()
        self._check(self.lib.csvgpu_cigar_scan(self.h, C.byref(rs), depth_len, min_oplen, min_mapq, ptr(out), C.byref(n)))
        return out[: n.value].copy()

    def aln_intervals(self, reads: Reads):
        n = reads.n_reads
        ref_end, qs, qe = (np.zeros(max(n, 1), np.int32) for _ in range(3))
        rs = reads.c_struct()
        self._check(self.lib.csvgpu_aln_intervals(self.h, C.byref(rs), ptr(ref_end), ptr(qs), ptr(qe)))
        return ref_end[:n], qs[:n], qe[:n]

    def depth(self, reads: Reads, depth_len: int, want_array: bool = True):
        d = np.zeros(max(depth_len, 1), np.uint32) if want_array else None
        s, nz = C.c_uint64(0), C.c_uint32(0)
        rs = reads.c_struct()
        self._check(self.lib.csvgpu_depth(self.h, C.byref(rs), depth_len, ptr(d), C.byref(s), C.byref(nz)))
        return (d[:depth_len] if d is not None else None), s.value, nz.value

    def dbscan_iv(self, start, end, eps: float, min_pts: int):
        start = np.ascontiguousarray(start, np.uint32)
        end = np.ascontiguousarray(end, np.uint32)
        n = len(start)
        labels = np.zeros(max(n, 1), np.int32)
        self._check(self.lib.csvgpu_dbscan_iv(self.h, ptr(start), ptr(end), n, eps, min_pts, ptr(labels)))
        return labels[:n]

    def dbscan_iv_batch(self, start, end, seg_off, eps: float, min_pts: int):
        start = np.ascontiguousarray(start, np.uint32)
        end = np.ascontiguousarray(end, np.uint32)
        seg_off = np.ascontiguousarray(seg_off, np.uint64)
        labels = np.zeros(max(len(start), 1), np.int32)
        self._check(self.lib.csvgpu_dbscan_iv_batch(self.h, ptr(start), ptr(end), ptr(seg_off), len(seg_off) - 1, eps, min_pts, ptr(labels)))
        return labels[: len(start)]

    def dbscan_1d(self, pts, seg_off, eps: float, min_pts: int):
        pts = np.ascontiguousarray(pts, np.int32)
        seg_off = np.ascontiguousarray(seg_off, np.uint64)
        labels = np.zeros(max(len(pts), 1), np.int32)
        self._check(self.lib.csvgpu_dbscan_1d(self.h, ptr(pts), ptr(seg_off), len(seg_off) - 1, eps, min_pts, ptr(labels)))
        return labels[: len(pts)]

    def window_log2(self, depth, region_start, region_end, sample_size, mean_cov: float):
        depth = np.ascontiguousarray(depth, np.uint32)
        rs = np.ascontiguousarray(region_start, np.uint32)
        re = np.ascontiguousarray(region_end, np.uint32)
        ss = np.ascontiguousarray(sample_size, np.int32)
        off = np.zeros(len(rs) + 1, np.uint64)
        off[1:] = np.cumsum(ss.astype(np.int64))
        nw = int(off[-1])
        l2 = np.zeros(max(nw, 1), np.float64)
        ws, we = np.zeros(max(nw, 1), np.uint32), np.zeros(max(nw, 1), np.uint32)
        self._check(self.lib.csvgpu_window_log2(self.h, ptr(depth), len(depth), ptr(rs), ptr(re), ptr(ss), ptr(off), len(rs),
                                                mean_cov, ptr(l2), ptr(ws), ptr(we)))
        return l2[:nw], ws[:nw], we[:nw], off

    def viterbi(self, hmm: csv_hmm, o1, o2, pfb, seq_off):
        o1 = np.ascontiguousarray(o1, np.float64)
        o2 = np.ascontiguousarray(o2, np.float64)
        pfb = np.ascontiguousarray(pfb, np.float64)
        seq_off = np.ascontiguousarray(seq_off, np.uint64)
        n_seq = len(seq_off) - 1
        states = np.zeros(max(len(o1), 1), np.int32)
        ll = np.zeros(max(n_seq, 1), np.float64)
        self._check(self.lib.csvgpu_viterbi(self.h, C.byref(hmm), ptr(o1), ptr(o2), ptr(pfb), ptr(seq_off), n_seq, ptr(states), ptr(ll)))
        return states[: len(o1)], ll[:n_seq]

    def split_order(self, shards, min_mapq: int, supp_hash, capacity: int | None = None):
        """csvgpu_split_order: per contig (shard with query-name hashes attached) the records of the surviving primaries in the
        iteration order of the reference's qname map -> list of uint32 arrays."""
        supp_hash = np.ascontiguousarray(supp_hash, np.uint64)
        n = len(shards)
        hs = (C.c_void_p * max(n, 1))(*[s.h for s in shards])
        cap = capacity if capacity is not None else max(1024, 2 * len(supp_hash))
        out = np.zeros(max(cap, 1), np.uint32)
        off = np.zeros(n + 1, np.uint64)
        rc = self.lib.csvgpu_split_order(self.h, n, hs, min_mapq, ptr(supp_hash), len(supp_hash), ptr(out), cap, ptr(off))
        if rc == _lib.CSV_ECAPACITY and capacity is None:
            return self.split_order(shards, min_mapq, supp_hash, capacity=int(off[n]))
        self._check(rc)
        return [out[int(off[k]): int(off[k + 1])].copy() for k in range(n)]

    def split_order_two_calls(self, shards, min_mapq: int, supp_hash, capacity: int = 4):
        """csvgpu_split_order_begin (everything that needs no supplementary record, queued) + csvgpu_split_order_finish; a too small
        `capacity` exercises the retry of _finish after CSV_ECAPACITY."""
        supp_hash = np.ascontiguousarray(supp_hash, np.uint64)
        n = len(shards)
        hs = (C.c_void_p * max(n, 1))(*[s.h for s in shards])
        self._check(self.lib.csvgpu_split_order_begin(self.h, n, hs, min_mapq))
        out = np.zeros(max(capacity, 1), np.uint32)
        off = np.zeros(n + 1, np.uint64)
        rc = self.lib.csvgpu_split_order_finish(self.h, ptr(supp_hash), len(supp_hash), ptr(out), capacity, ptr(off))
        if rc == _lib.CSV_ECAPACITY:
            capacity = int(off[n])
            out = np.zeros(max(capacity, 1), np.uint32)
            rc = self.lib.csvgpu_split_order_finish(self.h, ptr(supp_hash), len(supp_hash), ptr(out), capacity, ptr(off))
        self._check(rc)
        return [out[int(off[k]): int(off[k + 1])].copy() for k in range(n)]

    def split_order_self(self, shards, min_mapq: int, capacity: int = 4):
        """csvgpu_split_order_begin_self (the supplementary hashes are those of the same shards' supplementary records; the whole order is
        queued) + csvgpu_split_order_finish without hashes."""
        n = len(shards)
        hs = (C.c_void_p * max(n, 1))(*[s.h for s in shards])
        self._check(self.lib.csvgpu_split_order_begin_self(self.h, n, hs, min_mapq))
        out = np.zeros(max(capacity, 1), np.uint32)
        off = np.zeros(n + 1, np.uint64)
        rc = self.lib.csvgpu_split_order_finish(self.h, None, 0, ptr(out), capacity, ptr(off))
        if rc == _lib.CSV_ECAPACITY:
            capacity = int(off[n])
            out = np.zeros(max(capacity, 1), np.uint32)
            rc = self.lib.csvgpu_split_order_finish(self.h, None, 0, ptr(out), capacity, ptr(off))
        self._check(rc)
        return [out[int(off[k]): int(off[k + 1])].copy() for k in range(n)]

    # -------------------------------------------------------------------------------- timing
    def synchronize(self):
        self._check(self.lib.csvgpu_synchronize(self.h))

    def timing_enable(self, on=True):
        """0 / False: off; 1 / True: HIP-event timers around every kernel group; 2: only around the CIGAR scan and the depth pass."""
        self._check(self.lib.csvgpu_timing_enable(self.h, int(on)))

    def timing_reset(self):
        self._check(self.lib.csvgpu_timing_reset(self.h))

    def timing(self) -> dict:
        out = {}
        for k, name in enumerate(_lib.KERNEL_NAMES):
            ms, n = C.c_double(0), C.c_uint64(0)
            self._check(self.lib.csvgpu_timing_get(self.h, k, C.byref(ms), C.byref(n)))
            out[name] = (ms.value, n.value)
        return out

    # -------------------------------------------------------------------------------- resident shards
    def upload(self, reads: Reads, depth_len: int) -> "Shard":
        rs = reads.c_struct()
        h = self.lib.csvgpu_shard_upload(self.h, C.byref(rs), depth_len)
        if not h:
            raise CsvError(_lib.CSV_ENOMEM, (self.lib.csvgpu_last_error(self.h) or b"").decode())
        return Shard(self, h, reads.n_reads, depth_len)

    def wrap_device_ptrs(self, n_reads: int, n_cigar: int, pos: int, flag: int, mapq: int, cigar_off: int, cigar: int, depth_len: int, keep=None) -> "Shard":
        """Arrays that already live in HBM, given as device addresses (csv_reads layouts), wrapped without a copy
        (csvgpu_shard_wrap_dev). They must outlive the shard; `keep` is held by it for that purpose."""
        r = csv_reads()
        r.n_reads, r.n_cigar = n_reads, n_cigar
        r.pos, r.flag, r.mapq, r.tid, r.cigar_off, r.cigar = pos, flag, mapq, None, cigar_off, cigar       # (void * fields)
        h = self.lib.csvgpu_shard_wrap_dev(self.h, C.byref(r), depth_len)
        if not h:
            raise CsvError(_lib.CSV_EINVAL, (self.lib.csvgpu_last_error(self.h) or b"").decode())
        sh = Shard(self, h, n_reads, depth_len)
        sh._keep = keep
        return sh

    def wrap_device(self, pos, flag, mapq, cigar_off, cigar, depth_len: int) -> "Shard":
        """The same for torch tensors on this context's device (int32 / int16 / uint8 / int64 / int32 bit patterns of csv_reads)."""
        n, m = int(pos.numel()), int(cigar.numel())
        if not (flag.numel() == n and mapq.numel() == n and cigar_off.numel() == n + 1):
            raise ValueError("wrap_device: array lengths disagree")
        for t, size in ((pos, 4), (flag, 2), (mapq, 1), (cigar_off, 8), (cigar, 4)):
            if not t.is_cuda or not t.is_contiguous() or t.element_size() != size:
                raise ValueError("wrap_device: tensors must be contiguous device tensors of the csv_reads element sizes")
        return self.wrap_device_ptrs(n, m, pos.data_ptr(), flag.data_ptr(), mapq.data_ptr(), cigar_off.data_ptr(), cigar.data_ptr(), depth_len,
                                     keep=(pos, flag, mapq, cigar_off, cigar))


@dataclass
class ChrResult:
    n_sig: int
    n_del: int
    n_ins: int
    depth_sum: int
    depth_nonzero: int
    min_pts: int
    mean_cov: float
    raw: csv_chr_result


class Shard:
    """Reads resident in HBM + the per-chromosome device pipeline (csvgpu_chr_pipeline_dev)."""

    def __init__(self, ctx: Context, handle, n_reads: int, depth_len: int):
        self.ctx, self.h, self.n_reads, self.depth_len = ctx, handle, n_reads, depth_len

    def free(self):
        if self.h and self.ctx.h:
            self.ctx.lib.csvgpu_shard_free(self.ctx.h, self.h)
        self.h = None

    def set_qname_hash(self, qhash):
        qhash = np.ascontiguousarray(qhash, np.uint64)
        assert len(qhash) == self.n_reads
        self.ctx._check(self.ctx.lib.csvgpu_shard_set_qname_hash(self.ctx.h, self.h, ptr(qhash)))

    def pipeline(self, eps: float = 0.1, min_pts_pct: float = 0.1, min_oplen: int = 50, min_mapq: int = 20) -> ChrResult:
        res = csv_chr_result()
        self.ctx._check(self.ctx.lib.csvgpu_chr_pipeline_dev(self.ctx.h, self.h, min_oplen, min_mapq, eps, min_pts_pct, C.byref(res)))
        return ChrResult(res.n_sig, res.n_del, res.n_ins, res.depth_sum, res.depth_nonzero, res.min_pts, res.mean_cov, res)

    def fetch(self, res: ChrResult, want_depth: bool = False):
        """Copy the pipeline's device results to host numpy arrays (test / host-merge plumbing)."""
        def d2h(p, nbytes, dtype):
            if not p or nbytes == 0:
                return np.zeros(0, dtype)
            out = np.zeros(nbytes // np.dtype(dtype).itemsize, dtype)
            self.ctx._check(self.ctx.lib.csvgpu_download(self.ctx.h, out.ctypes.data, p, nbytes))
            return out

        r = res.raw
        out = {
            "sig_del": d2h(r.sig_del, res.n_del * 16, SIG_DTYPE),
            "sig_ins": d2h(r.sig_ins, res.n_ins * 16, SIG_DTYPE),
            "label_del": d2h(r.label_del, res.n_del * 4, np.int32),
            "label_ins": d2h(r.label_ins, res.n_ins * 4, np.int32),
            "ref_end": d2h(r.ref_end, self.n_reads * 4, np.int32),
            "q_start": d2h(r.q_start, self.n_reads * 4, np.int32),
            "q_end": d2h(r.q_end, self.n_reads * 4, np.int32),
        }
        if want_depth:
            out["depth"] = d2h(r.depth, self.depth_len * 4, np.uint32)
        return out

    def depth_lookup(self, pos) -> np.ndarray:
        """depth[pos[i]] on the resident depth map; -1 where pos[i] is outside it (csvgpu_depth_lookup_resident)."""
        pos = np.ascontiguousarray(pos, np.uint32)
        out = np.zeros(max(len(pos), 1), np.int32)
        self.ctx._check(self.ctx.lib.csvgpu_depth_lookup_resident(self.ctx.h, self.h, ptr(pos), len(pos), ptr(out)))
        return out[: len(pos)]
