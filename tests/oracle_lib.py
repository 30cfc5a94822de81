"""ctypes loaders for the CHECKERS (tests only): oracle/_build/libcsvoracle.so (our C restatement) and
oracle/_ref/libcsvref.so (the reference's own dbscan.cpp, dbscan1d.cpp, kc.cpp). Never imported by
the product package."""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_SO = os.path.join(ROOT, "oracle", "_build", "libcsvoracle.so")
REF_SO = os.path.join(ROOT, "oracle", "_ref", "libcsvref.so")

SIG_DTYPE = np.dtype([("start", "<u4"), ("end", "<u4"), ("read", "<u4"), ("qpos_kind", "<u4")])
CALL_DTYPE = np.dtype([("start", "<u4"), ("end", "<u4"), ("sv_type", "<i4"), ("cluster_size", "<i4"),
                       ("hmm_likelihood", "<f8"), ("id", "<i8")])
P = C.c_void_p
LABEL_FN = C.CFUNCTYPE(None, P, P, C.c_uint64, C.c_double, C.c_int32, P)


class orc_hmm(C.Structure):
    _fields_ = [("A", C.c_double * 36), ("pi", C.c_double * 6), ("B1_mean", C.c_double * 6), ("B1_sd", C.c_double * 6),
                ("B1_uf", C.c_double), ("B2_mean", C.c_double * 5), ("B2_sd", C.c_double * 5), ("B2_uf", C.c_double)]


def _p(a):
    return a.ctypes.data if a is not None else None


class Oracle:
    def __init__(self, lib):
        self.lib = lib
        lib.orc_cigar_scan.restype = C.c_int64
        lib.orc_cigar_scan.argtypes = [C.c_uint64, P, P, P, P, P, C.c_uint32, C.c_uint32, C.c_uint8, P, C.c_uint64]
        lib.orc_aln_intervals.restype = None
        lib.orc_aln_intervals.argtypes = [C.c_uint64, P, P, P, P, P, P, P]
        lib.orc_depth.restype = None
        lib.orc_depth.argtypes = [C.c_uint64, P, P, P, P, C.c_uint32, P, C.POINTER(C.c_uint64), C.POINTER(C.c_uint32)]
        lib.orc_dbscan_iv.restype = None
        lib.orc_dbscan_iv.argtypes = [P, P, C.c_uint64, C.c_double, C.c_int32, P]
        lib.orc_dbscan_1d.restype = None
        lib.orc_dbscan_1d.argtypes = [P, C.c_uint64, C.c_double, C.c_int32, P]
        lib.orc_largest_cluster.restype = C.c_int64
        lib.orc_largest_cluster.argtypes = [P, P, C.c_uint64, P]
        lib.orc_window_log2.restype = None
        lib.orc_window_log2.argtypes = [P, C.c_uint32, C.c_uint32, C.c_uint32, C.c_int32, C.c_double, P, P, P]
        lib.orc_viterbi_batch.restype = None
        lib.orc_viterbi_batch.argtypes = [C.POINTER(orc_hmm), P, P, P, P, C.c_uint64, P, P]
        for f in (lib.orc_pdf_normal, lib.orc_cdf_normal):
            f.restype = C.c_double
            f.argtypes = [C.c_double] * 3
        lib.orc_merge_svs.restype = C.c_int64
        lib.orc_merge_svs.argtypes = [P, C.c_uint64, C.c_double, C.c_int32, C.c_int, LABEL_FN, P]
        lib.orc_merge_duplicates.restype = C.c_int64
        lib.orc_merge_duplicates.argtypes = [P, C.c_uint64]
        lib.orc_fasta_query.restype = C.c_int64
        lib.orc_fasta_query.argtypes = [C.c_char_p, C.c_char_p, C.c_uint32, C.c_uint32, P, C.c_uint64]
        lib.orc_fasta_describe.restype = C.c_int64
        lib.orc_fasta_describe.argtypes = [C.c_char_p, P, C.c_uint64]
        lib.orc_save_vcf.restype = C.c_int
        lib.orc_save_vcf.argtypes = [C.c_char_p, C.c_char_p, C.c_char_p, C.c_char_p, C.c_int, P, P, P, P, P, P, P]

    def read_snp_af(self, snp_txt, pfb_txt, chr, chr_gnomad, start, end, af_key, cap=1 << 20):
        """readSNPAlleleFrequencies restated for one region over plain-text VCFs -> (pos, baf, (pfb_pos, pfb) | None)."""
        self.lib.orc_read_snp_af.restype = C.c_int64
        self.lib.orc_read_snp_af.argtypes = [C.c_char_p, C.c_char_p, C.c_char_p, C.c_char_p, C.c_uint32, C.c_uint32, C.c_char_p, P, P, C.c_uint64,
                                             C.POINTER(C.c_int), C.POINTER(C.c_uint32), C.POINTER(C.c_double)]
        pos, baf = np.zeros(cap, np.uint32), np.zeros(cap, np.float64)
        has, pp, pv = C.c_int(0), C.c_uint32(0), C.c_double(0)
        n = self.lib.orc_read_snp_af(snp_txt.encode(), pfb_txt.encode() if pfb_txt else None, chr.encode(), chr_gnomad.encode(), start, end,
                                     af_key.encode(), pos.ctypes.data, baf.ctypes.data, cap, C.byref(has), C.byref(pp), C.byref(pv))
        assert n >= 0, n
        return pos[:n].copy(), baf[:n].copy(), ((pp.value, pv.value) if has.value else None)

    def fasta_query(self, fasta, chr, a, b):
        """ReferenceGenome::query restated; None for an unknown contig."""
        n = self.lib.orc_fasta_query(fasta.encode(), chr.encode(), a, b, None, 0)
        if n == -1:
            return None
        assert n >= 0, n
        buf = C.create_string_buffer(max(int(n), 1))
        self.lib.orc_fasta_query(fasta.encode(), chr.encode(), a, b, buf, n)
        return buf.raw[:n]

    def fasta_describe(self, fasta):
        """(getContigHeader, getChromosomes) restated."""
        n = self.lib.orc_fasta_describe(fasta.encode(), None, 0)
        assert n >= 0, n
        buf = C.create_string_buffer(max(int(n), 1))
        self.lib.orc_fasta_describe(fasta.encode(), buf, n)
        hdr, names = buf.raw[:n].split(b"\x01")
        return hdr, (names.split(b"\n") if names else [])

    def save_vcf(self, out_path, fasta, contigs, gap_path=None, file_date=None):
        """saveToVCF restated. contigs = [(name, calls[48-byte POD], alts, depth uint32 array or None)]. Returns (rc, counts)."""
        n = len(contigs)
        names = (C.c_char_p * max(n, 1))(*[c[0].encode() for c in contigs])
        off = np.zeros(n + 1, np.uint64)
        off[1:] = np.cumsum([len(c[1]) for c in contigs])
        calls = np.ascontiguousarray(np.concatenate([c[1] for c in contigs])) if n else np.zeros(0, np.uint8)
        assert calls.dtype.itemsize == 48
        alt_list = [a for c in contigs for a in c[2]]
        alts = (C.c_char_p * max(len(alt_list), 1))(*alt_list)
        arrs = [(np.ascontiguousarray(c[3], np.uint32) if c[3] is not None else None) for c in contigs]
        dptr = (C.c_void_p * max(n, 1))(*[(a.ctypes.data if a is not None else None) for a in arrs])
        dlen = np.array([(len(a) if a is not None else 0) for a in arrs] + [0], np.uint64)
        counts = np.zeros(3, np.int32)
        rc = self.lib.orc_save_vcf(out_path.encode(), fasta.encode(), gap_path.encode() if gap_path else None,
                                   file_date.encode() if file_date else None, n, names, off.ctypes.data, calls.ctypes.data, alts,
                                   dptr, dlen.ctypes.data, counts.ctypes.data)
        return rc, tuple(int(x) for x in counts)

    def cigar_scan(self, reads, depth_len, min_oplen=50, min_mapq=20):
        cap = max(reads.n_cigar, 1)
        out = np.zeros(cap, SIG_DTYPE)
        n = self.lib.orc_cigar_scan(reads.n_reads, _p(reads.pos), _p(reads.flag), _p(reads.mapq), _p(reads.cigar_off),
                                    _p(reads.cigar), depth_len, min_oplen, min_mapq, _p(out), cap)
        return out[:n].copy()

    def aln_intervals(self, reads):
        n = reads.n_reads
        a, b, c = (np.zeros(max(n, 1), np.int32) for _ in range(3))
        self.lib.orc_aln_intervals(n, _p(reads.pos), _p(reads.flag), _p(reads.cigar_off), _p(reads.cigar), _p(a), _p(b), _p(c))
        return a[:n], b[:n], c[:n]

    def depth(self, reads, depth_len):
        d = np.zeros(max(depth_len, 1), np.uint32)
        s, nz = C.c_uint64(0), C.c_uint32(0)
        self.lib.orc_depth(reads.n_reads, _p(reads.pos), _p(reads.flag), _p(reads.cigar_off), _p(reads.cigar), depth_len,
                           _p(d), C.byref(s), C.byref(nz))
        return d[:depth_len], s.value, nz.value

    def dbscan_iv(self, start, end, eps, min_pts):
        start = np.ascontiguousarray(start, np.uint32); end = np.ascontiguousarray(end, np.uint32)
        lab = np.zeros(max(len(start), 1), np.int32)
        self.lib.orc_dbscan_iv(_p(start), _p(end), len(start), eps, min_pts, _p(lab))
        return lab[: len(start)]

    def dbscan_iv_windowed(self, start, end, eps, min_pts):
        """oracle/dbscan_window.cpp: the order-free windowed form, O(n * window) — pinned against the reference in tests/test_oracle_windowed.py"""
        start = np.ascontiguousarray(start, np.uint32); end = np.ascontiguousarray(end, np.uint32)
        lab = np.zeros(max(len(start), 1), np.int32)
        self.lib.orc_dbscan_iv_windowed.restype = C.c_int
        self.lib.orc_dbscan_iv_windowed.argtypes = [P, P, C.c_uint64, C.c_double, C.c_int32, P]
        rc = self.lib.orc_dbscan_iv_windowed(_p(start), _p(end), len(start), eps, min_pts, _p(lab))
        if rc:
            raise ValueError("orc_dbscan_iv_windowed: eps outside [0, 1)")
        return lab[: len(start)]

    def dbscan_1d(self, pts, eps, min_pts):
        pts = np.ascontiguousarray(pts, np.int32)
        lab = np.zeros(max(len(pts), 1), np.int32)
        self.lib.orc_dbscan_1d(_p(pts), len(pts), eps, min_pts, _p(lab))
        return lab[: len(pts)]

    def largest_cluster(self, pts, labels):
        pts = np.ascontiguousarray(pts, np.int32); labels = np.ascontiguousarray(labels, np.int32)
        out = np.zeros(max(len(pts), 1), np.int32)
        m = self.lib.orc_largest_cluster(_p(pts), _p(labels), len(pts), _p(out))
        return out[:m]

    def window_log2(self, depth, start, end, sample_size, mean_cov):
        depth = np.ascontiguousarray(depth, np.uint32)
        l2 = np.zeros(sample_size, np.float64); ws = np.zeros(sample_size, np.uint32); we = np.zeros(sample_size, np.uint32)
        self.lib.orc_window_log2(_p(depth), len(depth), start, end, sample_size, mean_cov, _p(l2), _p(ws), _p(we))
        return l2, ws, we

    def viterbi(self, hmm, o1, o2, pfb, seq_off):
        """hmm: contextsv_amd._lib.csv_hmm or orc_hmm (same layout)."""
        h = orc_hmm.from_buffer_copy(bytes(hmm))
        o1 = np.ascontiguousarray(o1, np.float64); o2 = np.ascontiguousarray(o2, np.float64); pfb = np.ascontiguousarray(pfb, np.float64)
        seq_off = np.ascontiguousarray(seq_off, np.uint64)
        st = np.zeros(max(len(o1), 1), np.int32); ll = np.zeros(max(len(seq_off) - 1, 1), np.float64)
        self.lib.orc_viterbi_batch(C.byref(h), _p(o1), _p(o2), _p(pfb), _p(seq_off), len(seq_off) - 1, _p(st), _p(ll))
        return st[: len(o1)], ll[: len(seq_off) - 1]

    # ---- copy-number pass (oracle/cnv_oracle.cpp); calls use the 48-byte layout of contextsv_amd.host.CALL_DTYPE
    def _snps(self, snps):
        return (np.ascontiguousarray(snps["pos"], np.uint32), np.ascontiguousarray(snps["baf"], np.float64),
                np.ascontiguousarray(snps["pfb"], np.float64), np.ascontiguousarray(snps["has_pfb"], np.uint8))

    def query_snp_region(self, depth, start, end, mean_cov, sample_size, snps, cap=1 << 16):
        depth = np.ascontiguousarray(depth, np.uint32)
        pos, baf, pfb, has = self._snps(snps)
        o_pos = np.zeros(cap, np.uint32); o_baf = np.zeros(cap); o_pfb = np.zeros(cap); o_l2 = np.zeros(cap); o_is = np.zeros(cap, np.uint8)
        f = self.lib.orc_query_snp_region
        f.restype = C.c_int64
        f.argtypes = [P, C.c_uint32, C.c_uint32, C.c_uint32, C.c_double, C.c_int, P, P, P, P, C.c_uint64, P, P, P, P, P, C.c_uint64]
        k = f(_p(depth), len(depth), start, end, mean_cov, sample_size, _p(pos), _p(baf), _p(pfb), _p(has), len(pos),
              _p(o_pos), _p(o_baf), _p(o_pfb), _p(o_l2), _p(o_is), cap)
        return {"pos": o_pos[:k], "baf": o_baf[:k], "pfb": o_pfb[:k], "log2_cov": o_l2[:k], "is_snp": o_is[:k].astype(bool)}

    def cn_prediction(self, depth, calls, hmm, mean_cov, snps, split, sample_size=20, min_cnv=2000):
        depth = np.ascontiguousarray(depth, np.uint32)
        pos, baf, pfb, has = self._snps(snps)
        h = orc_hmm.from_buffer_copy(bytes(hmm))
        cap = 2 * len(calls) + 16
        buf = np.zeros(cap, calls.dtype)
        buf[: len(calls)] = calls
        if split:
            f = self.lib.orc_split_cn_prediction
            f.restype = C.c_int64
            f.argtypes = [P, C.c_uint32, P, C.c_uint64, C.c_uint64, C.POINTER(orc_hmm), C.c_double, C.c_int, P, P, P, P, C.c_uint64]
            n = f(_p(depth), len(depth), _p(buf), len(calls), cap, C.byref(h), mean_cov, sample_size, _p(pos), _p(baf), _p(pfb), _p(has), len(pos))
            return buf[:n].copy()
        f = self.lib.orc_cigar_cn_prediction
        f.restype = None
        f.argtypes = [P, C.c_uint32, P, C.c_uint64, C.POINTER(orc_hmm), C.c_double, C.c_int, C.c_uint32, P, P, P, P, C.c_uint64]
        f(_p(depth), len(depth), _p(buf), len(calls), C.byref(h), mean_cov, sample_size, min_cnv, _p(pos), _p(baf), _p(pfb), _p(has), len(pos))
        return buf[: len(calls)].copy()

    def split_signatures(self, tid, pos, flag, mapq, ref_end, q_start, q_end, qname_id, min_mapq=20):
        a = [np.ascontiguousarray(x, dt) for x, dt in ((tid, np.int32), (pos, np.int32), (flag, np.uint16), (mapq, np.uint8), (ref_end, np.int32),
                                                        (q_start, np.int32), (q_end, np.int32), (qname_id, np.uint32))]
        dt = np.dtype([("start", "<u4"), ("end", "<u4"), ("sv_type", "<i4"), ("cluster_size", "<i4"), ("aln_offset", "<i4"), ("aln_flags", "<u4"), ("tid", "<i4")])
        n = len(a[0])
        out = np.zeros(4 * n + 16, dt)
        f = self.lib.orc_split_signatures
        f.restype = C.c_int64
        f.argtypes = [C.c_uint64, P, P, P, P, P, P, P, P, C.c_int, P, C.c_uint64]
        k = f(n, *[_p(x) for x in a], min_mapq, _p(out), len(out))
        return out[:k].copy()

    def merge_svs(self, calls, eps, min_pts, keep_noise, label_fn=None):
        """calls: CALL_DTYPE array. label_fn(start,end,eps,min_pts)->labels, default = oracle DBSCAN."""
        calls = np.ascontiguousarray(calls, CALL_DTYPE)
        fn = label_fn or self.dbscan_iv

        def cb(ps, pe, n, e, mp, pl):
            s = np.ctypeslib.as_array(C.cast(ps, C.POINTER(C.c_uint32)), shape=(n,))
            en = np.ctypeslib.as_array(C.cast(pe, C.POINTER(C.c_uint32)), shape=(n,))
            lab = np.ctypeslib.as_array(C.cast(pl, C.POINTER(C.c_int32)), shape=(n,))
            lab[:] = fn(s.copy(), en.copy(), e, mp)
        out = np.zeros(max(len(calls), 1), CALL_DTYPE)
        m = self.lib.orc_merge_svs(_p(calls), len(calls), eps, min_pts, int(keep_noise), LABEL_FN(cb), _p(out))
        return out[:m].copy()

    def merge_duplicates(self, calls):
        calls = np.ascontiguousarray(calls, CALL_DTYPE).copy()
        m = self.lib.orc_merge_duplicates(_p(calls), len(calls))
        return calls[:m].copy()


class Ref:
    def __init__(self, lib):
        self.lib = lib
        lib.ref_dbscan_iv.restype = None
        lib.ref_dbscan_iv.argtypes = [P, P, C.c_uint64, C.c_double, C.c_int32, P]
        lib.ref_dbscan_1d.restype = None
        lib.ref_dbscan_1d.argtypes = [P, C.c_uint64, C.c_double, C.c_int32, P]
        lib.ref_dbscan_1d_largest.restype = C.c_int64
        lib.ref_dbscan_1d_largest.argtypes = [P, C.c_uint64, C.c_double, C.c_int32, P]
        for f in (lib.ref_pdf_normal, lib.ref_cdf_normal):
            f.restype = C.c_double
            f.argtypes = [C.c_double] * 3

    def dbscan_iv(self, start, end, eps, min_pts):
        start = np.ascontiguousarray(start, np.uint32); end = np.ascontiguousarray(end, np.uint32)
        lab = np.zeros(max(len(start), 1), np.int32)
        self.lib.ref_dbscan_iv(_p(start), _p(end), len(start), eps, min_pts, _p(lab))
        return lab[: len(start)]

    def dbscan_1d(self, pts, eps, min_pts):
        pts = np.ascontiguousarray(pts, np.int32)
        lab = np.zeros(max(len(pts), 1), np.int32)
        self.lib.ref_dbscan_1d(_p(pts), len(pts), eps, min_pts, _p(lab))
        return lab[: len(pts)]

    def largest(self, pts, eps, min_pts):
        pts = np.ascontiguousarray(pts, np.int32)
        out = np.zeros(max(len(pts), 1), np.int32)
        m = self.lib.ref_dbscan_1d_largest(_p(pts), len(pts), eps, min_pts, _p(out))
        return out[:m]


def load_oracle() -> Oracle:
    if not os.path.exists(ORACLE_SO):
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle")])
    return Oracle(C.CDLL(ORACLE_SO))


def load_ref_O0():
    """the reference's dbscan.cpp / dbscan1d.cpp / kc.cpp at its SHIPPED flags (Makefile:14 has no -O): oracle/_ref/libcsvref_O0.so"""
    so = os.path.join(ROOT, "oracle", "_ref", "libcsvref_O0.so")
    if not os.path.exists(so):
        if os.path.isdir("/root/reference/src"):
            subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "ref-O0"])
        else:
            return None
    return Ref(C.CDLL(so))


def load_ref():
    if not os.path.exists(REF_SO):
        if os.path.isdir("/root/reference/src"):
            subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "ref"])
        else:
            return None
    return Ref(C.CDLL(REF_SO))
