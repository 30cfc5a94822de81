"""ctypes bindings of the C++ host mirror (csrc/host/ -> lib/libcontextsv_host.so): mergeSVs and friends,
the per-chromosome SVCaller path, the HMM file parser and the synthetic shard generator. Plumbing only."""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

from . import _lib
from .context import Context, Reads, Shard

# CONTEXTSV_HOST_LIB: another build of the host mirror (tests: the AddressSanitizer build, csrc/_obj/libcontextsv_host_asan.so)
HOST_LIB_PATH = os.environ.get("CONTEXTSV_HOST_LIB") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "lib", "libcontextsv_host.so")

CALL_DTYPE = np.dtype([("start", "<u4"), ("end", "<u4"), ("sv_type", "<i4"), ("cluster_size", "<i4"), ("hmm_likelihood", "<f8"),
                       ("id", "<i8"), ("aln_flags", "<u4"), ("genotype", "<i4"), ("cn_state", "<i4"), ("aln_offset", "<i4")])


class chr_stats(C.Structure):
    _fields_ = [("n_signatures", C.c_uint64), ("n_del", C.c_uint64), ("n_ins", C.c_uint64), ("depth_sum", C.c_uint64),
                ("depth_nonzero", C.c_uint32), ("min_pts", C.c_int32), ("mean_cov", C.c_double), ("ms_device", C.c_double),
                ("ms_host_merge", C.c_double), ("n_calls", C.c_uint64)]


class bam_stats(C.Structure):
    _fields_ = [("n_contigs", C.c_uint64), ("n_reads", C.c_uint64), ("n_cigar", C.c_uint64), ("bam_bytes", C.c_uint64),
                ("ms_decode", C.c_double), ("ms_total", C.c_double)]


class stage_times(C.Structure):
    _fields_ = [(k, C.c_double) for k in ("ms_cigar", "ms_cigar_cn", "ms_split_fetch", "ms_split", "ms_split_cn", "ms_merge_split", "ms_merge_final",
                                          "ms_vcf", "ms_total", "ms_split_prepare")] + \
               [(k, C.c_uint64) for k in ("n_reads", "n_signatures", "n_cigar_calls", "n_cigar_cn_regions", "n_split_calls", "n_final_calls")]


_P = C.c_void_p
_hlib = None


def load() -> C.CDLL:
    global _hlib
    if _hlib is not None:
        return _hlib
    _lib.load()          # libcsvgpu.so first (the host mirror links against it)
    if not os.path.exists(HOST_LIB_PATH):
        raise RuntimeError(f"host mirror not built: {HOST_LIB_PATH} is missing — run __graft_entry__.build()")
    lib = C.CDLL(HOST_LIB_PATH)
    lib.csvhost_last_error.restype = C.c_char_p
    lib.csvhost_set_context.argtypes = [_P]
    lib.csvhost_set_quiet.argtypes = [C.c_int]
    lib.csvhost_merge_svs.argtypes = [_P, C.c_uint64, C.c_double, C.c_int32, C.c_int, _P, C.POINTER(C.c_uint64)]
    lib.csvhost_merge_type_with_labels.argtypes = [_P, _P, C.c_uint64, C.c_int, _P, C.POINTER(C.c_uint64)]
    lib.csvhost_merge_duplicates.argtypes = [_P, C.c_uint64, C.POINTER(C.c_uint64)]
    lib.csvhost_add_sv_calls.argtypes = [_P, C.c_uint64, _P, C.POINTER(C.c_uint64)]
    lib.csvhost_synth_generate.restype = _P
    lib.csvhost_synth_generate.argtypes = [C.c_uint64, C.c_uint32, C.c_double, C.c_int, C.c_int, C.c_int]
    lib.csvhost_synth_view.argtypes = [_P, C.POINTER(_lib.csv_reads), C.POINTER(C.c_uint32), C.POINTER(_P), C.POINTER(_P)]
    lib.csvhost_synth_free.argtypes = [_P]
    lib.csvhost_process_resident_chromosome.argtypes = [_P, _P, _P, _P, C.c_double, C.c_double, _P, _P, C.c_uint64, C.POINTER(chr_stats)]
    lib.csvhost_process_resident_pipelined.argtypes = [_P, _P, C.c_uint64, _P, _P, C.c_double, C.c_double, _P, _P, C.c_uint64,
                                                       C.POINTER(chr_stats), C.POINTER(C.c_double), C.POINTER(C.c_uint64)]
    lib.csvhost_query_snp_region.argtypes = [_P, _P, C.c_uint32, C.c_uint32, C.c_double, C.c_int, _P, _P, _P, _P, C.c_uint64,
                                             _P, _P, _P, _P, _P, C.c_uint64, C.POINTER(C.c_uint64)]
    lib.csvhost_cn_prediction.argtypes = [_P, _P, C.c_int, _P, C.c_uint64, C.c_uint64, C.POINTER(C.c_uint64), C.POINTER(_lib.csv_hmm),
                                          C.c_double, C.c_int, C.c_uint32, _P, _P, _P, _P, C.c_uint64]
    lib.csvhost_split_signatures.argtypes = [_P, C.c_uint64, _P, _P, _P, _P, _P, _P, _P, _P, C.c_int, C.c_int, _P, C.c_uint64, C.POINTER(C.c_uint64)]
    lib.csvhost_run.argtypes = [_P, C.c_int, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, C.POINTER(_lib.csv_hmm), C.c_double, C.c_double,
                                C.c_int, C.c_uint32, _P, _P, C.c_uint64, C.POINTER(C.c_uint64), _P, C.c_char_p, C.c_char_p, C.c_char_p, _P, C.c_uint64, _P, C.c_int]
    lib.csvhost_read_chmm.argtypes = [C.c_char_p, C.POINTER(_lib.csv_hmm), C.POINTER(C.c_int32)]
    lib.csvhost_sort_select_check.argtypes = [_P, C.c_uint64, C.c_uint64, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
    lib.csvhost_partition_check.argtypes = [_P, C.c_uint64, C.POINTER(C.c_int)]
    lib.csvhost_fasta_open.restype = _P
    lib.csvhost_fasta_open.argtypes = [C.c_char_p]
    lib.csvhost_fasta_free.argtypes = [_P]
    lib.csvhost_fasta_query.restype = C.c_int64
    lib.csvhost_fasta_query.argtypes = [_P, C.c_char_p, C.c_uint32, C.c_uint32, _P, C.c_uint64]
    lib.csvhost_fasta_compare.argtypes = [_P, C.c_char_p, C.c_uint32, C.c_uint32, C.c_char_p, C.c_float]
    lib.csvhost_fasta_length.restype = C.c_uint32
    lib.csvhost_fasta_length.argtypes = [_P, C.c_char_p]
    lib.csvhost_fasta_contig_header.restype = C.c_int64
    lib.csvhost_fasta_contig_header.argtypes = [_P, _P, C.c_uint64]
    lib.csvhost_fasta_chromosomes.restype = C.c_int64
    lib.csvhost_fasta_chromosomes.argtypes = [_P, _P, C.c_uint64]
    lib.csvhost_save_vcf.argtypes = [_P, C.c_char_p, _P, C.c_char_p, C.c_char_p, C.c_int, _P, _P, _P, _P, _P, _P, _P, C.c_int, _P]
    lib.csvhost_bam_open.restype = _P
    lib.csvhost_bam_open.argtypes = [C.c_char_p, C.c_int]
    lib.csvhost_bam_close.argtypes = [_P]
    lib.csvhost_bam_n_ref.argtypes = [_P]
    lib.csvhost_bam_names.restype = C.c_char_p
    lib.csvhost_bam_names.argtypes = [_P]
    lib.csvhost_bam_text.restype = C.c_char_p
    lib.csvhost_bam_text.argtypes = [_P]
    lib.csvhost_bam_ref_len.restype = C.c_uint32
    lib.csvhost_bam_ref_len.argtypes = [_P, C.c_int]
    lib.csvhost_bam_read.argtypes = [_P, C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_uint32, C.POINTER(C.c_uint64)]
    lib.csvhost_bam_shard.argtypes = [_P, C.c_int, C.POINTER(_lib.csv_reads), C.POINTER(C.c_int32), C.POINTER(C.c_uint32), C.POINTER(_P), C.POINTER(_P)]
    lib.csvhost_bam_shard_qnames.restype = C.c_int64
    lib.csvhost_bam_shard_qnames.argtypes = [_P, C.c_int, _P, C.c_uint64]
    lib.csvhost_bam_write.argtypes = [C.c_char_p, C.c_char_p, C.c_int, C.c_char_p, _P, C.c_uint64, _P, _P, _P, _P, _P, _P, C.c_char_p, _P, _P, _P,
                                      C.c_int, C.c_int]
    lib.csvhost_synth_write_bam.argtypes = [_P, C.c_char_p, C.c_char_p, C.c_int, C.c_int, C.POINTER(C.c_uint64)]
    lib.csvhost_run_bam.argtypes = [_P, C.c_char_p, C.c_char_p, C.c_int, C.POINTER(_lib.csv_hmm), C.c_double, C.c_double, C.c_int, C.c_uint32, C.c_int,
                                    _P, C.c_char_p, C.c_char_p, C.c_char_p, _P, _P, C.c_uint64, C.POINTER(C.c_uint64), C.POINTER(bam_stats),
                                    C.c_char_p, C.c_char_p, C.c_char_p]
    lib.csvhost_snp_open.restype = _P
    lib.csvhost_snp_open.argtypes = [C.c_char_p, C.c_int]
    lib.csvhost_snp_free.argtypes = [_P]
    lib.csvhost_snp_kept.restype = C.c_uint64
    lib.csvhost_snp_kept.argtypes = [_P]
    lib.csvhost_snp_query.restype = C.c_int64
    lib.csvhost_snp_query.argtypes = [_P, C.c_char_p, C.c_char_p, C.c_char_p, C.c_uint32, C.c_uint32, _P, _P, C.c_uint64, C.POINTER(C.c_int),
                                      C.POINTER(C.c_uint32), C.POINTER(C.c_double), C.c_int]
    lib.csvhost_pfb_path.restype = C.c_int64
    lib.csvhost_pfb_path.argtypes = [C.c_char_p, C.c_char_p, _P, C.c_uint64]
    lib.csvhost_gnomad_contig.restype = C.c_int64
    lib.csvhost_gnomad_contig.argtypes = [C.c_char_p, C.c_char_p, _P, C.c_uint64]
    lib.csvhost_process_resident_lanes.argtypes = [C.c_int, _P, _P, _P, C.c_double, C.c_double, _P, C.c_uint64, C.POINTER(chr_stats),
                                                   C.POINTER(C.c_double), C.POINTER(C.c_uint64)]
    lib.csvhost_bam_writer_open.restype = _P
    lib.csvhost_bam_writer_open.argtypes = [C.c_char_p, C.c_int, C.c_char_p, _P, C.c_int, C.c_int]
    lib.csvhost_bam_writer_append_synth.argtypes = [_P, _P, C.c_int]
    lib.csvhost_bam_writer_close.argtypes = [_P]
    lib.csvhost_umap_order_check.argtypes = [C.c_char_p, C.c_uint64, _P, _P, _P, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64),
                                             C.POINTER(C.c_uint64)]
    lib.csvhost_synth_qname_ids.restype = _P
    lib.csvhost_synth_qname_ids.argtypes = [_P]
    lib.csvhost_genome_create.restype = _P
    lib.csvhost_genome_free.argtypes = [_P]
    lib.csvhost_genome_add.argtypes = [_P, _P, C.c_char_p, C.c_int32, C.POINTER(_lib.csv_reads), C.c_uint32, _P, C.c_int, _P, _P, _P, _P, C.c_uint64]
    lib.csvhost_synth_generate2.restype = _P
    lib.csvhost_synth_generate2.argtypes = [C.c_uint64, C.c_uint32, C.c_double, C.c_int, C.c_int, C.c_int, C.c_double]
    lib.csvhost_genome_add_synth.argtypes = [_P, _P, C.c_char_p, C.c_int32, _P, C.c_uint64, C.c_int]
    lib.csvhost_genome_n_contigs.restype = C.c_uint64
    lib.csvhost_genome_n_contigs.argtypes = [_P]
    lib.csvhost_genome_contig_info.argtypes = [_P, C.c_uint64, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_uint32), C.POINTER(C.c_int32), C.POINTER(_P)]
    lib.csvhost_genome_run.argtypes = [_P, _P, C.c_int, _P, C.POINTER(_lib.csv_hmm), C.c_double, C.c_double, C.c_int, C.c_uint32, C.c_int, C.c_int, _P, _P,
                                       C.c_uint64, C.POINTER(C.c_uint64), C.POINTER(stage_times), _P]
    lib.csvhost_sig_alts.argtypes = [_P, C.c_uint64, _P, _P, _P, C.c_uint64, C.POINTER(C.c_uint64)]
    lib.csvhost_process_resident_chromosome_alts.argtypes = [_P, _P, _P, _P, C.c_double, C.c_double, _P, C.c_uint64, C.POINTER(C.c_uint64)]
    lib.csvhost_par_selftest.argtypes = [C.c_int, C.c_int]
    lib.csvhost_string_hashes.argtypes = [C.c_char_p, C.c_uint64, _P]
    lib.csvhost_assign_shards.argtypes = [_P, C.c_uint64, C.c_int, _P]
    lib.csvhost_set_quiet(1)
    _hlib = lib
    return lib


def _check(rc):
    if rc != 0:
        raise RuntimeError((load().csvhost_last_error() or b"host error").decode())


def set_context(ctx: Context):
    load().csvhost_set_context(ctx.h)


def make_calls(start, end, sv_type, cluster_size=None, hmm_likelihood=None) -> np.ndarray:
    n = len(start)
    c = np.zeros(n, CALL_DTYPE)
    c["start"], c["end"], c["sv_type"] = start, end, sv_type
    c["cluster_size"] = 0 if cluster_size is None else cluster_size
    c["hmm_likelihood"] = 0.0 if hmm_likelihood is None else hmm_likelihood
    c["id"] = np.arange(n)
    c["genotype"] = 3
    return c


def merge_svs(calls: np.ndarray, eps: float, min_pts: int, keep_noise: bool) -> np.ndarray:
    """mergeSVs (host mirror; DBSCAN labels from the GPU). Needs set_context()."""
    calls = np.ascontiguousarray(calls, CALL_DTYPE)
    out = np.zeros(max(len(calls), 1), CALL_DTYPE)
    n = C.c_uint64(0)
    _check(load().csvhost_merge_svs(calls.ctypes.data, len(calls), eps, min_pts, int(keep_noise), out.ctypes.data, C.byref(n)))
    return out[: n.value].copy()


def merge_type_with_labels(calls: np.ndarray, labels: np.ndarray, keep_noise: bool) -> np.ndarray:
    calls = np.ascontiguousarray(calls, CALL_DTYPE)
    labels = np.ascontiguousarray(labels, np.int32)
    out = np.zeros(max(len(calls), 1), CALL_DTYPE)
    n = C.c_uint64(0)
    _check(load().csvhost_merge_type_with_labels(calls.ctypes.data, labels.ctypes.data, len(calls), int(keep_noise), out.ctypes.data, C.byref(n)))
    return out[: n.value].copy()


def merge_duplicates(calls: np.ndarray) -> np.ndarray:
    calls = np.ascontiguousarray(calls, CALL_DTYPE).copy()
    n = C.c_uint64(0)
    _check(load().csvhost_merge_duplicates(calls.ctypes.data, len(calls), C.byref(n)))
    return calls[: n.value].copy()


def add_sv_calls_order(calls: np.ndarray) -> np.ndarray:
    calls = np.ascontiguousarray(calls, CALL_DTYPE)
    ids = np.zeros(max(len(calls), 1), np.int64)
    n = C.c_uint64(0)
    _check(load().csvhost_add_sv_calls(calls.ctypes.data, len(calls), ids.ctypes.data, C.byref(n)))
    return ids[: n.value].copy()


class SynthShard:
    """A generated shard living in the host library's memory; `reads` are zero-copy numpy views."""

    def __init__(self, seed: int, chr_len: int, depth: float, tech: int = 0, threads: int = 8, with_seq: bool = False, sv_per_bp: float = 0.0):
        lib = load()
        self.h = lib.csvhost_synth_generate2(seed, chr_len, depth, tech, threads, int(with_seq), sv_per_bp)
        if not self.h:
            raise RuntimeError((lib.csvhost_last_error() or b"").decode())
        r = _lib.csv_reads()
        dl = C.c_uint32(0)
        so, sq = _P(), _P()
        lib.csvhost_synth_view(self.h, C.byref(r), C.byref(dl), C.byref(so), C.byref(sq))
        n, m = r.n_reads, r.n_cigar

        def view(p, count, dtype):
            if not p or count == 0:
                return np.zeros(0, dtype)
            buf = (C.c_char * (count * np.dtype(dtype).itemsize)).from_address(p)
            return np.frombuffer(buf, dtype=dtype)

        self.depth_len = dl.value
        self.reads = Reads.__new__(Reads)
        self.reads.pos = view(r.pos, n, np.int32)
        self.reads.flag = view(r.flag, n, np.uint16)
        self.reads.mapq = view(r.mapq, n, np.uint8)
        self.reads.tid = view(r.tid, n, np.int32)
        self.reads.cigar_off = view(r.cigar_off, n + 1, np.uint64)
        self.reads.cigar = view(r.cigar, m, np.uint32)
        self.seq_off_ptr, self.seq_ptr = so.value, sq.value
        self.qname_id = view(lib.csvhost_synth_qname_ids(self.h), n, np.uint32)

    def write_bam(self, path: str, chr_name: str = "chr22", level: int = 1, threads: int = 8) -> int:
        """The shard as a coordinate-sorted BAM + BAI on one contig; returns the BAM's size in bytes."""
        nbytes = C.c_uint64(0)
        _check(load().csvhost_synth_write_bam(self.h, os.fsencode(path), chr_name.encode(), level, threads, C.byref(nbytes)))
        return nbytes.value

    def free(self):
        if self.h:
            self.reads = None
            load().csvhost_synth_free(self.h)
            self.h = None


class Genome:
    """A staged genome: contigs resident in HBM + the small host arrays of the host-side passes; run() = SVCaller::runResident
    (one step of the whole-genome benchmark). Contigs are added through a Context (they may use different contexts of one GPU)."""

    def __init__(self):
        self.h = load().csvhost_genome_create()
        self.names = []

    def add(self, ctx: Context, name: str, global_tid: int, reads: Reads, depth_len: int, qname_id=None, snps=None, name_style: int = 0):
        rs = reads.c_struct()
        q = np.ascontiguousarray(qname_id, np.uint32) if qname_id is not None else None
        sp = np.ascontiguousarray(snps["pos"], np.uint32) if snps is not None else np.zeros(0, np.uint32)
        sb = np.ascontiguousarray(snps["baf"], np.float64) if snps is not None else np.zeros(0, np.float64)
        sf = np.ascontiguousarray(snps["pfb"], np.float64) if snps is not None and "pfb" in snps else None
        sh = np.ascontiguousarray(snps["has_pfb"], np.uint8) if snps is not None and "has_pfb" in snps else None
        _check(load().csvhost_genome_add(self.h, ctx.h, name.encode(), global_tid, C.byref(rs), depth_len, q.ctypes.data if q is not None else None,
                                         name_style, sp.ctypes.data, sb.ctypes.data, sf.ctypes.data if sf is not None else None,
                                         sh.ctypes.data if sh is not None else None, len(sp)))
        self.names.append(name)

    def add_synth(self, ctx: Context, name: str, global_tid: int, syn: "SynthShard", snp_seed: int = 0, with_snps: bool = True):
        _check(load().csvhost_genome_add_synth(self.h, ctx.h, name.encode(), global_tid, syn.h, snp_seed, int(with_snps)))
        self.names.append(name)

    def __len__(self):
        return int(load().csvhost_genome_n_contigs(self.h))

    def contig_info(self, i: int) -> dict:
        nr, nc, dl, gt, sh = C.c_uint64(0), C.c_uint64(0), C.c_uint32(0), C.c_int32(0), _P()
        load().csvhost_genome_contig_info(self.h, i, C.byref(nr), C.byref(nc), C.byref(dl), C.byref(gt), C.byref(sh))
        return {"n_reads": nr.value, "n_cigar": nc.value, "depth_len": dl.value, "global_tid": gt.value, "shard": sh.value}

    def run(self, ctx: Context, hmm, lanes=None, eps=0.1, min_pts_pct=0.1, sample_size=20, min_cnv=2000, split_svs=True, cigar_cn=True, merges=True,
            host_threads=0, capacity: int = 1 << 20, host_split_order: bool = False, overlap_split: bool = True, copy: bool = True):
        """-> (calls[CALL_DTYPE], global tid per call, stage_times, per-contig chr_stats list). copy=False: the two arrays are views of buffers
        the genome owns (two sets, used alternately) and stay valid until the run after the next one."""
        n = len(self)
        if getattr(self, "_cap", 0) < capacity:              # result buffers live with the genome (tens of megabytes of page faults per call otherwise)
            self._bufs = [(np.empty(capacity, CALL_DTYPE), np.empty(capacity, np.int32)) for _ in range(2)]
            self._cap, self._flip = capacity, 0
        self._flip ^= 1
        out, tid = self._bufs[self._flip]
        k = C.c_uint64(0)
        st = stage_times()
        cs = (chr_stats * max(n, 1))()
        lanes = lanes or []
        lp = (C.c_void_p * max(len(lanes), 1))(*[c.h for c in lanes])
        _check(load().csvhost_genome_run(self.h, ctx.h, len(lanes), lp, C.byref(hmm), eps, min_pts_pct, sample_size, min_cnv,
                                         int(split_svs) | (int(cigar_cn) << 1) | (int(merges) << 2) | (int(host_split_order) << 3) | (int(not overlap_split) << 4), host_threads, out.ctypes.data, tid.ctypes.data, capacity,
                                         C.byref(k), C.byref(st), cs))
        if k.value > capacity:
            raise RuntimeError("Genome.run: capacity too small")
        if copy:
            return out[: k.value].copy(), tid[: k.value].copy(), st, list(cs)[:n]
        return out[: k.value], tid[: k.value], st, list(cs)[:n]

    def free(self):
        if self.h:
            load().csvhost_genome_free(self.h)
            self.h = None


def assign_shards(weights, world: int) -> list:
    """SVCaller::assignShards (the C++ mirror of parallel.assign_shards): the shard indices of every rank, ascending."""
    w = np.ascontiguousarray(weights, np.float64)
    r = np.zeros(max(len(w), 1), np.int32)
    _check(load().csvhost_assign_shards(w.ctypes.data, len(w), world, r.ctypes.data))
    return [[int(i) for i in np.nonzero(r[: len(w)] == k)[0]] for k in range(world)]


def string_hashes(names) -> np.ndarray:
    """libstdc++ std::hash<std::string> of every name (uint64)."""
    out = np.zeros(max(len(names), 1), np.uint64)
    _check(load().csvhost_string_hashes("\n".join(names).encode(), len(names), out.ctypes.data))
    return out[: len(names)]


def umap_order_check(keys, erase_mask=None):
    """(order of a real std::unordered_map<std::string,int>, order from the replay in umap_order.h, bucket counts) — CPU only."""
    n = len(keys)
    blob = "\n".join(keys).encode()
    a, b = np.zeros(max(n, 1), np.int64), np.zeros(max(n, 1), np.int64)
    na, nb, ba, bb = C.c_uint64(0), C.c_uint64(0), C.c_uint64(0), C.c_uint64(0)
    em = np.ascontiguousarray(erase_mask, np.uint8) if erase_mask is not None else None
    _check(load().csvhost_umap_order_check(blob, n, em.ctypes.data if em is not None else None, a.ctypes.data, b.ctypes.data, C.byref(na), C.byref(nb),
                                           C.byref(ba), C.byref(bb)))
    return a[: na.value].copy(), b[: nb.value].copy(), (ba.value, bb.value)


def sig_alts(sig: np.ndarray, seq_off=None, seq=None):
    """ALT string of every signature (SVCaller::toSVCall; host only). seq_off: uint64[n_reads + 1] byte offsets, seq: uint8 4-bit packed."""
    sig = np.ascontiguousarray(sig, _lib.SIG_DTYPE)
    so = np.ascontiguousarray(seq_off, np.uint64) if seq_off is not None else None
    sq = np.ascontiguousarray(seq, np.uint8) if seq is not None else None
    cap = 64 * len(sig) + 16
    buf = C.create_string_buffer(cap)
    ln = C.c_uint64(0)
    _check(load().csvhost_sig_alts(sig.ctypes.data, len(sig), so.ctypes.data if so is not None else None, sq.ctypes.data if sq is not None else None,
                                   buf, cap, C.byref(ln)))
    return buf.raw[: ln.value].decode().split("\n")[:-1]


def process_resident_chromosome_alts(ctx: Context, shard: Shard, eps: float, min_pts_pct: float, seq_off=None, seq=None):
    """processChromosome on a resident shard -> [(start, end, ALT)] of the merged calls."""
    so = np.ascontiguousarray(seq_off, np.uint64) if seq_off is not None else None
    sq = np.ascontiguousarray(seq, np.uint8) if seq is not None else None
    cap = 1 << 22
    buf = C.create_string_buffer(cap)
    ln = C.c_uint64(0)
    _check(load().csvhost_process_resident_chromosome_alts(ctx.h, shard.h, so.ctypes.data if so is not None else None,
                                                           sq.ctypes.data if sq is not None else None, eps, min_pts_pct, buf, cap, C.byref(ln)))
    out = []
    for line in buf.raw[: ln.value].decode().split("\n")[:-1]:
        a, b, alt = line.split("\t")
        out.append((int(a), int(b), alt))
    return out


def process_resident_chromosome(ctx: Context, shard: Shard, eps: float, min_pts_pct: float, seq_off_ptr=None, seq_ptr=None, capacity: int = 1 << 20):
    """SVCaller::processChromosome mirror on a resident shard -> (merged calls, alt tags, stats)."""
    out = np.zeros(capacity, CALL_DTYPE)
    tag = np.zeros(capacity, np.uint8)
    st = chr_stats()
    _check(load().csvhost_process_resident_chromosome(ctx.h, shard.h, seq_off_ptr, seq_ptr, eps, min_pts_pct, out.ctypes.data, tag.ctypes.data,
                                                      capacity, C.byref(st)))
    n = min(st.n_calls, capacity)
    return out[:n].copy(), tag[:n].copy(), st


def process_resident_pipelined(ctx: Context, shard: Shard, n_steps: int, eps: float, min_pts_pct: float, seq_off_ptr=None, seq_ptr=None,
                               capacity: int = 1 << 20):
    """n_steps pipelined passes over one resident shard (device chain of step i+1 overlaps the host merge of step i).
    -> (merged calls of the last step, alt tags, stats averaged over steps, wall ms, total merged calls)."""
    out = np.zeros(capacity, CALL_DTYPE)
    tag = np.zeros(capacity, np.uint8)
    st = chr_stats()
    ms, tot = C.c_double(0), C.c_uint64(0)
    _check(load().csvhost_process_resident_pipelined(ctx.h, shard.h, n_steps, seq_off_ptr, seq_ptr, eps, min_pts_pct, out.ctypes.data,
                                                     tag.ctypes.data, capacity, C.byref(st), C.byref(ms), C.byref(tot)))
    n = min(st.n_calls, capacity)
    return out[:n].copy(), tag[:n].copy(), st, ms.value, tot.value


def process_resident_lanes(ctxs, shards, steps, eps: float, min_pts_pct: float, capacity: int = 1 << 20):
    """Several contexts of one GPU, lane l doing steps[l] pipelined passes over shards[l] at the same time (attach one Gate to all
    contexts first). -> (lane 0's last merged calls, stats averaged over all steps, wall ms, total merged calls)."""
    n = len(ctxs)
    cp = (C.c_void_p * n)(*[c.h for c in ctxs])
    sp = (C.c_void_p * n)(*[s.h for s in shards])
    st_n = np.asarray(steps, np.uint64)
    out = np.zeros(capacity, CALL_DTYPE)
    st = chr_stats()
    ms, tot = C.c_double(0), C.c_uint64(0)
    _check(load().csvhost_process_resident_lanes(n, cp, sp, st_n.ctypes.data, eps, min_pts_pct, out.ctypes.data, capacity, C.byref(st), C.byref(ms),
                                                 C.byref(tot)))
    return out[: min(st.n_calls, capacity)].copy(), st, ms.value, tot.value


def _snp_arrays(snps):
    pos = np.ascontiguousarray(snps["pos"], np.uint32)
    baf = np.ascontiguousarray(snps["baf"], np.float64)
    pfb = np.ascontiguousarray(snps["pfb"], np.float64)
    has = np.ascontiguousarray(snps["has_pfb"], np.uint8)
    return pos, baf, pfb, has


def query_snp_region(ctx: Context, shard: Shard, start: int, end: int, mean_cov: float, sample_size: int, snps: dict, cap: int = 1 << 16):
    """CNVCaller::querySNPRegion mirror on the shard's resident depth map -> dict of observation arrays."""
    pos, baf, pfb, has = _snp_arrays(snps)
    o_pos = np.zeros(cap, np.uint32); o_baf = np.zeros(cap); o_pfb = np.zeros(cap); o_l2 = np.zeros(cap); o_is = np.zeros(cap, np.uint8)
    n = C.c_uint64(0)
    _check(load().csvhost_query_snp_region(ctx.h, shard.h, start, end, mean_cov, sample_size, pos.ctypes.data, baf.ctypes.data, pfb.ctypes.data,
                                           has.ctypes.data, len(pos), o_pos.ctypes.data, o_baf.ctypes.data, o_pfb.ctypes.data, o_l2.ctypes.data,
                                           o_is.ctypes.data, cap, C.byref(n)))
    k = n.value
    return {"pos": o_pos[:k], "baf": o_baf[:k], "pfb": o_pfb[:k], "log2_cov": o_l2[:k], "is_snp": o_is[:k].astype(bool)}


def cn_prediction(ctx: Context, shard: Shard, calls: np.ndarray, hmm, mean_cov: float, snps: dict, split: bool, sample_size: int = 20,
                  min_cnv: int = 2000) -> np.ndarray:
    """runCIGARCopyNumberPrediction (split=False) / runSplitReadCopyNumberPredictions (split=True) mirror."""
    pos, baf, pfb, has = _snp_arrays(snps)
    cap = 2 * len(calls) + 16
    buf = np.zeros(cap, CALL_DTYPE)
    buf[: len(calls)] = np.ascontiguousarray(calls, CALL_DTYPE)
    n = C.c_uint64(0)
    _check(load().csvhost_cn_prediction(ctx.h, shard.h, int(split), buf.ctypes.data, len(calls), cap, C.byref(n), C.byref(hmm), mean_cov,
                                        sample_size, min_cnv, pos.ctypes.data, baf.ctypes.data, pfb.ctypes.data, has.ctypes.data, len(pos)))
    return buf[: n.value].copy()


SPLIT_CALL_DTYPE = np.dtype([("start", "<u4"), ("end", "<u4"), ("sv_type", "<i4"), ("cluster_size", "<i4"), ("aln_offset", "<i4"),
                             ("aln_flags", "<u4"), ("tid", "<i4")])


def split_signatures(ctx: Context, tid, pos, flag, mapq, ref_end, q_start, q_end, qname_id, n_targets: int, min_mapq: int = 20) -> np.ndarray:
    """findSplitSVSignatures mirror (qname of record i = "r<qname_id[i]>") -> calls sorted by contig id."""
    a = [np.ascontiguousarray(x, dt) for x, dt in ((tid, np.int32), (pos, np.int32), (flag, np.uint16), (mapq, np.uint8), (ref_end, np.int32),
                                                    (q_start, np.int32), (q_end, np.int32), (qname_id, np.uint32))]
    n = len(a[0])
    cap = 4 * n + 16
    out = np.zeros(cap, SPLIT_CALL_DTYPE)
    k = C.c_uint64(0)
    _check(load().csvhost_split_signatures(ctx.h, n, *[x.ctypes.data for x in a], n_targets, min_mapq, out.ctypes.data, cap, C.byref(k)))
    return out[: k.value].copy()


def run(ctx: Context, contigs: list, hmm, eps=0.1, min_pts_pct=0.1, sample_size=20, min_cnv=2000, genome=None, vcf_dir=None,
        gap_path=None, file_date=None, want_alts=False, save_cnv=False):
    """SVCaller::run mirror. contigs: list of dicts {reads: Reads, depth_len, qname_id: uint32[n], snps: dict}; contig t is named
    "contig<t>". With genome (ReferenceGenome) and vcf_dir the run ends by writing <vcf_dir>/output.vcf.
    -> (merged calls, contig id per call[, ALT strings])."""
    read_off = np.zeros(len(contigs) + 1, np.uint64)
    read_off[1:] = np.cumsum([c["reads"].n_reads for c in contigs])
    cig_base = np.concatenate([[0], np.cumsum([c["reads"].n_cigar for c in contigs])]).astype(np.uint64)
    pos = np.ascontiguousarray(np.concatenate([c["reads"].pos for c in contigs]), np.int32)
    flag = np.ascontiguousarray(np.concatenate([c["reads"].flag for c in contigs]), np.uint16)
    mapq = np.ascontiguousarray(np.concatenate([c["reads"].mapq for c in contigs]), np.uint8)
    cigar = np.ascontiguousarray(np.concatenate([c["reads"].cigar for c in contigs]), np.uint32)
    coff = np.ascontiguousarray(np.concatenate([c["reads"].cigar_off[:-1] + cig_base[i] for i, c in enumerate(contigs)] + [cig_base[-1:]]), np.uint64)
    qid = np.ascontiguousarray(np.concatenate([c["qname_id"] for c in contigs]), np.uint32)
    dl = np.asarray([c["depth_len"] for c in contigs], np.uint32)
    snp_off = np.zeros(len(contigs) + 1, np.uint64)
    snp_off[1:] = np.cumsum([len(c["snps"]["pos"]) for c in contigs])
    sp = np.ascontiguousarray(np.concatenate([c["snps"]["pos"] for c in contigs]), np.uint32)
    sb = np.ascontiguousarray(np.concatenate([c["snps"]["baf"] for c in contigs]), np.float64)
    sf = np.ascontiguousarray(np.concatenate([c["snps"]["pfb"] for c in contigs]), np.float64)
    sh = np.ascontiguousarray(np.concatenate([c["snps"]["has_pfb"] for c in contigs]), np.uint8)
    cap = len(cigar) + 1024
    out = np.zeros(cap, CALL_DTYPE)
    tid = np.zeros(cap, np.int32)
    n = C.c_uint64(0)
    alt_cap = 64 * cap if want_alts else 0
    alt_buf = np.zeros(max(alt_cap, 1), np.uint8)
    alt_off = np.zeros(cap + 1, np.uint64)
    _check(load().csvhost_run(ctx.h, len(contigs), read_off.ctypes.data, dl.ctypes.data, pos.ctypes.data, flag.ctypes.data, mapq.ctypes.data,
                              coff.ctypes.data, cigar.ctypes.data, qid.ctypes.data, snp_off.ctypes.data, sp.ctypes.data, sb.ctypes.data,
                              sf.ctypes.data, sh.ctypes.data, C.byref(hmm), eps, min_pts_pct, sample_size, min_cnv, out.ctypes.data,
                              tid.ctypes.data, cap, C.byref(n),
                              genome.h if genome is not None and vcf_dir else None, os.fsencode(vcf_dir) if vcf_dir else None,
                              os.fsencode(gap_path) if gap_path else None, file_date.encode() if file_date else None,
                              alt_buf.ctypes.data if want_alts else None, alt_cap, alt_off.ctypes.data if want_alts else None, int(save_cnv)))
    if want_alts:
        raw = alt_buf.tobytes()
        alts = [raw[int(alt_off[i]):int(alt_off[i + 1])] for i in range(n.value)]
        return out[: n.value].copy(), tid[: n.value].copy(), alts
    return out[: n.value].copy(), tid[: n.value].copy()


def sort_select_check(keys: np.ndarray, nth: int):
    """(id std::sort leaves at slot nth, id std_sort_select returns) for a descending sort by key."""
    keys = np.ascontiguousarray(keys, np.uint32)
    a, b = C.c_int64(-1), C.c_int64(-1)
    _check(load().csvhost_sort_select_check(keys.ctypes.data, len(keys), nth, C.byref(a), C.byref(b)))
    return a.value, b.value


def partition_check(keys: np.ndarray) -> bool:
    """One partition step of sort_select.h: True when the block-wise form leaves the cut and the arrangement of the library's loop."""
    keys = np.ascontiguousarray(keys, np.uint32)
    d = C.c_int(1)
    _check(load().csvhost_partition_check(keys.ctypes.data, len(keys), C.byref(d)))
    return d.value == 0


def read_chmm(path: str):
    h = _lib.csv_hmm()
    n = C.c_int32(0)
    _check(load().csvhost_read_chmm(path.encode(), C.byref(h), C.byref(n)))
    return h, n.value


class ReferenceGenome:
    """ReferenceGenome of the host mirror (src/fasta_query.cpp): whole FASTA in memory, 1-based inclusive queries."""

    def __init__(self, path: str):
        self.path = path
        self.h = load().csvhost_fasta_open(os.fsencode(path))
        if not self.h:
            raise RuntimeError((load().csvhost_last_error() or b"cannot open FASTA").decode())

    def close(self):
        if self.h:
            load().csvhost_fasta_free(self.h)
        self.h = None

    def __del__(self):
        self.close()

    def _text(self, fn) -> bytes:
        n = fn(self.h, None, 0)
        buf = C.create_string_buffer(max(int(n), 1))
        fn(self.h, buf, n)
        return buf.raw[:n]

    def query(self, chr: str, pos_start: int, pos_end: int) -> bytes:
        """Raises KeyError for an unknown contig (std::out_of_range in the C++ interface)."""
        lib = load()
        n = lib.csvhost_fasta_query(self.h, chr.encode(), pos_start, pos_end, None, 0)
        if n < 0:
            raise KeyError(chr)
        buf = C.create_string_buffer(max(int(n), 1))
        lib.csvhost_fasta_query(self.h, chr.encode(), pos_start, pos_end, buf, n)
        return buf.raw[:n]

    def compare(self, chr: str, pos_start: int, pos_end: int, seq: bytes, threshold: float) -> bool:
        rc = load().csvhost_fasta_compare(self.h, chr.encode(), pos_start, pos_end, seq, threshold)
        if rc < 0:
            raise KeyError(chr)
        return bool(rc)

    def getChromosomeLength(self, chr: str) -> int:
        return int(load().csvhost_fasta_length(self.h, chr.encode()))

    def getContigHeader(self) -> bytes:
        return self._text(load().csvhost_fasta_contig_header)

    def getChromosomes(self):
        t = self._text(load().csvhost_fasta_chromosomes)
        return t.split(b"\n") if t else []


def save_vcf(out_dir: str, genome: ReferenceGenome, contigs, gap_path: str | None = None, file_date: str | None = None,
             ctx: Context | None = None, map_order: bool = False):
    """saveToVCF (host mirror) -> <out_dir>/output.vcf. `contigs` = [(name, calls[CALL_DTYPE], alts[list of bytes], depth)] where depth is a
    resident Shard (SUPPORT/DP gathered on the device; needs ctx), a uint32 numpy array, or None. Returns (total, unclassified, gap_filtered)."""
    lib = load()
    n = len(contigs)
    names = (C.c_char_p * n)(*[c[0].encode() for c in contigs])
    off = np.zeros(n + 1, np.uint64)
    off[1:] = np.cumsum([len(c[1]) for c in contigs])
    calls = np.ascontiguousarray(np.concatenate([np.asarray(c[1], CALL_DTYPE) for c in contigs]) if n else np.zeros(0, CALL_DTYPE))
    alt_list = [a for c in contigs for a in c[2]]
    assert len(alt_list) == len(calls)
    alts = (C.c_char_p * max(len(alt_list), 1))(*alt_list)
    use_shards = n > 0 and all(hasattr(c[3], "h") or c[3] is None for c in contigs) and any(c[3] is not None for c in contigs)
    keep = []
    if use_shards:
        shards = (C.c_void_p * n)(*[(c[3].h if c[3] is not None else None) for c in contigs])
        depth_p, len_p, shard_p = None, None, shards
    else:
        arrs = [(np.ascontiguousarray(c[3], np.uint32) if c[3] is not None else None) for c in contigs]
        keep.append(arrs)
        dptr = (C.c_void_p * max(n, 1))(*[(a.ctypes.data if a is not None else None) for a in arrs])
        dlen = np.array([(len(a) if a is not None else 0) for a in arrs] + [0], np.uint64)
        depth_p, len_p, shard_p = dptr, dlen.ctypes.data, None
    counts = np.zeros(3, np.int32)
    _check(lib.csvhost_save_vcf(ctx.h if ctx is not None else None, os.fsencode(out_dir), genome.h, os.fsencode(gap_path) if gap_path else None,
                                file_date.encode() if file_date else None, n, names, off.ctypes.data, calls.ctypes.data, alts,
                                shard_p, depth_p, len_p, int(map_order), counts.ctypes.data))
    return tuple(int(x) for x in counts)


def _np_from(ptr_, count, dtype):
    if not ptr_ or count == 0:
        return np.zeros(0, dtype)
    buf = (C.c_char * (count * np.dtype(dtype).itemsize)).from_address(ptr_)
    return np.frombuffer(buf, dtype).copy()


class BamFile:
    """BamReader of the host mirror (BGZF/BAM/BAI without htslib). Shards come back as dicts of numpy arrays
    {tid, name, target_len, reads: Reads, seq_off, seq, qnames}."""

    def __init__(self, path: str, load_index: bool = True):
        self.h = load().csvhost_bam_open(os.fsencode(path), int(load_index))
        if not self.h:
            raise RuntimeError((load().csvhost_last_error() or b"cannot open BAM").decode())
        lib = load()
        n = lib.csvhost_bam_n_ref(self.h)
        names = lib.csvhost_bam_names(self.h).decode()
        self.names = names.split("\n") if n else []
        self.lens = [int(lib.csvhost_bam_ref_len(self.h, i)) for i in range(n)]
        self.text = lib.csvhost_bam_text(self.h).decode()

    def close(self):
        if self.h:
            load().csvhost_bam_close(self.h)
        self.h = None

    def __del__(self):
        self.close()

    def _shard(self, i, want_seq, want_qnames):
        lib = load()
        rs, tid, tl, so, sq = _lib.csv_reads(), C.c_int32(0), C.c_uint32(0), C.c_void_p(), C.c_void_p()
        lib.csvhost_bam_shard(self.h, i, C.byref(rs), C.byref(tid), C.byref(tl), C.byref(so), C.byref(sq))
        n, m = rs.n_reads, rs.n_cigar
        addr = lambda p: C.cast(p, C.c_void_p).value
        reads = Reads.__new__(Reads)
        reads.pos, reads.flag, reads.mapq, reads.tid = _np_from(addr(rs.pos), n, np.int32), _np_from(addr(rs.flag), n, np.uint16), _np_from(addr(rs.mapq), n, np.uint8), None
        reads.cigar_off, reads.cigar = _np_from(addr(rs.cigar_off), n + 1, np.uint64), _np_from(addr(rs.cigar), m, np.uint32)
        out = {"tid": tid.value, "name": self.names[tid.value], "target_len": tl.value, "reads": reads}
        if want_seq:
            out["seq_off"] = _np_from(so.value, n + 1, np.uint64)
            out["seq"] = _np_from(sq.value, int(out["seq_off"][-1]), np.uint8)
        if want_qnames:
            ln = lib.csvhost_bam_shard_qnames(self.h, i, None, 0)
            buf = C.create_string_buffer(max(int(ln), 1))
            lib.csvhost_bam_shard_qnames(self.h, i, buf, ln)
            out["qnames"] = buf.raw[:ln].decode(errors="replace").split("\n")[:-1]
        return out

    def read_contig(self, chr: str, want_seq=False, want_qnames=False, threads=8, window_blocks=0):
        k = load().csvhost_bam_read(self.h, chr.encode(), int(want_seq), int(want_qnames), threads, window_blocks, None)
        if k < 0:
            raise RuntimeError((load().csvhost_last_error() or b"BAM read failed").decode())
        return self._shard(0, want_seq, want_qnames)

    def read_all(self, want_seq=False, want_qnames=False, threads=8, window_blocks=0):
        un = C.c_uint64(0)
        k = load().csvhost_bam_read(self.h, None, int(want_seq), int(want_qnames), threads, window_blocks, C.byref(un))
        if k < 0:
            raise RuntimeError((load().csvhost_last_error() or b"BAM read failed").decode())
        return [self._shard(i, want_seq, want_qnames) for i in range(k)], un.value


def write_bam(path: str, ref_names, ref_lens, tid, reads: Reads, qnames, seq_off=None, seq=None, l_seq=None, text="", level=1, threads=4):
    """BamWriter of the host mirror: coordinate-sorted records -> <path> + <path>.bai."""
    n = reads.n_reads
    tid = np.ascontiguousarray(tid, np.int32)
    lens = np.ascontiguousarray(ref_lens, np.uint32)
    pos, flag, mapq = (np.ascontiguousarray(a, t) for a, t in ((reads.pos, np.int32), (reads.flag, np.uint16), (reads.mapq, np.uint8)))
    coff, cig = np.ascontiguousarray(reads.cigar_off, np.uint64), np.ascontiguousarray(reads.cigar, np.uint32)
    assert len(qnames) == n
    so = np.ascontiguousarray(seq_off, np.uint64) if seq_off is not None else None
    sq = np.ascontiguousarray(seq, np.uint8) if seq is not None else None
    ls = np.ascontiguousarray(l_seq, np.int32) if l_seq is not None else None
    p = lambda a: a.ctypes.data if a is not None else None
    _check(load().csvhost_bam_write(os.fsencode(path), text.encode(), len(ref_names), "\n".join(ref_names).encode(), p(lens), n, p(tid), p(pos), p(flag),
                                    p(mapq), p(coff), p(cig), "\n".join(qnames).encode(), p(so), p(sq), p(ls), level, threads))


def run_bam(ctx: Context, bam_path: str, hmm, chromosomes=None, threads=8, eps=0.1, min_pts_pct=0.1, sample_size=20, min_cnv=2000, split_svs=True, cigar_cn=True, save_cnv=False,
            genome: ReferenceGenome | None = None, vcf_dir=None, gap_path=None, file_date=None, capacity: int = 1 << 20,
            snp_vcf=None, pfb_table=None, ethnicity=""):
    """SVCaller::runBam: the whole run fed from a coordinate-sorted, indexed BAM. -> (calls, contig index per call, stats dict)."""
    out = np.zeros(capacity, CALL_DTYPE)
    tid = np.zeros(capacity, np.int32)
    n = C.c_uint64(0)
    st = bam_stats()
    _check(load().csvhost_run_bam(ctx.h, os.fsencode(bam_path), "\n".join(chromosomes).encode() if chromosomes else None, threads, C.byref(hmm), eps,
                                  min_pts_pct, sample_size, min_cnv, int(split_svs) | (int(cigar_cn) << 1) | (int(save_cnv) << 2), genome.h if genome is not None and vcf_dir else None,
                                  os.fsencode(vcf_dir) if vcf_dir else None, os.fsencode(gap_path) if gap_path else None,
                                  file_date.encode() if file_date else None, out.ctypes.data, tid.ctypes.data, capacity, C.byref(n), C.byref(st),
                                  os.fsencode(snp_vcf) if snp_vcf else None, os.fsencode(pfb_table) if pfb_table else None, ethnicity.encode()))
    if n.value > capacity:
        raise RuntimeError("run_bam: capacity too small")
    return out[: n.value].copy(), tid[: n.value].copy(), {f: getattr(st, f) for f, _ in bam_stats._fields_}


class SNPFile:
    """The sample's SNP VCF (.vcf or bgzipped + indexed .vcf.gz) parsed once with the reference's filters; query() gives what
    readSNPAlleleFrequencies returns for a region: (positions in file order, BAF at those positions, (pfb_pos, pfb) or None)."""

    def __init__(self, snp_vcf: str, threads: int = 4):
        self.h = load().csvhost_snp_open(os.fsencode(snp_vcf), threads)
        if not self.h:
            raise RuntimeError((load().csvhost_last_error() or b"cannot load SNP VCF").decode())

    def close(self):
        if self.h:
            load().csvhost_snp_free(self.h)
        self.h = None

    def __del__(self):
        self.close()

    @property
    def records_kept(self) -> int:
        return int(load().csvhost_snp_kept(self.h))

    def query(self, chr: str, start: int, end: int, pfb_vcf: str | None = None, ethnicity: str = "", threads: int = 4, cap: int = 1 << 20):
        pos, baf = np.zeros(cap, np.uint32), np.zeros(cap, np.float64)
        has, pp, pv = C.c_int(0), C.c_uint32(0), C.c_double(0)
        n = load().csvhost_snp_query(self.h, chr.encode(), os.fsencode(pfb_vcf) if pfb_vcf else None, ethnicity.encode(), start, end,
                                     pos.ctypes.data, baf.ctypes.data, cap, C.byref(has), C.byref(pp), C.byref(pv), threads)
        if n < 0:
            raise RuntimeError("SNP query: capacity too small")
        return pos[:n].copy(), baf[:n].copy(), ((pp.value, pv.value) if has.value else None)


def pfb_path(table_path: str, chr: str) -> str:
    """InputData::getAlleleFreqFilepath over a --pfb table file."""
    buf = C.create_string_buffer(4096)
    n = load().csvhost_pfb_path(os.fsencode(table_path), chr.encode(), buf, 4096)
    if n < 0:
        raise RuntimeError((load().csvhost_last_error() or b"").decode())
    return buf.raw[:n].decode()


def gnomad_contig(chr: str, pfb_path_: str) -> str:
    buf = C.create_string_buffer(4096)
    n = load().csvhost_gnomad_contig(chr.encode(), pfb_path_.encode(), buf, 4096)
    return buf.raw[:n].decode()


class SynthBamWriter:
    """Several SynthShards into one coordinate-sorted BAM + BAI, one contig each (append in contig order)."""

    def __init__(self, path: str, ref_names, ref_lens, level: int = 1, threads: int = 8):
        lens = np.ascontiguousarray(ref_lens, np.uint32)
        self.h = load().csvhost_bam_writer_open(os.fsencode(path), len(ref_names), "\n".join(ref_names).encode(), lens.ctypes.data, level, threads)
        if not self.h:
            raise RuntimeError((load().csvhost_last_error() or b"cannot create BAM").decode())

    def append(self, syn: SynthShard, tid: int):
        _check(load().csvhost_bam_writer_append_synth(self.h, syn.h, tid))

    def close(self):
        if self.h:
            h, self.h = self.h, None
            _check(load().csvhost_bam_writer_close(h))
