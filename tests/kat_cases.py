"""Known-answer CIGAR cases shared by the oracle and the GPU tests. Expected values were derived by hand
from the reference source (sv_caller.cpp:526, 539-661, 663-690; sv_object.cpp:22-33; cnv_caller.cpp:488-543).
expect = list of (start, end, read, qpos, kind) in chr_sv_calls order; kind 0 CIGARINS, 1 CIGARDEL, 2 CIGARCLIP."""
from contextsv_amd import Reads

M, I, D, N, S, H, P, EQ, X = range(9)
_C1 = [(S, 10), (M, 20), (I, 60), (M, 5), (D, 70), (M, 10), (I, 49), (D, 50), (M, 3), (S, 55)]

KAT_SCAN = {
    # every emitting op kind, thresholds (49 < 50 <= 50), cursor rules ④, coordinates ①②③
    "basic": dict(depth_len=1001, reads=[(99, 0, 60, _C1)],
                  expect=[(120, 179, 0, 30, 0), (125, 194, 0, 95, 1), (205, 254, 0, 154, 1), (258, 312, 0, 157, 2)],
                  intervals=[(257, 10, 212)]),
    # the filter of sv_caller.cpp:526: mapq < 20, secondary, unmapped, dup, qcfail, supplementary are all skipped
    "filters": dict(depth_len=1001, reads=[(99, 0, 19, _C1), (99, 0x100, 60, _C1), (99, 0x4, 60, _C1), (99, 0x400, 60, _C1),
                                           (99, 0x200, 60, _C1), (99, 0x800, 60, _C1), (99, 0x10, 20, _C1)],
                    expect=[(120, 179, 6, 30, 0), (125, 194, 6, 95, 1), (205, 254, 6, 154, 1), (258, 312, 6, 157, 2)],
                    intervals=[(257, 10, 212)] * 2 + [(100, 10, 212)] + [(257, 10, 212)] * 4),
    # quirk ②: a soft clip at pos+1 >= depth_len is skipped by `continue`, which also skips the query cursor update,
    # so the following insertion reports qpos 65, not 145
    "clip_continue": dict(depth_len=1001, reads=[(995, 0, 60, [(S, 60), (M, 5), (S, 80), (I, 70)])],
                          expect=[(996, 1055, 0, 0, 2), (1001, 1070, 0, 65, 0)], intervals=[(1000, 60, 215)]),
    # addSVCall: equal (start,end) end up in REVERSE insertion order — across reads and inside one read
    "ties": dict(depth_len=5001, reads=[(500, 0, 60, [(M, 30), (I, 55), (M, 10)]), (500, 0, 60, [(M, 30), (I, 55), (M, 10)]),
                                        (700, 0, 60, [(M, 20), (I, 60), (S, 60)])],
                 expect=[(531, 585, 1, 30, 0), (531, 585, 0, 30, 0), (721, 780, 2, 80, 2), (721, 780, 2, 20, 0)],
                 intervals=[(540, 0, 95), (540, 0, 95), (720, 0, 140)]),
    # hard clips / pads never emit and consume nothing; N consumes reference only; = and X behave like M
    "h_p_n_eq_x": dict(depth_len=5001, reads=[(1000, 0, 60, [(H, 100), (EQ, 30), (P, 100), (N, 100), (X, 5), (D, 50), (EQ, 1)]),
                                              (2000, 0, 60, [(H, 60), (S, 60), (H, 60)]), (2500, 0, 60, [])],
                       expect=[(1136, 1185, 0, 35, 1), (2001, 2060, 1, 0, 2)],
                       intervals=[(1186, 0, 36), (2001, 0, 60), (2501, 0, 0)]),
    # thresholds are parameters of the seam
    "min_oplen_1": dict(depth_len=1001, min_oplen=1, min_mapq=0, reads=[(9, 0, 0, [(M, 2), (I, 1), (D, 2), (M, 1)])],
                        expect=[(12, 12, 0, 2, 0), (12, 13, 0, 3, 1)], intervals=[(14, 0, 4)]),
}

KAT_DEPTH = {
    # M/=/X count, D/N skip reference, I/S/H/P consume nothing; no mapq filter, supplementary counts,
    # UNMAP|SECONDARY|QCFAIL|DUP do not (cnv_caller.cpp:491-495); bases at p >= depth_len are dropped
    "ops": dict(depth_len=16, reads=[(0, 0, 0, [(S, 4), (M, 3), (D, 2), (M, 2), (I, 1), (EQ, 2), (X, 1), (H, 9)]),
                                     (2, 0x800, 0, [(M, 4), (N, 3), (M, 30)]), (3, 0x100, 60, [(M, 5)]), (3, 0x400, 60, [(M, 5)]),
                                     (3, 0x200, 60, [(M, 5)]), (3, 0x4, 60, [(M, 5)]), (14, 0x10, 60, [(P, 2), (M, 2)])],
                depth=[0, 1, 1, 2, 1, 1, 2, 1, 1, 1, 2, 1, 1, 1, 1, 2]),
    "empty": dict(depth_len=8, reads=[], depth=[0] * 8),
}


def kat_reads(case):
    r = case["reads"]
    return Reads.from_cigar_lists([x[0] for x in r], [x[1] for x in r], [x[2] for x in r], [x[3] for x in r])
