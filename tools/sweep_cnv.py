"""One-off sweep: window log2 kernel and Viterbi against the oracle on extreme inputs."""
import sys
sys.path.insert(0, 'tests'); sys.path.insert(0, '.')
import numpy as np
import contextsv_amd as cs
from contextsv_amd import make_hmm
import oracle_lib
from hmm_params import WGS_HMM, WGS_TEST_HMM
orc = oracle_lib.load_oracle()
ctx = cs.Context(0)
rng = np.random.default_rng(11)
bad = 0
# ---- windows
for it in range(60):
    L = int(rng.choice([1, 2, 50, 1000, 200_000]))
    depth = rng.choice([rng.poisson(30, L), np.zeros(L, np.int64), rng.integers(0, 2**31, L), np.full(L, 2**32 - 1, np.int64)]).astype(np.uint32)
    k = 12
    rs = rng.integers(0, L + 50, k).astype(np.uint32)
    re = (rs + rng.choice([0, 1, 5, 19, 20, 21, 1000, 10**6], k)).astype(np.uint32)
    ss = rng.choice([1, 2, 5, 20, 137], k).astype(np.int32)
    mean = float(rng.choice([29.7, 1e-300, 1.0, 1e6]))
    try:
        l2, ws, we, off = ctx.window_log2(depth, rs, re, ss, mean)
    except Exception as e:
        print('window raise', it, e); bad += 1; continue
    for r in range(k):
        o_l2, o_ws, o_we = orc.window_log2(depth, int(rs[r]), int(re[r]), int(ss[r]), mean)
        a, b = int(off[r]), int(off[r + 1])
        if not (np.array_equal(ws[a:b], o_ws) and np.array_equal(we[a:b], o_we) and np.allclose(l2[a:b], o_l2, rtol=0, atol=1e-6, equal_nan=True)):
            bad += 1; print('WINDOW FAIL', it, r, L, rs[r], re[r], ss[r], mean, l2[a:b][:4], o_l2[:4]); break
# ---- viterbi
for params in (WGS_HMM, WGS_TEST_HMM):
    hmm = make_hmm(**params)
    for it in range(40):
        o1s, o2s, pfs, off = [], [], [], [0]
        for _ in range(25):
            T = int(rng.choice([0, 1, 2, 3, 20, 200, 1001]))
            o1 = rng.choice([rng.normal(0, 0.4, T), rng.choice([-50.0, -9.966, 0.0, 5.0, 50.0, 1e300, -1e300], T), np.zeros(T)])
            o2 = rng.choice([-1.0, 0.0, 1.0, 0.5, 1e-12, 1 - 1e-12, 0.3333], T)
            pf = rng.choice([0.0, 1.0, 0.5, 0.01, 0.99, 1e-9], T)
            o1s.append(o1); o2s.append(o2); pfs.append(pf); off.append(off[-1] + T)
        o1, o2, pf = np.concatenate(o1s), np.concatenate(o2s), np.concatenate(pfs)
        st, ll = ctx.viterbi(hmm, o1, o2, pf, np.asarray(off, np.uint64))
        ost, oll = orc.viterbi(hmm, o1, o2, pf, np.asarray(off, np.uint64))
        if not (np.array_equal(st, ost) and np.allclose(ll, oll, rtol=0, atol=1e-6, equal_nan=True)):
            bad += 1
            d = np.flatnonzero(st != ost)
            print('VITERBI FAIL', it, len(d), 'first diff at', d[:3], np.max(np.abs(ll - oll)))
print('done, failures:', bad)
