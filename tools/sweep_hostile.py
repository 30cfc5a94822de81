"""One-off wider sweep of tests/test_gpu_hostile.py's generator (not part of the suite)."""
import sys
sys.path.insert(0, 'tests'); sys.path.insert(0, '.')
import numpy as np
import contextsv_amd as cs
import oracle_lib
import test_gpu_hostile as th
orc = oracle_lib.load_oracle()
ctx = cs.Context(0)
bad = 0
rng = np.random.default_rng(99)
for seed in range(100, 160):
    n = int(rng.choice([1, 3, 30, 200, 700]))
    dl = int(rng.choice([1, 2, 17, 300, 5000, 80_000]))
    srt = bool(rng.random() < 0.5)
    try:
        th.test_seams_on_hostile_shards(ctx, orc, seed, n, dl, srt)
    except AssertionError as e:
        bad += 1
        print('FAIL', seed, n, dl, srt, str(e)[:300].replace('\n', ' '))
print('done, failures:', bad)
