import sys
sys.path.insert(0, 'tests'); sys.path.insert(0, '.')
import numpy as np
import contextsv_amd as cs
from contextsv_amd import Reads
import oracle_lib
from test_gpu_hostile import _hostile
orc = oracle_lib.load_oracle()
ctx = cs.Context(0)
reads = _hostile(5, 30, 17, False)
for r in range(reads.n_reads):
    a, b = int(reads.cigar_off[r]), int(reads.cigar_off[r + 1])
    one = Reads(reads.pos[r:r+1], reads.flag[r:r+1], reads.mapq[r:r+1], np.array([0, b - a], np.uint64), reads.cigar[a:b])
    od, s, nz = orc.depth(one, 17)
    gd, gs, gnz = ctx.depth(one, 17)
    if not np.array_equal(od, gd):
        print('read', r, 'pos', reads.pos[r], 'flag', reads.flag[r], 'ops', [(int(w & 15), int(w >> 4)) for w in reads.cigar[a:b]][:12], 'n', b - a)
        print(' oracle', od.tolist()); print(' gpu   ', gd.tolist())
gd, _, _ = ctx.depth(reads, 17); od, _, _ = orc.depth(reads, 17)
print('all', gd.tolist(), od.tolist())
