import sys, time
sys.path.insert(0, '.')
import contextsv_amd as cs
from contextsv_amd import host
syn = host.SynthShard(0x5EED0000 + 1000 + 22, 50818468, 30.0, 0, 8)
ctx = cs.Context(0)
ctx.timing_enable(True)
for i in range(3):
    ctx.timing_reset()
    ctx.depth(syn.reads, syn.depth_len, want_array=False)
    t = ctx.timing()
    print('emit=0 scan ms', t['cigar_scan'], 'depth', t['depth'])
for i in range(2):
    ctx.timing_reset()
    s = ctx.cigar_scan(syn.reads, syn.depth_len, capacity=1<<20)
    t = ctx.timing()
    print('emit=1 scan ms', t['cigar_scan'], 'sort', t['sort'], len(s))
