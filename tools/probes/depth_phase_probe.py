"""Where a depth tile's time goes (build with `make -C contextsv_amd/csrc EXTRA=-DDEPTH_PHASE_PROBE` after touching kernels/depth.hip):
cycle counts of thread 0 of every workgroup at the phase boundaries, summed over tiles. Investigation only; the symbol does not exist
in the normal build."""
import ctypes as C, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import contextsv_amd as cs
from contextsv_amd import host, _lib

contig = int(sys.argv[1]) if len(sys.argv) > 1 else 22
tech = sys.argv[2] if len(sys.argv) > 2 else "ont"
L = {1: 248956422, 22: 50818468}[contig]
syn = host.SynthShard(0x5EED0000 + 1000 * (3 if tech == "ont" else 4) + contig, L, 30.0 if tech == "ont" else 60.0, 0 if tech == "ont" else 1, 32)
ctx = cs.Context(0)
sh = ctx.upload(syn.reads, syn.depth_len)
syn.free()
lib = _lib.load()
sh.pipeline(); ctx.synchronize()
out = (C.c_ulonglong * 8)()
lib.csvgpu_debug_depth_phase(out, 1)
for _ in range(5):
    sh.pipeline()
ctx.synchronize()
lib.csvgpu_debug_depth_phase(out, 0)
v = list(out)[:5]
tot = sum(v) or 1
names = ["zero + range/count loads", "work list (load or build) + barrier", "walk (wave 0)", "barrier wait after the walk", "scan + write"]
print(json.dumps({"contig": contig, "tech": tech, "cycles": dict(zip(names, v)), "share": {n: round(x / tot, 3) for n, x in zip(names, v)}}))
