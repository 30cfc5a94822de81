import numpy as np, sys
sys.path.insert(0, '.')
import contextsv_amd as cs
from contextsv_amd import host
syn = host.SynthShard(0x5EED0000 + 1022, 50818468, 30.0, 0, 16)
ctx = cs.Context(0)
sh = ctx.upload(syn.reads, syn.depth_len)
res = sh.pipeline()
out = sh.fetch(res)
np.savez('gpurun_out/sig_dump.npz', sig=np.concatenate([out['sig_del'], out['sig_ins']]), lab=np.concatenate([out['label_del'], out['label_ins']]), n_del=res.n_del, n_ins=res.n_ins)
print(res.n_sig, res.n_del)
