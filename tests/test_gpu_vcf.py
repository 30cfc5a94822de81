"""§8f-1 on the device: the VCF writer's SUPPORT / DP lookups (SVCaller::getReadDepth, sv_caller.cpp:1332-1344) as a gather
on the depth map resident in a shard, against the oracle's depth map; and the writer fed from shards against the writer fed
from host arrays."""
import numpy as np
import pytest

from contextsv_amd import host
from contextsv_amd.host import CALL_DTYPE
import synth_small as ss

pytestmark = pytest.mark.gpu


def test_depth_lookup_resident(ctx, oracle):
    rng = np.random.default_rng(2)
    reads, depth_len = ss.random_shard(2, n_reads=3000, mean_ops=40, chr_len=200_000)
    want, _, _ = oracle.depth(reads, depth_len)
    sh = ctx.upload(reads, depth_len)
    try:
        sh.pipeline()
        pos = np.concatenate([rng.integers(0, depth_len, 5000), [0, depth_len - 1, depth_len, depth_len + 1, 2**32 - 1]]).astype(np.uint32)
        got = sh.depth_lookup(pos)
        inside = pos < depth_len
        assert np.array_equal(got[inside], want[pos[inside]].astype(np.int32))
        assert (got[~inside] == -1).all() and (~inside).sum() >= 3
        assert len(sh.depth_lookup(np.zeros(0, np.uint32))) == 0
        assert want[pos[inside]].max() > 0
    finally:
        sh.free()


def test_writer_from_shards_equals_writer_from_host_arrays(ctx, oracle, tmp_path):
    rng = np.random.default_rng(4)
    fasta = tmp_path / "g.fa"
    names, items_dev, items_host, shards = ["c1", "c2"], [], [], []
    with open(fasta, "w") as f:
        for n in names:
            f.write(">%s\n%s\n" % (n, "".join(rng.choice(list("ACGTN"), 50_000))))
    genome = host.ReferenceGenome(str(fasta))
    try:
        for n in names:
            reads, depth_len = ss.random_shard(len(shards) + 9, n_reads=1500, mean_ops=30, chr_len=50_000)
            sh = ctx.upload(reads, depth_len)
            shards.append(sh)
            sh.pipeline()
            want, _, _ = oracle.depth(reads, depth_len)
            c = np.zeros(300, CALL_DTYPE)
            c["sv_type"] = rng.choice([0, 1, 2, 3], 300)
            c["start"] = rng.integers(1, depth_len + 40, 300)          # some past the end of the map
            c["end"] = c["start"] + rng.integers(0, 300, 300)
            c["genotype"], c["cn_state"] = rng.integers(0, 4, 300), rng.integers(0, 7, 300)
            alts = [{0: b"<DEL>", 1: b"<DUP>", 2: b"<INV>", 3: b"ACGT"}[int(t)] for t in c["sv_type"]]
            items_dev.append((n, c, alts, sh))
            items_host.append((n, c, alts, want))
        (tmp_path / "dev").mkdir(); (tmp_path / "hst").mkdir()
        cd = host.save_vcf(str(tmp_path / "dev"), genome, items_dev, file_date="20250926", ctx=ctx)
        ch = host.save_vcf(str(tmp_path / "hst"), genome, items_host, file_date="20250926")
        rc, co = oracle.save_vcf(str(tmp_path / "orc.vcf"), str(fasta), items_host, file_date="20250926")
        a = (tmp_path / "dev" / "output.vcf").read_bytes()
        assert a == (tmp_path / "hst" / "output.vcf").read_bytes() == (tmp_path / "orc.vcf").read_bytes()
        assert rc == 0 and cd == ch == co and cd[0] == 600
        assert b"SUPPORT=0;" in a
    finally:
        for sh in shards:
            sh.free()
