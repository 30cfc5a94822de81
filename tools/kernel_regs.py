"""Register / LDS / scratch use of every kernel of one .hip file (compiled to assembly in a scratch directory, gfx950):
    python tools/kernel_regs.py scan [depth ...]
"""
import os, re, subprocess, sys, tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    for name in sys.argv[1:]:
        src = os.path.join(ROOT, "contextsv_amd", "csrc", "kernels", name + ".hip")
        with tempfile.TemporaryDirectory() as d:
            out = os.path.join(d, name + ".s")
            subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-ffp-contract=off", "--cuda-device-only", "-S", src, "-o", out])
            s = open(out).read()
            keep = os.environ.get("KEEP_ASM")
            if keep:
                open(keep, "w").write(s)
        for b in s.split("  - .agpr_count:")[1:]:
            sym = re.search(r"\.name:\s+(\S+)", b).group(1)
            dn = subprocess.run(["c++filt", sym], capture_output=True, text=True).stdout.split("(")[0].strip()
            g = lambda k: re.search(r"\." + k + r":\s+(\d+)", b).group(1)
            print(f"{name:8s} {dn:60s} vgpr {g('vgpr_count'):>3s} spill {g('vgpr_spill_count'):>3s} sgpr {g('sgpr_count'):>3s} lds {g('group_segment_fixed_size'):>6s} scratch {g('private_segment_fixed_size'):>4s}")


if __name__ == "__main__":
    main()
