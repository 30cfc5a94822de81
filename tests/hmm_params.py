"""HMM parameter sets used as test constants. The values are the numbers in the reference's data files
(data/wgs.hmm and data/wgs_test.hmm, which differ only in B2_uf), re-typed here as data because
/root/reference does not exist on the GPU box."""
import numpy as np

_A = [[0.899997, 0.009, 0.091, 0.000001, 0.000001, 0.000001],
      [0.009, 0.899997, 0.091, 0.000001, 0.000001, 0.000001],
      [0.00001, 0.00005, 0.99987, 0.00001, 0.00005, 0.00001],
      [0.000001, 0.000001, 0.00003, 0.999966, 0.000001, 0.000001],
      [0.000001, 0.000001, 0.091, 0.000001, 0.899997, 0.009],
      [0.000001, 0.000001, 0.091, 0.000001, 0.009, 0.899997]]
WGS_HMM = dict(A=_A, pi=[0.000001, 0.000500, 0.999000, 0.000001, 0.000500, 0.000001],
               B1_mean=[-3.739099, -0.727964, 0.000000, 100, 0.395454, 0.658622],
               B1_sd=[2.564467, 0.303606, 0.163877, 0.163877, 0.127181, 0.124527], B1_uf=0.01,
               B2_mean=[0.0, 0.25, 0.333333, 0.5, 0.5], B2_sd=[0.155241, 0.157236, 0.166946, 0.057305, 0.044416], B2_uf=0.01)
WGS_TEST_HMM = dict(WGS_HMM, B2_uf=0.001)


def write_hmm_file(path, p):
    """Write a parameter set in the .hmm text grammar ReadCHMM parses (khmm.cpp:395-553)."""
    def row(v):
        return " ".join(f"{x:.6f}" for x in v)
    B = np.full((6, 6), 0.000001)
    lines = ["M=6", "N=6", "A:"] + [row(r) for r in p["A"]] + ["B:"] + [row(r) + " " for r in B] + ["pi:", row(p["pi"]) + " ",
             "B1_mean:", row(p["B1_mean"]) + " ", "B1_sd:", row(p["B1_sd"]) + " ", "B1_uf:", f"{p['B1_uf']:.6f}",
             "B2_mean:", row(p["B2_mean"]) + " ", "B2_sd:", row(p["B2_sd"]), "B2_uf:", f"{p['B2_uf']:.6f}",
             "B3_mean:", row(p["B1_mean"]) + " ", "B3_sd:", row(p["B1_sd"]) + " ", "B3_uf:", "0.010000"]
    with open(path, "w") as f:
        f.write("\n".join(lines) + "\n")
    return path
