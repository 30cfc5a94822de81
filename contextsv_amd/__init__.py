"""contextsv_amd — MI355X-native hot path of ContextSV (CIGAR scan, depth map, DBSCAN/DBSCAN1D,
6-state Viterbi) behind the C-ABI of include/csvgpu.h. Python here is plumbing for tests and the
benchmark; the product is the HIP library (csrc/) and the C++ host mirror (csrc/host/)."""
from ._lib import CsvError, KIND_CLIP, KIND_DEL, KIND_INS, SIG_DTYPE, make_hmm  # noqa: F401
from .context import ChrResult, Context, Gate, Reads, Shard  # noqa: F401
from .api import DBSCAN, DBSCAN1D, CHMM, ReadCHMM, testVit_CHMM, default_context  # noqa: F401
