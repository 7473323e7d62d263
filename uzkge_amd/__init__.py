"""uzkge_amd -- MI355X (gfx950) backend for the two hot paths of uzkge's PlonK prover:
BN254 G1 MSM (KZG commit) and the BN254 Fr NTT (EvaluationDomain fft/ifft).

The product is `libuzkge_gpu.so` (hand-written HIP, C ABI in include/uzkge_gpu.h).  This package
is the host-side mirror of the reference's interface for that path (`FpPolynomial`,
`KZGCommitmentSchemeBN254`) plus numpy-facing wrappers used by tests and bench.py.
Importing it loads the native library and fails loudly if it has not been built.
"""
from . import _native  # noqa: F401  (loads libuzkge_gpu.so or raises ImportError)
from . import backend
from .errors import UzkgeError

__all__ = ["backend", "UzkgeError"]
