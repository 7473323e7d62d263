"""Error type mirroring `UzkgeError` (reference: uzkge/src/errors.rs:5-44) for the variants the
hot path can produce; DEVICE is new (no GPU / HIP failure -- there is no CPU fallback)."""
from __future__ import annotations

from . import _native as N


class UzkgeError(Exception):
    NAMES = {
        N.UZK_ERR_PARAMETER: "ParameterError",
        N.UZK_ERR_DEGREE: "DegreeError",
        N.UZK_ERR_FFT: "FFTError",
        N.UZK_ERR_COMMITMENT: "CommitmentError",
        N.UZK_ERR_DEVICE: "DeviceError",
    }

    def __init__(self, code: int, detail: str = ""):
        self.code = code
        self.kind = self.NAMES.get(code, f"Unknown({code})")
        super().__init__(f"{self.kind}: {detail}" if detail else self.kind)


def check(rc: int) -> None:
    if rc != N.UZK_OK:
        raise UzkgeError(rc, (N.lib.uzk_last_error() or b"").decode(errors="replace"))
