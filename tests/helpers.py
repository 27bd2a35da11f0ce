"""Shared helpers of the parity tests."""
import numpy as np


def bits_equal(a, b):
    """Bit-exact float64 comparison in which any NaN equals any NaN (NaN payloads and signs are
    not part of the contract; x86 and gfx950 produce different default NaNs)."""
    a = np.ascontiguousarray(a, dtype=np.float64)
    b = np.ascontiguousarray(b, dtype=np.float64)
    both_nan = np.isnan(a) & np.isnan(b)
    return (a.view(np.uint64) == b.view(np.uint64)) | both_nan


def lsb_stats(img_a, img_b):
    """img_*: int32 packed 0xRRGGBB.  Returns (fraction of pixels within 1 LSB in every channel, max abs diff)."""
    def unpack(i):
        return np.stack([(i >> 16) & 255, (i >> 8) & 255, i & 255], -1).astype(np.int32)
    d = np.abs(unpack(img_a) - unpack(img_b))
    return float((d.max(-1) <= 1).mean()), int(d.max())
