"""Shared comparison helpers for the parity tests.

Tolerance (BASELINE.json north_star): the GPU frame must match the CPU oracle's frame of the same scene on the
pre-tonemap float buffer with PSNR >= 60 dB (peak 255, the range the reference quantises to) and a per-pixel
max-abs bound, and primary-ray hit ids must be identical.  Geometry is bit-exact by construction (same IEEE
operations in the same order); the only float differences come from libm-style functions (powf, expf, acosf,
atan2f, sinf, cosf) whose last ulp differs between glibc and the ROCm device library, so the max-abs bound is
relative: |gpu - cpu| <= 1e-3 * max(1, |cpu|)  (observed: ~1e-5 relative).
"""
import numpy as np

PSNR_MIN_DB = 60.0
MAX_REL = 1e-3


def psnr(a, b, peak=255.0):
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    mse = np.mean((a - b) ** 2)
    return float("inf") if mse == 0 else 10.0 * np.log10(peak * peak / mse)


def compare_frames(gpu_hdr, cpu_hdr, gpu_id=None, cpu_id=None, gpu_t=None, cpu_t=None):
    """Returns a dict of metrics and raises AssertionError when outside tolerance."""
    gpu_hdr = np.asarray(gpu_hdr)
    cpu_hdr = np.asarray(cpu_hdr)
    assert gpu_hdr.shape == cpu_hdr.shape
    # non-finite values must sit at the same places on both sides, NaN against NaN and infinities of the same sign
    assert np.array_equal(np.isfinite(gpu_hdr), np.isfinite(cpu_hdr)), "non-finite values at different pixels"
    assert np.array_equal(np.isnan(gpu_hdr), np.isnan(cpu_hdr)), "NaN on one side, infinity on the other"
    inf = np.isinf(cpu_hdr)
    assert np.array_equal(np.sign(gpu_hdr[inf]), np.sign(cpu_hdr[inf])), "infinities of different sign"
    fin = np.isfinite(cpu_hdr) & np.isfinite(gpu_hdr)
    diff = np.abs(np.where(fin, gpu_hdr, 0).astype(np.float64) - np.where(fin, cpu_hdr, 0).astype(np.float64))
    bound = MAX_REL * np.maximum(1.0, np.abs(np.where(fin, cpu_hdr, 0)))
    m = {"psnr": psnr(np.where(fin, gpu_hdr, 0), np.where(fin, cpu_hdr, 0)), "max_abs": float(diff.max()),
         "bad_pixels": int((diff > bound).any(axis=-1).sum()), "bit_equal": float((gpu_hdr == cpu_hdr).mean())}
    if gpu_id is not None:
        m["id_mismatch"] = int((np.asarray(gpu_id) != np.asarray(cpu_id)).sum())
        assert m["id_mismatch"] == 0, m
    if gpu_t is not None:
        m["t_mismatch"] = int((np.asarray(gpu_t) != np.asarray(cpu_t)).sum())
        assert m["t_mismatch"] == 0, m
    assert m["psnr"] >= PSNR_MIN_DB, m
    assert m["bad_pixels"] == 0, m
    return m
