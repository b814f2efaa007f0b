"""Shared helpers of the parity tests.

Tolerance (BASELINE.json north_star): 1e-10 relative in fp64.  "Relative" is taken per column profile:
    |a - b| <= RTOL * max_k |ref(column, :)|      for profile fields,
    |a - b| <= RTOL * max(|a|, |b|, FLOOR * max|field|)   for per-column scalars,
because individual elements of a tendency profile pass through zero (cancellation leaves 1e-19-size
residues next to 1e-3-size neighbours) and an element-wise relative error is meaningless there.
Integer / index outputs must be identical.
"""
from __future__ import annotations

import numpy as np

RTOL = 1.0e-10
FLOOR = 1.0e-6


def rel_err_profile(a: np.ndarray, ref: np.ndarray) -> float:
    """max over columns of max_k|a-ref| / max_k|ref|; the last axis is the level axis"""
    a = np.asarray(a, dtype=np.float64); ref = np.asarray(ref, dtype=np.float64)
    assert a.shape == ref.shape, (a.shape, ref.shape)
    if a.size == 0:                      # e.g. the gathered set of a run in which no column convects
        return 0.0
    if a.ndim == 1:
        scale = np.maximum(np.maximum(np.abs(a), np.abs(ref)), FLOOR * max(np.abs(ref).max(), 1e-300))
        return float((np.abs(a - ref) / scale).max()) if a.size else 0.0
    den = np.abs(ref).max(axis=-1, keepdims=True)
    num = np.abs(a - ref).max(axis=-1, keepdims=True)
    fieldmax = max(np.abs(ref).max(), 1e-300)
    den = np.maximum(den, FLOOR * fieldmax)
    return float((num / den).max()) if a.size else 0.0


def assert_close(name: str, a, ref, rtol: float = RTOL):
    a = np.asarray(a); ref = np.asarray(ref)
    if np.issubdtype(ref.dtype, np.integer):
        assert np.array_equal(a, ref), f"{name}: integer field differs at {np.argwhere(a != ref)[:5].tolist()}"
        return 0.0
    assert np.isfinite(a).all(), f"{name}: non-finite values"
    e = rel_err_profile(a, ref)
    assert e <= rtol, f"{name}: relative error {e:.3e} > {rtol:.1e}"
    return e


def gathered_mask(lengath: np.ndarray, ncol: int, pcols: int) -> np.ndarray:
    """True for rows of the flat gathered layout that hold a gathered column (slot < lengath[chunk])."""
    rows = np.arange(ncol)
    return (rows % pcols) < np.asarray(lengath)[rows // pcols]
