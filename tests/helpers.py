"""Shared helpers for the parity tests."""
import numpy as np
import scipy.sparse as sp


def csc_to_scipy(csc, n):
    outer, inner, values = csc
    return sp.csc_matrix((values, inner, outer), shape=(n, n))


def rel_err(a, b):
    """max |a-b| / max |b| (norm-wise relative error; entries that cancel to ~0 are measured against the
    largest entry, as an entry-wise ratio is meaningless for them)."""
    a, b = np.asarray(a), np.asarray(b)
    scale = np.abs(b).max() if b.size else 1.0
    return np.abs(a - b).max() / (scale if scale > 0 else 1.0)
