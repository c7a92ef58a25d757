"""Shared helpers for the parity tests."""
import numpy as np


def field_groups(n_rw):
    g = {"r": slice(0, 3), "v": slice(3, 6), "sigma": slice(6, 9), "omega": slice(9, 12)}
    if n_rw:
        g["Omega"] = slice(12, 12 + n_rw)
        g["u"] = slice(12 + n_rw + 3, 12 + n_rw + 3 + n_rw)
    return g


def rel_err(a, b, sl):
    """max |a-b| over the group, relative to the group's largest magnitude in b."""
    return float(np.abs(a[sl] - b[sl]).max() / max(np.abs(b[sl]).max(), 1e-300))


def max_group_err(a, b, n_rw):
    return {k: rel_err(a, b, sl) for k, sl in field_groups(n_rw).items()}
