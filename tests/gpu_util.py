"""Helpers for the -m gpu parity tests: the HIP path (through the C ABI) vs the CPU oracle."""
import numpy as np
import torch

DEV = "cuda:0"


def to_dev(a, ld=None):
    """numpy (rows, cols) -> column-major device buffer with leading dimension ld; returns (buf, view[row, col])."""
    a = np.asarray(a, dtype=np.float64)
    rows, cols = a.shape
    ld = ld or rows
    buf = torch.full((cols, ld), float("nan"), dtype=torch.float64, device=DEV)
    buf[:, :rows] = torch.from_numpy(np.ascontiguousarray(a.T)).to(DEV)
    return buf, buf[:, :rows].t()


def to_host(view):
    return view.detach().cpu().numpy().copy()


def relerr(x, ref):
    d = np.linalg.norm(np.asarray(x) - np.asarray(ref))
    n = np.linalg.norm(ref)
    return d / n if n > 0 else d
