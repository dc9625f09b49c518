"""Plumbing helpers: device pointers, streams, scratch.  torch is used ONLY for device memory/streams."""
import torch

from . import _lib


def dptr(t):
    """Raw device address of a torch tensor (must live on a HIP device) or a pass-through int."""
    if t is None:
        return None
    if isinstance(t, int):
        return t
    if isinstance(t, torch.Tensor):
        if not t.is_cuda:
            raise _lib.CapitalError("capital_amd operates on device (HBM) buffers only - got a CPU tensor; "
                                    "there is no CPU path")
        if t.dtype != torch.float64 and t.dtype != torch.int32:
            raise _lib.CapitalError("fp64 buffers expected, got %s" % t.dtype)
        return t.data_ptr()
    raise TypeError("expected a torch tensor or a device address")


def cur_stream(stream=None):
    if stream is None:
        return torch.cuda.current_stream().cuda_stream
    if isinstance(stream, int):
        return stream
    return stream.cuda_stream


def scratch(n_doubles, like):
    dev = like.device if isinstance(like, torch.Tensor) else torch.device("cuda", torch.cuda.current_device())
    return torch.empty(max(int(n_doubles), 2), dtype=torch.float64, device=dev)


def colmajor_empty(rows, cols, device, ld=None, zero=False):
    """A (rows x cols) column-major device matrix with leading dimension ld: returns (buffer, view)
    where buffer has shape (cols, ld) and view = buffer[:, :rows].t() indexes [row, col]."""
    ld = ld or rows
    buf = (torch.zeros if zero else torch.empty)(cols, ld, dtype=torch.float64, device=device)
    return buf, buf[:, :rows].t()
