"""capital_amd - MI355X-native (gfx950) fp64 Cholesky / CholeskyQR2 hot path of CAPITAL.

Host-side mirror of the reference's operator / algorithm interface over the C ABI of
libcapital_amd.so (include/capital_amd.h).  PyTorch is used only as plumbing: device
memory, streams, torch.distributed bootstrap."""
__version__ = "0.1.0"
