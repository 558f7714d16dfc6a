"""cpg - Python host side of the MI355X-native peptide WAE-training / CLaSS-sampling hot path.

The arithmetic lives in libcpg_hip.so (hand-written HIP for gfx950, C ABI in include/cpg_api.h); this package binds
it with ctypes, wraps the kernels as torch.autograd Functions, and mirrors the reference's Python interface one level
up (models.model.RNN_VAE, losses, train_vae, density_modeling, sample_pipeline live next to this package).
There is no CPU fallback: without the library and a GPU every op raises.
"""
from ._lib import lib, LibraryMissing, build_library, LIB_PATH  # noqa: F401
