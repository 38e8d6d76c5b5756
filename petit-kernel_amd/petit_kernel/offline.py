"""petit_kernel.offline -- checkpoint-side format tooling.

Convert the native NVFP4 / MXFP4 tensors of a checkpoint into the packed gfx950 layout on the CPU,
so the load-time GPU repack (`repack_nvfp4`, `process_*_scales`) becomes optional: the outputs are
bit-identical to what those produce on the device (tests/test_layout_and_abi.py,
tests/test_gpu_parity.py::test_offline_repack_matches_device) and can be saved with the
checkpoint and copied to the GPU as they are.  The reference has no counterpart: its repack exists
only as GPU kernels (quantization_utils.cu:208-304).

This is format tooling, not a CPU fallback: only the GPU kernels consume the packed tensors.
"""
import torch

from . import _lib
from .ops import _LAYOUT_M, _LAYOUT_N, _PACK, _check, _raise_on


def _cpu(t: torch.Tensor, name: str) -> None:
    _check(not t.is_cuda, f"{name} must be a CPU tensor (use petit_kernel.repack_* for GPU tensors)")
    _check(t.is_contiguous(), f"{name} is not contiguous")


def repack_nvfp4_cpu(qw: torch.Tensor, size_n: int, size_k: int) -> torch.Tensor:
    """CPU twin of repack_nvfp4 / repack_mxfp4: int32 [N, K/8] -> int32 [N/16, 2K]."""
    _check(size_k % _LAYOUT_M == 0, f"size_k = {size_k} is not divisible by tile_k_size = {_LAYOUT_M}")
    _check(size_n % _LAYOUT_N == 0, f"size_n = {size_n} is not divisible by tile_n_size = {_LAYOUT_N}")
    _check(qw.dim() == 2 and qw.size(0) == size_n and qw.size(1) == size_k // _PACK,
           f"qw must be [size_n, size_k / {_PACK}]")
    _check(qw.dtype == torch.int32, "qw type is not kInt")
    _cpu(qw, "qw")
    out = torch.empty((size_n // _LAYOUT_N, size_k * _LAYOUT_N // _PACK), dtype=torch.int32)
    _raise_on(_lib.lib.petit_repack_nvfp4_weights_host(out.data_ptr(), qw.data_ptr(), size_k, size_n), "repack_nvfp4_cpu")
    return out


repack_mxfp4_cpu = repack_nvfp4_cpu


def process_nvfp4_scales_cpu(scales: torch.Tensor, size_n: int, size_k: int) -> torch.Tensor:
    """CPU twin of process_nvfp4_scales: e4m3 [N, K/16] -> e4m3 [N, K/16]."""
    _check(size_k % (2 * _LAYOUT_M) == 0, f"size_k = {size_k} is not divisible by tile_k_size = {2 * _LAYOUT_M}")
    _check(size_n % _LAYOUT_N == 0, f"size_n = {size_n} is not divisible by tile_n_size = {_LAYOUT_N}")
    _check(scales.dim() == 2 and scales.size(0) == size_n and scales.size(1) * 16 == size_k,
           "Only groupsize = 16 is supported.")
    _check(scales.dtype == torch.float8_e4m3fn, "scales type is not float8_e4m3fn")
    _cpu(scales, "scales")
    out = torch.empty_like(scales)
    _raise_on(_lib.lib.petit_repack_nvfp4_scales_host(out.data_ptr(), scales.data_ptr(), size_k, size_n),
              "process_nvfp4_scales_cpu")
    return out


def process_mxfp4_scales_cpu(scales: torch.Tensor, size_n: int, size_k: int) -> torch.Tensor:
    """CPU twin of process_mxfp4_scales: e8m0 uint8 [N, K/32] -> uint8 [N/32, K]."""
    _check(size_k % (2 * _LAYOUT_M) == 0, f"size_k = {size_k} is not divisible by tile_k_size = {2 * _LAYOUT_M}")
    _check(size_n % 32 == 0, f"size_n = {size_n} is not divisible by 32")
    _check(scales.dim() == 2 and scales.size(0) == size_n and scales.size(1) * 32 == size_k,
           "Only groupsize = 32 is supported.")
    _check(scales.dtype == torch.uint8, "scales type is not uint8")
    _cpu(scales, "scales")
    out = torch.empty((size_n // 32, size_k), dtype=torch.uint8)
    _raise_on(_lib.lib.petit_repack_mxfp4_scales_host(out.data_ptr(), scales.data_ptr(), size_k, size_n),
              "process_mxfp4_scales_cpu")
    return out


# --- tensors already packed by the REFERENCE wheel -> this build's layout (include/petit_amd.h, petit_convert_reference_*) -----

def from_reference_packed_weights(b_packed: torch.Tensor, size_n: int, size_k: int) -> torch.Tensor:
    """int32 [N/16, 2K] as returned by the REFERENCE's repack_nvfp4 / repack_mxfp4 -> the same shape in this build's layout."""
    _check(size_k % _LAYOUT_M == 0 and size_n % 32 == 0, "needs size_k % 128 == 0 and size_n % 32 == 0")
    _check(b_packed.dtype == torch.int32 and b_packed.numel() * 4 == size_n * size_k // 2, "b_packed does not hold size_n * size_k 4-bit weights")
    _cpu(b_packed, "b_packed")
    out = torch.empty((size_n // _LAYOUT_N, size_k * _LAYOUT_N // _PACK), dtype=torch.int32)
    _raise_on(_lib.lib.petit_convert_reference_weights_host(out.data_ptr(), b_packed.contiguous().data_ptr(), size_k, size_n),
              "from_reference_packed_weights")
    return out


def from_reference_packed_nvfp4_scales(s_packed: torch.Tensor, size_n: int, size_k: int) -> torch.Tensor:
    """[N, K/16] bytes as returned by the REFERENCE's process_nvfp4_scales ("e5m3" bytes in its layout) -> float8_e4m3fn
    [N, K/16] in this build's layout."""
    _check(size_k % (2 * _LAYOUT_M) == 0 and size_n % 64 == 0, "needs size_k % 256 == 0 and size_n % 64 == 0")
    _check(s_packed.numel() * s_packed.element_size() == size_n * size_k // 16, "s_packed does not hold size_n * size_k / 16 scale bytes")
    _cpu(s_packed, "s_packed")
    out = torch.empty((size_n, size_k // 16), dtype=torch.uint8)
    _raise_on(_lib.lib.petit_convert_reference_nvfp4_scales_host(out.data_ptr(), s_packed.contiguous().data_ptr(), size_k, size_n),
              "from_reference_packed_nvfp4_scales")
    return out.view(torch.float8_e4m3fn)


def from_reference_packed_mxfp4_scales(s_packed: torch.Tensor, size_n: int, size_k: int) -> torch.Tensor:
    """uint8 [N/32, K] as returned by the REFERENCE's process_mxfp4_scales -> the same shape in this build's layout."""
    _check(size_k % (2 * _LAYOUT_M) == 0 and size_n % 32 == 0, "needs size_k % 256 == 0 and size_n % 32 == 0")
    _check(s_packed.dtype == torch.uint8 and s_packed.numel() == size_n * size_k // 32, "s_packed does not hold size_n * size_k / 32 scale bytes")
    _cpu(s_packed, "s_packed")
    out = torch.empty((size_n // 32, size_k), dtype=torch.uint8)
    _raise_on(_lib.lib.petit_convert_reference_mxfp4_scales_host(out.data_ptr(), s_packed.contiguous().data_ptr(), size_k, size_n),
              "from_reference_packed_mxfp4_scales")
    return out


# --- the MFMA-native image of NVFP4 weights, offline (include/petit_amd.h "NVFP4 weights on the native class") ---------------------------

def nvfp4_native_image_cpu(b_packed: torch.Tensor, s_packed: torch.Tensor, size_n: int, size_k: int) -> torch.Tensor:
    """CPU twin of petit_kernel.nvfp4_native_image: the PACKED tensors (repack_nvfp4_cpu / process_nvfp4_scales_cpu) -> the image, uint8, bit-identical
    to what the device kernel builds; save it with the checkpoint and copy it to the GPU as it is."""
    nbytes = int(_lib.lib.petit_nvfp4_native_image_bytes(size_k, size_n))
    _check(nbytes > 0, f"Incompatible problem shape (n={size_n}, k={size_k})")
    _check(b_packed.numel() * b_packed.element_size() == size_n * size_k // 2, "b_packed does not hold size_n * size_k 4-bit weights")
    _check(s_packed.numel() * s_packed.element_size() == size_n * size_k // 16, "s_packed does not hold size_n * size_k / 16 scales")
    _cpu(b_packed, "b_packed")
    _cpu(s_packed, "s_packed")
    image = torch.empty(nbytes, dtype=torch.uint8)
    _raise_on(_lib.lib.petit_nvfp4_native_image_host(image.data_ptr(), b_packed.data_ptr(), s_packed.data_ptr(), size_k, size_n), "nvfp4_native_image_cpu")
    return image


def nvfp4_native_image_dequant_cpu(image: torch.Tensor, size_n: int, size_k: int) -> torch.Tensor:
    """Dense expansion of an image on the CPU (test / debug aid): f32 [N, K] = element x 2^(scale - 127), WITHOUT the global scale."""
    _check(image.dtype == torch.uint8 and image.numel() == int(_lib.lib.petit_nvfp4_native_image_bytes(size_k, size_n)) and image.numel() > 0,
           "image does not hold the native image of size_n x size_k NVFP4 weights")
    _cpu(image, "image")
    out = torch.empty((size_n, size_k), dtype=torch.float32)
    _raise_on(_lib.lib.petit_nvfp4_native_image_dequant_host(out.data_ptr(), image.data_ptr(), size_k, size_n), "nvfp4_native_image_dequant_cpu")
    return out
