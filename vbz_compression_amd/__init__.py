"""vbz_compression_amd -- the VBZ int16 hot path (delta zig-zag + streamvbyte + zstd-format entropy
stage) as hand-written HIP kernels for MI355X (gfx950), behind the reference's C ABI.

  vbz_compression_amd.vbz      numpy front end with pyvbz's interface (single buffers, host memory)
  vbz_compression_amd.batch    batched device-resident codec over torch tensors
  vbz_compression_amd.shard    read sharding across the GPUs of a node (torch.distributed / RCCL)
  vbz_compression_amd.build    hipcc build of lib/libvbz_hip.so and lib/libvbz_hdf_plugin.so
"""
from . import _lib  # noqa: F401

__all__ = ["_lib", "vbz", "batch", "shard", "build"]
