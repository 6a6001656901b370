"""vbz_compression_amd -- the VBZ int16 hot path (delta zig-zag + streamvbyte + zstd-format entropy
stage) as hand-written HIP kernels for MI355X (gfx950), behind the reference's C ABI.

  vbz_compression_amd.vbz      numpy front end with pyvbz's interface (single buffers, host memory)
  vbz_compression_amd.batch    batched device-resident codec over torch tensors
  vbz_compression_amd.shard    read sharding across the GPUs of a node (torch.distributed / RCCL)
  vbz_compression_amd.fast5    bulk re-packer of fast5 files (fast5vbz.py's interface over bin/vbz_fast5_repack)
  vbz_compression_amd.build    hipcc build of lib/libvbz_hip.so, lib/libvbz_hdf_plugin.so and bin/vbz_fast5_repack
"""
from . import _lib  # noqa: F401

__all__ = ["_lib", "vbz", "batch", "shard", "fast5", "build"]
