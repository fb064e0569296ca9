"""cris.pytorch_amd - MI355X-native (gfx950) CRIS training path.

Python host code on PyTorch-ROCm (device memory, streams, torch.distributed/RCCL only) calling a
C-ABI HIP library (csrc/, include/cris_hip.h) through ctypes.  See DESIGN.md.
"""
__version__ = "0.1.0"
