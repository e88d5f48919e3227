"""nasrec_amd — MI355X-native engine for the NASRec supernet forward/backward/optimizer hot path.

Host code is Python on PyTorch-ROCm (device memory, streams, torch.distributed); every kernel is hand-written
HIP for gfx950 behind the C-ABI in include/nasrec_hip.h (nasrec_amd/_lib.py is the ctypes binding).
"""
__version__ = "0.1.0"
