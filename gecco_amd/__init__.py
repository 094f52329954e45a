"""gecco_amd — MI355X-native GECCO denoiser (drop-in for the `gecco_torch` module API).

The compute path is the hand-written HIP library `libgecco_hip.so` (C ABI: include/gecco_hip.h).
There is no CPU or PyTorch fallback: if the library is missing or the tensors are not on a HIP
device, the operators raise.
"""
__version__ = "0.1.0"
