"""reed_amd — MI355X-native (gfx950) implementation of REED's image/ SiT training + sampling
hot path. Python host on PyTorch-ROCm over hand-written HIP kernels behind a C ABI
(include/reed_hip.h). See DESIGN.md."""
__version__ = "0.1.0"
