"""bmhrl_amd -- MI355X-native hot path of Berghojo/bmhrl (bimodal transformer fwd/bwd + per-token loss step).

Host code is Python on PyTorch-ROCm; all arithmetic of the hot path runs in the hand-written gfx950 HIP
kernels of ``bmhrl_amd/csrc`` behind the C ABI declared in ``include/bmhrl_hip.h``.  The HIP library is
loaded lazily by ``bmhrl_amd._lib``; there is no CPU fallback.
"""
__version__ = "0.1.0"
