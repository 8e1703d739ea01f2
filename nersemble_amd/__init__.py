"""nersemble_amd -- MI355X (gfx950) native per-sample hot path of NeRSemble.

The compute lives in ``csrc/libnsx.so`` (hand-written HIP, C ABI in ``include/nsx.h``); the Python
modules mirror the reference's plugin/operator surface for this path (tcnn-, nerfacc- and
distloss-shaped operators, HashEnsemble / Field / Sampler / Model classes).  There is no CPU or
eager-PyTorch fallback for the native ops: if ``libnsx.so`` is missing they raise.
"""
__version__ = "0.1.0"
