"""uzu_amd -- MI355X (gfx950) backend for uzu's transformer forward path.

Layout: ``csrc/`` hand-written HIP kernels + the C ABI (``include/uzu_hip.h``), ``backend.py`` the
host-side mirror of the reference's backend trait surface, ``engine.py`` the model driver handle,
``desc.py`` / ``synthetic.py`` model description and synthetic format-exact weights.
Compute only ever runs through ``lib/libuzu_hip.so`` on an AMD GPU; there is no CPU fallback.
"""
__all__ = ["backend", "desc", "engine", "synthetic"]
