"""kcftools_amd -- MI355X-native engine for the kcftools `getVariations` hot path.

Only what the path needs lives here: `csrc/` (hand-written HIP kernels + the C ABI of
include/kcfx.h + the C++ host side mirroring GetVariants), a ctypes binding of that ABI
(`engine`), and seeded synthetic-input generators (`synth`).
"""
__version__ = "0.6.0"          # the library says the same: kcfx_version()
