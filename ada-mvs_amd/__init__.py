"""MI355X-native Ada-MVS depth-inference hot path (gfx950 HIP kernels behind a
C-ABI library, Python host that mirrors the reference's models/adamvs.py)."""
__version__ = "0.1.0"
