"""MI355X-native FaceDetector hot path (gfx950 HIP kernels behind a C ABI).

The product path never falls back to a CPU implementation: if libfv_hotpath.so is missing
or no MI355X is visible, calls raise."""
__version__ = '0.1.0'
