'''
radar-camera-fusion-depth on MI355X: the FusionNet encoder/decoder forward+backward hot path of
nesl/radar-camera-fusion-depth as hand-written HIP kernels for gfx950 behind a C ABI
(csrc/, include/rcf_hip.h), with a Python host that mirrors the reference's FusionNetModel surface.

The directory name is the one the build contract prescribes; it is not a valid Python identifier,
so import it as ``import rcf_amd`` (repo-root alias) or ``importlib.import_module(...)``.
'''

__version__ = '0.1.0'
