"""tfmq-dm_amd: MI355X-native hot path of TFMQ-DM (w4a8 DDIM sampling + PTQ calibration).

Layout: csrc/ (hand-written gfx950 HIP kernels + the C ABI of include/tfmq_hip.h),
_lib.py (ctypes binding), engine/ (device plans), quant/ (mirror of the reference's quant/
surface), ddim/ (pixel-space UNet description + sampler).  Import as `tfmq_dm_amd`.
"""
__version__ = "0.1.0"
