"""MI355X-native hot path of davidmhunt/mmwave_radar_processing (range-Doppler-angle FFT chain + CFAR).

Sub-packages mirror the reference: ``config_managers``, ``processors``, ``detectors``, ``logging``;
``batch`` adds the device-resident multi-frame pipeline and per-frame GPU sharding.  All arithmetic runs in
hand-written HIP kernels behind the C ABI of ``include/mmwgpu.h`` (``_lib``); there is no CPU fallback.
"""
__version__ = "0.1.0"
