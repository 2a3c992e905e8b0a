"""Processor plugin base (reference: mmwave_radar_processing/processors/_processor.py:6-64).

Protocol kept: ``Cls(config_manager, **params)`` stores the config, calls
``configure()``; ``process(adc_cube, **kwargs)`` takes one virtual-array cube
``[virtRx, sample, chirp]`` and must ignore unknown kwargs (the plugin host
re-passes constructor params, visualization/backends/view_controller.py:56,85,94-101).
Added here: a lazily created device context + buffer set for the HIP path.
"""
from __future__ import annotations

import numpy as np

from .. import _lib
from ..logging import get_logger


def as_cube_c64(adc_cube: np.ndarray) -> np.ndarray:
    """Host cube -> C-contiguous complex64 [V, S, C] (ADC samples are 16-bit integers: exact)."""
    a = np.asarray(adc_cube)
    if a.ndim != 3:
        raise ValueError("adc_cube must be (rx antennas) x (adc samples) x (chirps)")
    return np.ascontiguousarray(a, dtype=np.complex64)


class _Processor:
    def __init__(self, config_manager, **kwargs) -> None:
        self.config_manager = config_manager
        self.logger = get_logger(__name__)
        self._ctx = self._bufs = None
        _Processor.reset(self)              # the two history lists
        self.configure()

    # device plumbing ---------------------------------------------------
    def _device(self):
        """(context, buffers); raises MmwGpuError if the HIP library / GPU is unavailable."""
        if self._ctx is None:
            self._ctx = _lib.default_context()
            self._bufs = _lib.BufferSet(self._ctx)
        return self._ctx, self._bufs

    def _upload_cube(self, adc_cube):
        ctx, bufs = self._device()
        cube = as_cube_c64(adc_cube)
        d_cube = bufs.get("cube", cube.nbytes)
        d_cube.upload(cube)
        return ctx, bufs, d_cube, cube.shape

    # reference protocol ------------------------------------------------
    def configure(self):
        pass

    def reset(self):
        self.history_estimated, self.history_gt = [], []

    def update_history(self, estimated: np.ndarray = np.empty(0), ground_truth: np.ndarray = np.empty(0)) -> None:
        for log, sample in ((self.history_estimated, estimated), (self.history_gt, ground_truth)):
            if np.size(sample):             # empty = nothing to record this frame
                log.append(np.array(sample))

    def process(self, adc_cube: np.ndarray, **kwargs) -> np.ndarray:
        raise NotImplementedError
