"""Range-Doppler response (reference: mmwave_radar_processing/processors/range_doppler_resp.py:8-110)."""
from __future__ import annotations

import numpy as np

from .. import _lib
from ._processor import _Processor


class RangeDopplerProcessor(_Processor):
    """fftshift_C(FFT_S FFT_C(hann(S) hann(C) x)) for every virtual antenna, on the GPU (``mmw_range_doppler``)."""

    def __init__(self, config_manager, **kwargs) -> None:
        self.vel_bins = None
        self.range_bins = None
        super().__init__(config_manager)

    def configure(self):
        cm = self.config_manager
        self.vel_bins = np.arange(start=-1 * cm.vel_max_m_s, stop=cm.vel_max_m_s - cm.vel_res_m_s + 1e-3,
                                  step=cm.vel_res_m_s)
        self.range_bins = np.arange(start=0, step=cm.range_res_m, stop=cm.range_max_m - cm.range_res_m / 2 + 1e-3)

    def _range_doppler_device(self, adc_cube, want_mag: bool):
        """Runs the chain; leaves rd (c64 [V,S,C]) and optionally |rd| (f32) in device buffers."""
        ctx, bufs, d_cube, (V, S, C) = self._upload_cube(adc_cube)
        n = V * S * C
        d_rd = bufs.get("rd", n * 8)
        d_mag = bufs.get("rd_mag", n * 4) if want_mag else None
        _lib.check(ctx.lib.mmw_range_doppler(ctx.handle, d_cube.ptr, d_rd.ptr, d_mag.ptr if want_mag else None,
                                             1, V, S, C))
        return ctx, bufs, d_cube, d_rd, d_mag, (V, S, C)

    def process(self, adc_cube: np.ndarray, rx_idx: int = 0, return_magnitude: bool = True, **kwargs) -> np.ndarray:
        """``rx_idx`` >= 0 selects one antenna AFTER all are computed, -1 returns all (reference :108-110)."""
        _, _, _, d_rd, d_mag, (V, S, C) = self._range_doppler_device(adc_cube, return_magnitude)
        plane = S * C
        if return_magnitude:
            if rx_idx >= 0:
                return d_mag.download((S, C), np.float32, rx_idx * plane * 4).astype(np.float64)
            return d_mag.download((V, S, C), np.float32).astype(np.float64)
        if rx_idx >= 0:
            return d_rd.download((S, C), np.complex64, rx_idx * plane * 8).astype(np.complex128)
        return d_rd.download((V, S, C), np.complex64).astype(np.complex128)
