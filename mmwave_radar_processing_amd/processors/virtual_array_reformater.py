"""MIMO-TDM de-interleave (reference: mmwave_radar_processing/processors/virtual_array_reformater.py:6-65)."""
from __future__ import annotations

import numpy as np

from .. import _lib
from ._processor import _Processor, as_cube_c64


class VirtualArrayReformatter(_Processor):
    """raw ``[num_rx, S, num_tx*loops]`` -> virtual ``[num_tx*num_rx, S, loops]`` complex128.

    Virtual antenna ``t*num_rx + r`` is Rx ``r`` under chirp-config slot ``t``
    (every ``num_tx``-th chirp starting at ``t``), gathered by ``mmw_virtual_array_reformat``.
    """

    def __init__(self, config_manager, **kwargs) -> None:
        self.chirp_cfg_idxs = None
        self.chirp_cfg_idxs_for_frame = None
        self.chirp_cfgs_per_loop = 0
        self.adc_samples_per_chirp = 0
        super().__init__(config_manager)

    def configure(self):
        cm = self.config_manager
        self.chirp_cfg_idxs = np.arange(cm.frameCfg_start_index, cm.frameCfg_end_index + 1)
        self.chirp_cfg_idxs_for_frame = np.tile(self.chirp_cfg_idxs, cm.frameCfg_loops)
        self.chirp_cfgs_per_loop = cm.frameCfg_end_index - cm.frameCfg_start_index + 1
        self.adc_samples_per_chirp = cm.get_num_adc_samples(profile_idx=0)

    def process(self, adc_cube: np.ndarray, **kwargs) -> np.ndarray:
        cm = self.config_manager
        num_rx, num_tx, loops, S = cm.num_rx_antennas, self.chirp_cfgs_per_loop, cm.frameCfg_loops, self.adc_samples_per_chirp
        raw = as_cube_c64(adc_cube)
        if raw.shape[0] < num_rx or raw.shape[1] != S or raw.shape[2] != num_tx * loops:
            raise ValueError(f"raw cube {raw.shape} does not match cfg ({num_rx}, {S}, {num_tx * loops})")
        raw = np.ascontiguousarray(raw[:num_rx])
        ctx, bufs = self._device()
        d_raw = bufs.get("raw", raw.nbytes)
        d_virt = bufs.get("virt", raw.nbytes)
        d_raw.upload(raw)
        _lib.check(ctx.lib.mmw_virtual_array_reformat(ctx.handle, d_raw.ptr, d_virt.ptr, 1, num_rx, num_tx, S, loops))
        return d_virt.download((num_rx * num_tx, S, loops), np.complex64).astype(complex)
