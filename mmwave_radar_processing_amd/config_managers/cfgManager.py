"""TI mmWave ``.cfg`` parser and derived radar scalars (host-side metadata only).

Mirrors the public surface of the reference's ``ConfigManager``
(mmwave_radar_processing/config_managers/cfgManager.py:15-360): same attribute
names, same float64 expressions for the derived quantities, so processors built
on it produce identical bin tables.  Implementation is table-driven: each cfg
command maps to a tuple of (field name, converter) pairs.
"""
from __future__ import annotations

import numpy as np

SPEED_OF_LIGHT = 299792458.0   # scipy.constants.c (cfgManager.py:119)


class InvalidConfiguration(Exception):
    pass


class ConfigNotLoaded(Exception):
    pass


_PROFILE_FIELDS = (
    ("profileId", int), ("startFreq_GHz", float), ("idleTime_us", float), ("adcStartTime_us", float),
    ("rampEndTime_us", float), ("txOutPower", float), ("txPhaseShifter", float),
    ("freqSlope_MHz_us", float), ("txStartTime_us", float), ("adcSamples", int),
    ("sampleRate_kSps", int), ("hpfCornerFreq1", int), ("hpfCornerFreq2", int), ("rxGain_dB", float),
)
_CHIRP_FIELDS = (
    ("startIndex", int), ("endIndex", int), ("profile", int), ("startFreqVariation_Hz", float),
    ("freqSlopVariation_MHz_us", float), ("idleTimeVariation_us", float),
    ("ADCStartTimeVariation_us", float), ("txMask", int),
)
_ADC_BITS = {0: 12, 1: 14, 2: 16}


def _record(fields, tokens):
    return {name: conv(tok) for (name, conv), tok in zip(fields, tokens)}


class ConfigManager:
    def __init__(self):
        self.channelCfg_tx_chan_enabled = 0
        self.channelCfg_rx_chan_enabled = 0
        self.channelCfg_cascading = 0
        self.adcCfg_num_adc_bits = 0
        self.adcCfg_adcOutputFmt = 0
        self.adcbufCfg_adc_output_fmt = 0
        self.adcbufCfg_sample_swap = False
        self.adcbufCfg_channel_interleave = False
        self.adcbufCfg_chirp_threshold = 1
        blank_profile = _record(_PROFILE_FIELDS, ["0"] * len(_PROFILE_FIELDS))
        blank_profile.update(profileId=-1, startFreq_GHz=77.0)
        self.profile_cfgs = [blank_profile]
        blank_chirp = _record(_CHIRP_FIELDS, ["0"] * len(_CHIRP_FIELDS))
        blank_chirp.update(startIndex=-1, endIndex=-1)
        self.chirp_cfgs = [blank_chirp]
        self.frameCfg_start_index = 0
        self.frameCfg_end_index = 0
        self.frameCfg_loops = 0
        self.frameCfg_frames = 0
        self.frameCfg_periodicity_ms = 0.0
        self.frameCfg_hardware_trigger_enabled = False
        self.frameCfg_trigger_delay_ms = 0.0
        self.range_res_m = 0.0
        self.range_bin_size_m = 0.0
        self.range_max_m = 0.0
        self.range_bins_m = np.empty(0, dtype=float)
        self.vel_res_m_s = 0.0
        self.vel_max_m_s = 0.0
        self.num_tx_antennas = 3
        self.num_rx_antennas = 4
        self.virtual_antennas_enabled = False
        self.config_loaded = False
        self.array_geometry = "standard"
        self.array_direction = "down"

    # ---------------------------------------------------------------- loading
    def load_cfg(self, cfg_file_path: str, array_geometry: str = "standard", array_direction: str = "down"):
        with open(cfg_file_path) as fh:
            self.load_cfg_text(fh.read(), array_geometry, array_direction)

    def load_cfg_text(self, text: str, array_geometry: str = "standard", array_direction: str = "down"):
        """Same as load_cfg for cfg text already in memory (lines containing '%' are comments, :234)."""
        self.array_geometry = array_geometry
        self.array_direction = array_direction
        handlers = {
            "channelCfg": self._on_channel, "adcCfg": self._on_adc, "adcbufCfg": self._on_adcbuf,
            "profileCfg": self._on_profile, "chirpCfg": self._on_chirp, "frameCfg": self._on_frame,
        }
        for line in text.splitlines():
            if "%" in line:
                continue
            tokens = line.strip("\n").split(" ")
            fn = handlers.get(tokens[0])
            if fn is not None:
                fn(tokens)
        self.config_loaded = True
        self.compute_radar_perforance(profile_idx=0)

    def _on_channel(self, t):
        self.channelCfg_rx_chan_enabled = int(t[1])
        self.channelCfg_tx_chan_enabled = int(t[2])
        self.channelCfg_cascading = int(t[3])
        self.num_rx_antennas = bin(self.channelCfg_rx_chan_enabled).count("1")
        self.num_tx_antennas = bin(self.channelCfg_tx_chan_enabled).count("1")

    def _on_adc(self, t):
        code = int(t[1])
        if code in _ADC_BITS:
            self.adcCfg_num_adc_bits = _ADC_BITS[code]
        self.adcCfg_adcOutputFmt = int(t[2])

    def _on_adcbuf(self, t):
        self.adcbufCfg_adc_output_fmt = int(t[-4])
        self.adcbufCfg_sample_swap = int(t[-3]) != 0
        self.adcbufCfg_channel_interleave = int(t[-2]) == 0
        self.adcbufCfg_chirp_threshold = int(t[-1])

    def _append_unique(self, table, rec, id_key, what):
        if table[0][id_key] == -1:
            table[0] = rec
        elif rec[id_key] < len(table):
            print(f"cfgManager: attempted to load multiple {what} with the same ID")
        else:
            table.append(rec)

    def _on_profile(self, t):
        self._append_unique(self.profile_cfgs, _record(_PROFILE_FIELDS, t[1:15]), "profileId", "profiles")

    def _on_chirp(self, t):
        self._append_unique(self.chirp_cfgs, _record(_CHIRP_FIELDS, t[1:9]), "startIndex", "chirps")

    def _on_frame(self, t):
        self.frameCfg_start_index = int(t[1])
        self.frameCfg_end_index = int(t[2])
        self.frameCfg_loops = int(t[3])
        self.frameCfg_frames = int(t[4])
        self.frameCfg_periodicity_ms = float(t[5])
        self.frameCfg_hardware_trigger_enabled = int(t[6]) != 1
        self.frameCfg_trigger_delay_ms = float(t[7])

    # ---------------------------------------------------------------- derived scalars
    def compute_radar_perforance(self, profile_idx: int = 0):   # (sic) name kept for callers
        self._compute_range_performance(profile_idx)
        self._compute_vel_performance(profile_idx)
        self._compute_angular_performance()

    compute_radar_performance = compute_radar_perforance

    def _compute_range_performance(self, profile_idx: int = 0):
        samples = self.get_num_adc_samples(profile_idx)
        fs_hz = self.get_adc_sample_rate_kSps(profile_idx) * 1e3
        slope_hz_s = self.get_chirp_slope_MHz_us(profile_idx) * (1e6 / 1e-6)
        nfft = np.power(2, np.ceil(np.log2(samples)))
        self.range_res_m = (SPEED_OF_LIGHT * fs_hz) / (2 * slope_hz_s * samples)
        self.range_bin_size_m = (SPEED_OF_LIGHT * fs_hz) / (2 * slope_hz_s * nfft)
        self.range_max_m = (SPEED_OF_LIGHT * fs_hz) / (2 * slope_hz_s)

    def _compute_vel_performance(self, profile_idx: int = 0):
        prof = self.profile_cfgs[profile_idx]
        lambda_m = SPEED_OF_LIGHT / (float(prof["startFreq_GHz"]) * 1e9)
        chirps_per_loop = self.frameCfg_end_index - self.frameCfg_start_index + 1
        loops = float(self.frameCfg_loops)
        chirp_period_us = float(prof["rampEndTime_us"]) + float(prof["idleTime_us"])
        self.vel_res_m_s = lambda_m / (2 * chirp_period_us * chirps_per_loop * 1e-6 * loops)
        self.vel_max_m_s = lambda_m / (4 * chirp_period_us * chirps_per_loop * 1e-6)

    def _compute_angular_performance(self):
        self.virtual_antennas_enabled = (self.frameCfg_end_index - self.frameCfg_start_index + 1) > 1

    # ---------------------------------------------------------------- getters
    def get_adc_sample_rate_kSps(self, profile_idx: int = 0) -> int:
        return self.profile_cfgs[profile_idx]["sampleRate_kSps"]

    def get_num_adc_samples(self, profile_idx: int = 0) -> int:
        return self.profile_cfgs[profile_idx]["adcSamples"]

    def get_chirp_slope_MHz_us(self, profile_idx: int = 0) -> float:
        return self.profile_cfgs[profile_idx]["freqSlope_MHz_us"]

    def print_cfg_overview(self):
        prof = self.profile_cfgs[0]
        period_us = prof["idleTime_us"] + prof["rampEndTime_us"]
        n_tx = self.frameCfg_end_index - self.frameCfg_start_index + 1
        print("---- Radar Configuration Overview ----")
        print(f"range res {self.range_res_m:.2f} m, max {self.range_max_m:.2f} m; "
              f"vel res {self.vel_res_m_s:.2f} m/s, max {self.vel_max_m_s:.2f} m/s")
        print(f"profile: {prof}")
        print(f"loops {self.frameCfg_loops}, chirp period {period_us} us, "
              f"active frame {n_tx * self.frameCfg_loops * period_us * 1e-3} ms of {self.frameCfg_periodicity_ms} ms")
        print(f"geometry {self.array_geometry}, start {prof['startFreq_GHz']} GHz")
