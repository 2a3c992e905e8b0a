#!/usr/bin/env python3
"""One mmw_diag_mfma_peak probe (tools/pmc_coexec.sh profiles it under rocprofv3 --pmc):  python tools/mfma_probe.py KIND"""
import ctypes as ct
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mmwave_radar_processing_amd import _lib  # noqa: E402

kind = int(sys.argv[1])
ctx = _lib.Context(0)
v = ct.c_double(0)
_lib.check(ctx.lib.mmw_diag_mfma_peak(ctx.handle, kind, ct.byref(v)))
print(f"kind {kind}: {v.value:.1f} TFLOP/s (MFMA flops only)")
