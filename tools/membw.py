#!/usr/bin/env python3
"""In-situ HBM ceilings of the device (copy / write / read, several launch shapes) via mmw_diag_membw."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mmwave_radar_processing_amd import _lib  # noqa: E402

ctx = _lib.Context(0)
L = ctx.lib
N = 8 << 30
buf = ctx.alloc(N)
names = {0: "copy", 1: "write", 2: "read", 3: "write_nt", 4: "write_x4", 5: "copy_x4"}
out = {}
for mode in (0, 1, 2, 3, 4, 5):
    for bpc in (4, 8, 16, 32, 64):
        blocks = 256 * bpc
        nb = N // 2 if mode in (0, 5) else N
        dst = buf.ptr + (N // 2 if mode in (0, 5) else 0)
        fn = lambda: _lib.check(L.mmw_diag_membw(ctx.handle, buf.ptr, dst, nb, mode, blocks))
        fn(); ctx.sync(); ctx.timer_start()
        for _ in range(5):
            fn()
        ms = ctx.timer_stop() / 5
        moved = 2 * nb if mode in (0, 5) else nb
        out[f"{names[mode]}_b{bpc}"] = round(moved / ms / 1e6)
print(json.dumps(out))
