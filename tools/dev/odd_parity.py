#!/usr/bin/env python3
"""Developer helper (GPU box): parity of 8-bit 4:2:x pictures with an ODD number of blocks per line against the oracle, for a
library variant (VFGS_LIB).  Widths around position / group boundaries, several heights, garbage in the stride padding."""
import os, sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import numpy as np
import torch
import vfgs_testlib as T
from gpu_util import DevFrame, stream_ptr
from versatilefilmgrain_amd import hw
if os.environ.get("VFGS_LIB"):
    hw.load(os.environ["VFGS_LIB"])
h = hw.VfgsHip(device=0)
bad = 0
for name in ("fgs_afgs1_test1_8_420", "fgs_sei_8_420", "fgs_sei_ff_test6_8_422"):
    rec = T.load_trace(name)
    depth, sx, sy = T.trace_geometry(rec)
    for width in (144, 176, 496, 528, 720, 1008, 1040, 1968, 2032, 2064, 3856, 4080, 4112, 8176):
        assert ((width + 15) // 16) % 2 == 1
        for height in (16, 17, 33, 70):
            h.lib.vfgs_hip_reset_state(); T.replay(h, rec)
            ora = T.OracleHW(); T.replay(ora, rec)
            rng = np.random.default_rng(width + height)
            f = T.Frame(width, height, depth, sx, sy)
            for p in f.planes():
                p[...] = rng.integers(0, 256, p.shape).astype(f.dtype)
            want = f.copy(); ora.add_grain_frame(want)
            d = DevFrame(f)
            h.add_grain_frame_dev(*d.ptrs(), f.width, f.height, f.stride, f.cstride, stream_ptr())
            ok = d.download().equal_all(want) and h.seed_state() == ora.seed_state()
            bad += not ok
            if not ok: print("DIFF", name, width, height)
print("odd-block parity:", "ok" if not bad else f"{bad} differences")
sys.exit(1 if bad else 0)
