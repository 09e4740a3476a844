#!/usr/bin/env python3
"""Developer helper (GPU box): 8-bit 4:2:0 with an ODD number of blocks per line (rows are not whole 16-byte units: the tiled
kernels with shifted accesses run, vfgs_kernel.hip has_shifted()).  VFGS_LIB / VFGS_ALLOW_DEV_BUILD select a variant."""
import os, sys, json
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import torch
import vfgs_testlib as T
from versatilefilmgrain_amd import hw
if os.environ.get("VFGS_LIB"):
    hw.load(os.environ["VFGS_LIB"])
h = hw.VfgsHip(device=0)
for trace in ("fgs_afgs1_test1_8_420", "fgs_sei_8_420"):
    h.lib.vfgs_hip_reset_state()
    T.replay(h, T.load_trace(trace))
    w, hh, stride, batch = 3856, 2160, 3872, 8
    mk = lambda r, c: torch.randint(0, 256, (batch, r, c), dtype=torch.int32, device="cuda").to(torch.uint8)
    sets = [(mk(hh, stride), mk(hh // 2, stride // 2), mk(hh // 2, stride // 2)) for _ in range(6)]
    st = torch.cuda.current_stream().cuda_stream
    def step(i):
        Y, U, V = sets[i % len(sets)]
        h.add_grain_frames_dev(Y.data_ptr(), U.data_ptr(), V.data_ptr(), w, hh, stride, stride // 2, batch, Y[0].numel(), U[0].numel(), st)
    for i in range(30): step(i)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(100): step(i)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 100 / batch
    print(json.dumps({"trace": trace, "width": w, "us_per_frame": round(us, 3), "frac_of_8TBps": round(2 * 1.5 * w * hh / us / 1e3 / 8000, 4)}))
