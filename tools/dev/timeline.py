"""Developer tool (GPU box): average wave timeline of the grain kernel (luma waves), from a build of the instrumented kernel
source tools/dev/vfgs_kernel_timeline.hip.txt (s_memrealtime marks, 100 MHz).  usage: VFGS_LIB=<lib> python3 tools/dev/timeline.py"""
import ctypes as C, os, sys, json
from pathlib import Path
ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import torch
import vfgs_testlib as T
from versatilefilmgrain_amd import hw
hw.load(os.environ["VFGS_LIB"])
h = hw.VfgsHip(device=0)
T.replay(h, T.load_trace(os.environ.get("TL_TRACE", "fgs_sei_10_420")))
w, hh, batch = (int(x) for x in os.environ.get("TL_SHAPE", "7680x4320x8").split("x"))
g = torch.Generator(device="cuda").manual_seed(3)
mk = lambda r, c: torch.randint(0, 1024, (batch, r, c), dtype=torch.int32, device="cuda", generator=g).to(torch.int16)
npool = max(3, min(16, int(1.2e9 // (w * hh * 3 * batch)) + 1))
sets = [(mk(hh, w), mk(hh // 2, w // 2), mk(hh // 2, w // 2)) for _ in range(npool)]
st = torch.cuda.current_stream().cuda_stream
def step(i):
    Y, U, V = sets[i % npool]
    h.add_grain_frames_dev(Y.data_ptr(), U.data_ptr(), V.data_ptr(), w, hh, w, w // 2, batch, Y[0].numel() * 2, U[0].numel() * 2, st)
for i in range(200): step(i)
torch.cuda.synchronize()
out = (C.c_ulonglong * 16)()
h.lib.vfgs_hip_debug_timeline(out, 1)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for i in range(50): step(i)
e1.record(); torch.cuda.synchronize()
h.lib.vfgs_hip_debug_timeline(out, 0)
n = out[0]
names = {7: "geometry done (no load issued yet)", 8: "all loads issued", 1: "image + block parameters in LDS", 2: "after barrier", 4: "first row done", 5: "last row done", 6: "stores drained"}
print(f"{w}x{hh} x {batch} frames per launch: launch us", round(e0.elapsed_time(e1) / 50 * 1e3, 2), "sampled luma waves", n)
for i in (7, 8, 1, 2, 4, 5, 6):
    print(f"  {names[i]:32s} {out[i] / max(n, 1) * 0.01:8.2f} us after wave start")
