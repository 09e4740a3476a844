"""Developer helper (GPU box): run one frame through the device path and the oracle, print where they differ.
usage: python tools/dev/mismatch.py <trace name> <width> <height>"""
import sys
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import torch
import vfgs_testlib as T
from gpu_util import DevFrame, stream_ptr
from versatilefilmgrain_amd import hw

name, W, H = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
import os
if os.environ.get('VFGS_LIB'): hw.load(os.environ['VFGS_LIB'])
hip = hw.VfgsHip(device=0)
hip.lib.vfgs_hip_reset_state()
rec = T.load_trace(name)
T.replay(hip, rec)
ora = T.OracleHW(); T.replay(ora, rec)
depth, sx, sy = T.trace_geometry(rec)
f, _ = T.lcg_frames(W, H, depth, sx, sy, 1)
want = f[0].copy()
pad0 = torch.zeros(int(os.environ.get('MM_PAD', '0')) + 16, dtype=torch.uint8, device='cuda')
d = DevFrame(f[0])
pad1 = torch.zeros(int(os.environ.get('MM_PAD', '0')) + 16, dtype=torch.uint8, device='cuda')
print('ptrs', hex(d.Y.data_ptr()), hex(d.U.data_ptr()), hex(d.V.data_ptr()))
if os.environ.get('MM_SYNC'): torch.cuda.synchronize()
hip.add_grain_frame_dev(d.Y.data_ptr(), d.U.data_ptr(), d.V.data_ptr(), f[0].width, f[0].height, f[0].stride, f[0].cstride, stream_ptr())
ora.add_grain_frame(want)
got = d.download()
for nm, a, b, src in zip("YUV", got.planes(), want.planes(), f[0].planes()):
    bad = np.argwhere(a != b)
    print(nm, "shape", a.shape, "mismatches", len(bad))
    if len(bad):
        rows = sorted(set(bad[:, 0].tolist())); cols = sorted(set(bad[:, 1].tolist()))
        print("  rows", rows[:20], "... n", len(rows))
        # column runs
        runs = []; s0 = cols[0]; prev = cols[0]
        for c in cols[1:]:
            if c != prev + 1: runs.append((s0, prev)); s0 = c
            prev = c
        runs.append((s0, prev))
        print("  col runs", runs[:40], "n", len(runs))
        r, c = bad[0]
        print("  first", r, c, "got", a[r, c], "want", b[r, c], "src", src[r, c])
        vals, cnt = np.unique(a[a != b], return_counts=True)
        order = np.argsort(-cnt)[:6]
        print("  got values at mismatches (most frequent)", {hex(int(vals[i])): int(cnt[i]) for i in order})
        r = bad[0][0]
        c0 = (bad[0][1] // 8) * 8
        print("  row", r, "cols", c0 - 8, "..", c0 + 16)
        print("   got ", a[r, c0 - 8:c0 + 16].tolist())
        print("   want", b[r, c0 - 8:c0 + 16].tolist())
        print("   src ", src[r, c0 - 8:c0 + 16].tolist())
        byrow = {}
        for rr, cc in bad: byrow.setdefault(int(rr), []).append(int(cc))
        for rr in sorted(byrow)[:12]: print("   row", rr, "n", len(byrow[rr]), "first cols", byrow[rr][:12])
        if nm == "V":
            worst = max(byrow, key=lambda k: len(byrow[k]))
            bpu = 16 // a.itemsize
            items = sorted(set((c // (256 * bpu), (c // (64 * bpu)) % 4, (c // bpu) % 64, (c % bpu) * a.itemsize // 4) for c in byrow[worst]))
            print("  worst row", worst, "(group of four positions, position, lane, dword):", items)
