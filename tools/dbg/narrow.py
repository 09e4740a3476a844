import sys
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np, torch
import vfgs_testlib as T
from gpu_util import DevFrame, stream_ptr
from versatilefilmgrain_amd import hw
w, h = int(sys.argv[1]), int(sys.argv[2])
hip = hw.VfgsHip(device=0)
rec = T.load_trace(sys.argv[3] if len(sys.argv) > 3 else "fgs_sei_ff_test6_10_420")
T.replay(hip, rec)
ora = T.OracleHW(); T.replay(ora, rec)
f, _ = T.lcg_frames(w, h, 10, 2, 2, 1)
want = f[0].copy()
d = DevFrame(f[0])
hip.add_grain_frame_dev(d.Y.data_ptr(), d.U.data_ptr(), d.V.data_ptr(), w, h, f[0].stride, f[0].cstride, stream_ptr())
ora.add_grain_frame(want)
got = d.download()
for name, a, b in zip("YUV", got.planes(), want.planes()):
    bad = np.argwhere(a != b)
    print(name, a.shape, "mismatches", len(bad))
    if len(bad):
        rows = np.unique(bad[:, 0]); cols = np.unique(bad[:, 1])
        print("  rows", rows[:40], "... n", len(rows)); print("  rows mod", np.unique(rows % (16 if name == "Y" else 8)))
        print("  cols", cols[:60], "n", len(cols))
        print("  first", bad[:10].tolist(), [(int(a[tuple(x)]), int(b[tuple(x)])) for x in bad[:10]])
