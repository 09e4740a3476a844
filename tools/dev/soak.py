"""Developer tool (GPU box): sustained parity check -- many full-size frames through queued 8-frame launches, every frame compared
with the oracle afterwards (catches timing-dependent faults such as the store data hazard, DESIGN.md 4, finding 6).
usage: python3 tools/dev/soak.py [launches]"""
import sys, time
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import torch
import vfgs_testlib as T
from versatilefilmgrain_amd import hw

launches = int(sys.argv[1]) if len(sys.argv) > 1 else 6
h = hw.VfgsHip(device=0)
st = torch.cuda.current_stream().cuda_stream
bad = 0
for name, w, hh in [("fgs_sei_10_420", 7680, 4320), ("fgs_afgs1_test1_8_444", 3840, 2160), ("fgs_sei_ar_test1_8_420", 3840, 2160),
                    ("fgs_sei_10_420", 1920, 1080), ("fgs_sei_ff_test6_8_422", 1928, 1080)]:
    h.lib.vfgs_hip_reset_state()
    rec = T.load_trace(name)
    T.replay(h, rec)
    ora = T.OracleHW(); T.replay(ora, rec)
    depth, sx, sy = T.trace_geometry(rec)
    dt = torch.int16 if depth > 8 else torch.uint8
    sz = 2 if depth > 8 else 1
    batch = 8
    nb = (w + 15) // 16
    stride, cstride = nb * 16, (nb * 16 // sx + 15) // 16 * 16
    g = torch.Generator(device="cuda").manual_seed(7)
    mk = lambda r, c: torch.randint(0, 1 << depth, (batch, r, c), dtype=torch.int32, device="cuda", generator=g).to(dt)
    sets = [(mk(hh, stride), mk(hh // sy, cstride), mk(hh // sy, cstride)) for _ in range(launches)]
    src = [tuple(t.cpu().numpy() for t in s_) for s_ in sets]
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for Y, U, V in sets:      # all launches queued back to back
        h.add_grain_frames_dev(Y.data_ptr(), U.data_ptr(), V.data_ptr(), w, hh, stride, cstride, batch, Y[0].numel() * sz, U[0].numel() * sz, st)
    torch.cuda.synchronize()
    gpu_t = time.perf_counter() - t0
    nbad = 0
    for (Y, U, V), (sYa, sUa, sVa) in zip(sets, src):
        gY, gU, gV = Y.cpu().numpy(), U.cpu().numpy(), V.cpu().numpy()
        for f in range(batch):
            fr = T.Frame(w, hh, depth, sx, sy, stride=stride, cstride=cstride)
            npd = np.uint16 if depth > 8 else np.uint8
            fr.Y[:hh] = sYa[f].view(npd); fr.U[:hh // sy] = sUa[f].view(npd); fr.V[:hh // sy] = sVa[f].view(npd)
            ora.add_grain_frame(fr)
            ok = np.array_equal(fr.Y[:hh], gY[f].view(npd)) and np.array_equal(fr.U[:hh // sy], gU[f].view(npd)) and np.array_equal(fr.V[:hh // sy], gV[f].view(npd))
            nbad += not ok
    print(f"{name} {w}x{hh}: {launches * batch} frames, {nbad} differ from the oracle (GPU {gpu_t * 1e3:.1f} ms)", flush=True)
    assert h.seed_state() == ora.seed_state()
    bad += nbad
print("soak:", "FAILED" if bad else "ok")
sys.exit(1 if bad else 0)
