#!/usr/bin/env python3
"""Developer utility (GPU box): sustained parity soak.  For every case: `launches` launches of 8 frames queued back to back
(no host synchronisation in between), plain and then again inside an overlap region, EVERY frame compared with the oracle
afterwards (bit-exact, seed registers included).  The GPU suite's soak tests check 1-3 launches per shape; this one keeps the
chip saturated for seconds.  Prints one line per case; exit code 1 on any difference.
  python3 tools/soak_parity.py [launches]"""
import sys, time
from pathlib import Path
import numpy as np
import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import vfgs_testlib as T
from versatilefilmgrain_amd import hw

CASES = [("fgs_sei_10_420", 7680, 4320), ("fgs_afgs1_test1_8_444", 3840, 2160), ("fgs_afgs1_test1_8_420", 3840, 2160),
         ("fgs_sei_ar_test1_8_420", 3840, 2160), ("fgs_sei_10_420", 1920, 1080), ("fgs_sei_ff_test6_8_422", 1952, 1080)]


def main():
    launches = int(sys.argv[1]) if len(sys.argv) > 1 else 12
    h = hw.VfgsHip(device=0)
    st = torch.cuda.current_stream().cuda_stream
    bad_total = 0
    for name, w, hh in CASES:
        for region in (False, True):
            rec = T.load_trace(name)
            h.lib.vfgs_hip_reset_state()
            T.replay(h, rec)
            ora = T.OracleHW(); T.replay(ora, rec)
            depth, sx, sy = T.trace_geometry(rec)
            dt, npd, sz = (torch.int16, np.uint16, 2) if depth > 8 else (torch.uint8, np.uint8, 1)
            g = torch.Generator(device="cuda").manual_seed(17)
            mk = lambda r, c: torch.randint(0, 1 << depth, (8, r, c), dtype=torch.int32, device="cuda", generator=g).to(dt)
            sets = [(mk(hh, w), mk(hh // sy, w // sx), mk(hh // sy, w // sx)) for _ in range(launches)]
            src = [tuple(t.cpu().numpy().view(npd) for t in s_) for s_ in sets]
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            if region:
                h.overlap_begin(st)
            for Y, U, V in sets:
                h.add_grain_frames_dev(Y.data_ptr(), U.data_ptr(), V.data_ptr(), w, hh, w, w // sx, 8, Y[0].numel() * sz, U[0].numel() * sz, st)
            if region:
                h.overlap_end(st)
            torch.cuda.synchronize()
            gpu_ms = (time.perf_counter() - t0) * 1e3
            bad = 0
            for (Y, U, V), (sY, sU, sV) in zip(sets, src):
                gY, gU, gV = (t.cpu().numpy().view(npd) for t in (Y, U, V))
                for f in range(8):
                    fr = T.Frame(w, hh, depth, sx, sy, stride=w, cstride=w // sx)
                    fr.Y[:hh], fr.U[:hh // sy], fr.V[:hh // sy] = sY[f], sU[f], sV[f]
                    ora.add_grain_frame(fr)
                    bad += not (np.array_equal(fr.Y[:hh], gY[f]) and np.array_equal(fr.U[:hh // sy], gU[f]) and np.array_equal(fr.V[:hh // sy], gV[f]))
            bad += h.seed_state() != ora.seed_state()
            bad_total += bad
            print(f"{name} {w}x{hh} {'overlap region' if region else 'plain':14s}: {launches * 8} frames, {bad} differ from the oracle (GPU {gpu_ms:.1f} ms)", flush=True)
            del sets, src
            torch.cuda.empty_cache()
    print("soak: ok" if bad_total == 0 else f"soak: {bad_total} DIFFERENCES")
    sys.exit(1 if bad_total else 0)


if __name__ == "__main__":
    main()
