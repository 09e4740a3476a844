#!/usr/bin/env python3
"""Developer tool (GPU box): consistency soak of the persistent luma workgroups.  The same frames from the same seed registers through
(a) ONE batch call (persistent workgroups: several tasks per workgroup, the parameter table refilled behind a barrier) and (b) one
call per frame (ordinary workgroups) must give the same bytes; repeated with fresh random frames.  A race on the parameter table or a
task handed out twice / not at all shows up as a difference; tests/ and bench.py check single launches against the oracle, this
checks thousands of them against the other launch shape on the GPU itself."""
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import torch  # noqa: E402
import vfgs_testlib as T  # noqa: E402
from versatilefilmgrain_amd import hw  # noqa: E402


def main():
    seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
    h = hw.VfgsHip(device=0)
    st = torch.cuda.current_stream().cuda_stream
    shapes = [("fgs_sei_10_420", 1920, 1080, 32, 2, 2), ("fgs_sei_10_420", 1920, 1080, 8, 2, 2), ("fgs_sei_10_420", 3840, 2160, 8, 2, 2),
              ("fgs_sei_10_444", 1280, 720, 24, 1, 1), ("fgs_sei_10_422", 1920, 1080, 16, 2, 1)]
    g = torch.Generator(device="cuda").manual_seed(5)
    t_end, it, bad, persisted = time.time() + seconds, 0, 0, 0
    while time.time() < t_end:
        name, w, hh, n, sx, sy = shapes[it % len(shapes)]
        rec = T.load_trace(name)
        cw, ch = w // sx, hh // sy
        mk = lambda r, c: torch.randint(0, 1024, (n, r, c), dtype=torch.int32, device="cuda", generator=g).to(torch.int16)
        Y, U, V = mk(hh, w), mk(ch, cw), mk(ch, cw)
        Y2, U2, V2 = Y.clone(), U.clone(), V.clone()
        h.lib.vfgs_hip_reset_state(); T.replay(h, rec); h.set_seed(1000 + it)
        h.add_grain_frames_dev(Y.data_ptr(), U.data_ptr(), V.data_ptr(), w, hh, w, cw, n, Y[0].numel() * 2, U[0].numel() * 2, st)
        persisted += h.last_launch_info()["persistent_luma_workgroups"] > 0
        s1 = h.seed_state()
        h.lib.vfgs_hip_reset_state(); T.replay(h, rec); h.set_seed(1000 + it)
        for f in range(n):
            h.add_grain_frame_dev(Y2[f].data_ptr(), U2[f].data_ptr(), V2[f].data_ptr(), w, hh, w, cw, st)
        torch.cuda.synchronize()
        ok = torch.equal(Y, Y2) and torch.equal(U, U2) and torch.equal(V, V2) and s1 == h.seed_state()
        bad += not ok
        it += 1
    print(f"persist_soak: {it} iterations in {seconds:.0f} s, {persisted} batch launches with persistent workgroups, {bad} mismatches")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
