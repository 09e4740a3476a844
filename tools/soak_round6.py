#!/usr/bin/env python3
"""Developer utility (GPU box): randomised parity soak of what is new in round 6 -- the packed 16-bit form of the 8-bit one-pattern
kernels (scale LUTs with random maxima on both sides of the form's range proof, every scale shift, extreme patterns) and chained batches of
stripes whose LFSR windows are reached by jumps (random rank splits) -- mixed with round 5's frame / batch / frame-list calls in ONE seed
sequence per configuration.  Every frame of every call is compared with the oracle
(bit-exact, seed registers after every call).  Time bounded; one progress line every ~15 s; exit code 1 on any difference.
  python3 tools/soak_round6.py [seconds] [seed]"""
import sys, time
from pathlib import Path
import numpy as np
import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import vfgs_testlib as T
from versatilefilmgrain_amd import hw

TRACES = ["fgs_afgs1_test1_8_444", "fgs_afgs1_test1_8_420", "fgs_sei_8_420", "fgs_sei_8_422", "fgs_afgs1_test1_8_440", "fgs_sei_ff_test6_8_444", "fgs_sei_ff_test6_8_422",
          "fgs_afgs1_test12_8_420", "fgs_sei_ar_test1_8_420", "fgs_sei_10_420", "fgs_sei_ar_test1_10_420", "fgs_afgs1_test1_10_444"]


def main():
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 2026)
    h = hw.VfgsHip(device=0)
    st = torch.cuda.current_stream().cuda_stream
    t_start = t_print = time.perf_counter()
    calls = frames_done = bad = jumps = 0
    kinds = {}
    while time.perf_counter() - t_start < budget:
        name = TRACES[rng.integers(len(TRACES))]
        rec = T.load_trace(name)
        h.lib.vfgs_hip_reset_state(); T.replay(h, rec)
        ora = T.OracleHW(); T.replay(ora, rec)
        depth, sx, sy = T.trace_geometry(rec)
        if depth == 8 and rng.random() < 0.6:
            # the packed 16-bit form multiplies in 16 bits while max(scale) * 127 + 2^(shift-1) <= 32767 (vfgs_host.cpp image_form): LUTs on both sides of it
            shift = int(rng.integers(2, 8))
            limit = (32767 - (1 << (shift + 5))) // 127
            top = int(min(255, max(1, limit + rng.integers(-3, 4)))) if rng.random() < 0.7 else int(rng.integers(1, 256))
            for obj in (h, ora):
                obj.set_scale_shift(shift)
            for c in range(3):
                lut = rng.integers(0, top + 1, 256).astype(np.uint8)
                lut[rng.integers(0, 256, 24)] = top
                for obj in (h, ora):
                    obj.set_scale_lut(c, lut.tobytes())
            if rng.random() < 0.5:
                P = rng.choice(np.array([-127, 127, 126, -126, -1, 0, 1, 90], dtype=np.int8), size=4096).astype(np.int8)
                for obj in (h, ora):
                    for k in range(2):
                        obj.set_luma_pattern(k, P.tobytes()); obj.set_chroma_pattern(k, P[::-1].tobytes())
        wide = rng.random() < 0.12
        w = int(rng.integers(8208 // 8, 16400 // 8)) * 8 if wide else int(rng.integers(136 // 8, 2600 // 8)) * 8
        hh = int(rng.integers(16, 40 if wide else 400))
        if sy == 2:
            hh += hh & 1
        for _ in range(int(rng.integers(2, 6))):          # several calls in one seed sequence
            kind = ["frame", "frames", "list", "list_copy", "list_copy8", "parts", "parts", "parts"][rng.integers(8)]
            if kind == "parts" and (wide or hh < 96):
                kind = "frames"
            if kind == "list_copy8" and depth != 10:
                kind = "list"
            n = 1 if kind == "frame" else int(rng.integers(1, 6 if wide else 41))
            fr = []
            for i in range(n):
                f = T.Frame(w, hh, depth, sx, sy)
                for p in f.planes():
                    p[...] = rng.integers(0, (1 << 16) - 2 if depth > 8 else 256, p.shape).astype(f.dtype)
                fr.append(f)
            want = [f.copy() for f in fr]
            for x in want:
                ora.add_grain_frame(x)
            f0 = fr[0]
            up = lambda a: torch.from_numpy(a.view(np.uint8).copy()).cuda()
            ok = True
            if kind == "parts":
                # one rank's stripe of n frames per call, three chained calls (the second and third continue the chain of jumps)
                ranks = int(rng.integers(3, 9)); nbr = (hh + 15) // 16; rows = -(-nbr // ranks); rank = int(rng.integers(0, ranks))
                py = min(rank * rows * 16, (nbr - 1) * 16); ph = min(rows * 16, hh - py)
                cy0, cy1 = py // sy, -(-(py + ph) // sy)
                for rep in range(3):
                    if rep:
                        fr = []
                        for i in range(n):
                            f = T.Frame(w, hh, depth, sx, sy)
                            for p in f.planes():
                                p[...] = rng.integers(0, (1 << 16) - 2 if depth > 8 else 256, p.shape).astype(f.dtype)
                            fr.append(f)
                        want = [f.copy() for f in fr]
                        for x in want:
                            ora.add_grain_frame(x)
                    Y = up(np.stack([x.Y[py:py + ph] for x in fr])); U = up(np.stack([x.U[cy0:cy1] for x in fr])); V = up(np.stack([x.V[cy0:cy1] for x in fr]))
                    h.add_grain_frames_part_dev(Y.data_ptr(), U.data_ptr(), V.data_ptr(), w, hh, py, ph, f0.stride, f0.cstride, n, Y[0].numel(), U[0].numel(), st)
                    torch.cuda.synchronize()
                    for i, x in enumerate(want):
                        ok &= np.array_equal(Y[i].cpu().numpy().view(x.dtype).reshape(ph, -1), x.Y[py:py + ph])
                        ok &= np.array_equal(U[i].cpu().numpy().view(x.dtype).reshape(cy1 - cy0, -1), x.U[cy0:cy1])
                        ok &= np.array_equal(V[i].cpu().numpy().view(x.dtype).reshape(cy1 - cy0, -1), x.V[cy0:cy1])
                    ok &= h.seed_state() == ora.seed_state()
                    jumps = jumps + int(h.stripe_stream_stats()["last_launch_used_it"])
                    frames_done += n if rep else 0
            elif kind in ("frame", "frames"):
                Y = up(np.stack([x.Y for x in fr])); U = up(np.stack([x.U for x in fr])); V = up(np.stack([x.V for x in fr]))
                if kind == "frame":
                    h.add_grain_frame_dev(Y.data_ptr(), U.data_ptr(), V.data_ptr(), w, hh, f0.stride, f0.cstride, st)
                else:
                    h.add_grain_frames_dev(Y.data_ptr(), U.data_ptr(), V.data_ptr(), w, hh, f0.stride, f0.cstride, n, Y[0].numel(), U[0].numel(), st)
                torch.cuda.synchronize()
                for i, x in enumerate(want):
                    ok &= all(np.array_equal(t[i].cpu().numpy().view(x.dtype).reshape(p.shape), p) for t, p in zip((Y, U, V), x.planes()))
            else:
                order = rng.permutation(n)
                dev = [None] * n
                for i in order:
                    dev[i] = tuple(up(p) for p in fr[i].planes())
                src = [tuple(t.data_ptr() for t in d) for d in dev]
                if kind == "list":
                    h.add_grain_frame_list_dev(src, w, hh, f0.stride, f0.cstride, st)
                    torch.cuda.synchronize()
                    for d, x in zip(dev, want):
                        ok &= all(np.array_equal(t.cpu().numpy().view(x.dtype).reshape(p.shape), p) for t, p in zip(d, x.planes()))
                else:
                    f8 = T.Frame(w, hh, 8, sx, sy) if kind == "list_copy8" else f0
                    dst = [tuple(torch.full((p.size * p.itemsize,), 0x5a, dtype=torch.uint8, device="cuda") for p in f8.planes()) for _ in range(n)]
                    dp = [tuple(t.data_ptr() for t in d) for d in dst]
                    if kind == "list_copy":
                        h.add_grain_frame_list_copy_dev(src, dp, w, hh, f0.stride, f0.cstride, st)
                    else:
                        h.add_grain_frame_list_copy8_dev(src, dp, w, hh, f0.stride, f0.cstride, f8.stride, f8.cstride, st)
                    torch.cuda.synchronize()
                    cols = (w + 15) // 16 * 16
                    for d, x in zip(dst, want):
                        for t, p, (rows, cc) in zip(d, x.planes(), ((hh, cols), ((hh + sy - 1) // sy, cols // sx), ((hh + sy - 1) // sy, cols // sx))):
                            if kind == "list_copy":
                                g = t.cpu().numpy().view(x.dtype).reshape(p.shape)
                                ok &= np.array_equal(g[:rows, :cc], p[:rows, :cc]) and bool((g[rows:].view(np.uint8) == 0x5a).all()) and bool((g[:, cc:].view(np.uint8) == 0x5a).all())
                            else:
                                g = t.cpu().numpy().reshape(f8.Y.shape if t is d[0] else f8.U.shape)
                                exp = ((p[:rows, :cc].astype(np.int32) + 2) >> 2).astype(np.uint8)
                                ok &= np.array_equal(g[:rows, :cc], exp) and bool((g[rows:] == 0x5a).all()) and bool((g[:, cc:] == 0x5a).all())
                    for d, x in zip(dev, fr):          # the sources are untouched
                        ok &= all(np.array_equal(t.cpu().numpy().view(x.dtype).reshape(p.shape), p) for t, p in zip(d, x.planes()))
            ok &= h.seed_state() == ora.seed_state()
            calls += 1; frames_done += n
            kinds[kind] = kinds.get(kind, 0) + 1
            if not ok:
                bad += 1
                print(f"DIFFERENCE: {name} {w}x{hh} {kind} x{n}", flush=True)
        if time.perf_counter() - t_print > 15:
            t_print = time.perf_counter()
            print(f"{t_print - t_start:6.0f} s: {calls} calls, {frames_done} frames, {bad} calls with differences", flush=True)
    print(f"soak: {calls} calls {kinds} ({jumps} stripe launches read jumped segments), {frames_done} frames, " + ("ok" if bad == 0 else f"{bad} CALLS WITH DIFFERENCES"))
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
