"""Parity of the launch shapes the benchmarks time (`-m gpu`): full-size frames, 8 per launch, several launches queued
back to back, in place, every frame of every launch compared with the oracle afterwards.

A small test cannot see faults that need a full chip: the store-data hazard of DESIGN.md 4 (finding 6) corrupted one dword
in 16 of a wave's units only in waves that ran next to another workgroup on their CU.  The 4320p case goes through
vfgs_hip_add_grain_frames_part_dev -- the call bench.py times (reference semantics: vfgs_hw.c:288-312 once per line)."""
import numpy as np
import pytest

import vfgs_testlib as T

pytestmark = pytest.mark.gpu

CASES = [
    # trace, width, height, launches, entry point
    ("fgs_sei_10_420", 7680, 4320, 2, "part"),          # bench.py's shape: 7680x4320 10-bit 4:2:0, fgs_sei, 8 frames per launch
    ("fgs_afgs1_test1_8_444", 3840, 2160, 2, "frames"),  # BASELINE config 4
    ("fgs_afgs1_test1_8_420", 3840, 2160, 2, "frames"),  # the mainstream AFGS1 case (8-bit 4:2:0, vfgs_hw.c:352-362)
    ("fgs_sei_8_420", 3840, 2160, 1, "frames"),          # 8-bit 4:2:0 with per-sample pattern selection
]


@pytest.mark.parametrize("name,w,hh,launches,entry", CASES, ids=[f"{c[0]}_{c[1]}x{c[2]}" for c in CASES])
def test_queued_full_size_batches_equal_oracle(name, w, hh, launches, entry):
    import torch
    from versatilefilmgrain_amd import hw

    h = hw.VfgsHip(device=0)
    st = torch.cuda.current_stream().cuda_stream
    rec = T.load_trace(name)
    T.replay(h, rec)
    ora = T.OracleHW()
    T.replay(ora, rec)
    depth, sx, sy = T.trace_geometry(rec)
    dt = torch.int16 if depth > 8 else torch.uint8
    npd = np.uint16 if depth > 8 else np.uint8
    sz = 2 if depth > 8 else 1
    batch = 8
    stride, cstride = w, w // sx
    g = torch.Generator(device="cuda").manual_seed(11)
    mk = lambda r, c: torch.randint(0, 1 << depth, (batch, r, c), dtype=torch.int32, device="cuda", generator=g).to(dt)
    sets = [(mk(hh, stride), mk(hh // sy, cstride), mk(hh // sy, cstride)) for _ in range(launches)]
    src = [tuple(t.cpu().numpy().view(npd) for t in s_) for s_ in sets]
    torch.cuda.synchronize()
    for Y, U, V in sets:      # all launches queued back to back, nothing in between
        if entry == "part":
            h.add_grain_frames_part_dev(Y.data_ptr(), U.data_ptr(), V.data_ptr(), w, hh, 0, hh, stride, cstride, batch,
                                        Y[0].numel() * sz, U[0].numel() * sz, st)
        else:
            h.add_grain_frames_dev(Y.data_ptr(), U.data_ptr(), V.data_ptr(), w, hh, stride, cstride, batch,
                                   Y[0].numel() * sz, U[0].numel() * sz, st)
    torch.cuda.synchronize()
    bad = []
    for li, ((Y, U, V), (sY, sU, sV)) in enumerate(zip(sets, src)):
        gY, gU, gV = (t.cpu().numpy().view(npd) for t in (Y, U, V))
        for f in range(batch):
            fr = T.Frame(w, hh, depth, sx, sy, stride=stride, cstride=cstride)
            fr.Y[:hh], fr.U[:hh // sy], fr.V[:hh // sy] = sY[f], sU[f], sV[f]
            ora.add_grain_frame(fr)
            if not (np.array_equal(fr.Y[:hh], gY[f]) and np.array_equal(fr.U[:hh // sy], gU[f]) and np.array_equal(fr.V[:hh // sy], gV[f])):
                bad.append((li, f))
    assert not bad, f"frames (launch, index) that differ from the oracle: {bad}"
    assert h.seed_state() == ora.seed_state()
