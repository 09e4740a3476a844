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
    ("fgs_sei_10_420", 7680, 4320, 3, "part_region"),   # the same launches inside an overlap region (two run at a time, one frame front each)
    ("fgs_sei_10_420", 7680, 4320, 1, "part3"),         # odd frame count: the two frame fronts of one launch, last front half empty
    ("fgs_afgs1_test1_8_420", 7680, 4320, 1, "part3"),  # two frame fronts in the 8-bit one-pattern kernels (negated banks, five waves per SIMD)
    ("fgs_sei_10_420", 7680, 4320, 1, "copy3"),         # two frame fronts, out of place
    ("fgs_afgs1_test1_8_444", 3840, 2160, 2, "frames"),  # BASELINE config 4
    ("fgs_afgs1_test1_8_420", 3840, 2160, 2, "frames"),  # the mainstream AFGS1 case (8-bit 4:2:0, vfgs_hw.c:352-362)
    ("fgs_sei_8_420", 3840, 2160, 1, "frames"),          # 8-bit 4:2:0 with per-sample pattern selection
    # round 4: the persistent luma workgroups at full-chip scale (the timed shapes of bench.py's `configs` leg and of the 2160p SEI case)
    ("fgs_sei_10_420", 1920, 1080, 2, "frames32"),       # BASELINE config 1 at 32 frames per launch: 5 tasks per persistent workgroup
    ("fgs_sei_10_420", 1920, 1080, 3, "frames"),         # ... at 8: two tasks each
    ("fgs_sei_10_420", 3840, 2160, 1, "frames"),         # general-form luma at 2160p: 5 tasks each
    ("fgs_sei_10_420", 1920, 1080, 2, "frames32_region"),  # ... inside an overlap region (two persistent launches at a time)
    ("fgs_sei_ar_test1_10_420", 3840, 2160, 1, "copy8_8"),  # fused 8-bit output, one-pattern forms, the timed shape
]


@pytest.mark.parametrize("name,w,hh,launches,entry", CASES, ids=[f"{c[0]}_{c[1]}x{c[2]}_{c[4]}" for c in CASES])
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
    batch = 3 if entry in ("part3", "copy3") else (32 if entry.startswith("frames32") else 8)
    stride, cstride = w, w // sx
    g = torch.Generator(device="cuda").manual_seed(11)
    mk = lambda r, c: torch.randint(0, 1 << depth, (batch, r, c), dtype=torch.int32, device="cuda", generator=g).to(dt)
    sets = [(mk(hh, stride), mk(hh // sy, cstride), mk(hh // sy, cstride)) for _ in range(launches)]
    src = [tuple(t.cpu().numpy().view(npd) for t in s_) for s_ in sets]
    torch.cuda.synchronize()
    if entry in ("part_region", "frames32_region"):
        h.overlap_begin(st)
    out8 = []
    for Y, U, V in sets:      # all launches queued back to back, nothing in between
        if entry == "copy8_8":
            d8 = tuple(torch.zeros(t.shape, dtype=torch.uint8, device="cuda") for t in (Y, U, V))
            h.add_grain_copy8_dev(Y.data_ptr(), U.data_ptr(), V.data_ptr(), d8[0].data_ptr(), d8[1].data_ptr(), d8[2].data_ptr(), w, hh, 0, hh,
                                  stride, cstride, stride, cstride, batch, Y[0].numel() * sz, U[0].numel() * sz, d8[0][0].numel(), d8[1][0].numel(), st)
            out8.append(d8)
        elif entry == "copy3":
            dst = tuple(torch.zeros_like(t) for t in (Y, U, V))
            h.add_grain_copy_dev(Y.data_ptr(), U.data_ptr(), V.data_ptr(), dst[0].data_ptr(), dst[1].data_ptr(), dst[2].data_ptr(), w, hh, 0, hh,
                                 stride, cstride, batch, Y[0].numel() * sz, U[0].numel() * sz, st)
            torch.cuda.synchronize()
            assert all(torch.equal(t, torch.from_numpy(s_.view(np.int16 if depth > 8 else np.uint8)).cuda()) for t, s_ in zip((Y, U, V), src[0])), "source changed"
            Y.copy_(dst[0]); U.copy_(dst[1]); V.copy_(dst[2])
        elif entry in ("part", "part3", "part_region"):
            h.add_grain_frames_part_dev(Y.data_ptr(), U.data_ptr(), V.data_ptr(), w, hh, 0, hh, stride, cstride, batch,
                                        Y[0].numel() * sz, U[0].numel() * sz, st)
        else:
            h.add_grain_frames_dev(Y.data_ptr(), U.data_ptr(), V.data_ptr(), w, hh, stride, cstride, batch,
                                   Y[0].numel() * sz, U[0].numel() * sz, st)
    if entry in ("part_region", "frames32_region"):
        h.overlap_end(st)
    torch.cuda.synchronize()
    if entry.startswith("frames") and name == "fgs_sei_10_420" and w <= 3840:
        assert h.last_launch_info()["persistent_luma_workgroups"] > 0
    bad = []
    for li, ((Y, U, V), (sY, sU, sV)) in enumerate(zip(sets, src)):
        gY, gU, gV = (t.cpu().numpy().view(npd) for t in (Y, U, V))
        if out8:
            gY, gU, gV = (t.cpu().numpy() for t in out8[li])
        for f in range(batch):
            fr = T.Frame(w, hh, depth, sx, sy, stride=stride, cstride=cstride)
            fr.Y[:hh], fr.U[:hh // sy], fr.V[:hh // sy] = sY[f], sU[f], sV[f]
            ora.add_grain_frame(fr)
            if out8:      # yuv_to_8bit (yuv.c:216-258)
                fr.Y[...], fr.U[...], fr.V[...] = (fr.Y + 2) >> 2, (fr.U + 2) >> 2, (fr.V + 2) >> 2
            if not (np.array_equal(fr.Y[:hh], gY[f]) and np.array_equal(fr.U[:hh // sy], gU[f]) and np.array_equal(fr.V[:hh // sy], gV[f])):
                bad.append((li, f))
    assert not bad, f"frames (launch, index) that differ from the oracle: {bad}"
    assert h.seed_state() == ora.seed_state()


@pytest.mark.parametrize("name,w,hh,calls", [("fgs_sei_10_420", 7680, 4320, 5), ("fgs_sei_10_420", 1920, 1080, 9), ("fgs_afgs1_test1_8_420", 3840, 2160, 6)],
                         ids=["4320p", "1080p", "2160p_8bit"])
def test_overlap_region_one_frame_per_call_equals_oracle(name, w, hh, calls):
    """vfgs_hip_overlap_begin/_end: independent frames, one per call (the reference's call pattern, vfgs_main.c:771-790), run
    alternately on two internal streams.  The frames are produced on the caller's stream right before the region (the fork must
    order them) and read on it right after (the join must order that); results and seeds as without the region."""
    import torch
    from versatilefilmgrain_amd import hw

    h = hw.VfgsHip(device=0)
    rec = T.load_trace(name)
    T.replay(h, rec)
    ora = T.OracleHW()
    T.replay(ora, rec)
    depth, sx, sy = T.trace_geometry(rec)
    dt = torch.int16 if depth > 8 else torch.uint8
    npd = np.uint16 if depth > 8 else np.uint8
    g = torch.Generator(device="cuda").manual_seed(5)
    mk = lambda r, c: torch.randint(0, 1 << depth, (calls, r, c), dtype=torch.int32, device="cuda", generator=g).to(dt)
    Y0, U0, V0 = mk(hh, w), mk(hh // sy, w // sx), mk(hh // sy, w // sx)
    src = tuple(t.cpu().numpy().view(npd) for t in (Y0, U0, V0))
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        st = side.cuda_stream
        Y, U, V = Y0.clone(), U0.clone(), V0.clone()          # producer on the caller's stream: the region must wait for it
        h.overlap_begin(st)
        for f in range(calls):
            h.add_grain_frame_dev(Y[f].data_ptr(), U[f].data_ptr(), V[f].data_ptr(), w, hh, w, w // sx, st)
        h.overlap_end(st)
        oY, oU, oV = Y.clone(), U.clone(), V.clone()          # consumer on the caller's stream: must see every frame finished
    side.synchronize()
    gY, gU, gV = (t.cpu().numpy().view(npd) for t in (oY, oU, oV))
    bad = []
    for f in range(calls):
        fr = T.Frame(w, hh, depth, sx, sy, stride=w, cstride=w // sx)
        fr.Y[:hh], fr.U[:hh // sy], fr.V[:hh // sy] = src[0][f], src[1][f], src[2][f]
        ora.add_grain_frame(fr)
        if not (np.array_equal(fr.Y[:hh], gY[f]) and np.array_equal(fr.U[:hh // sy], gU[f]) and np.array_equal(fr.V[:hh // sy], gV[f])):
            bad.append(f)
    assert not bad, f"frames that differ from the oracle: {bad}"
    assert h.seed_state() == ora.seed_state()


def test_overlap_region_errors():
    import torch
    from versatilefilmgrain_amd import hw

    h = hw.VfgsHip(device=0)
    st = torch.cuda.current_stream().cuda_stream
    with pytest.raises(hw.VfgsHipError):
        h.overlap_end(st)                 # nothing open
    h.overlap_begin(st)
    with pytest.raises(hw.VfgsHipError):
        h.overlap_begin(st)               # one region at a time
    other = torch.cuda.Stream()
    with pytest.raises(hw.VfgsHipError):
        h.overlap_end(other.cuda_stream)  # not the region's stream
    h.overlap_end(st)
    torch.cuda.synchronize()


def test_overlap_region_with_configuration_changes_between_frames():
    """A new model (banks, LUTs, seed: a per-picture SEI / AFGS1 message) before every frame of a region: the table and LFSR
    images uploaded for a frame on one internal stream must be complete before the kernel on the other one reads them, and an
    image still being read there must not be overwritten (the slot guards of vfgs_host.cpp)."""
    import torch
    from gpu_util import DevFrame
    from versatilefilmgrain_amd import hw

    h = hw.VfgsHip(device=0)
    names = ["fgs_sei_10_420", "fgs_sei_ff_test6_10_420", "fgs_afgs1_test1_10_420", "fgs_sei_ar_test1_10_420"] * 4
    frames, _ = T.lcg_frames(1936, 528, 10, 2, 2, len(names))
    devs = [DevFrame(f) for f in frames]
    torch.cuda.synchronize()
    ora = T.OracleHW()
    st = torch.cuda.current_stream().cuda_stream
    h.overlap_begin(st)
    for name, d, f in zip(names, devs, frames):
        rec = T.load_trace(name)
        T.replay(h, rec)
        T.replay(ora, rec)
        h.add_grain_frame_dev(d.Y.data_ptr(), d.U.data_ptr(), d.V.data_ptr(), f.width, f.height, f.stride, f.cstride, st)
        ora.add_grain_frame(f)
    h.overlap_end(st)
    torch.cuda.synchronize()
    for i, (d, f) in enumerate(zip(devs, frames)):
        assert d.download().equal_all(f), i
    assert h.seed_state() == ora.seed_state()


def test_lfsr_window_lookahead_across_refills():
    """A seed that is never reset over 128 frames of 7680x4320 consumes 16.6 Mbit of the LFSR stream: the device window grows to its
    full 1 MiB, the NEXT window is built ahead on the library's copy stream once a call reaches the second half, and a later call
    switches to it (vfgs_host.cpp StreamCache::ensure).  Every frame of every launch against the oracle; two launches are always in
    flight (the second one is queued before the first is checked), so a window swapped too early would be seen."""
    import torch
    from versatilefilmgrain_amd import hw

    h = hw.VfgsHip(device=0)
    st = torch.cuda.current_stream().cuda_stream
    rec = T.load_trace("fgs_sei_10_420")
    T.replay(h, rec)
    ora = T.OracleHW()
    T.replay(ora, rec)
    w, hh, batch, launches = 7680, 4320, 8, 16
    g = torch.Generator(device="cuda").manual_seed(23)
    mk = lambda r, c: torch.randint(0, 1024, (batch, r, c), dtype=torch.int32, device="cuda", generator=g).to(torch.int16)

    def launch():
        Y, U, V = mk(hh, w), mk(hh // 2, w // 2), mk(hh // 2, w // 2)
        src = tuple(t.cpu().numpy().view(np.uint16) for t in (Y, U, V))
        h.add_grain_frames_dev(Y.data_ptr(), U.data_ptr(), V.data_ptr(), w, hh, w, w // 2, batch, Y[0].numel() * 2, U[0].numel() * 2, st)
        return (Y, U, V), src

    bad = []
    pending = launch()
    for li in range(launches):
        nxt = launch() if li + 1 < launches else None
        (Y, U, V), (sY, sU, sV) = pending
        gY, gU, gV = (t.cpu().numpy().view(np.uint16) for t in (Y, U, V))
        for f in range(batch):
            fr = T.Frame(w, hh, 10, 2, 2, stride=w, cstride=w // 2)
            fr.Y[:hh], fr.U[:hh // 2], fr.V[:hh // 2] = sY[f], sU[f], sV[f]
            ora.add_grain_frame(fr)
            if not (np.array_equal(fr.Y[:hh], gY[f]) and np.array_equal(fr.U[:hh // 2], gU[f]) and np.array_equal(fr.V[:hh // 2], gV[f])):
                bad.append((li, f))
        pending = nxt
    assert not bad, f"frames (launch, index) that differ from the oracle: {bad}"
    assert h.seed_state() == ora.seed_state()
    stats = h.stream_stats()
    assert stats["windows_built_ahead"] >= 1 and stats["switches_to_built_ahead"] >= 1, stats


def test_overlap_region_mixed_with_host_memory_calls():
    """Inside a region the drop-in host-memory calls (vfgs_add_grain_stripe: complete on return, on the library's own stream) may
    be mixed with the device-pointer calls: frames are independent, the seed registers advance in CALL order whatever stream a
    frame runs on."""
    import torch
    from gpu_util import DevFrame
    from versatilefilmgrain_amd import hw

    h = hw.VfgsHip(device=0)
    rec = T.load_trace("fgs_sei_10_420")
    T.replay(h, rec)
    ora = T.OracleHW()
    T.replay(ora, rec)
    frames, _ = T.lcg_frames(1936, 272, 10, 2, 2, 9)
    want = [f.copy() for f in frames]
    devs = {i: DevFrame(f) for i, f in enumerate(frames) if i % 3 != 1}
    torch.cuda.synchronize()
    st = torch.cuda.current_stream().cuda_stream
    h.overlap_begin(st)
    for i, f in enumerate(frames):
        if i % 3 == 1:      # a host-memory frame in between
            h.add_grain_stripe(f.Y.ctypes.data, f.U.ctypes.data, f.V.ctypes.data, 0, f.width, f.height, f.stride, f.cstride)
        else:
            d = devs[i]
            h.add_grain_frame_dev(d.Y.data_ptr(), d.U.data_ptr(), d.V.data_ptr(), f.width, f.height, f.stride, f.cstride, st)
        ora.add_grain_frame(want[i])
    h.overlap_end(st)
    torch.cuda.synchronize()
    for i, f in enumerate(frames):
        got = f if i % 3 == 1 else devs[i].download()
        assert got.equal_all(want[i]), i
    assert h.seed_state() == ora.seed_state()
