"""GPU (MI355X): the HIP path, called through the C ABI, against the oracle and the golden
vectors of the real reference.  Bit-exact is the bar (integer path)."""
import json

import numpy as np
import pytest

import vfgs_testlib as T

pytestmark = pytest.mark.gpu

MD5 = json.loads((T.GOLDEN / "md5.json").read_text())
SUB = {"420": (2, 2), "422": (2, 1), "444": (1, 1), "440": (1, 2)}   # "440": csubx 1, csuby 2 (hw layer only; derived traces, make_golden.py)
W, H, N = 192, 144, 3


@pytest.fixture(scope="module")
def hip():
    import torch
    assert torch.cuda.is_available(), "these tests need the MI355X"
    from versatilefilmgrain_amd import hw
    h = hw.VfgsHip(device=0)
    info = h.device_info()
    assert "gfx950" in torch.cuda.get_device_properties(0).gcnArchName
    print(info)
    return h


def program(hip, name):
    hip.lib.vfgs_hip_reset_state()
    rec = T.load_trace(name)
    T.replay(hip, rec)
    ora = T.OracleHW()
    T.replay(ora, rec)
    return ora, T.trace_geometry(rec)


def host_frame_call(hip, f):
    hip.add_grain_stripe(f.Y.ctypes.data, f.U.ctypes.data, f.V.ctypes.data, 0, f.width, f.height, f.stride, f.cstride)


@pytest.mark.parametrize("name", sorted(MD5["small"]))
def test_small_frames_host_stripe_vs_golden_md5_and_oracle(hip, name):
    """Every cfg x depth x format: 3 frames 192x144 through vfgs_add_grain_stripe (host memory)."""
    ora, (depth, sx, sy) = program(hip, name)
    frames, _ = T.lcg_frames(W, H, depth, sx, sy, N)
    want = [f.copy() for f in frames]
    for f, w in zip(frames, want):
        host_frame_call(hip, f)
        ora.add_grain_frame(w)
        assert f.equal_all(w)
        assert hip.seed_state() == ora.seed_state()
    assert T.md5_frames(frames) == MD5["small"][name]


@pytest.mark.parametrize("name", ["fgs_sei_10_420", "fgs_afgs1_test1_8_444", "fgs_sei_ff_test6_8_422", "fgs_sei_ar_test1_8_420", "fgs_sei_ff_test6_10_440"])
def test_line_api_matches_oracle_line_by_line(hip, name):
    """The drop-in entry point itself: one call per line, host pointers, in order."""
    ora, (depth, sx, sy) = program(hip, name)
    f, _ = T.lcg_frames(W, H, depth, sx, sy, 1)
    a, b = f[0].copy(), f[0].copy()
    for fr, hwimpl in ((a, hip), (b, ora)):
        for y in range(fr.height):
            hwimpl.add_grain_line(fr.Y[y].ctypes.data, fr.U[y // sy].ctypes.data, fr.V[y // sy].ctypes.data, y, fr.width)
    assert a.equal_all(b)
    assert hip.seed_state() == ora.seed_state()


def test_line_api_arbitrary_order(hip):
    ora, (depth, sx, sy) = program(hip, "fgs_sei_10_420")
    f, _ = T.lcg_frames(W, H, depth, sx, sy, 1)
    a, b = f[0].copy(), f[0].copy()
    order = [0, 1, 16, 17, 5, 32, 32, 33, 48, 2, 64, 65, 80, 15, 16, 96]
    for fr, hwimpl in ((a, hip), (b, ora)):
        for y in order:
            hwimpl.add_grain_line(fr.Y[y].ctypes.data, fr.U[y // 2].ctypes.data, fr.V[y // 2].ctypes.data, y, fr.width)
    assert a.equal_all(b)
    assert hip.seed_state() == ora.seed_state()


@pytest.mark.parametrize("name", ["fgs_sei_10_420", "fgs_sei_8_422", "fgs_afgs1_test1_10_444", "fgs_sei_ff_test6_10_444", "fgs_sei_10_440", "fgs_afgs1_test1_8_440"])
@pytest.mark.parametrize("width,height", [(200, 152), (136, 130), (1000, 160)])
def test_ragged_sizes_garbage_padding_and_out_of_range(hip, name, width, height):
    """W % 16 != 0 (whole last block is processed in the stride padding, quirk 7), stripes that
    end mid block row, garbage everywhere incl. > 10-bit values (quirk 8)."""
    ora, (depth, sx, sy) = program(hip, name)
    f = T.Frame(width, height, depth, sx, sy)
    rng = np.random.default_rng(width)
    for p in f.planes():
        p[...] = rng.integers(0, 1 << (16 if depth > 8 else 8), p.shape).astype(f.dtype)
    a, b = f.copy(), f.copy()
    host_frame_call(hip, a)
    ora.add_grain_frame(b)
    assert a.equal_all(b)


@pytest.mark.parametrize("name", ["fgs_sei_ar_test1_8_420", "fgs_sei_8_422", "fgs_sei_ff_test6_8_422"])
@pytest.mark.parametrize("width", [208, 224, 720, 736, 2032, 2048])
def test_8bit_subsampled_even_and_odd_block_counts(hip, name, width):
    """8-bit planes with 8-sample blocks: rows are whole 16-byte units only for an even number of blocks per line.
    Even counts run the aligned kernels, odd ones (208 = 13 blocks, 720 = 45, 2032 = 127) the kernels with shifted
    accesses and partly valid lanes (vfgs_kernel.hip aligned_ok()); both against the oracle, on the device, with
    row pitches that are multiples of 16 bytes (the device entry points' rule) but not of a cache line."""
    from gpu_util import DevFrame, stream_ptr
    ora, (depth, sx, sy) = program(hip, name)
    assert depth == 8 and sx == 2
    nblk = (width + 15) // 16
    f = T.Frame(width, 70, depth, sx, sy, stride=nblk * 16 + 48, cstride=(nblk * 8 + 15) // 16 * 16 + 16)
    rng = np.random.default_rng(width)
    for p in f.planes():
        p[...] = rng.integers(0, 256, p.shape).astype(f.dtype)
    want = f.copy()
    ora.add_grain_frame(want)
    d = DevFrame(f)
    hip.add_grain_frame_dev(d.Y.data_ptr(), d.U.data_ptr(), d.V.data_ptr(), f.width, f.height, f.stride, f.cstride, stream_ptr())
    assert d.download().equal_all(want)
    assert hip.seed_state() == ora.seed_state()


@pytest.mark.parametrize("name,width", [("fgs_sei_10_420", 512), ("fgs_sei_10_420", 2048), ("fgs_sei_10_420", 4096), ("fgs_sei_10_420", 8192),
                                        ("fgs_afgs1_test1_10_444", 2048), ("fgs_sei_10_422", 4096), ("fgs_afgs1_test1_8_444", 1024),
                                        ("fgs_afgs1_test1_8_444", 4096), ("fgs_sei_ar_test1_8_420", 8192), ("fgs_sei_10_440", 6144)])
def test_rows_that_end_on_a_segment_or_tile_boundary(hip, name, width):
    """Aligned kernels: a wave computes lanes shifted by half a block against the 16-byte units it moves, so a row whose
    bytes are an exact multiple of 1 KiB (segment) or 4 KiB (tile) ends with a lane that lives alone in the NEXT segment /
    tile, whose wave moves nothing but the tail of the unit in front of it.  Two block rows + a ragged end, device, oracle."""
    from gpu_util import DevFrame, stream_ptr
    ora, (depth, sx, sy) = program(hip, name)
    f, _ = T.lcg_frames(width, 38 if sy == 1 else 40, depth, sx, sy, 1)
    want = f[0].copy()
    ora.add_grain_frame(want)
    d = DevFrame(f[0])
    hip.add_grain_frame_dev(d.Y.data_ptr(), d.U.data_ptr(), d.V.data_ptr(), f[0].width, f[0].height, f[0].stride, f[0].cstride, stream_ptr())
    assert d.download().equal_all(want)
    assert hip.seed_state() == ora.seed_state()


def test_stripes_equal_whole_frame(hip):
    """A frame fed as uneven host stripes (not multiples of 16) == line by line."""
    ora, (depth, sx, sy) = program(hip, "fgs_sei_10_420")
    f, _ = T.lcg_frames(320, 208, depth, sx, sy, 1)
    a, b = f[0].copy(), f[0].copy()
    y = 0
    for h in (6, 26, 32, 1, 15, 64, 64):
        hip.add_grain_stripe(a.Y[y].ctypes.data, a.U[y // 2].ctypes.data, a.V[y // 2].ctypes.data, y, a.width, h, a.stride, a.cstride)
        y += h
    assert y == 208
    ora.add_grain_frame(b)
    assert a.equal_all(b)
    assert hip.seed_state() == ora.seed_state()


@pytest.mark.parametrize("key", sorted(MD5["full"]))
def test_full_size_device_frames_md5(hip, key):
    """BASELINE.json configs at full size, device-resident, vs the reference CLI's md5."""
    from gpu_util import DevFrame, stream_ptr
    g = MD5["full"][key]
    sx, sy = SUB[g["format"]]
    ora, _ = program(hip, f'{g["cfg"]}_{g["depth"]}_{g["format"]}')
    frames, _ = T.lcg_frames(g["width"], g["height"], g["depth"], sx, sy, g["frames"])
    assert T.md5_frames(frames) == g["input_md5"]
    out = []
    for f in frames:
        d = DevFrame(f)
        hip.add_grain_frame_dev(*d.ptrs(), f.width, f.height, f.stride, f.cstride, stream_ptr())
        out.append(d.download())
    assert T.md5_frames(out) == g["output_md5"]


def test_batch_launch_equals_sequential(hip):
    import torch
    from gpu_util import stream_ptr
    ora, (depth, sx, sy) = program(hip, "fgs_sei_10_420")
    n, w, h = 5, 640, 368
    frames, _ = T.lcg_frames(w, h, depth, sx, sy, n)
    want = [f.copy() for f in frames]
    for f in want:
        ora.add_grain_frame(f)
    f0 = frames[0]
    Y = torch.from_numpy(np.stack([f.Y for f in frames]).view(np.uint8)).cuda()
    U = torch.from_numpy(np.stack([f.U for f in frames]).view(np.uint8)).cuda()
    V = torch.from_numpy(np.stack([f.V for f in frames]).view(np.uint8)).cuda()
    hip.add_grain_frames_dev(Y.data_ptr(), U.data_ptr(), V.data_ptr(), w, h, f0.stride, f0.cstride, n,
                             Y[0].numel(), U[0].numel(), stream_ptr())
    torch.cuda.synchronize()
    for i, f in enumerate(want):
        assert np.array_equal(Y[i].cpu().numpy().view(f.dtype).reshape(f.Y.shape), f.Y)
        assert np.array_equal(U[i].cpu().numpy().view(f.dtype).reshape(f.U.shape), f.U)
        assert np.array_equal(V[i].cpu().numpy().view(f.dtype).reshape(f.V.shape), f.V)
    assert hip.seed_state() == ora.seed_state()


def test_frame_parts_equal_whole_frame_and_keep_seeds(hip):
    """The multi-GPU split on one GPU: 3 uneven 16-aligned parts, each by a freshly seeded
    library state (as a separate rank would), assemble to the whole frame."""
    from gpu_util import DevFrame, stream_ptr
    ora, (depth, sx, sy) = program(hip, "fgs_sei_10_420")
    f, _ = T.lcg_frames(640, 368, depth, sx, sy, 2)
    want = [x.copy() for x in f]
    for x in want:
        ora.add_grain_frame(x)
    parts = [(0, 128), (128, 112), (240, 128)]
    devs = [DevFrame(x) for x in f]
    for (py, ph) in parts:
        program(hip, "fgs_sei_10_420")            # every "rank" starts from the same programmed state
        for d, x in zip(devs, f):
            hip.add_grain_frame_part_dev(*d.ptrs(py), x.width, x.height, py, ph, x.stride, x.cstride, stream_ptr())
        assert hip.seed_state() == ora.seed_state()
    for d, wnt in zip(devs, want):
        assert d.download().equal_all(wnt)


def test_out_of_place_equals_in_place_and_keeps_source(hip):
    from gpu_util import DevFrame, stream_ptr
    for name in ("fgs_sei_10_420", "fgs_afgs1_test1_8_444", "fgs_sei_ff_test6_8_422", "fgs_afgs1_test1_10_440"):
        ora, (depth, sx, sy) = program(hip, name)
        f, _ = T.lcg_frames(456, 304, depth, sx, sy, 1)
        f = f[0]
        want = f.copy()
        ora.add_grain_frame(want)
        src, dst = DevFrame(f), DevFrame(f)
        for t in (dst.Y, dst.U, dst.V):
            t.fill_(0x5a)
        hip.add_grain_copy_dev(*src.ptrs(), *dst.ptrs(), f.width, f.height, 0, f.height, f.stride, f.cstride, 1, 0, 0, stream_ptr())
        assert src.download().equal_all(f)                    # source untouched
        got = dst.download()
        assert got.equal_picture(want)
        # whole 16-sample blocks are written (quirk 7); everything else in dst stays as it was
        nb = (f.width + 15) // 16 * 16
        assert np.array_equal(got.Y[:f.height, :nb], want.Y[:f.height, :nb])
        assert (got.Y[:f.height, nb:].view(np.uint8) == 0x5a).all()
        assert (got.Y[f.height:].view(np.uint8) == 0x5a).all()
        assert hip.seed_state() == ora.seed_state()


def test_zero_scale_is_clip_only_at_full_size(hip):
    """Size-independent property on a 4320p frame: all-zero scale LUTs -> out = clip(in)
    (quirk 1: 10-bit full range clips at 1020), and a second pass changes nothing."""
    import torch
    from gpu_util import stream_ptr
    hip.lib.vfgs_hip_reset_state()
    hip.set_depth(10)
    w, h = 7680, 4320
    g = torch.Generator(device="cuda").manual_seed(1)
    Y = torch.randint(0, 1024, (h, w), dtype=torch.int16, device="cuda", generator=g)
    U = torch.randint(0, 1024, (h // 2, w // 2), dtype=torch.int16, device="cuda", generator=g)
    V = torch.randint(0, 1024, (h // 2, w // 2), dtype=torch.int16, device="cuda", generator=g)
    ref = [t.clamp(0, 1020).clone() for t in (Y, U, V)]
    for _ in range(2):
        hip.add_grain_frame_dev(Y.data_ptr(), U.data_ptr(), V.data_ptr(), w, h, w, w // 2, stream_ptr())
        torch.cuda.synchronize()
        for a, b in zip((Y, U, V), ref):
            assert torch.equal(a, b)


def test_unchanged_reference_cli_linked_against_hip_library(hip, tmp_path):
    """DROP-IN: the reference's own vfgs_main.c + vfgs_fw.c + yuv.c, compiled unmodified and
    linked against libvfgs_hip.so (oracle/_ref/vfgs_hip_cli), must write the same file as the
    all-reference binary (oracle/_ref/vfgs_ref).  Both prebuilt in the build container; the cfg
    below is written here (the reference's cfg corpus does not travel)."""
    import subprocess
    cli, ref = T.REF_DIR / "vfgs_hip_cli", T.REF_DIR / "vfgs_ref"
    if not (cli.exists() and ref.exists()):
        pytest.skip("oracle/_ref binaries were not prebuilt")
    cfg = tmp_path / "two_patterns.cfg"
    cfg.write_text("\n".join([
        "SEIFGCModelID : 0", "SEIFGCLog2ScaleFactor : 4",
        "SEIFGCCompModelPresentComp0 : 1", "SEIFGCCompModelPresentComp1 : 1", "SEIFGCCompModelPresentComp2 : 1",
        "SEIFGCNumIntensityIntervalMinus1Comp0 : 2", "SEIFGCNumIntensityIntervalMinus1Comp1 : 0",
        "SEIFGCNumIntensityIntervalMinus1Comp2 : 1",
        "SEIFGCNumModelValuesMinus1Comp0 : 2", "SEIFGCNumModelValuesMinus1Comp1 : 2", "SEIFGCNumModelValuesMinus1Comp2 : 2",
        "SEIFGCIntensityIntervalLowerBoundComp0 : 0 70 150", "SEIFGCIntensityIntervalUpperBoundComp0 : 69 149 255",
        "SEIFGCIntensityIntervalLowerBoundComp1 : 0", "SEIFGCIntensityIntervalUpperBoundComp1 : 255",
        "SEIFGCIntensityIntervalLowerBoundComp2 : 0 128", "SEIFGCIntensityIntervalUpperBoundComp2 : 127 255",
        "SEIFGCCompModelValuesComp0 : 90 6 9 120 10 10 60 13 4",
        "SEIFGCCompModelValuesComp1 : 70 5 5", "SEIFGCCompModelValuesComp2 : 40 4 6 80 7 3", ""]))
    w, h, n = 208, 160, 3
    for depth in (10, 8):
        frames, _ = T.lcg_frames(w, h, depth, 2, 2, n)
        inp = tmp_path / f"in{depth}.yuv"
        inp.write_bytes(b"".join(f.picture_bytes() for f in frames))
        outs = []
        for exe in (ref, cli):
            out = tmp_path / f"{exe.name}_{depth}.yuv"
            subprocess.run([str(exe), "-w", str(w), "-h", str(h), "-b", str(depth), "-n", str(n), "-r", "777",
                            "-c", str(cfg), str(inp), str(out)], check=True, stdout=subprocess.DEVNULL, timeout=600)
            outs.append(out.read_bytes())
        assert len(outs[0]) == len(inp.read_bytes())
        assert outs[0] != inp.read_bytes()          # grain was really added
        assert outs[0] == outs[1]


@pytest.mark.parametrize("name", sorted(MD5["small_outdepth8"]))
def test_fused_8bit_output_matches_reference_outdepth8(hip, name):
    """grain on 10-bit input with the 10->8 bit narrowing of yuv_to_8bit
    (yuv.c:216-258) fused into the store; golden = reference CLI run with --outdepth 8."""
    import hashlib
    import torch
    from gpu_util import DevFrame, stream_ptr
    ora, (depth, sx, sy) = program(hip, name)
    assert depth == 10
    frames, _ = T.lcg_frames(W, H, depth, sx, sy, N)
    m = hashlib.md5()
    for f in frames:
        want = f.copy()
        ora.add_grain_frame(want)
        src = DevFrame(f)
        f8 = T.Frame(W, H, 8, sx, sy)
        dY = torch.full(f8.Y.shape, 0x5a, dtype=torch.uint8, device="cuda")
        dU = torch.full(f8.U.shape, 0x5a, dtype=torch.uint8, device="cuda")
        dV = torch.full(f8.V.shape, 0x5a, dtype=torch.uint8, device="cuda")
        hip.add_grain_copy8_dev(*src.ptrs(), dY.data_ptr(), dU.data_ptr(), dV.data_ptr(), W, H, 0, H, f.stride, f.cstride,
                                f8.stride, f8.cstride, 1, 0, 0, 0, 0, stream_ptr())
        torch.cuda.synchronize()
        assert src.download().equal_all(f)
        for got, w16, hh, ww in ((dY, want.Y, H, W), (dU, want.U, H // sy, W // sx), (dV, want.V, H // sy, W // sx)):
            g = got.cpu().numpy()
            exp = ((w16[:hh, :ww].astype(np.int32) + 2) >> 2).astype(np.uint8)
            assert np.array_equal(g[:hh, :ww], exp)
            m.update(g[:hh, :ww].tobytes())
        assert (dY.cpu().numpy()[H:] == 0x5a).all()      # rows below the picture untouched
        assert hip.seed_state() == ora.seed_state()
    assert m.hexdigest() == MD5["small_outdepth8"][name]


@pytest.mark.parametrize("name", ["fgs_sei_ff_test6_10_444", "fgs_sei_ff_test6_10_440", "fgs_sei_10_422"])
def test_fused_8bit_output_batch_and_parts_other_formats(hip, name):
    """copy8 with nframes > 1, a 16-aligned part, and the chroma formats the CLI cannot drive (16-sample chroma blocks,
    with and without vertical subsampling; 8-sample blocks without)."""
    import torch
    from gpu_util import stream_ptr
    ora, (depth, sx, sy) = program(hip, name)
    n, w, h = 3, 328, 176
    frames, _ = T.lcg_frames(w, h, depth, sx, sy, n)
    want = [f.copy() for f in frames]
    for f in want:
        ora.add_grain_frame(f)
    f0 = frames[0]
    f8 = T.Frame(w, h, 8, sx, sy)
    Y = torch.from_numpy(np.stack([f.Y for f in frames]).view(np.uint8)).cuda()
    U = torch.from_numpy(np.stack([f.U for f in frames]).view(np.uint8)).cuda()
    V = torch.from_numpy(np.stack([f.V for f in frames]).view(np.uint8)).cuda()
    dY = torch.zeros((n,) + f8.Y.shape, dtype=torch.uint8, device="cuda")
    dU = torch.zeros((n,) + f8.U.shape, dtype=torch.uint8, device="cuda")
    dV = torch.zeros((n,) + f8.V.shape, dtype=torch.uint8, device="cuda")
    py, ph = 48, 96
    cy, ch, cw = py // sy, ph // sy, w // sx
    sz = 2
    hip.add_grain_copy8_dev(Y.data_ptr() + py * f0.stride * sz, U.data_ptr() + cy * f0.cstride * sz, V.data_ptr() + cy * f0.cstride * sz,
                            dY.data_ptr() + py * f8.stride, dU.data_ptr() + cy * f8.cstride, dV.data_ptr() + cy * f8.cstride,
                            w, h, py, ph, f0.stride, f0.cstride, f8.stride, f8.cstride, n,
                            Y[0].numel(), U[0].numel(), dY[0].numel(), dU[0].numel(), stream_ptr())
    torch.cuda.synchronize()
    for i, wf in enumerate(want):
        for got, w16, r0, nr, nc in ((dY[i], wf.Y, py, ph, w), (dU[i], wf.U, cy, ch, cw), (dV[i], wf.V, cy, ch, cw)):
            g = got.cpu().numpy()
            exp = ((w16[r0:r0 + nr, :nc].astype(np.int32) + 2) >> 2).astype(np.uint8)
            assert np.array_equal(g[r0:r0 + nr, :nc], exp)
            assert (g[:r0] == 0).all() and (g[r0 + nr:] == 0).all()
    assert hip.seed_state() == ora.seed_state()


def test_frames_part_batched_equals_whole(hip):
    """What one rank of the multi-GPU bench runs: its stripe of several consecutive frames in ONE launch."""
    import torch
    from gpu_util import stream_ptr
    ora, (depth, sx, sy) = program(hip, "fgs_sei_10_420")
    n, w, h = 4, 640, 368
    frames, _ = T.lcg_frames(w, h, depth, sx, sy, n)
    want = [f.copy() for f in frames]
    for f in want:
        ora.add_grain_frame(f)
    py, ph = 128, 112
    f0 = frames[0]
    Y = torch.from_numpy(np.stack([f.Y[py:py + ph] for f in frames]).view(np.uint8)).cuda()
    U = torch.from_numpy(np.stack([f.U[py // 2:(py + ph) // 2] for f in frames]).view(np.uint8)).cuda()
    V = torch.from_numpy(np.stack([f.V[py // 2:(py + ph) // 2] for f in frames]).view(np.uint8)).cuda()
    hip.add_grain_frames_part_dev(Y.data_ptr(), U.data_ptr(), V.data_ptr(), w, h, py, ph, f0.stride, f0.cstride, n,
                                  Y[0].numel(), U[0].numel(), stream_ptr())
    torch.cuda.synchronize()
    for i, wf in enumerate(want):
        assert np.array_equal(Y[i].cpu().numpy().view(wf.dtype).reshape(ph, -1), wf.Y[py:py + ph])
        assert np.array_equal(U[i].cpu().numpy().view(wf.dtype).reshape(ph // 2, -1), wf.U[py // 2:(py + ph) // 2])
        assert np.array_equal(V[i].cpu().numpy().view(wf.dtype).reshape(ph // 2, -1), wf.V[py // 2:(py + ph) // 2])
    assert hip.seed_state() == ora.seed_state()


def _line_loop(hwimpl, fr, sy):
    for y in range(fr.height):
        hwimpl.add_grain_line(fr.Y[y].ctypes.data, fr.U[y // sy].ctypes.data, fr.V[y // sy].ctypes.data, y, fr.width)


@pytest.mark.parametrize("name,w,h", [("fgs_sei_10_420", 192, 144), ("fgs_sei_10_420", 200, 150), ("fgs_afgs1_test1_8_444", 192, 144),
                                       ("fgs_sei_ff_test6_8_422", 264, 136), ("fgs_sei_10_420", 1920, 1080)])
def test_line_api_lookahead_many_frames(hip, name, w, h):
    """The drop-in loop of vfgs_main.c:664-682 over several frames: from the second call on the
    library serves lines from pre-computed stripes (line_call() in vfgs_host.cpp); results and seed
    registers must be those of the oracle called line by line."""
    ora, (depth, sx, sy) = program(hip, name)
    nfr = 2 if w > 1000 else 4
    frames, _ = T.lcg_frames(w, h, depth, sx, sy, nfr, garbage_padding=True)
    for f in frames:
        a, b = f.copy(), f.copy()
        _line_loop(hip, a, sy)
        _line_loop(ora, b, sy)
        assert a.equal_all(b)
        assert hip.seed_state() == ora.seed_state()


def test_line_api_lookahead_sees_late_changes_and_irregular_calls(hip):
    """Look-ahead must never be observable: the caller edits lines it has not handed over yet,
    skips lines, repeats a line, changes the width, calls a setter in the middle of a frame."""
    ora, (depth, sx, sy) = program(hip, "fgs_sei_10_420")
    frames, _ = T.lcg_frames(320, 176, depth, sx, sy, 3)
    rng = np.random.default_rng(5)
    for fi, f in enumerate(frames):
        a, b = f.copy(), f.copy()
        y = 0
        while y < f.height:
            for fr, hwimpl in ((a, hip), (b, ora)):
                hwimpl.add_grain_line(fr.Y[y].ctypes.data, fr.U[y // 2].ctypes.data, fr.V[y // 2].ctypes.data, y, fr.width)
            assert np.array_equal(a.Y[y], b.Y[y]) and np.array_equal(a.U[y // 2], b.U[y // 2]), (fi, y)
            if y % 23 == 5 and y + 3 < f.height:          # edit a line that was (probably) read ahead already
                patch = rng.integers(0, 1024, f.width).astype(a.dtype)
                a.Y[y + 2, :f.width] = patch
                b.Y[y + 2, :f.width] = patch
                a.V[(y + 2) // 2, :16] = 7
                b.V[(y + 2) // 2, :16] = 7
            if fi == 1 and y == 40:                        # repeat a line
                for fr, hwimpl in ((a, hip), (b, ora)):
                    hwimpl.add_grain_line(fr.Y[y].ctypes.data, fr.U[y // 2].ctypes.data, fr.V[y // 2].ctypes.data, y, fr.width)
            if fi == 1 and y == 70:                        # skip ahead
                y += 5
            if fi == 2 and y == 50:                        # a setter in the middle of the frame
                hip.set_scale_shift(3)
                ora.set_scale_shift(3)
            if fi == 2 and y == 90:                        # narrower calls from here on
                for yy in range(y + 1, y + 4):
                    for fr, hwimpl in ((a, hip), (b, ora)):
                        hwimpl.add_grain_line(fr.Y[yy].ctypes.data, fr.U[yy // 2].ctypes.data, fr.V[yy // 2].ctypes.data, yy, 200)
                y += 3
            y += 1
        assert a.equal_all(b)
        assert hip.seed_state() == ora.seed_state()


def test_line_api_lookahead_ring_of_stripes_irregular(hip):
    """1080p: the rest of a frame is computed ahead in stripes of 352 lines through a ring of three slots (round 4).  Late edits of
    lines in stripes that are already in flight or back (just before, on and behind stripe boundaries), a line repeated on a
    boundary, a skip across one, a setter between two stripes -- never observable."""
    ora, (depth, sx, sy) = program(hip, "fgs_sei_ff_test1_10_420")
    frames, _ = T.lcg_frames(1920, 1080, depth, sx, sy, 1)
    rng = np.random.default_rng(9)
    edits = {3: 351, 349: 352, 350: 704, 351: 353, 600: 1079, 703: 705, 704: 1056, 1000: 1057}    # at line k: rewrite line v (v > k)
    for fi in range(4):
        a, b = frames[0].copy(), frames[0].copy()
        y = 0
        while y < a.height:
            for fr, hwimpl in ((a, hip), (b, ora)):
                hwimpl.add_grain_line(fr.Y[y].ctypes.data, fr.U[y // 2].ctypes.data, fr.V[y // 2].ctypes.data, y, fr.width)
            assert np.array_equal(a.Y[y], b.Y[y]) and np.array_equal(a.U[y // 2], b.U[y // 2]) and np.array_equal(a.V[y // 2], b.V[y // 2]), (fi, y)
            if fi >= 1 and y in edits:
                v = edits[y]
                patch = rng.integers(0, 1024, a.width).astype(a.dtype)
                a.Y[v, :a.width] = patch
                b.Y[v, :a.width] = patch
                a.U[v // 2, 5:40] = 513
                b.U[v // 2, 5:40] = 513
            if fi == 2 and y in (351, 352, 704):           # repeat lines on both sides of a boundary
                for fr, hwimpl in ((a, hip), (b, ora)):
                    hwimpl.add_grain_line(fr.Y[y].ctypes.data, fr.U[y // 2].ctypes.data, fr.V[y // 2].ctypes.data, y, fr.width)
            if fi == 2 and y == 700:                       # skip across a boundary
                y += 9
            if fi == 3 and y == 352:                       # a setter between two stripes
                hip.set_legal_range(1)
                ora.set_legal_range(1)
            y += 1
        assert a.equal_all(b), fi
        assert hip.seed_state() == ora.seed_state()


def test_fuzz_sizes_formats_and_stripes(hip):
    """Randomised geometry: widths that give every segment/tile shape (partial last segment, odd
    number of segments, one- and many-tile rows), heights that are not multiples of 16, uneven
    host stripes and device stripes with extra row pitch -- against the oracle, whole buffers."""
    from gpu_util import DevFrame, stream_ptr
    rng = np.random.default_rng(20260403)
    names = ["fgs_sei_10_420", "fgs_sei_8_422", "fgs_afgs1_test1_10_444", "fgs_sei_ff_test6_10_444", "fgs_afgs1_test1_8_444",
             "fgs_sei_ff_test6_8_422", "fgs_sei_ar_test1_8_420", "fgs_sei_10_422", "fgs_sei_10_440", "fgs_afgs1_test1_8_440",
             "fgs_sei_ff_test6_10_440"]
    for it in range(48):
        name = names[it % len(names)]
        ora, (depth, sx, sy) = program(hip, name)
        w = int(rng.choice([int(rng.integers(130, 700)), int(rng.integers(700, 2600)), 1024 + 16 * int(rng.integers(0, 3)), 2040, 129 + int(rng.integers(0, 16))]))
        h = int(rng.integers(17, 120))
        if sy == 2:
            h += h & 1
        pad = 64 * int(rng.integers(0, 3))
        f = T.Frame(w, h, depth, sx, sy, stride=((w + 63) // 64 * 64) + pad, cstride=((w // sx + 63) // 64 * 64) + pad)
        for p in f.planes():
            p[...] = rng.integers(0, 1 << (10 if depth > 8 else 8), p.shape).astype(f.dtype)
        want = f.copy()
        ora.add_grain_frame(want)
        if it % 2 == 0:      # host stripes of random heights
            a = f.copy()
            y = 0
            while y < h:
                n = min(h - y, int(rng.integers(1, 40)))
                hip.add_grain_stripe(a.Y[y].ctypes.data, a.U[y // sy].ctypes.data, a.V[y // sy].ctypes.data, y, w, n, a.stride, a.cstride)
                y += n
            got = a
        else:                # device, 16-aligned parts
            d = DevFrame(f)
            cut = 16 * int(rng.integers(0, h // 16 + 1))
            for (y0, n) in ((0, cut), (cut, h - cut)):
                if n:
                    hip.add_grain_stripe_dev(*d.ptrs(y0), y0, w, n, f.stride, f.cstride, stream_ptr())
            got = d.download()
        assert got.equal_all(want), (it, name, w, h, pad)
        assert hip.seed_state() == ora.seed_state()


@pytest.mark.parametrize("name,seed", [("fgs_sei_10_420", None), ("fgs_afgs1_test1_10_420", 40000)])
def test_long_stream_across_lfsr_window_refills(hip, name, seed):
    """240 frames in 20 batched launches: the seed positions walk through ~2.3 Mbit of LFSR stream,
    i.e. across several refills of the host/device stream window (16 KiB growing to 1 MiB, the
    blocked word recurrence) -- the last launch and the registers must still equal the oracle, which
    steps its registers bit by bit through every frame."""
    import torch
    from gpu_util import stream_ptr
    ora, (depth, sx, sy) = program(hip, name)
    if seed is not None:
        hip.set_seed(seed)
        ora.set_seed(seed)
    per, launches, w, h = 12, 20, 1928, 1088
    frames, _ = T.lcg_frames(w, h, depth, sx, sy, per)
    f0 = frames[0]
    srcY = torch.from_numpy(np.stack([f.Y for f in frames]).view(np.uint8)).cuda()
    srcU = torch.from_numpy(np.stack([f.U for f in frames]).view(np.uint8)).cuda()
    srcV = torch.from_numpy(np.stack([f.V for f in frames]).view(np.uint8)).cuda()
    dY, dU, dV = srcY.clone(), srcU.clone(), srcV.clone()   # (the stride padding beyond the last block is never written)
    for _ in range(launches):       # out of place from the same pristine frames every time
        hip.add_grain_copy_dev(srcY.data_ptr(), srcU.data_ptr(), srcV.data_ptr(), dY.data_ptr(), dU.data_ptr(), dV.data_ptr(),
                               w, h, 0, h, f0.stride, f0.cstride, per, srcY[0].numel(), srcU[0].numel(), stream_ptr())
    torch.cuda.synchronize()
    want = None
    for _ in range(launches):
        want = [f.copy() for f in frames]
        for f in want:
            ora.add_grain_frame(f)
    for i, f in enumerate(want):
        assert np.array_equal(dY[i].cpu().numpy().view(f.dtype).reshape(f.Y.shape), f.Y), i
        assert np.array_equal(dU[i].cpu().numpy().view(f.dtype).reshape(f.U.shape), f.U), i
        assert np.array_equal(dV[i].cpu().numpy().view(f.dtype).reshape(f.V.shape), f.V), i
    assert hip.seed_state() == ora.seed_state()


def test_line_api_lookahead_only_in_proven_rows(hip):
    """Working ahead reads lines the caller has not handed over yet, so it is confined to rows the caller has proven to
    own (vfgs_hip.h): the SAME buffer walked once before, or a declared frame.  Pictures of different heights follow
    each other in one buffer and in fresh buffers; a declared frame works ahead from its first line; results and seed
    registers are the oracle's in every case."""
    ora, (depth, sx, sy) = program(hip, "fgs_sei_10_420")
    big, _ = T.lcg_frames(320, 176, depth, sx, sy, 6)
    small, _ = T.lcg_frames(320, 112, depth, sx, sy, 3)
    buf = big[0].copy()                                     # ONE buffer for the whole sequence

    def run(src, hgt):
        want = src.copy()
        buf.Y[:hgt] = src.Y[:hgt]; buf.U[:hgt // 2] = src.U[:hgt // 2]; buf.V[:hgt // 2] = src.V[:hgt // 2]
        for y in range(hgt):
            hip.add_grain_line(buf.Y[y].ctypes.data, buf.U[y // 2].ctypes.data, buf.V[y // 2].ctypes.data, y, buf.width)
            ora.add_grain_line(want.Y[y].ctypes.data, want.U[y // 2].ctypes.data, want.V[y // 2].ctypes.data, y, want.width)
        assert np.array_equal(buf.Y[:hgt], want.Y[:hgt]) and np.array_equal(buf.U[:hgt // 2], want.U[:hgt // 2]) \
            and np.array_equal(buf.V[:hgt // 2], want.V[:hgt // 2])
        assert hip.seed_state() == ora.seed_state()

    run(big[0], 176)        # first walk: line by line
    run(big[1], 176)        # same buffer: works ahead up to line 175
    run(small[0], 112)      # a smaller picture in the same buffer (the rows read ahead beyond 111 belong to the buffer)
    run(small[1], 112)
    run(big[2], 176)        # larger again: only 112 rows are proven now
    run(big[3], 176)
    for f in (big[4], small[2]):                            # fresh buffers: nothing is proven
        a, b = f.copy(), f.copy()
        _line_loop(hip, a, sy)
        _line_loop(ora, b, sy)
        assert a.equal_all(b)
    # a declared frame works ahead from the first line of the first walk
    d, want = big[5].copy(), big[5].copy()
    hip.declare_frame(d.Y.ctypes.data, d.U.ctypes.data, d.V.ctypes.data, d.width, d.height, d.stride, d.cstride)
    _line_loop(hip, d, sy)
    _line_loop(ora, want, sy)
    assert d.equal_all(want)
    hip.declare_frame(None, None, None, 0, 0, 0, 0)
    hip.line_lookahead(False)
    d, want = big[0].copy(), big[0].copy()
    _line_loop(hip, d, sy)
    _line_loop(ora, want, sy)
    assert d.equal_all(want) and hip.seed_state() == ora.seed_state()
    hip.line_lookahead(True)


def test_new_seed_per_frame_queued_behind_a_long_launch(hip):
    """A new seed before every frame (the AFGS1 case, vfgs_fw.c:672) switches the LFSR stream image every call.  The
    calls are asynchronous: many of them queue up behind a long launch, and every one must still read ITS image when it
    finally runs (the images live in a small ring whose slots are only reused after their readers)."""
    import torch
    from gpu_util import DevFrame, stream_ptr
    ora, (depth, sx, sy) = program(hip, "fgs_sei_10_420")
    # something long in front: 6 x 4320p frames in one launch, twice
    Wb, Hb = 7680, 4320
    g = torch.Generator(device="cuda").manual_seed(3)
    bY = torch.randint(0, 1024, (6, Hb, Wb), dtype=torch.int16, device="cuda", generator=g)
    bU = torch.randint(0, 1024, (6, Hb // 2, Wb // 2), dtype=torch.int16, device="cuda", generator=g)
    bV = torch.randint(0, 1024, (6, Hb // 2, Wb // 2), dtype=torch.int16, device="cuda", generator=g)
    frames, _ = T.lcg_frames(W, H, depth, sx, sy, 12)
    devs = [DevFrame(f) for f in frames]
    torch.cuda.synchronize()
    for _ in range(2):
        hip.add_grain_frames_dev(bY.data_ptr(), bU.data_ptr(), bV.data_ptr(), Wb, Hb, Wb, Wb // 2, 6, Hb * Wb * 2, Hb * Wb // 2, stream_ptr())
    for i, d in enumerate(devs):                            # all queued while the long launches still run
        hip.set_seed(1000 + 17 * i)
        hip.add_grain_frame_dev(*d.ptrs(), W, H, frames[i].stride, frames[i].cstride, stream_ptr())
    torch.cuda.synchronize()
    # the oracle: the two long launches advance the registers like 12 whole frames, but every small frame starts from its own seed
    for i, f in enumerate(frames):
        want = f.copy()
        ora.set_seed(1000 + 17 * i)
        ora.add_grain_frame(want)
        assert devs[i].download().equal_all(want), i
    assert hip.seed_state() == ora.seed_state()


def test_line_api_frame_height_promised_through_the_environment(hip, monkeypatch):
    """VFGS_HIP_FRAME_HEIGHT: an unchanged binary's first walk through a buffer is computed ahead too (after the lines that show
    the pitches); results and registers as ever; a walk in a NEW buffer keeps the promise; without the variable the first walk
    is line by line (one launch per line)."""
    ora, (depth, sx, sy) = program(hip, "fgs_sei_10_420")
    w, h = 1920, 540
    frames, _ = T.lcg_frames(w, h, depth, sx, sy, 3)
    n0 = hip.last_launch_info()["launches"] if hip.last_launch_info() else 0
    a, b = frames[0].copy(), frames[0].copy()
    _line_loop(hip, a, sy)
    _line_loop(ora, b, sy)
    assert a.equal_all(b)
    n1 = hip.last_launch_info()["launches"]
    assert n1 - n0 >= h            # no promise: a launch per line
    assert hip.last_launch_info()["internal"] == 1      # ... on the library's staging buffers, and the record says so
    monkeypatch.setenv("VFGS_HIP_FRAME_HEIGHT", str(h))
    for f in frames[1:]:           # fresh buffers: never walked before
        a, b = f.copy(), f.copy()
        _line_loop(hip, a, sy)
        _line_loop(ora, b, sy)
        assert a.equal_all(b)
        assert hip.seed_state() == ora.seed_state()
    n2 = hip.last_launch_info()["launches"]
    assert n2 - n1 < 40, n2 - n1    # two frames of 540 lines: a few single lines + a handful of stripes each


def test_line_api_promised_height_with_buffers_of_different_pitches(hip, monkeypatch):
    """VFGS_HIP_FRAME_HEIGHT and a walk through ANOTHER buffer of the same width but a smaller row pitch: the pitches the previous
    walk showed prove nothing about this buffer -- the library learns them again from its first lines before it reads ahead
    (with the stale pitches it would snapshot rows beyond the smaller allocation).  Results and registers as ever."""
    ora, (depth, sx, sy) = program(hip, "fgs_sei_10_420")
    w, h = 1000, 200
    monkeypatch.setenv("VFGS_HIP_FRAME_HEIGHT", str(h))
    rng = np.random.default_rng(77)
    for i, (stride, cstride) in enumerate([(1280, 640), (1024, 512), (1024, 512), (1152, 576), (1024, 512)]):
        a = T.Frame(w, h, depth, sx, sy, stride=stride, cstride=cstride)
        for p in a.planes():
            p[...] = rng.integers(0, 1024, p.shape).astype(a.dtype)
        b = a.copy()
        _line_loop(hip, a, sy)
        _line_loop(ora, b, sy)
        assert a.equal_all(b), i
        assert hip.seed_state() == ora.seed_state()


@pytest.mark.parametrize("name,w,h,ranks,rank", [("fgs_sei_10_420", 640, 368, 8, 0), ("fgs_sei_10_420", 640, 368, 8, 5), ("fgs_sei_10_420", 640, 368, 8, 7),
                                                 ("fgs_afgs1_test1_8_444", 520, 400, 4, 2), ("fgs_sei_ff_test6_8_422", 456, 272, 4, 1),
                                                 ("fgs_sei_10_420", 1920, 1080, 8, 3)])
def test_chained_stripe_batches_read_jumped_stream_segments(hip, name, w, h, ranks, rank):
    """One rank of a stripe split over several steps (bench.py --gpus N): its stripe of `n` consecutive frames per call, call after
    call.  Such a call reads its LFSR windows from segments the library reaches by jumps -- (nbr - 1) x nblk register steps per
    frame, vfgs_hw.c:291-298,309-310, as one GF(2) matrix (vfgs_host.cpp StripeStream) -- and the next call's segments are built
    ahead.  A whole-frame call in between and a new seed break the chain; results and registers are the oracle's throughout."""
    import torch
    from gpu_util import DevFrame, stream_ptr
    ora, (depth, sx, sy) = program(hip, name)
    n = 16                              # (an image of the stripe stream holds four such calls: the fifth finds its segments built ahead)
    nbr = (h + 15) // 16
    rows = -(-nbr // ranks)
    py = rank * rows * 16
    ph = min(rows * 16, h - py)
    sz = 2 if depth > 8 else 1
    st0 = hip.stripe_stream_stats()
    used = 0
    for call in range(10):
        frames, _ = T.lcg_frames(w, h, depth, sx, sy, n, state=call + 1)
        want = [f.copy() for f in frames]
        for f in want:
            ora.add_grain_frame(f)
        f0 = frames[0]
        cy0, cy1 = py // sy, -(-(py + ph) // sy)
        Y = torch.from_numpy(np.stack([f.Y[py:py + ph] for f in frames]).view(np.uint8)).cuda()
        U = torch.from_numpy(np.stack([f.U[cy0:cy1] for f in frames]).view(np.uint8)).cuda()
        V = torch.from_numpy(np.stack([f.V[cy0:cy1] for f in frames]).view(np.uint8)).cuda()
        hip.add_grain_frames_part_dev(Y.data_ptr(), U.data_ptr(), V.data_ptr(), w, h, py, ph, f0.stride, f0.cstride, n,
                                      Y[0].numel(), U[0].numel(), stream_ptr())
        used += hip.stripe_stream_stats()["last_launch_used_it"]
        torch.cuda.synchronize()
        for i, wf in enumerate(want):
            assert np.array_equal(Y[i].cpu().numpy().view(wf.dtype).reshape(ph, -1), wf.Y[py:py + ph]), (call, i)
            assert np.array_equal(U[i].cpu().numpy().view(wf.dtype).reshape(cy1 - cy0, -1), wf.U[cy0:cy1]), (call, i)
            assert np.array_equal(V[i].cpu().numpy().view(wf.dtype).reshape(cy1 - cy0, -1), wf.V[cy0:cy1]), (call, i)
        assert hip.seed_state() == ora.seed_state(), call
        if call == 6:                       # something else in between: the chain starts over behind it
            g, _ = T.lcg_frames(w, h, depth, sx, sy, 1, state=99)
            d = DevFrame(g[0])
            hip.add_grain_frame_dev(d.Y.data_ptr(), d.U.data_ptr(), d.V.data_ptr(), w, h, g[0].stride, g[0].cstride, stream_ptr())
            ora.add_grain_frame(g[0])
            assert d.download().equal_all(g[0])
        if call == 7:
            hip.set_seed(4711)
            ora.set_seed(4711)
    st1 = hip.stripe_stream_stats()
    # (rank 0's first call after a new seed begins at bit 0, in front of which there is nothing to jump from: the ordinary window)
    assert used >= (8 if rank == 0 else 10), used
    assert st1["built_ahead"] > st0["built_ahead"] and st1["switches_to_built_ahead"] > st0["switches_to_built_ahead"]
