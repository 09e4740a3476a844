"""GPU (MI355X): the firmware layer with device-side pattern generation (include/vfgs_hip_fw.h)
against the REAL reference firmware.

Inputs: the parameter structures the reference CLI handed to vfgs_init_sei / vfgs_init_afgs1
(tests/golden/fwcfg).  Expected: what the reference firmware then programmed into the hardware
layer (tests/golden/traces, same name) and the md5 of the reference CLI's output.  Where the
prebuilt reference library travelled to the box (oracle/_ref/libvfgs_ref.so) random parameter
sets are also run through both firmwares."""
import ctypes as C
import json

import numpy as np
import pytest

import vfgs_testlib as T

pytestmark = pytest.mark.gpu

MD5 = json.loads((T.GOLDEN / "md5.json").read_text())
W, H, N = 192, 144, 3
NAMES = sorted(p.stem for p in T.FWCFG.glob("*.npz"))


@pytest.fixture(scope="module")
def hip():
    import torch
    assert torch.cuda.is_available(), "these tests need the MI355X"
    from versatilefilmgrain_amd import hw
    return hw.VfgsHip(device=0)


def program_by_firmware(hip, name):
    """vfgs_main.c:750-781 with OUR firmware: depth, subsampling, init(default), seed, init(cfg)."""
    from versatilefilmgrain_amd import fw
    depth, sx, sy = T.trace_geometry(T.load_trace(name))
    seed, cfgs = T.load_fwcfg(name)
    hip.lib.vfgs_hip_reset_state()
    hip.set_depth(depth)
    hip.set_chroma_subsampling(sx, sy)
    for i, (kind, raw) in enumerate(cfgs):
        fw.init(fw.struct_from_bytes(kind, raw))
        if i == 0:
            hip.set_seed(seed)
    return depth, sx, sy


def host_frame_call(hip, f):
    hip.add_grain_stripe(f.Y.ctypes.data, f.U.ctypes.data, f.V.ctypes.data, 0, f.width, f.height, f.stride, f.cstride)


@pytest.mark.parametrize("name", NAMES)
def test_generated_patterns_equal_reference_firmware(hip, name):
    from versatilefilmgrain_amd import fw
    _, sx, sy = program_by_firmware(hip, name)
    want = T.BankModel()
    T.replay(want, T.load_trace(name))
    assert want.luma, "trace programs no luma pattern?"
    for i, p in want.luma.items():
        got = np.frombuffer(fw.get_pattern(0, i), dtype=np.int8).reshape(64, 64)
        assert np.array_equal(got, p), f"luma slot {i}"
    for i, p in want.chroma.items():
        got = np.frombuffer(fw.get_pattern(1, i), dtype=np.int8).reshape(64, 64)[:64 // sy, :64 // sx]
        assert np.array_equal(got, p), f"chroma slot {i}"


@pytest.mark.parametrize("name", NAMES)
def test_firmware_programmed_frames_equal_reference_cli_md5(hip, name):
    """cfg structure -> our firmware (GPU patterns) -> our hardware layer == the reference CLI's output."""
    depth, sx, sy = program_by_firmware(hip, name)
    frames, _ = T.lcg_frames(W, H, depth, sx, sy, N)
    for f in frames:
        host_frame_call(hip, f)
    assert T.md5_frames(frames) == MD5["small"][name]


def test_host_set_pattern_overrides_generated_and_back(hip):
    """Slots remember where their pattern came from: setter after generation wins, and vice versa."""
    from versatilefilmgrain_amd import fw
    name = "fgs_sei_10_420"
    depth, sx, sy = program_by_firmware(hip, name)
    rng = np.random.default_rng(7)
    mine = rng.integers(-127, 128, 4096).astype(np.int8)
    hip.set_luma_pattern(2, mine.tobytes())
    assert fw.get_pattern(0, 2) == mine.tobytes()
    ora = T.OracleHW()
    T.replay(ora, T.load_trace(name))
    ora.set_luma_pattern(2, mine.tobytes())
    f, _ = T.lcg_frames(W, H, depth, sx, sy, 1)
    a, b = f[0].copy(), f[0].copy()
    host_frame_call(hip, a)
    ora.add_grain_frame(b)
    assert a.equal_all(b)
    # generate over it again: the device pattern is back
    program_by_firmware(hip, name)
    want = T.BankModel()
    T.replay(want, T.load_trace(name))
    assert fw.get_pattern(0, 2) == want.luma[2].tobytes()


def test_config_switch_between_frames_without_sync(hip):
    """vfgs_main.c:773-781: a new configuration before every frame, device-resident frames,
    nothing synchronises between the firmware call and the grain launches."""
    import torch
    from gpu_util import DevFrame, stream_ptr
    from versatilefilmgrain_amd import fw
    names = ["fgs_sei_10_420", "fgs_afgs1_test1_10_420", "fgs_sei_ar_test1_10_420", "fgs_sei_ff_test6_10_420"]
    frames, _ = T.lcg_frames(W, H, 10, 2, 2, len(names))
    devs = [DevFrame(f) for f in frames]
    hip.lib.vfgs_hip_reset_state()
    hip.set_depth(10)
    hip.set_chroma_subsampling(2, 2)
    ora = T.OracleHW()
    ora.set_depth(10)
    ora.set_chroma_subsampling(2, 2)
    for name, d, f in zip(names, devs, frames):
        _, cfgs = T.load_fwcfg(name)
        kind, raw = cfgs[-1]
        fw.init(fw.struct_from_bytes(kind, raw))
        # expected: the reference firmware's programming for the same structure = the tail of the trace
        rec = T.load_trace(name)
        last_init = max(i for i, r in enumerate(rec) if r[0] == T.OP_SEED) + 1 if kind == 0 else \
            max(i for i, r in enumerate(rec) if r[0] == T.OP_SEED)
        T.replay(ora, rec[last_init:])
        hip.add_grain_frame_dev(d.Y.data_ptr(), d.U.data_ptr(), d.V.data_ptr(), f.width, f.height, f.stride, f.cstride, stream_ptr())
        ora.add_grain_frame(f)
    torch.cuda.synchronize()
    for d, f in zip(devs, frames):
        assert d.download().equal_all(f)


# ---------------------------------------------------------------------------------------
# random parameter sets through both firmwares (real reference library, prebuilt)

def _random_sei(rng, model_id):
    from versatilefilmgrain_amd import fw
    s = fw.FgsSei()
    s.model_id = model_id
    s.log2_scale_factor = int(rng.integers(3, 8)) if model_id else int(rng.integers(2, 8))
    for c in range(3):
        s.comp_model_present_flag[c] = 1 if c == 0 else int(rng.integers(0, 2))
        n = int(rng.integers(1, 12))
        s.num_intensity_intervals[c] = n
        s.num_model_values[c] = 6 if model_id else 3
        cuts = np.sort(rng.choice(np.arange(1, 255), size=n, replace=False))
        lo = 0
        for k in range(n):
            s.intensity_interval_lower_bound[c][k] = lo
            s.intensity_interval_upper_bound[c][k] = int(cuts[k]) if rng.integers(0, 4) else min(255, int(cuts[k]) + 3)
            lo = int(cuts[k]) + int(rng.integers(1, 3))
            lo = min(lo, 255)
            v = s.comp_model_value[c][k]
            v[0] = int(rng.integers(0, 256))
            if model_id:
                v[1] = int(rng.integers(-100, 101)); v[2] = 0
                v[3] = int(rng.integers(-60, 61)); v[4] = int(rng.integers(0, 1 << s.log2_scale_factor))
                v[5] = int(rng.integers(-40, 41))
            else:
                choices = [2, 5, 8, 11, 14] if n > 6 else list(range(0, 16))
                v[1] = int(rng.choice(choices)); v[2] = int(rng.choice(choices))
    return s


def _random_afgs1(rng):
    from versatilefilmgrain_amd import fw
    a = fw.FgsAfgs1()
    a.grain_seed = int(rng.integers(1, 65536))
    for pts, vals, scal, mx in (("num_y_points", a.point_y_values, a.point_y_scaling, 14),
                                ("num_cb_points", a.point_cb_values, a.point_cb_scaling, 10),
                                ("num_cr_points", a.point_cr_values, a.point_cr_scaling, 10)):
        n = int(rng.integers(1, mx + 1))
        setattr(a, pts, n)
        xs = np.sort(rng.choice(np.arange(0, 256), size=n, replace=False))
        for k in range(n):
            vals[k] = int(xs[k]); scal[k] = int(rng.integers(0, 256))
    a.chroma_scaling_from_luma = int(rng.integers(0, 2))
    a.grain_scaling = int(rng.integers(8, 12))
    a.ar_coeff_lag = int(rng.integers(1, 4))
    for arr, n in ((a.ar_coeffs_y, 24), (a.ar_coeffs_cb, 25), (a.ar_coeffs_cr, 25)):
        for k in range(n):
            arr[k] = int(rng.integers(-40, 41))
    a.ar_coeff_shift = int(rng.integers(6, 10))
    a.grain_scale_shift = int(rng.integers(0, 4))
    a.clip_to_restricted_range = int(rng.integers(0, 2))
    return a


@pytest.mark.reference
@pytest.mark.parametrize("seed", range(24))
def test_random_parameter_sets_vs_real_reference_firmware(hip, seed):
    if not T.have_reference():
        pytest.skip("oracle/_ref/libvfgs_ref.so did not travel to this box")
    from versatilefilmgrain_amd import fw
    rng = np.random.default_rng(1000 + seed)
    kind = seed % 3            # 0: SEI frequency filtering, 1: SEI auto-regressive, 2: AFGS1
    depth = 10 if seed % 2 else 8
    sx, sy = [(2, 2), (2, 2), (1, 1), (2, 1)][(seed // 3) % 4]
    cfg = _random_afgs1(rng) if kind == 2 else _random_sei(rng, kind)
    ref = T.ReferenceHW()
    hip.lib.vfgs_hip_reset_state()
    for impl in (ref, hip):
        impl.set_depth(depth)
        impl.set_chroma_subsampling(sx, sy)
    (ref.lib.vfgs_init_afgs1 if kind == 2 else ref.lib.vfgs_init_sei)(C.byref(cfg))
    fw.init(cfg)
    if kind != 2:
        ref.set_seed(777 + seed)
        hip.set_seed(777 + seed)
    f, _ = T.lcg_frames(W, H, depth, sx, sy, 1, state=seed + 1)
    a, b = f[0].copy(), f[0].copy()
    host_frame_call(hip, a)
    ref.add_grain_frame(b)
    assert a.equal_all(b)


def test_reference_cli_with_our_firmware_and_hardware_layer(hip, tmp_path):
    """DROP-IN at the firmware interface: only the reference's vfgs_main.c + yuv.c (compiled
    unmodified, oracle/_ref/vfgs_hip_cli_fw); vfgs_init_sei / vfgs_init_afgs1 and the hardware
    layer come from libvfgs_hip.so.  Must write the same file as the all-reference binary,
    including configuration switches at frames 1 and 2 (`-c <poc>:file`, vfgs_main.c:773-781).
    The cfg files are written here (the reference's corpus does not travel)."""
    import subprocess
    cli, ref = T.REF_DIR / "vfgs_hip_cli_fw", T.REF_DIR / "vfgs_ref"
    if not (cli.exists() and ref.exists()):
        pytest.skip("oracle/_ref binaries were not prebuilt")
    ff = tmp_path / "ff.cfg"
    ff.write_text("\n".join([
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
    ar = tmp_path / "ar.cfg"
    ar.write_text("\n".join([
        "SEIFGCModelID : 1", "SEIFGCLog2ScaleFactor : 5",
        "SEIFGCCompModelPresentComp0 : 1", "SEIFGCCompModelPresentComp1 : 0", "SEIFGCCompModelPresentComp2 : 0",
        "SEIFGCNumIntensityIntervalMinus1Comp0 : 1", "SEIFGCNumModelValuesMinus1Comp0 : 5",
        "SEIFGCIntensityIntervalLowerBoundComp0 : 0 100", "SEIFGCIntensityIntervalUpperBoundComp0 : 99 255",
        "SEIFGCCompModelValuesComp0 : 80 40 0 -12 20 9 120 -30 0 14 25 -6", ""]))
    av = tmp_path / "afgs1.cfg"
    av.write_text("\n".join([
        "AFGS1GrainSeed : 4711", "AFGS1NumYPoints : 3", "AFGS1PointYValues : 0 120 255", "AFGS1PointYScaling : 30 90 50",
        "AFGS1ChromaScalingFromLuma : 0", "AFGS1NumCbPoints : 2", "AFGS1PointCbValues : 10 240", "AFGS1PointCbScaling : 60 20",
        "AFGS1NumCrPoints : 2", "AFGS1PointCrValues : 0 255", "AFGS1PointCrScaling : 25 70",
        "AFGS1GrainScaling : 10", "AFGS1ARCoeffLag : 2",
        "AFGS1ARCoeffsY : 3 -5 8 -2 1 6 -20 30 -9 4 12 -40",
        "AFGS1ARCoeffsCb : -2 4 -7 3 0 5 18 -25 7 -3 10 35 0",
        "AFGS1ARCoeffsCr : 1 -3 6 -4 2 -6 15 22 -8 5 -11 30 0",
        "AFGS1ARCoeffShift : 7", "AFGS1GrainScaleShift : 1", "AFGS1OverlapFlag : 1", "AFGS1ClipToRestrictedRange : 1", ""]))
    w, h, n = 208, 160, 4
    for depth in (10, 8):
        frames, _ = T.lcg_frames(w, h, depth, 2, 2, n)
        inp = tmp_path / f"in{depth}.yuv"
        inp.write_bytes(b"".join(f.picture_bytes() for f in frames))
        for cfgs in ([str(ff)], [str(ar)], [str(av)], [str(ff), f"1:{av}", f"2:{ar}", f"3:{ff}"]):
            outs = []
            for exe in (ref, cli):
                out = tmp_path / f"{exe.name}_{depth}.yuv"
                cmd = [str(exe), "-w", str(w), "-h", str(h), "-b", str(depth), "-n", str(n), "-r", "777"]
                for c in cfgs:
                    cmd += ["-c", c]
                subprocess.run(cmd + [str(inp), str(out)], check=True, stdout=subprocess.DEVNULL, timeout=600)
                outs.append(out.read_bytes())
            assert len(outs[0]) == len(inp.read_bytes())
            assert outs[0] != inp.read_bytes()          # grain was really added
            assert outs[0] == outs[1], cfgs


# ---------------------------------------------------------------------------------------
# the whole chain without any reference code: cfg FILE -> vfgs_hip_cfg_* -> firmware on the GPU
# -> hardware layer, against the reference CLI's md5 for the same command line

@pytest.fixture(scope="module")
def corpus(tmp_path_factory):
    d = tmp_path_factory.mktemp("cfg")
    with np.load(T.GOLDEN / "cfg_corpus.npz") as z:
        for name in z.files:
            (d / name).write_bytes(z[name].tobytes())
    return d


@pytest.mark.parametrize("name", [n for n in NAMES if n.endswith("_420")])
def test_cfg_file_through_library_equals_reference_cli_md5(hip, corpus, name):
    from test_cfg_cpu import split_name
    from versatilefilmgrain_amd import fw
    file, depth, fmt = split_name(name)
    seed, _ = T.load_fwcfg(name)
    hip.lib.vfgs_hip_reset_state()
    hip.set_depth(depth)                       # vfgs_main.c:750-760
    hip.set_chroma_subsampling(2, 2)
    st = fw.Cfg.defaults()
    assert st.check(fmt, depth) == 0
    st.adjust_chroma(fmt)
    st.apply_gain(100)
    st.program()
    hip.set_seed(seed)
    if file is not None and st.read(corpus / file) == 0 and st.check(fmt, depth) == 0:   # vfgs_main.c:773-781
        st.adjust_chroma(fmt)
        st.apply_gain(100)
        st.program()
    frames, _ = T.lcg_frames(W, H, depth, 2, 2, N)
    for f in frames:
        host_frame_call(hip, f)
    assert T.md5_frames(frames) == MD5["small"][name]


def test_pattern_jobs_directly_against_the_numpy_restatement(hip):
    """vfgs_hip_generate_patterns (the extension under the firmware entry points) with random jobs,
    against oracle/vfgs_fw_oracle.py -- including taps far larger than any real model uses."""
    import sys
    sys.path.insert(0, str(T.ROOT / "oracle"))
    import vfgs_fw_oracle as F
    from versatilefilmgrain_amd import fw
    rng = np.random.default_rng(2024)
    hip.lib.vfgs_hip_reset_state()
    hip.set_chroma_subsampling(2, 2)
    for rnd in range(6):
        jobs, want = [], []
        for chroma in (0, 1):
            for slot in rng.permutation(8)[:3]:
                j = fw.PatternJob()
                j.chroma, j.index = chroma, int(slot)
                if rnd % 2 == 0:
                    j.kind, j.seed_index = 0, int(rng.integers(0, 3))
                    j.fh, j.fv = int(rng.integers(-1, 17)), int(rng.integers(-1, 17))
                    want.append(F.ff_pattern(32 if chroma else 64, j.fh, j.fv, j.seed_index))
                else:
                    lag = int(rng.integers(1, 4))
                    big = 2000 if rnd == 5 else 60
                    taps = np.zeros((4, 7), dtype=np.int64)
                    for jj in range(-lag, 1):
                        for ii in range(-lag, lag + 1):
                            if ii < 0 or jj < 0:
                                taps[3 + jj, 3 + ii] = int(rng.integers(-big, big + 1))
                    j.kind, j.seed_index = 1, int(rng.integers(0, 3))
                    j.scale, j.shift = int(rng.integers(5, 12)), int(rng.integers(1, 5))
                    for k, v in enumerate(taps.reshape(-1)):
                        j.coef[k] = int(v)
                    want.append(F.ar_pattern(bool(chroma), taps, j.scale, j.shift, j.seed_index))
                jobs.append(j)
        fw.generate_patterns(jobs)
        for j, w in zip(jobs, want):
            n = 32 if j.chroma else 64
            got = np.frombuffer(fw.get_pattern(j.chroma, j.index), dtype=np.int8).reshape(64, 64)[:n, :n]
            assert np.array_equal(got, w), (rnd, j.kind, j.chroma, j.index)
