#!/usr/bin/env python3
"""Configuration-switch cost: the reference firmware on one host core vs. the firmware layer
of this library (host tables + pattern generation on the GPU).

Prints one JSON object; tools/gpu_bench_profile.sh stores it under gpurun_out/.
The reference leg uses the prebuilt oracle/_ref/libvfgs_ref.so (kind "reference") and is
skipped when that file did not travel.
"""
import ctypes as C
import json
import statistics
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))

import torch  # noqa: E402

import vfgs_testlib as T  # noqa: E402
from versatilefilmgrain_amd import fw, hw  # noqa: E402

CASES = ["fgs_sei_10_420", "fgs_sei_ff_test1_10_420", "fgs_sei_ar_test1_10_420", "fgs_afgs1_test1_10_420"]
W, H = 1920, 1080


def med(f, n):
    out = []
    for _ in range(n):
        t0 = time.perf_counter()
        f()
        out.append((time.perf_counter() - t0) * 1e3)
    return round(statistics.median(out), 4)


def main():
    assert torch.cuda.is_available()
    h = hw.VfgsHip(device=0)
    res = {}
    for name in CASES:
        _, cfgs = T.load_fwcfg(name)
        kind, raw = cfgs[-1]
        cfg = fw.struct_from_bytes(kind, raw)
        rec = T.load_trace(name)
        h.lib.vfgs_hip_reset_state()
        h.set_depth(10)
        h.set_chroma_subsampling(2, 2)
        fw.init(cfg)
        fw.get_pattern(0, 0)
        r = {"patterns": sum(1 for op, *_ in rec if op in (T.OP_LUMA_PATTERN, T.OP_CHROMA_PATTERN)) -
             sum(1 for op, *_ in T.load_trace("default_10_420") if op in (T.OP_LUMA_PATTERN, T.OP_CHROMA_PATTERN))}
        # a second, slightly different parameter set: alternating between the two defeats the
        # "same request as last time" shortcut, so every switch really generates patterns
        other = type(cfg).from_buffer_copy(bytes(cfg))
        if kind:
            other.ar_coeffs_y[0] += 1
            other.ar_coeffs_cb[0] += 1
            other.ar_coeffs_cr[0] += 1
        else:
            for c in range(3):
                for k in range(other.num_intensity_intervals[c]):
                    v = other.comp_model_value[c][k]
                    if other.model_id:
                        v[1] += 1
                    else:
                        v[1] = v[1] - 1 if v[1] > 8 else v[1] + 1

        # 1. our firmware: time until the call returns (everything queued) and until the patterns exist
        flip = [0]

        def call_only():
            flip[0] ^= 1
            fw.init(other if flip[0] else cfg)
        r["hip_call_returns_ms"] = med(call_only, 50)
        torch.cuda.synchronize()

        def gen_and_wait():
            flip[0] ^= 1
            fw.init(other if flip[0] else cfg)
            fw.get_pattern(0, 0)
        r["hip_patterns_ready_ms"] = med(gen_and_wait, 50)
        # 2. config switch before EVERY frame of a device-resident 1080p sequence
        Y = torch.randint(0, 1024, (8, H, W), dtype=torch.int16, device="cuda")
        U = torch.randint(0, 1024, (8, H // 2, W // 2), dtype=torch.int16, device="cuda")
        V = torch.randint(0, 1024, (8, H // 2, W // 2), dtype=torch.int16, device="cuda")
        s = torch.cuda.current_stream().cuda_stream

        def frames(switch, n=64):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for i in range(n):
                if switch == "device":
                    fw.init(other if i & 1 else cfg)
                elif switch == "same":
                    fw.init(cfg)
                elif switch == "setters":
                    T.replay(h, tail)
                k = i % 8
                h.add_grain_frame_dev(Y[k].data_ptr(), U[k].data_ptr(), V[k].data_ptr(), W, H, W, W // 2, s)
            torch.cuda.synchronize()
            return (time.perf_counter() - t0) / n * 1e3

        last_seed = max(i for i, x in enumerate(rec) if x[0] == T.OP_SEED)
        tail = rec[last_seed + 1:] if kind == 0 else rec[last_seed:]
        frames("none", 16)
        r["frame_ms_no_switch"] = round(frames("none"), 4)
        r["frame_ms_switch_device_firmware"] = round(frames("device"), 4)
        r["frame_ms_same_cfg_resent_every_frame"] = round(frames("same"), 4)   # e.g. AFGS1 with only a new seed per frame
        r["frame_ms_switch_host_setters_only"] = round(frames("setters"), 4)   # identical state re-sent through the setters (recognised as unchanged)
        # 3. the reference firmware on one host core
        if T.have_reference():
            ref = T.ReferenceHW()
            ref.set_depth(10)
            ref.set_chroma_subsampling(2, 2)
            fn = ref.lib.vfgs_init_afgs1 if kind else ref.lib.vfgs_init_sei
            r["reference_cpu_ms"] = med(lambda: fn(C.byref(cfg)), 20)
        res[name] = r
    print(json.dumps({"what": "configuration switch (firmware layer), 10-bit 4:2:0; frame = 1920x1080 device-resident",
                      "cases": res}, indent=1))


if __name__ == "__main__":
    main()
