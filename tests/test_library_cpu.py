"""CPU: the C-ABI library loads, exports every symbol include/vfgs_hip.h declares, keeps the
host-side state machine in step with the oracle where no kernel is needed, and fails LOUDLY
(error code, no silent fallback) when asked to process without a GPU."""
import ctypes as C
import re
import subprocess
import sys

import numpy as np
import pytest

import vfgs_testlib as T

import versatilefilmgrain_amd.build as B
from versatilefilmgrain_amd import hw


@pytest.fixture(scope="module")
def lib():
    B.build()
    return hw.load()


def test_header_symbols_all_exported(lib):
    from versatilefilmgrain_amd import fw
    header = (T.ROOT / "include" / "vfgs_hip.h").read_text() + (T.ROOT / "include" / "vfgs_hip_fw.h").read_text()
    declared = set(re.findall(r"\b(vfgs_[a-z0-9_]+)\s*\(", header))
    assert declared == set(hw.EXPORTS) | set(fw.EXPORTS), declared ^ (set(hw.EXPORTS) | set(fw.EXPORTS))
    for name in declared:
        assert hasattr(lib, name), name


def test_reference_interface_names_present(lib):
    # the ten entry points of /root/reference/src/vfgs_hw.h:51-62
    for name in ["vfgs_set_luma_pattern", "vfgs_set_chroma_pattern", "vfgs_set_scale_lut", "vfgs_set_pattern_lut",
                 "vfgs_set_seed", "vfgs_set_scale_shift", "vfgs_set_depth", "vfgs_set_legal_range",
                 "vfgs_set_chroma_subsampling", "vfgs_add_grain_line"]:
        assert hasattr(lib, name)


def test_firmware_interface_names_and_structure_layout(lib):
    """vfgs_fw.h:51-92: both entry points exported, parameter structures laid out like the
    reference's (the fixtures are raw dumps of the reference's own structures)."""
    import ctypes
    from versatilefilmgrain_amd import fw
    assert hasattr(lib, "vfgs_init_sei") and hasattr(lib, "vfgs_init_afgs1")
    sizes = {0: ctypes.sizeof(fw.FgsSei), 1: ctypes.sizeof(fw.FgsAfgs1)}
    for name in ("fgs_sei_10_420", "fgs_afgs1_test1_10_420"):
        seed, cfgs = T.load_fwcfg(name)
        assert seed == 12345
        for kind, raw in cfgs:
            assert len(raw) == sizes[kind]
            fw.struct_from_bytes(kind, raw)
    # the default SEI of the reference CLI (vfgs_main.c:69-110), as dumped: 8 luma intervals, scale 100.. , cut-offs 7..14
    _, cfgs = T.load_fwcfg("default_10_420")
    sei = fw.struct_from_bytes(*cfgs[0])
    assert sei.model_id == 0 and sei.log2_scale_factor == 5 and list(sei.num_intensity_intervals) == [8, 8, 8]
    assert [sei.comp_model_value[0][k][1] for k in range(8)] == [7, 8, 9, 10, 11, 12, 13, 14]
    _, cfgs = T.load_fwcfg("fgs_afgs1_test1_10_420")
    a = fw.struct_from_bytes(*cfgs[1])
    assert a.num_y_points >= 1 and 1 <= a.ar_coeff_lag <= 3 and 6 <= a.ar_coeff_shift <= 9


def test_seed_registers_after_set_seed_match_oracle(lib):
    h = hw.VfgsHip()
    o = T.OracleHW()
    assert h.seed_state() == o.seed_state() == (0xdeadbeef,) * 4
    for seed in (1, 12345, 0x7fffffff, 0xb0b3b0b3):
        h.set_seed(seed)
        o.set_seed(seed)
        assert h.seed_state() == o.seed_state()


def test_replaying_every_trace_does_not_need_a_gpu(lib):
    h = hw.VfgsHip()
    for name in T.list_traces():
        T.replay(h, T.load_trace(name))


def test_no_gpu_means_error_not_fallback():
    """In a GPU-less process a device call must return an error (run in a subprocess because
    other tests may run where a GPU exists)."""
    code = r"""
import sys, ctypes
sys.path.insert(0, %r)
from versatilefilmgrain_amd import hw
lib = hw.load()
import torch
if torch.cuda.is_available():
    print("HAVE_GPU"); sys.exit(0)
buf = ctypes.create_string_buffer(1 << 20)
p = (ctypes.addressof(buf) + 255) & ~255
lib.vfgs_set_depth(10)
rc = lib.vfgs_hip_add_grain_frame_dev(p, p, p, 192, 144, 192, 128, None)
print("RC", rc, lib.vfgs_hip_last_error_string().decode())
sys.exit(0 if rc != 0 else 3)
""" % str(T.ROOT)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "HAVE_GPU" in r.stdout or "RC" in r.stdout


def test_header_is_valid_c_and_links(lib, tmp_path):
    """gcc -std=c99: the public header as C, every entry point resolved, host-only calls executed."""
    exe = tmp_path / "c_abi_check"
    libdir = B.LIB.parent
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Werror", "-pedantic", f"-I{T.ROOT / 'include'}", str(T.ROOT / "tests" / "c_abi_check.c"),
                    f"-L{libdir}", "-lvfgs_hip", f"-Wl,-rpath,{libdir}", "-o", str(exe)], check=True)
    r = subprocess.run([str(exe)], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "56 entry points" in r.stdout and "0x00006072" in r.stdout


def test_product_library_is_not_a_developer_build(lib):
    """The shipped library carries every tuning / ablation knob at its default (vfgs_layout.h); the diagnostics of the
    benchmarks live in a library of their own and are not part of the product ABI."""
    assert lib.vfgs_hip_dev_build() == 0
    assert not hasattr(lib, "vfgs_hip_diag_stream")
    header = (T.ROOT / "include" / "vfgs_hip.h").read_text()
    assert "diag" not in header


@pytest.mark.parametrize("flag", ["-DVFGS_WAVES=8", "-DVFGS_RW_CONSEC=1", "-DVFGS_NO_FRONTS", "-DVFGS_LDAUX_ALIGNED=0"])
def test_a_stray_knob_is_a_build_error(flag, tmp_path):
    """A -D that changes what the kernels compute (or how) must not produce a product library: without VFGS_DEV_BUILD the
    layout header refuses it at compile time."""
    src = tmp_path / "knob.cpp"
    src.write_text('#include "vfgs_layout.h"\nint main() { return vfgs::kWavesPerWG; }\n')
    inc = T.ROOT / "versatilefilmgrain_amd" / "csrc"
    bad = subprocess.run(["g++", "-std=c++17", "-fsyntax-only", f"-I{inc}", flag, str(src)], capture_output=True, text=True)
    assert bad.returncode != 0 and "VFGS_DEV_BUILD" in bad.stderr
    ok = subprocess.run(["g++", "-std=c++17", "-fsyntax-only", f"-I{inc}", flag, "-DVFGS_DEV_BUILD", str(src)], capture_output=True, text=True)
    assert ok.returncode == 0, ok.stderr
    plain = subprocess.run(["g++", "-std=c++17", "-fsyntax-only", f"-I{inc}", str(src)], capture_output=True, text=True)
    assert plain.returncode == 0, plain.stderr
