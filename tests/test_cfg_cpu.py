"""CPU: the configuration-file reader of the library (include/vfgs_hip_fw.h, vfgs_hip_cfg_*)
against the reference CLI.

Input: the reference's configuration files (tests/golden/cfg_corpus.npz, data).  Expected: the
parameter structures the reference CLI handed to its firmware for `-b <depth> -f <fmt> -c <file>`
(tests/golden/fwcfg, raw dumps written by oracle/ref_harness.c --dump-cfg), byte for byte,
including the CLI's start-up pass over its built-in defaults and its refusals."""
import ctypes as C

import numpy as np
import pytest

import vfgs_testlib as T

import versatilefilmgrain_amd.build as B
from versatilefilmgrain_amd import fw, hw

NAMES = sorted(p.stem for p in T.FWCFG.glob("*.npz"))


@pytest.fixture(scope="module")
def corpus(tmp_path_factory):
    B.build()
    d = tmp_path_factory.mktemp("cfg")
    with np.load(T.GOLDEN / "cfg_corpus.npz") as z:
        for name in z.files:
            (d / name).write_bytes(z[name].tobytes())
    return d


def split_name(name):
    stem, depth, fmt = name.rsplit("_", 2)
    if stem == "default":
        return None, int(depth), int(fmt)
    for tag, ext in (("_tbl", ".tbl"), ("_txt", ".txt")):
        if stem.endswith(tag):
            return stem[:-len(tag)] + ext, int(depth), int(fmt)
    return stem + ".cfg", int(depth), int(fmt)


def raw(s):
    return C.string_at(C.addressof(s), C.sizeof(s))


@pytest.mark.parametrize("name", NAMES)
def test_cfg_file_to_structures_equals_reference_cli(corpus, name):
    file, depth, fmt = split_name(name)
    _, want = T.load_fwcfg(name)
    st = fw.Cfg.defaults()
    if fmt == 420:                      # the stock CLI checks its defaults first (vfgs_main.c:739)
        assert st.check(fmt, depth) == 0
    st.adjust_chroma(fmt)               # vfgs_main.c:752-753, on the defaults
    st.apply_gain(100)
    assert want[0][0] == 0 and raw(st.sei) == want[0][1]
    if file is None:
        assert len(want) == 1
        return
    ok = st.read(corpus / file) == 0    # vfgs_main.c:595-604 pop_cfg
    if ok and fmt == 420:               # the 4:2:2 / 4:4:4 fixtures were recorded with the check skipped (ref_harness --no-check)
        ok = st.check(fmt, depth) == 0
    if not ok:
        assert len(want) == 1, hw.load().vfgs_hip_last_error_string()
        return
    st.adjust_chroma(fmt)
    st.apply_gain(100)
    assert len(want) == 2
    kind = 1 if st.afgs1.num_y_points else 0
    assert kind == want[1][0]
    assert raw(st.active) == want[1][1]


def test_all_three_syntaxes_are_in_the_corpus(corpus):
    kinds = {"cfg": 0, "tbl": 0, "txt": 0}
    for p in corpus.iterdir():
        kinds[p.suffix[1:]] += 1
    assert kinds["cfg"] >= 20 and kinds["tbl"] >= 1 and kinds["txt"] >= 1


def test_refusals_carry_the_reason(corpus):
    lib = hw.load()
    st = fw.Cfg.defaults()
    assert st.read(corpus / "does_not_exist.cfg") == 1
    assert b"Can not open" in lib.vfgs_hip_last_error_string()
    junk = corpus / "junk.cfg"
    junk.write_text("Hello : 1\nWorld : 2\n")
    assert st.read(junk) == 1 and b"could not" in lib.vfgs_hip_last_error_string()
    st = fw.Cfg.defaults()
    assert st.check(444, 10) == 1 and b"color grain" in lib.vfgs_hip_last_error_string()
    st = fw.Cfg.defaults()
    st.sei.comp_model_value[0][3][1] = 15
    assert st.check(420, 10) == 1 and b"horizontal cutoff" in lib.vfgs_hip_last_error_string()
    bad = corpus / "bad_afgs1.cfg"
    bad.write_text("AFGS1NumYPoints : 2\nAFGS1GrainScaling : 12\n")
    st = fw.Cfg.defaults()
    assert st.read(bad) == 1 and b"8..11" in lib.vfgs_hip_last_error_string()


def test_gain_moves_powers_of_two_into_the_shift(corpus):
    st = fw.Cfg.defaults()
    st.apply_gain(400)          # vfgs_main.c:561-593: 400 -> 200 -> 100: two halvings of the shift, scale x 1.00
    assert st.sei.log2_scale_factor == 3 and st.sei.comp_model_value[0][0][0] == 100
    st = fw.Cfg.defaults()
    st.apply_gain(30)           # 30 -> 60: one more shift, scale x 0.60
    assert st.sei.log2_scale_factor == 6 and st.sei.comp_model_value[0][7][0] == 108
    st = fw.Cfg.defaults()
    assert st.read(corpus / "fgs_afgs1_test1.cfg") == 0
    g, y0 = st.afgs1.grain_scaling, st.afgs1.point_y_scaling[1]
    st.apply_gain(150)          # 150 -> 75
    assert st.afgs1.grain_scaling == g - 1 and st.afgs1.point_y_scaling[1] == y0 * 75 // 100
