"""CPU: oracle/vfgs_fw_oracle.py (numpy restatement of the reference's pattern generators) against
the pattern bytes the real reference firmware programmed (tests/golden/traces)."""
import sys

import numpy as np
import pytest

import vfgs_testlib as T

sys.path.insert(0, str(T.ROOT / "oracle"))
import vfgs_fw_oracle as F  # noqa: E402

from versatilefilmgrain_amd import fw  # noqa: E402


def programmed(name):
    want = T.BankModel()
    T.replay(want, T.load_trace(name))
    _, cfgs = T.load_fwcfg(name)
    return want, fw.struct_from_bytes(*cfgs[-1])


@pytest.mark.parametrize("name", ["fgs_sei_10_420", "fgs_sei_ff_test6_10_420", "default_8_420"])
def test_frequency_filtered_patterns(name):
    want, sei = programmed(name)
    # interval 0 of a component with ascending bounds is pattern 0 of its bank (vfgs_fw.c:552-568)
    v = sei.comp_model_value[0][0]
    assert np.array_equal(F.ff_pattern(64, v[1], v[2], 0), want.luma[0])
    if sei.comp_model_present_flag[1]:
        v = sei.comp_model_value[1][0]
        assert np.array_equal(F.ff_pattern(32, v[1], v[2], 1), want.chroma[0][:32, :32])


def test_sei_auto_regressive_pattern():
    want, sei = programmed("fgs_sei_ar_test1_10_420")
    v = list(sei.comp_model_value[0][0])
    taps = F.sei_ar_taps(v, sei.log2_scale_factor)
    assert np.array_equal(F.ar_pattern(False, taps, sei.log2_scale_factor, 1, 0), want.luma[0])


@pytest.mark.parametrize("name", ["fgs_afgs1_test1_10_420", "fgs_afgs1_test3_8_420"])
def test_afgs1_patterns(name):
    want, a = programmed(name)
    lag, scale, shift = a.ar_coeff_lag, a.ar_coeff_shift, a.grain_scale_shift + 1
    assert np.array_equal(F.ar_pattern(False, F.afgs1_taps(list(a.ar_coeffs_y), lag), scale, shift, 0), want.luma[0])
    assert np.array_equal(F.ar_pattern(True, F.afgs1_taps(list(a.ar_coeffs_cb), lag), scale, shift, 1), want.chroma[0][:32, :32])
    assert np.array_equal(F.ar_pattern(True, F.afgs1_taps(list(a.ar_coeffs_cr), lag), scale, shift, 2), want.chroma[1][:32, :32])
