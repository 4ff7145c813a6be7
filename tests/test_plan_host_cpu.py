"""CPU: the host-side tables of the library's plan layer (fdm_schedule_host, fdm_ddim_schedule_host, fdm_alibi_slopes_host,
fdm_pe_table_host, fdm_model_preset) against the golden vectors produced by the reference itself and against the
host-side Python restatement.  No device is touched."""
import ctypes as C

import numpy as np
import torch

from fdm_amd import _lib, presets, schedule
from fdm_amd.denoiser import model_desc


def ulps(a, b):
    a, b = np.asarray(a, np.float32), np.asarray(b, np.float32)
    return float(np.max(np.abs(a - b) / np.spacing(np.maximum(np.abs(a), np.abs(b)))))


def test_schedule_buffers_bit_equal_to_reference(golden):
    g = golden("schedule")
    out = np.zeros((12, 1000), np.float32)
    assert _lib.lib().fdm_schedule_host(1000, out.ctypes.data) == 0
    for i, name in enumerate(schedule.BUFFER_NAMES):
        assert np.array_equal(out[i], g[name]), name          # the reference's registered buffers, bit for bit


def test_ddim_pairs_and_tables():
    buf = schedule.make_buffers(1000)
    for steps in (1, 2, 3, 7, 50, 100, 250, 333, 500, 999, 1000):
        t, tn = np.zeros(steps, np.int32), np.zeros(steps, np.int32)
        sa, cn = np.zeros(steps, np.float32), np.zeros(steps, np.float32)
        n = _lib.lib().fdm_ddim_schedule_host(steps, 1000, t.ctypes.data, tn.ctypes.data, sa.ctypes.data, cn.ctypes.data)
        pairs = [p for p in schedule.ddim_time_pairs(steps) if p[1] >= 0]
        assert n == len(pairs)
        assert list(zip(t[:n].tolist(), tn[:n].tolist())) == [(int(a), int(b)) for a, b in pairs]
        if n:
            # sqrt(abar_next), sqrt(1 - abar_next): correctly rounded here; torch's vectorised CPU sqrt is within 1 ulp of that
            san, c = schedule.ddim_tables(buf, pairs)
            an = buf["alphas_cumprod"][torch.tensor([p[1] for p in pairs])].numpy()
            assert np.array_equal(sa[:n], np.sqrt(an)) and np.array_equal(cn[:n], np.sqrt(np.float32(1) - an))
            assert ulps(sa[:n], san.numpy()) <= 1 and ulps(cn[:n], c.numpy()) <= 1


def test_alibi_slopes_bit_equal(golden):
    for nh in (2, 4, 6, 8, 12, 16):
        o = np.zeros(nh, np.float32)
        assert _lib.lib().fdm_alibi_slopes_host(nh, o.ctypes.data) == 0
        assert np.array_equal(o, np.array(schedule.alibi_slopes(nh), np.float32)), nh


def test_positional_tables_are_a_valid_fp32_evaluation():
    """The reference builds PE with fp32 torch ops (exp, multiply, sin / cos); a 1-ulp difference in exp() is multiplied by
    the position, so no two fp32 libms agree bit for bit -- which is why the plan takes the registered buffer "PE.pe" when
    given.  The library's fallback must be as close to the exact table as torch's own evaluation is."""
    import math
    for d, periodic, period in ((1024, 1, 30), (512, 0, 30), (1024, 0, 25), (256, 1, 30)):
        o = np.zeros((630, d), np.float32)
        assert _lib.lib().fdm_pe_table_host(d, periodic, period, 630, o.ctypes.data) == 0
        ref = schedule.positional_table(d, "periodic" if periodic else "sinus", period, 630).numpy()
        pos = (np.arange(630) % period if periodic else np.arange(630)).astype(np.float64)[:, None]
        arg = pos * np.exp(np.arange(0, d, 2, dtype=np.float64) * (-math.log(10000.0) / d))[None, :]
        exact = np.zeros((630, d))
        exact[:, 0::2], exact[:, 1::2] = np.sin(arg), np.cos(arg)
        e_lib, e_torch = np.max(np.abs(o - exact)), np.max(np.abs(ref - exact))
        assert e_lib <= max(1.5 * e_torch, 2e-7), (d, periodic, e_lib, e_torch)
        assert np.max(np.abs(o - ref)) < 1e-4


def test_model_presets_match_python_presets():
    for name in ("vocaset", "mead", "biwi", "vocaset_tiny", "mead_tiny"):
        d = _lib.ModelDesc()
        assert _lib.lib().fdm_model_preset(name.encode(), C.byref(d)) == 0
        ref = model_desc(presets.get(name))
        assert all(getattr(d, f) == getattr(ref, f) for f, _ in _lib.ModelDesc._fields_), name
    assert _lib.lib().fdm_model_preset(b"nope", C.byref(_lib.ModelDesc())) != 0


def test_plan_create_validates_without_a_device():
    h = C.c_void_p()
    d = model_desc(presets.get("vocaset"))
    d.n_head = 3
    assert _lib.lib().fdm_plan_create(C.byref(d), 1, 10, 0, 0, C.byref(h)) != 0
    assert b"geometry" in _lib.lib().fdm_last_error()
    d = model_desc(presets.get("vocaset"))
    assert _lib.lib().fdm_plan_create(C.byref(d), 1, 601, 0, 0, C.byref(h)) != 0       # L > 600 (models/fdm_vocaset.py:44)
    if not _lib.lib().fdm_device_ok():
        assert _lib.lib().fdm_plan_create(C.byref(d), 1, 10, 0, 0, C.byref(h)) != 0     # no CPU fallback
        assert b"no gfx950" in _lib.lib().fdm_last_error()
