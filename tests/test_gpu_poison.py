"""No kernel may read device memory that nobody initialised.  vslam_debug_poison(byte) fills every device allocation made
from then on (and every reused scratch-pool block) with `byte`; the stage / closed-loop / batch parity tests are re-run
here, in this process, under two different fill values - a stage that depends on leftover memory content passes with
one and fails with the other (or differs from its unpoisoned run)."""
import ctypes as C
import numpy as np
import pytest
import synth

pytestmark = pytest.mark.gpu


@pytest.fixture
def poison(capi):
    L = capi.lib()
    L.vslam_debug_poison.restype = None

    def set_byte(b):
        L.vslam_debug_poison(C.c_int32(b))
    yield set_byte
    set_byte(-1)


@pytest.mark.parametrize("fill", [0xA5, 0xFF])
def test_tracking_stages_under_poisoned_allocations(oracle, capi, poison, fill):
    import test_gpu_track as tt
    import test_gpu_stereo as ts
    import test_gpu_extract as te
    poison(fill)
    tt.test_pipelined_two_extractor_pairs_match_serial(capi)
    for name in dir(ts):
        f = getattr(ts, name)
        if name.startswith("test_") and callable(f) and f.__code__.co_argcount == 2 and f.__code__.co_varnames[:2] == ("oracle", "capi"):
            f(oracle, capi)
    for name in dir(te):
        f = getattr(te, name)
        if name.startswith("test_") and callable(f) and f.__code__.co_argcount == 2 and f.__code__.co_varnames[:2] == ("oracle", "capi"):
            f(oracle, capi)


@pytest.mark.parametrize("fill", [0xA5, 0xFF])
def test_closed_loop_and_batch_under_poisoned_allocations(oracle, capi, poison, fill):
    import test_gpu_system as tsys
    import test_gpu_batch as tb
    poison(fill)
    ref, got, out = tsys._run(oracle, capi, "euroc", 1500, list(range(0, 44, 2)), use_imu=True)
    assert tsys._check(ref, got, out) >= 1
    tb.test_batch_lanes_equal_single_sessions(capi, False, 1, 0, 1)
    tb.test_batch_lanes_equal_single_sessions(capi, False, 2, 3, 1)
