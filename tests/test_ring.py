"""The host-buffer ring (include/psm.h: psm_ring_acquire / submit / wait, psm_host_register + psm_submit_grid_io,
psm_submit_grid / psm_wait_grid): every way in gives the synchronous entry's fields bit for bit, on the general and on
the geometry-bound path, with graphs and with plain launches; plus the two measurement entries bench.py uses."""
import ctypes as C
import os

import numpy as np
import pytest

from psm_amd import GridSurrogate, _lib, synthetic
from hipmem import DeviceArray

pytestmark = pytest.mark.gpu


def _three_ways(sur, grids, scales, n_cases=1):
    """-> fields through (zero-copy slots, registered memory, pageable memory), `depth` 3."""
    outs = {"zero_copy": [], "registered": [], "pageable": []}
    # zero-copy: pack into the slot, read the result from the slot
    pend = []
    for g, sc in zip(grids, scales):
        if len(pend) == 3:
            t, fo = pend.pop(0)
            sur.ring_wait(t)
            outs["zero_copy"].append(fo[:n_cases].copy())
        t, gi, fo = sur.ring_acquire()
        gi[:n_cases] = g
        sur.ring_submit(t, n_cases, out_scale=sc)
        pend.append((t, fo))
    for t, fo in pend:
        sur.ring_wait(t)
        outs["zero_copy"].append(fo[:n_cases].copy())
    # registered caller memory: direct DMA in and out
    gin = np.ascontiguousarray(np.stack(grids))
    gout = np.full((len(grids), n_cases, sur.ny, sur.nx, sur.model.c_out), np.nan, np.float32)
    sur.host_register(gin)
    sur.host_register(gout)
    pend = []
    for k, sc in enumerate(scales):
        if len(pend) == 3:
            sur.wait(pend.pop(0))
        pend.append(sur.submit(gin[k], out_scale=sc, out=gout[k]))
    while pend:
        sur.wait(pend.pop(0))
    sur.host_unregister(gin)
    sur.host_unregister(gout)
    outs["registered"] = [gout[k] for k in range(len(grids))]
    # pageable
    pend = []
    for g, sc in zip(grids, scales):
        if len(pend) == 3:
            outs["pageable"].append(sur.wait(pend.pop(0)))
        pend.append(sur.submit(g, out_scale=sc))
    while pend:
        outs["pageable"].append(sur.wait(pend.pop(0)))
    return outs


@pytest.mark.parametrize("graph,pull", [("1", "0"), ("0", "0"), ("1", "1"), ("0", "1")])
@pytest.mark.parametrize("bind", [False, True])
def test_every_way_into_the_ring_equals_the_synchronous_entry(bind, graph, pull, monkeypatch):
    """graph: the kernels of a ticket are one replay / plain launches; pull: hipMemcpyAsync copies around the kernels (0,
    default) / the GPU pulls the grid from and stores the field to the mapped pinned buffers itself (1)."""
    monkeypatch.setenv("PSM_RING_GRAPH", graph)
    monkeypatch.setenv("PSM_RING_PULL", pull)
    model = synthetic.make_model("gradp", p_in=48, p_out=40)
    grids = [synthetic.channel_grid(256, 256, seed=1 + s).astype(np.float32)[None] for s in range(9)]
    # the ring test rotates VELOCITY fields of one geometry (a case stream); scales change per step
    for g in grids[1:]:
        g[..., 2] = grids[0][..., 2]
    scales = [None if s % 3 == 0 else [0.7 + 0.1 * s] for s in range(9)]
    with GridSurrogate(model, 256, 256) as sur:
        if bind:
            assert sur.bind_geometry(grids[0][0])
        ref = [sur.solve(g, out_scale=sc) for g, sc in zip(grids, scales)]
        outs = _three_ways(sur, grids, scales)
        for name, got in outs.items():
            assert len(got) == len(ref)
            for a, b in zip(got, ref):
                np.testing.assert_array_equal(a, b, err_msg=name)
        # synchronous entry on registered memory: same field, no staging copies
        g0, o0 = grids[4].copy(), np.empty_like(ref[4])
        sur.host_register(g0); sur.host_register(o0)
        sur._chk(sur.lib.psm_solve_grid(sur.h, g0.ctypes.data_as(C.POINTER(C.c_float)), 1, None, o0.ctypes.data_as(C.POINTER(C.c_float))))
        np.testing.assert_array_equal(o0, sur.solve(grids[4]))
        sur.host_unregister(g0); sur.host_unregister(o0)


def test_ring_case_batch_and_rebinding():
    """Two cases per ticket; binding / unbinding between tickets re-captures the slot graphs."""
    model = synthetic.make_model("deltas", p_in=32, p_out=32)
    cases = synthetic.random_obstacle_cases(2, 256, 256, seed=9).astype(np.float32)
    steps = [cases * np.float32(1.0 + 0.1 * s) for s in range(6)]
    for g in steps:
        g[..., 2] = cases[..., 2]
    with GridSurrogate(model, 256, 256, max_cases=2) as sur:
        ref = [sur.solve(g) for g in steps]
        for bound in (False, True, False):
            if bound:
                assert sur.bind_geometry(cases)
            elif sur.geometry_bound:
                sur.unbind_geometry()
            outs = _three_ways(sur, steps, [None] * len(steps), n_cases=2)
            for name, got in outs.items():
                for a, b in zip(got, ref):
                    assert np.abs(a - b).max() <= 2e-5 * np.abs(b).max(), name


def test_ring_state_errors():
    model = synthetic.make_model("deltas", p_in=16, p_out=16)
    g = synthetic.channel_grid(256, 256, seed=3).astype(np.float32)
    with GridSurrogate(model, 256, 256) as sur:
        ts = []
        for _ in range(8):                              # PSM_RING_SLOTS
            t, gi, fo = sur.ring_acquire()
            gi[0] = g
            ts.append(t)
        with pytest.raises(_lib.PsmError) as e:
            sur.ring_acquire()                          # all slots handed out
        assert e.value.code == -2
        with pytest.raises(_lib.PsmError):
            sur.ring_wait(ts[0])                        # acquired, not submitted
        for t in ts:
            sur.ring_submit(t, 1)
        with pytest.raises(_lib.PsmError):
            sur.ring_submit(ts[0], 1)                   # already submitted
        for t in ts:
            sur.ring_wait(t)
        with pytest.raises(_lib.PsmError):
            sur.ring_wait(ts[0])                        # already waited for
        # psm_ring_release: an acquired ticket the caller decides not to submit gives its slot back
        t, gi, fo = sur.ring_acquire()
        sur.ring_release(t)
        for bad in (sur.ring_release, sur.ring_wait):
            with pytest.raises(_lib.PsmError):
                bad(t)                                  # released: neither acquired nor in flight any more
        with pytest.raises(_lib.PsmError):
            sur.ring_release(ts[0])                     # long gone
        ref = sur.solve(g)[0]
        for _ in range(9):                              # a full turn of the ring passes over the released slot
            t, gi, fo = sur.ring_acquire()
            gi[0] = g
            sur.ring_submit(t, 1)
            sur.ring_wait(t)
            np.testing.assert_array_equal(fo[0], ref)
        with pytest.raises(_lib.PsmError):
            sur.host_unregister(g)                      # never registered
        sur.host_register(g)
        with pytest.raises(_lib.PsmError):
            sur.host_register(g)                        # twice
        sur.host_unregister(g)


def test_bench_host_and_kernel_timing_entries():
    """psm_bench_host (C++ loop through the public entries) returns the synchronous entry's field in every mode;
    psm_time_kernels names every kernel of the bound 6-launch solve."""
    model = synthetic.make_model("gradp", p_in=32, p_out=32)
    grids = [synthetic.channel_grid(256, 256, seed=1).astype(np.float32) for _ in range(3)]
    for i, g in enumerate(grids):
        g[..., :2] *= np.float32(1.0 + 0.1 * i)
    host = np.ascontiguousarray(np.stack(grids))
    with GridSurrogate(model, 256, 256) as sur:
        assert sur.bind_geometry(grids[0])
        ref = [sur.solve(g) for g in grids]
        for mode, depth, steps, warm in ((0, 1, 7, 2), (1, 3, 11, 0), (2, 2, 9, 1), (2, 4, 12, 3), (3, 3, 10, 0)):
            last = np.empty((1, 256, 256, 2), np.float32)
            sec = C.c_double()
            sur._chk(sur.lib.psm_bench_host(sur.h, host.ctypes.data_as(C.POINTER(C.c_float)), 3, 1, mode, depth, steps, warm,
                                            C.byref(sec), last.ctypes.data_as(C.POINTER(C.c_float))))
            assert sec.value > 0
            if mode == 3:      # the slots were packed once, during the first turn of the ring: slot s holds input s % 3
                wu = max(warm, 8)
                want = ref[((wu + steps - 1) % 8) % 3]
            else:
                want = ref[(warm + steps - 1) % 3]
            np.testing.assert_array_equal(last, want, err_msg=f"mode {mode}")
        d_in, d_out = DeviceArray(grids[0]), DeviceArray(shape=(256, 256, 2))
        cap = 16
        names = C.create_string_buffer(cap * 64)
        ms, cnt, nk = (C.c_double * cap)(), (C.c_int64 * cap)(), C.c_int32()
        sur._chk(sur.lib.psm_time_kernels(sur.h, C.c_void_p(d_in.ptr), 1, C.c_void_p(d_out.ptr), 70, names, ms, cnt, cap, C.byref(nk)))
        got = {names.raw[k * 64:(k + 1) * 64].split(b"\0", 1)[0].decode(): (ms[k], cnt[k]) for k in range(nk.value)}
        base = {}
        for nm, (t, n) in got.items():
            b = nm.split("<")[0]
            base[b] = base.get(b, 0) + n
        assert {"psm_encode_kernel", "psm_reduce_dense1_kernel", "psm_dense_kernel", "psm_decode_paste_kernel"} <= set(base), got
        assert base["psm_encode_kernel"] == 70 and base["psm_dense_kernel"] == 3 * 70 and base["psm_decode_paste_kernel"] == 70
        dense = sorted(nm for nm in got if nm.startswith("psm_dense_kernel"))
        assert [nm.split("#")[1] for nm in dense] == ["layer1", "layer2", "layer3"]    # one entry per Dense launch; the last is head + strip dots
        assert all(got[nm][1] == 70 for nm in dense)
        assert all(0 < t / n < 1.0 for t, n in got.values())            # every dispatch took between 0 and 1 ms
        np.testing.assert_array_equal(d_out.numpy()[None], ref[0])
