"""The bound-geometry contract is checked on the device (include/psm.h, psm_bind_geometry): bind obstacle A, solve
obstacle B.  Guard waves riding in the strip-dots launch compare the flow-cell pattern of the grid being solved with the
bound one; on a mismatch the solve's field is NaN everywhere (never a plausible field of the wrong geometry), the
host-buffer entries drop the binding and solve again on the general path, and the device-pointer entry reports
PSM_ERR_GEOMETRY at the next psm_synchronize.  Needs a real MI355X."""
import numpy as np
import pytest

from psm_amd import GridSurrogate, _lib, synthetic
from hipmem import DeviceArray

pytestmark = pytest.mark.gpu


def two_obstacles(ny, nx, noise=0.02):
    a = synthetic.channel_grid(ny, nx, seed=1, noise=noise).astype(np.float32)
    b = synthetic.channel_grid(ny, nx, seed=1, noise=noise, cx=0.55, cy=0.42, r=0.1).astype(np.float32)
    assert not np.array_equal(a[..., 2] != 0, b[..., 2] != 0)
    return a, b


@pytest.mark.parametrize("variant,ny,nx,precision", [("gradp", 256, 256, "f32"), ("deltas", 256, 256, "f32"), ("deltas", 512, 512, "bf16"),
                                                     ("chapter5", 400, 1500, "f32")])
def test_host_entries_fall_back_to_the_general_path(variant, ny, nx, precision):
    """psm_solve_grid on obstacle B while A is bound: the field of B's general-path solve, binding dropped, one trip."""
    model = synthetic.make_model(variant, p_in=48, p_out=40)
    a, b = two_obstacles(ny, nx)
    with GridSurrogate(model, ny, nx, precision=precision) as sur:
        want_a, want_b = sur.solve(a)[0], sur.solve(b)[0]            # general path
        assert sur.bind_geometry(a)
        got_a = sur.solve(a)[0]
        assert sur.geometry_bound and sur.guard_trips == 0
        # same geometry, other velocities: no trip
        a2 = a.copy(); a2[..., :2] *= 0.5
        assert np.isfinite(sur.solve(a2)[0]).all() and sur.guard_trips == 0 and sur.geometry_bound
        got_b = sur.solve(b)[0]
        assert sur.guard_trips == 1 and not sur.geometry_bound
        assert "not the one bound" in _lib.last_error(sur.h)
        np.testing.assert_array_equal(got_b, want_b)                 # solved again on the general path
        assert np.abs(got_a - want_a).max() <= (2e-5 if precision == "f32" else 5e-3) * np.abs(want_a).max()
        # one pixel of difference is enough (a flow cell turned solid, last pixel of the grid included)
        for (y, x) in ((ny - 1, nx - 1), (0, 0), (ny // 2, 65)):
            c = a.copy(); c[y, x, 2] = 0.0 if a[y, x, 2] != 0 else 0.3
            assert sur.bind_geometry(a)
            got = sur.solve(c)[0]
            assert not sur.geometry_bound
            np.testing.assert_array_equal(got, sur.solve(c)[0])
        assert sur.guard_trips == 4


def test_device_entry_poisons_the_field_and_reports_at_synchronize(monkeypatch):
    model = synthetic.make_model("gradp", p_in=64, p_out=64)
    a, b = two_obstacles(256, 256)
    for graph in ("0", "1"):
        monkeypatch.setenv("PSM_GRAPH", graph)
        with GridSurrogate(model, 256, 256) as sur:
            want_b = sur.solve(b)[0]
            assert sur.bind_geometry(a)
            d_a, d_b = DeviceArray(a), DeviceArray(b)
            d_out = DeviceArray(shape=(256, 256, 2), dtype=np.float32)
            sur.solve_device(d_a.ptr, 1, d_out.ptr, 0)
            sur.synchronize()
            assert np.isfinite(d_out.numpy()).all() and sur.guard_trips == 0
            sur.solve_device(d_b.ptr, 1, d_out.ptr, 0)
            with pytest.raises(_lib.PsmError) as e:
                sur.synchronize()
            assert e.value.code == -7                                # PSM_ERR_GEOMETRY
            assert np.isnan(d_out.numpy()).all()                     # never a plausible field of the wrong geometry
            assert sur.guard_trips == 1 and not sur.geometry_bound
            sur.solve_device(d_b.ptr, 1, d_out.ptr, 0)               # binding dropped: the general path
            sur.synchronize()
            np.testing.assert_array_equal(d_out.numpy(), want_b)
            # the flags are per solve: A again after a re-bind is clean
            assert sur.bind_geometry(a)
            sur.solve_device(d_a.ptr, 1, d_out.ptr, 0)
            sur.synchronize()
            assert np.isfinite(d_out.numpy()).all() and sur.guard_trips == 1
            for d in (d_a, d_b, d_out):
                d.free()


def test_guard_can_be_switched_off(monkeypatch):
    monkeypatch.setenv("PSM_NO_GUARD", "1")
    model = synthetic.make_model("deltas", p_in=32, p_out=32)
    a, b = two_obstacles(256, 256)
    with GridSurrogate(model, 256, 256) as sur:
        assert sur.bind_geometry(a)
        got = sur.solve(b)[0]                                        # the contract is the caller's again: a wrong field
        assert sur.geometry_bound and sur.guard_trips == 0 and np.isfinite(got).all()


@pytest.mark.parametrize("pull", ["0", "1"])
def test_ring_tickets_on_another_geometry_are_solved_again(pull, monkeypatch):
    monkeypatch.setenv("PSM_RING_PULL", pull)
    model = synthetic.make_model("deltas", p_in=32, p_out=32)
    a, b = two_obstacles(256, 256)
    with GridSurrogate(model, 256, 256) as sur:
        want_a, want_b = sur.solve(a)[0], sur.solve(b)[0]
        assert sur.bind_geometry(a)
        bound_a = sur.solve(a)[0]
        t0 = sur.submit(a, out_scale=[1.0])
        t1 = sur.submit(b, out_scale=[0.7])
        t2 = sur.submit(a)
        f0 = sur.wait(t0)[0]
        np.testing.assert_array_equal(f0, bound_a)
        f1 = sur.wait(t1)[0]                                         # trips: binding dropped, ticket solved again
        assert sur.guard_trips == 1 and not sur.geometry_bound
        np.testing.assert_array_equal(f1, sur.solve(b, out_scale=[0.7])[0])
        f2 = sur.wait(t2)[0]                                         # was in flight on the bound path: still valid
        np.testing.assert_array_equal(f2, bound_a)
        # zero-copy slots
        assert sur.bind_geometry(a)
        t, gin, fout = sur.ring_acquire()
        gin[0] = b
        sur.ring_submit(t, 1)
        sur.ring_wait(t)
        assert sur.guard_trips == 2
        np.testing.assert_array_equal(fout[0], want_b)
        assert np.abs(bound_a - want_a).max() <= 2e-5 * np.abs(want_a).max()


def test_case_batch_guard():
    """One geometry per case slot: the same cases in another order are another geometry."""
    model = synthetic.make_model("deltas", p_in=32, p_out=32)
    grids = synthetic.random_obstacle_cases(8, 256, 256, seed=3).astype(np.float32)
    swapped = grids[[1, 0, 2, 3, 4, 5, 6, 7]].copy()
    with GridSurrogate(model, 256, 256, max_cases=8) as sur:
        want = sur.solve(swapped)
        assert sur.bind_geometry(grids)
        got = sur.solve(grids)
        assert np.isfinite(got).all() and sur.guard_trips == 0 and sur.geometry_bound
        got = sur.solve(swapped)
        assert sur.guard_trips == 1 and not sur.geometry_bound
        np.testing.assert_array_equal(got, want)


@pytest.mark.parametrize("odd_case", [0, 17, 39])
def test_large_batch_guard_riders_are_dealt_over_the_dense_launches(odd_case):
    """From 256 guard workgroups up (more than 16 cases of 256 x 256) the riders are dealt over ALL Dense launches of the step
    (psm_api_solve.cpp, PSM_GUARD_SPREAD), one flag per guard workgroup: a single foreign case anywhere in the batch -- in the
    share of the first hidden layer, of a middle one, of the head launch -- must still trip the guard."""
    n = 40
    model = synthetic.make_model("deltas", p_in=32, p_out=32)
    grids = synthetic.random_obstacle_cases(n, 256, 256, seed=3).astype(np.float32)
    other = synthetic.random_obstacle_cases(n, 256, 256, seed=11).astype(np.float32)
    foreign = grids.copy()
    foreign[odd_case] = other[odd_case]
    assert not np.array_equal(foreign[odd_case, ..., 2] != 0, grids[odd_case, ..., 2] != 0)
    with GridSurrogate(model, 256, 256, max_cases=n) as sur:
        want = sur.solve(foreign)                      # general path
        assert sur.bind_geometry(grids)
        got = sur.solve(grids)
        assert np.isfinite(got).all() and sur.guard_trips == 0 and sur.geometry_bound
        got = sur.solve(foreign)
        assert sur.guard_trips == 1 and not sur.geometry_bound
        np.testing.assert_array_equal(got, want)
        # the device-pointer entry: the whole field of the step is NaN (never a plausible field of the wrong geometry)
        assert sur.bind_geometry(grids)
        d_in, d_out = DeviceArray(foreign), DeviceArray(shape=(n, 256, 256, 1))
        sur.solve_device(d_in.ptr, n, d_out.ptr, 0)
        with pytest.raises(_lib.PsmError) as e:
            sur.synchronize()
        assert e.value.code == -7                          # PSM_ERR_GEOMETRY
        assert np.isnan(d_out.numpy()).all()
        d_in.free(); d_out.free()
