"""Handle life cycle on the GPU: no device-memory leak over create / plan / solve / destroy cycles, re-planning a
handle for another grid size, destroying a handle with tickets still in flight, and two handles interleaved on the
same device (what a host that owns several independent cases does)."""
import ctypes as C
import os

import numpy as np
import pytest

from oracle import psm_oracle as orc, unet_oracle as uo
from psm_amd import GridSurrogate, UNetSurrogate, _lib, synthetic
from test_oracle_golden import oracle_model

pytestmark = pytest.mark.gpu


def _free_bytes():
    import hipmem
    h = hipmem.hip()
    h.hipMemGetInfo.argtypes = [C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]
    assert h.hipDeviceSynchronize() == 0
    free, total = C.c_size_t(), C.c_size_t()
    assert h.hipMemGetInfo(C.byref(free), C.byref(total)) == 0
    return free.value


@pytest.mark.skipif(os.environ.get("PSM_GUARD_PAGES") in ("1", "2"), reason="guard-page allocator: freed ranges stay reserved and every buffer takes whole granules")
def test_no_device_memory_leak_over_handle_cycles():
    model = synthetic.make_model("deltas", p_in=16, p_out=16)
    grid = synthetic.channel_grid(256, 256, seed=1).astype(np.float32)
    W = uo.he_weights(uo.unet_specs(3, (16, 32), 1), seed=1)

    def cycle():
        with GridSurrogate(model, 256, 256, max_cases=2) as sur:
            sur.solve(grid)
            t = sur.submit(grid)                     # left in flight on purpose: destroy must drain it
            del t
        with UNetSurrogate(W, 64, 64, widths=(16, 32)) as net:
            net.forward(grid[:64, :64])
    cycle()                                          # first cycle: runtime-internal pools are created
    before = _free_bytes()
    for _ in range(5):
        cycle()
    after = _free_bytes()
    assert before - after < (8 << 20), f"device memory shrank by {(before - after) >> 20} MiB over 5 cycles"


def test_replan_and_interleaved_handles():
    m1 = synthetic.make_model("gradp", p_in=16, p_out=16, seed_pca=5)
    m2 = synthetic.make_model("deltas", p_in=24, p_out=24, seed_pca=6)
    g_small = synthetic.channel_grid(256, 256, seed=2).astype(np.float32)
    g_big = synthetic.channel_grid(300, 420, seed=3).astype(np.float32)
    ref = {}
    for name, m, g in (("a", m1, g_small), ("b", m2, g_big), ("c", m1, g_big)):
        ref[name] = orc.solve_grid(g.astype(np.float64), oracle_model(m)).fields
    with GridSurrogate(m1, 256, 256) as s1, GridSurrogate(m2, 300, 420) as s2:
        for _ in range(3):                           # alternate between the two handles
            a = s1.solve(g_small)[0]
            b = s2.solve(g_big)[0]
            assert np.abs(a - ref["a"]).max() <= 1e-4 * np.abs(ref["a"]).max()
            assert np.abs(b - ref["b"]).max() <= 1e-4 * np.abs(ref["b"]).max()
        # re-plan the first handle for the other grid (psm_plan_grid again): old buffers are released, results right
        s1._chk(s1.lib.psm_plan_grid(s1.h, 300, 420))
        s1.ny, s1.nx = 300, 420
        c = s1.solve(g_big)[0]
        assert np.abs(c - ref["c"]).max() <= 1e-4 * np.abs(ref["c"]).max()


def test_calls_in_the_wrong_order_are_reported():
    lib = _lib.load()
    cfg = _lib.psm_config(abi_version=_lib.PSM_ABI_VERSION, variant=1, block=128, c_in=3, c_out=1, p_in=8, p_out=8,
                          n_dense=2, sdf_channel=2, max_cases=1)
    h = C.c_void_p()
    assert lib.psm_create(C.byref(cfg), C.byref(h)) == 0
    try:
        assert lib.psm_plan_grid(h, 256, 256) == -2                       # model incomplete
        assert b"model incomplete" in lib.psm_last_error(h)
        out = np.zeros((256, 256, 1), np.float32)
        g = np.zeros((256, 256, 3), np.float32)
        assert lib.psm_solve_grid(h, g.ctypes.data_as(C.POINTER(C.c_float)), 1, None, out.ctypes.data_as(C.POINTER(C.c_float))) == -2
        t = C.c_int64()
        assert lib.psm_submit_grid(h, g.ctypes.data_as(C.POINTER(C.c_float)), 1, None, C.byref(t)) == -2
        assert lib.psm_wait_grid(h, 0, out.ctypes.data_as(C.POINTER(C.c_float))) == -1
        assert lib.psm_bind_geometry(h, g.ctypes.data_as(C.c_void_p), 0) == -2          # no plan yet
        assert lib.psm_bind_geometry_cases(h, g.ctypes.data_as(C.c_void_p), 1, 0) == -2
        assert lib.psm_geometry_bound(h) == 0 and lib.psm_unbind_geometry(h) == 0
        assert lib.psm_bind_geometry(None, g.ctypes.data_as(C.c_void_p), 0) == -1
    finally:
        lib.psm_destroy(h)


@pytest.mark.gpu
def test_guard_page_allocator_places_buffers_at_the_end_of_their_mapping():
    """PSM_GUARD_PAGES=1 (diagnostic, csrc/psm_alloc.cpp): every buffer ends where its mapping ends (16-byte granularity) with
    an unmapped granule behind it, data round-trips through it, and a whole bound solve runs on guarded buffers.  Run in a
    child process: the mode is read once per process."""
    import subprocess, sys, textwrap
    code = textwrap.dedent('''
        import ctypes as C, sys, os
        import numpy as np
        sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
        from psm_amd import _lib, GridSurrogate, synthetic
        from hipmem import DeviceArray, hip
        lib = _lib.load()
        assert lib.psm_debug_guard_pages() == 1
        for n in (1, 100, 4096, (1 << 20) + 4):
            p = C.c_void_p()
            assert lib.psm_debug_malloc(C.byref(p), n) == 0
            assert (p.value + (n + 15) // 16 * 16) % 4096 == 0, (n, hex(p.value))
            src = (np.arange(n) % 251).astype(np.uint8); dst = np.zeros(n, np.uint8)
            assert lib.psm_debug_copy_to_device(p.value, src.ctypes.data, n) == 0 and lib.psm_debug_copy_to_host(dst.ctypes.data, p.value, n) == 0
            assert np.array_equal(src, dst)
            assert lib.psm_debug_free(p.value) == 0
        model = synthetic.make_model("gradp", p_in=16, p_out=16)
        g = synthetic.channel_grid(256, 256, seed=2).astype(np.float32)
        with GridSurrogate(model, 256, 256) as sur:
            a = sur.solve(g)
            assert sur.bind_geometry(g)
            b = sur.solve(g)
            d_in, d_out = DeviceArray(g), DeviceArray(shape=(256, 256, 2))
            sur.solve_device(d_in.ptr, 1, d_out.ptr)
            c = d_out.numpy()
        assert np.abs(a - b).max() <= 5e-5 * np.abs(a).max() and np.array_equal(c[None], b)
        print("GUARDED OK")
    ''')
    env = dict(os.environ, PSM_GUARD_PAGES="1")
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600, env=env,
                         cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    assert out.returncode == 0 and "GUARDED OK" in out.stdout, (out.stdout[-500:], out.stderr[-1500:])
