"""psm_geometry_build / psm_init_geometry (csrc/psm_geometry.cpp): init_func's one-time tables built in C++ (own Delaunay
triangulation, hull, crossings test, SDF) against the SciPy-built tables of the Python host (the routines the reference
calls) -- and, on the GPU, a C++-only host going psm_create -> psm_init_geometry -> psm_solve."""
import os
import shutil
import struct
import subprocess

import numpy as np
import pytest

import cases
from oracle import psm_oracle as orc
from psm_amd import _lib, geometry, synthetic

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.dirname(_lib.LIB_PATH)


def _lattice(array, delta=5e-3):
    b = [round(array[:, 2].min(), 2), round(array[:, 2].max(), 2), round(array[:, 3].min(), 2), round(array[:, 3].max(), 2)]
    X0, Y0 = geometry.create_uniform_grid(b[0], b[1], b[2], b[3], delta)
    return np.c_[X0, Y0]


def test_tables_equal_the_scipy_built_ones_on_an_unstructured_mesh():
    from scipy.spatial import Delaunay
    array, top, obst, model, maxs = cases.build_mesh_case()
    ref = geometry.build_geometry(array, top, obst)
    nat = geometry.build_geometry_native(array, top, obst)
    assert (nat.ny, nat.nx) == (ref.ny, ref.nx)
    xy0 = _lattice(array)
    inside = Delaunay(array[:, 2:4]).find_simplex(xy0) >= 0
    # mesh -> grid: the Delaunay triangulation of points in general position is unique
    same = (np.sort(ref.vtx_m2g, 1) == np.sort(nat.vtx_m2g, 1)).all(1)
    assert same[inside].all()
    order_r, order_n = np.argsort(ref.vtx_m2g, 1), np.argsort(nat.vtx_m2g, 1)
    wr = np.take_along_axis(ref.wts_m2g, order_r, 1)
    wn = np.take_along_axis(nat.wts_m2g, order_n, 1)
    assert np.abs(wr - wn)[inside].max() < 1e-12
    # outside the hull: same set of points, "last simplex" weights with a negative entry in both
    assert np.array_equal((ref.wts_m2g < 0).any(1), (nat.wts_m2g < 0).any(1))
    assert np.abs(nat.wts_m2g.sum(1) - 1).max() < 1e-9
    # domain mask / index map / SDF image: identical
    np.testing.assert_array_equal(nat.indices, ref.indices)
    np.testing.assert_array_equal(nat.sdfunct, ref.sdfunct)
    # grid -> mesh: same cells fall outside the lattice; inside, fixed diagonals interpolate linear functions exactly
    out_r, out_n = (ref.wts_g2m < 0).any(1), (nat.wts_g2m < 0).any(1)
    np.testing.assert_array_equal(out_r, out_n)
    lin = 0.3 + 1.7 * xy0[:, 0] - 0.9 * xy0[:, 1]
    got = np.einsum("nj,nj->n", lin[nat.vtx_g2m], nat.wts_g2m)
    want = 0.3 + 1.7 * array[:, 2] - 0.9 * array[:, 3]
    assert np.abs(got - want)[~out_n].max() < 1e-12
    smooth = np.sin(2 * xy0[:, 0]) * np.cos(3 * xy0[:, 1])
    a = np.einsum("nj,nj->n", smooth[ref.vtx_g2m], ref.wts_g2m)
    b = np.einsum("nj,nj->n", smooth[nat.vtx_g2m], nat.wts_g2m)
    assert np.abs(a - b)[~out_n].max() < 1e-4               # the other diagonal of a 5 mm square: second-order difference


def test_structured_cell_centres_are_triangulated_consistently():
    """blockMesh-like cell centres: every quadruple is cocircular and the walls are collinear -- the triangulation must
    still be a valid one (weights of an interior point non-negative, linear functions reproduced)."""
    xs, ys = np.linspace(0.004, 1.196, 150), np.linspace(-0.296, 0.296, 75)
    X, Y = np.meshgrid(xs, ys)
    keep = (X - 0.4) ** 2 + Y ** 2 > 0.08 ** 2
    cx, cy = X[keep], Y[keep]
    array = np.c_[1.0 + cy, 0.1 * cx, cx, cy, np.zeros_like(cx)]
    th = np.linspace(0, 2 * np.pi, 120, endpoint=False)
    obst = np.c_[0.4 + 0.08 * np.cos(th), 0.08 * np.sin(th)]
    wall = np.linspace(0, 1.2, 241)
    top = np.concatenate([np.c_[wall, np.full_like(wall, 0.3)], np.c_[wall, np.full_like(wall, -0.3)]])
    nat = geometry.build_geometry_native(array, top, obst)
    xy0 = _lattice(array)
    inside = (nat.wts_m2g >= -1e-9).all(1)
    assert inside.mean() > 0.95
    lin = 2.0 - 0.7 * array[:, 2] + 1.3 * array[:, 3]
    got = np.einsum("nj,nj->n", lin[nat.vtx_m2g], nat.wts_m2g)
    assert np.abs(got - (2.0 - 0.7 * xy0[:, 0] + 1.3 * xy0[:, 1])).max() < 1e-10      # exact even for the extrapolated rows
    assert np.abs(nat.wts_m2g.sum(1) - 1).max() < 1e-9
    # Against the SciPy host: lattice points that lie ON an edge of the structured triangulation carry a weight of
    # 0 +- 1e-15, and `wts < 0` drops them by the sign of that noise (in the reference as well); the C++ builder counts
    # them as inside.  So: nothing the SciPy host keeps is lost, what it drops here is noise, the SDF values agree.
    ref = geometry.build_geometry(array, top, obst)
    flow_r, flow_n = ref.sdfunct != 0, nat.sdfunct != 0
    assert not (flow_r & ~flow_n).any()
    wmin_ref = ref.wts_m2g.min(1).reshape(nat.ny, nat.nx)
    assert (wmin_ref[flow_n & ~flow_r] > -1e-12).all()
    np.testing.assert_array_equal(nat.sdfunct[flow_r], ref.sdfunct[flow_r])


def test_large_cloud_and_degenerate_input():
    rng = np.random.default_rng(3)
    n = 120000
    cx, cy = rng.random(n) * 3.0, rng.random(n) * 0.8 - 0.4
    array = np.c_[np.ones(n), np.zeros(n), cx, cy, np.zeros(n)]
    obst = np.c_[1.0 + 0.1 * np.cos(np.linspace(0, 6.2, 50)), 0.1 * np.sin(np.linspace(0, 6.2, 50))]
    top = np.c_[np.linspace(0, 3, 100), np.full(100, 0.4)]
    nat = geometry.build_geometry_native(array, top, obst)
    xy0 = _lattice(array)
    lin = 1.0 + cx - 2 * cy
    got = np.einsum("nj,nj->n", lin[nat.vtx_m2g], nat.wts_m2g)
    assert np.abs(got - (1.0 + xy0[:, 0] - 2 * xy0[:, 1])).max() < 1e-9
    assert (nat.wts_m2g >= -1e-9).all(1).mean() > 0.97
    line = np.c_[np.ones(50), np.zeros(50), np.linspace(0, 1, 50), np.linspace(0, 1, 50) * 0.5, np.zeros(50)]
    with pytest.raises(ValueError):
        geometry.build_geometry_native(line, top, obst)     # collinear cell centres: no simplex


SOLVER_HOST = r"""
// C++-only solver-side host: the calls a DLPoissonFoam build makes instead of embedding CPython
// (PythonComm_init.H:53-94 -> psm_init_geometry, PythonComm.H:2-36 -> psm_solve).
#include <cstdint>
#include <cstdio>
#include <vector>
#include "psm.h"
template <typename T> static bool rd(FILE* f, std::vector<T>& v, size_t n) { v.resize(n); return std::fread(v.data(), sizeof(T), n, f) == n; }
#define CHECK(call) do { const int rc_ = (call); if (rc_ != PSM_OK) { std::fprintf(stderr, "%s -> %d: %s\n", #call, rc_, psm_last_error(sm)); return 2; } } while (0)
int main(int argc, char** argv) {
  if (argc != 4) return 1;
  FILE* fm = std::fopen(argv[1], "rb");
  int32_t hd[6];
  if (!fm || std::fread(hd, 4, 6, fm) != 6) return 1;
  const int p_in = hd[0], p_out = hd[1], n_dense = hd[2], n = hd[3], n_top = hd[4], n_obst = hd[5];
  const size_t K_in = 128u * 128u * 3, K_out = 128u * 128u;
  std::vector<double> ci, mi, co, mo, sc, maxs, cells, top, obst;
  if (!rd(fm, ci, p_in * K_in) || !rd(fm, mi, K_in) || !rd(fm, co, p_out * K_out) || !rd(fm, mo, K_out) || !rd(fm, sc, 2) || !rd(fm, maxs, 4)) return 1;
  psm_handle* sm = nullptr;
  psm_config cfg = {PSM_ABI_VERSION, PSM_VARIANT_CHAPTER5, 128, 0, 3, 1, p_in, p_out, n_dense, PSM_SCALER_MAX_ABS, 2, 0, 1, 0, PSM_PRECISION_F32};
  if (psm_create(&cfg, &sm) != PSM_OK) { std::fprintf(stderr, "psm_create: %s\n", psm_last_error(nullptr)); return 2; }
  CHECK(psm_set_pca(sm, ci.data(), mi.data(), co.data(), mo.data()));
  CHECK(psm_set_scaler(sm, &sc[0], &sc[0], &sc[1], &sc[1]));
  for (int l = 0; l < n_dense; ++l) {
    int32_t sh[2]; std::vector<float> W, b;
    if (std::fread(sh, 4, 2, fm) != 2 || !rd(fm, W, (size_t)sh[0] * sh[1]) || !rd(fm, b, sh[1])) return 1;
    CHECK(psm_set_dense(sm, l, sh[0], sh[1], W.data(), b.data()));
  }
  if (!rd(fm, cells, (size_t)n * 5) || !rd(fm, top, (size_t)n_top * 2) || !rd(fm, obst, (size_t)n_obst * 2)) return 1;
  std::fclose(fm);
  CHECK(psm_set_case(sm, maxs.data(), 5e-3, 10, 0.05));
  CHECK(psm_init_geometry(sm, cells.data(), n, top.data(), n_top, obst.data(), n_obst, 0));      // init_func
  std::vector<double> p(n), p2(n);
  CHECK(psm_solve(sm, cells.data(), n, 0, p.data()));                                             // py_func
  FILE* fs = std::fopen(argv[2], "rb");                                                           // a second time step
  if (!fs || !rd(fs, cells, (size_t)n * 5)) return 1;
  std::fclose(fs);
  CHECK(psm_solve(sm, cells.data(), n, 0, p2.data()));
  FILE* fo = std::fopen(argv[3], "wb");
  if (!fo || std::fwrite(p.data(), 8, n, fo) != (size_t)n || std::fwrite(p2.data(), 8, n, fo) != (size_t)n) return 1;
  std::fclose(fo);
  psm_destroy(sm);
  return 0;
}
"""


def _build_solver_host(tmp_path):
    src = tmp_path / "solver_host.cpp"
    src.write_text(SOLVER_HOST)
    exe = str(tmp_path / "solver_host")
    cmd = ["g++", "-std=c++17", "-O2", "-Wall", "-I", os.path.join(ROOT, "include"), str(src), "-L", PKG, "-lpsm_hip",
           "-Wl,-rpath," + PKG, "-Wl,-rpath,/opt/rocm/lib", "-L/opt/rocm/lib", "-o", exe]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    return exe


@pytest.mark.skipif(shutil.which("g++") is None, reason="needs g++")
def test_cpp_solver_host_builds(tmp_path):
    _build_solver_host(tmp_path)


def _oracle_p(array, tables, model, maxs):
    geo = orc.Geometry(tables.ny, tables.nx, tables.vtx_m2g, tables.wts_m2g, tables.vtx_g2m, tables.wts_g2m, tables.indices, tables.sdfunct)
    sc = orc.Scaler(model.scaler_kind, model.in_a, model.in_b, model.out_a, model.out_b)
    om = orc.Model(model.variant, model.c_in, model.c_out, model.comp_in, model.mean_in, model.comp_out, model.mean_out,
                   model.weights, sc, model.out_scale, model.S, model.ov, model.sdf_ch)
    return orc.py_func_mesh(array, geo, om, maxs)[0]


@pytest.mark.gpu
def test_cpp_only_init_geometry_and_solve(tmp_path):
    """psm_create -> psm_init_geometry -> psm_solve from a C++ program (no Python on the product side): p equals the
    oracle's py_func on the C++-built tables within the mesh tolerance (2e-4 max|p|), for two time steps; against the
    reference run's p (golden, SciPy tables) the difference is the documented one of the out-of-hull grid points."""
    exe = _build_solver_host(tmp_path)
    array, top, obst, model, maxs = cases.build_mesh_case()
    array2 = cases.build_mesh_case(step=1)[0]
    with open(tmp_path / "case.bin", "wb") as f:
        f.write(struct.pack("<6i", model.p_in, model.p_out, len(model.weights), array.shape[0], top.shape[0], obst.shape[0]))
        for a in (model.comp_in, model.mean_in, model.comp_out, model.mean_out, np.array([model.in_a, model.out_a]), np.array(maxs)):
            f.write(np.ascontiguousarray(a, "<f8").tobytes())
        for W, b in model.weights:
            f.write(struct.pack("<2i", *W.shape))
            f.write(np.ascontiguousarray(W, "<f4").tobytes()); f.write(np.ascontiguousarray(b, "<f4").tobytes())
        for a in (array, top, obst):
            f.write(np.ascontiguousarray(a, "<f8").tobytes())
    np.ascontiguousarray(array2, "<f8").tofile(tmp_path / "step2.bin")
    run = subprocess.run([exe, str(tmp_path / "case.bin"), str(tmp_path / "step2.bin"), str(tmp_path / "p.bin")],
                         capture_output=True, text=True, timeout=300)
    assert run.returncode == 0, (run.stdout, run.stderr)
    p = np.fromfile(tmp_path / "p.bin", np.float64).reshape(2, -1)
    nat = geometry.build_geometry_native(array, top, obst)
    for k, arr in enumerate((array, array2)):
        ref = _oracle_p(arr, nat, model, maxs)
        assert np.abs(p[k] - ref).max() <= 2e-4 * np.abs(ref).max()
        np.testing.assert_array_equal(p[k][p[k] == arr[:, 4]] , arr[:, 4][p[k] == arr[:, 4]])
        assert ((p[k] == arr[:, 4]) == (ref == arr[:, 4])).all()          # same near-wall / outside-lattice fallback cells
    # the Python mirror with geometry="native" goes through the same two entries
    from psm_amd import SolverModule
    sm = SolverModule(model, maxs, geometry="native")
    assert sm.init_func(array, top, obst) == 0
    np.testing.assert_array_equal(sm.py_func(array), p[0])
    # Against the reference run (golden p, SciPy / qhull tables): the two documented differences, separately.
    # (a) mesh -> grid built in C++, grid -> mesh tables handed over by the caller (qhull's): the only difference is the
    #     value that the out-of-hull grid points leave in image cell (0,0) -- < 1.5 % of max|p| here;
    # (b) grid -> mesh on fixed lattice diagonals: qhull's diagonal per lattice square is rounding noise (50/50), and this
    #     synthetic model's field is white noise from pixel to pixel, so (b) is checked on a smooth lattice function
    #     (test_tables_equal_the_scipy_built_ones_on_an_unstructured_mesh: < 1e-4), not on p.
    import ctypes as C
    from psm_amd import GridSurrogate
    gold = cases.load_golden("mesh_chapter5")["p"]
    ref = geometry.build_geometry(array, top, obst)
    with GridSurrogate(model, nat.ny, nat.nx) as sur:
        f64, i32 = C.POINTER(C.c_double), C.POINTER(C.c_int32)
        v1, w1 = np.ascontiguousarray(nat.vtx_m2g, np.int32), np.ascontiguousarray(nat.wts_m2g)
        idx, sdf = np.ascontiguousarray(nat.indices, np.int32), np.ascontiguousarray(nat.sdfunct)
        v2, w2 = np.ascontiguousarray(ref.vtx_g2m, np.int32), np.ascontiguousarray(ref.wts_g2m)
        mx = np.ascontiguousarray(maxs, np.float64)
        sur._chk(sur.lib.psm_set_geometry(sur.h, array.shape[0], nat.ny, nat.nx, v1.ctypes.data_as(i32), w1.ctypes.data_as(f64),
                                          idx.ctypes.data_as(i32), sdf.ctypes.data_as(f64), v2.ctypes.data_as(i32), w2.ctypes.data_as(f64),
                                          mx.ctypes.data_as(f64), 0, 0, 0.05))
        pa = np.empty(array.shape[0])
        a64 = np.ascontiguousarray(array, np.float64)
        sur._chk(sur.lib.psm_solve(sur.h, a64.ctypes.data_as(f64), array.shape[0], 0, pa.ctypes.data_as(f64)))
    assert ((pa == array[:, 4]) == (gold == array[:, 4])).all()
    rel = np.abs(pa - gold).max() / np.abs(gold).max()
    print("C++ mesh->grid tables + qhull grid->mesh tables vs the reference run: max |dp| / max|p| =", rel)
    assert rel <= 1.5e-2
