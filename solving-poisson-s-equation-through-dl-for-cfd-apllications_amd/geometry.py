"""One-time geometry set-up of the solver boundary (``init_func``, python_module.py:172-247;
``Evaluation.computeOnlyOnce``, SM_call.py:89-180): uniform grid, Delaunay barycentric weights
in both directions, signed-distance image and the grid-point -> image-cell index map.

This is the host (Python) side of the boundary, like in the reference; it uses the same
third-party routine the reference uses for the triangulation (SciPy's qhull ``Delaunay``).
shapely (``MultiPoint.convex_hull``, python_module.py:83-85) is replaced by ``scipy.spatial.ConvexHull`` with the ring
handed over in GEOS's clockwise order, and ``matplotlib.path.Path.contains_points`` (python_module.py:89-90) by a
restatement of its crossings test (``points_in_ring``; pinned against matplotlib itself by
tests/golden/domain_dist_case.npz, on-edge points included).  The tables go to the GPU library through
``psm_set_geometry``; the per-step work happens there.
"""
from __future__ import annotations

from dataclasses import dataclass

import numpy as np


@dataclass
class GeometryTables:
    ny: int
    nx: int
    vtx_m2g: np.ndarray      # [ny*nx, 3] int32
    wts_m2g: np.ndarray      # [ny*nx, 3] float64
    indices: np.ndarray      # [ny*nx, 2] int32 (ii, jj)
    sdfunct: np.ndarray      # [ny, nx] float64
    vtx_g2m: np.ndarray      # [N, 3] int32
    wts_g2m: np.ndarray      # [N, 3] float64
    domain_bool: np.ndarray  # [ny*nx] bool
    x0: float = 0.0
    y0: float = 0.0
    delta: float = 5e-3
    box: tuple = None        # (min_x, max_x, min_y, max_y) of the inside-domain test (evaluators)


def _cell_centres(lo: float, hi: float, delta: float) -> np.ndarray:
    """Centres of the ``int(round((hi - lo) / delta))`` cells of width ``delta`` that tile [lo, hi] (Python's round: half to even)."""
    n_cells = int(round((hi - lo) / delta))
    return np.linspace(lo + delta / 2, hi - delta / 2, n_cells)


def create_uniform_grid(x_min, x_max, y_min, y_max, delta):
    """The lattice of python_module.py:42-48 / utils.py:111-125: cell-centred, x fastest -> (X0, Y0), each [ny * nx]."""
    xs, ys = _cell_centres(x_min, x_max, delta), _cell_centres(y_min, y_max, delta)
    return np.tile(xs, ys.size), np.repeat(ys, xs.size)


def interp_weights(xyz, uvw, idw_fallback: bool = False):
    """python_module.py:52-62: simplex vertices and barycentric weights of ``uvw`` in the
    Delaunay triangulation of ``xyz`` (points outside the hull: simplex -1, i.e. the last
    simplex, with at least one negative weight).

    ``idw_fallback`` = the Improved_SM form (pressureSM_deltas/utils.py:22-55, pressureSM_Poisson/SM_call.py:139-172):
    targets outside the hull take their 3 nearest source points with inverse-square-distance weights
    (``1 / max(d**2, 1e-6)``, normalised) instead -- all positive, so ``interpolate_fill`` keeps them."""
    from scipy.spatial import Delaunay
    tri = Delaunay(xyz)
    simplex = tri.find_simplex(uvw)                      # -1 outside the hull: indexes the LAST simplex below, like the reference
    vertices = tri.simplices[simplex].copy()
    # SciPy stores, per simplex, the inverse edge matrix (rows 0, 1) and the simplex's last vertex (row 2): the first two
    # barycentric coordinates are Tinv . (target - r), the third closes the sum to one
    T = tri.transform[simplex]
    first_two = np.einsum("njk,nk->nj", T[:, :2, :], np.asarray(uvw) - T[:, 2, :])
    wts = np.concatenate([first_two, 1.0 - first_two.sum(axis=1, keepdims=True)], axis=1)
    if idw_fallback:
        out = simplex == -1
        if out.any():
            from scipy.spatial import cKDTree
            nndist, nni = cKDTree(np.asarray(xyz)).query(np.asarray(uvw)[out], k=3)
            w = 1.0 / np.maximum(nndist ** 2, 1e-6)
            vertices[out] = nni
            wts[out] = w / w.sum(axis=-1)[:, None]
    return vertices.astype(np.int32), wts


def convex_hull_ring(points):
    """Closed ring of the convex hull of ``points`` as ``shapely.geometry.MultiPoint(points).convex_hull.exterior``
    hands it over (python_module.py:83-86): clockwise (the orientation of GEOS's ConvexHull, e.g. the Shapely manual's
    ``POLYGON ((1 0, 0 0, 0 2, 2 2, 3 1, 1 0))``), first vertex repeated at the end.  SciPy's qhull vertices are
    counter-clockwise."""
    from scipy.spatial import ConvexHull
    hv = np.asarray(points, np.float64)[ConvexHull(points).vertices][::-1]
    return np.vstack([hv, hv[:1]])


def points_in_ring(ring, pts):
    """``matplotlib.path.Path(ring).contains_points(pts)`` with the default ``radius=0`` (python_module.py:89-90),
    restated: matplotlib's crossings test (src/_path.h ``point_in_path_impl``) toggles for every edge v0 -> v1 whose
    end points lie on different sides of the +x ray (``yflag = (v.y >= ty)``) when
    ``((v1.y - ty) * (v0.x - v1.x) >= (v1.x - tx) * (v0.y - v1.y)) == yflag1``; a path without CLOSEPOLY is closed by the
    edge back to its first vertex."""
    ring = np.asarray(ring, np.float64)
    tx, ty = np.asarray(pts, np.float64)[:, 0], np.asarray(pts, np.float64)[:, 1]
    inside = np.zeros(len(tx), bool)
    n = len(ring)
    for k in range(n):
        v0, v1 = ring[k], ring[(k + 1) % n]
        f0, f1 = v0[1] >= ty, v1[1] >= ty
        hit = ((v1[1] - ty) * (v0[0] - v1[0]) >= (v1[0] - tx) * (v0[1] - v1[1])) == f1
        inside ^= (f0 != f1) & hit
    return inside & np.isfinite(tx) & np.isfinite(ty)


def domain_dist(top, obst, xy0, every: int = 10):
    """python_module.py:72-99: inside the bounding box of the ``top`` patch and outside the convex
    hull of the obstacle; SDF = distance to the nearest of every ``every``-th boundary point."""
    from scipy.spatial.distance import cdist
    max_x, max_y, min_x, min_y = np.max(top[:, 0]), np.max(top[:, 1]), np.min(top[:, 0]), np.min(top[:, 1])
    inside_box = (xy0[:, 0] <= max_x) & (xy0[:, 0] >= min_x) & (xy0[:, 1] <= max_y) & (xy0[:, 1] >= min_y)
    inside_obst = points_in_ring(convex_hull_ring(obst), xy0)
    domain_bool = inside_box & ~inside_obst
    t, o = top[::every], obst[::every]
    sdf = np.minimum(cdist(xy0, o).min(axis=1), cdist(xy0, t).min(axis=1)) * domain_bool
    return domain_bool, sdf


def build_geometry(array, top, obst, delta: float = 5e-3, every: int = 10, round_digits: int = 2) -> GeometryTables:
    """``init_func`` on one rank (python_module.py:195-243).  ``array`` is the solver's
    [N,5] buffer (Ux, Uy, Cx, Cy, p).  ``indices`` is zero-initialised like SM_call.py:161
    (python_module.py:225 uses ``np.empty``: undefined content for out-of-domain points)."""
    array = np.asarray(array, np.float64)
    x_min, x_max = round(np.min(array[:, 2]), round_digits), round(np.max(array[:, 2]), round_digits)
    y_min, y_max = round(np.min(array[:, 3]), round_digits), round(np.max(array[:, 3]), round_digits)
    X0, Y0 = create_uniform_grid(x_min, x_max, y_min, y_max, delta)
    xy0 = np.stack([X0, Y0], axis=-1)
    points = array[:, 2:4]
    vtx_m2g, wts_m2g = interp_weights(points, xy0)
    vtx_g2m, wts_g2m = interp_weights(xy0, points)
    domain_bool, sdf = domain_dist(np.asarray(top, np.float64), np.asarray(obst, np.float64), xy0, every)
    ny, nx = int(round((y_max - y_min) / delta)), int(round((x_max - x_min) / delta))
    x0, y0 = np.min(X0), np.min(Y0)
    ux_interp = np.einsum("nj,nj->n", np.take(array[:, 0], vtx_m2g), wts_m2g)
    ux_interp[np.any(wts_m2g < 0, axis=1)] = np.nan                 # interpolate_fill (python_module.py:231)
    ok = domain_bool & ~np.isnan(ux_interp)
    jj = np.rint((X0 - x0) / delta).astype(np.int64)
    ii = np.rint((Y0 - y0) / delta).astype(np.int64)
    indices = np.zeros((X0.shape[0], 2), np.int32)
    indices[ok, 0], indices[ok, 1] = ii[ok], jj[ok]
    sdfunct = np.zeros((ny, nx))
    sdfunct[ii[ok], jj[ok]] = sdf[ok]
    return GeometryTables(ny, nx, vtx_m2g, np.ascontiguousarray(wts_m2g), indices, sdfunct, vtx_g2m,
                          np.ascontiguousarray(wts_g2m), domain_bool, float(x0), float(y0), delta)


def build_geometry_evaluator(points, p_values, top, obst, delta: float, every: int = 5, round_digits: int = 3,
                             box: str = "mixed", idw_fallback: bool = False) -> GeometryTables:
    """``Evaluation.computeOnlyOnce`` (pressureSM_deltas/SM_call.py:89-180): like ``build_geometry`` with the
    evaluator's differences -- bounds rounded to 3 digits (:103-107), the box test mixes the ``top`` patch with
    the data bounds (:119-122), every 5th boundary point for the SDF (:139-140), ``p`` decides which grid
    points are interpolable (:165-169), and there is no grid -> mesh direction.  ``idw_fallback``: the deltas and
    Poisson evaluators interpolate with ``utils.interp_weights`` (nearest-neighbour IDW outside the mesh hull,
    utils.py:47-53); the U_to_gradP evaluator's own method (Eval_dual_Dense_onlycil.py:69-85) has no fallback."""
    from scipy.spatial.distance import cdist
    points = np.asarray(points, np.float64)
    top, obst = np.asarray(top, np.float64), np.asarray(obst, np.float64)
    x_min, x_max = round(float(np.min(points[:, 0])), round_digits), round(float(np.max(points[:, 0])), round_digits)
    y_min, y_max = round(float(np.min(points[:, 1])), round_digits), round(float(np.max(points[:, 1])), round_digits)
    X0, Y0 = create_uniform_grid(x_min, x_max, y_min, y_max, delta)
    xy0 = np.stack([X0, Y0], axis=-1)
    vtx, wts = interp_weights(points, xy0, idw_fallback)          # utils.interp_weights (SM_call.py:115) has the IDW fallback
    if box == "mixed":
        max_x, max_y = max(top[:, 0].max(), x_max), min(top[:, 1].max(), y_max)      # SM_call.py:119
        min_x, min_y = max(top[:, 0].min(), x_min), min(top[:, 1].min(), y_min)      # SM_call.py:120
    else:
        max_x, max_y, min_x, min_y = top[:, 0].max(), top[:, 1].max(), top[:, 0].min(), top[:, 1].min()
    inside_box = (xy0[:, 0] <= max_x) & (xy0[:, 0] >= min_x) & (xy0[:, 1] <= max_y) & (xy0[:, 1] >= min_y)
    inside_obst = points_in_ring(convex_hull_ring(obst), xy0)
    domain_bool = inside_box & ~inside_obst
    sdf = np.minimum(cdist(xy0, obst[::every]).min(axis=1), cdist(xy0, top[::every]).min(axis=1)) * domain_bool
    ny, nx = int(round((y_max - y_min) / delta)), int(round((x_max - x_min) / delta))
    x0, y0 = np.min(X0), np.min(Y0)
    p_interp = np.einsum("nj,nj->n", np.take(np.asarray(p_values, np.float64), vtx), wts)
    p_interp[np.any(wts < 0, axis=1)] = np.nan
    ok = domain_bool & ~np.isnan(p_interp)
    jj = np.rint((X0 - x0) / delta).astype(np.int64)
    ii = np.rint((Y0 - y0) / delta).astype(np.int64)
    indices = np.zeros((X0.shape[0], 2), np.int32)
    indices[ok, 0], indices[ok, 1] = ii[ok], jj[ok]
    sdfunct = np.zeros((ny, nx))
    sdfunct[ii[ok], jj[ok]] = sdf[ok]
    t = GeometryTables(ny, nx, vtx, np.ascontiguousarray(wts), indices, sdfunct, None, None, domain_bool,
                       float(x0), float(y0), delta)
    t.box = (float(min_x), float(max_x), float(min_y), float(max_y))
    return t


def build_geometry_native(array, top, obst, delta: float = 5e-3, every: int = 10) -> GeometryTables:
    """The same tables from the library's own C++ builder (``psm_geometry_build``, csrc/psm_geometry.cpp: no SciPy) --
    what ``psm_init_geometry`` installs.  Differences from ``build_geometry`` are listed at psm.h ``psm_init_geometry``."""
    import ctypes as C
    from . import _lib
    lib = _lib.load()
    a = np.ascontiguousarray(array, np.float64)
    t, o = np.ascontiguousarray(top, np.float64), np.ascontiguousarray(obst, np.float64)
    ny, nx = C.c_int32(), C.c_int32()
    bd = np.zeros(4)
    f64, i32 = C.POINTER(C.c_double), C.POINTER(C.c_int32)
    if lib.psm_geometry_shape(a.ctypes.data_as(f64), a.shape[0], delta, C.byref(ny), C.byref(nx), bd.ctypes.data_as(f64)):
        raise ValueError(lib.psm_geometry_last_error().decode())
    ng, n = ny.value * nx.value, a.shape[0]
    v1, w1 = np.empty((ng, 3), np.int32), np.empty((ng, 3))
    idx, sdf = np.empty((ng, 2), np.int32), np.empty(ng)
    v2, w2 = np.empty((n, 3), np.int32), np.empty((n, 3))
    rc = lib.psm_geometry_build(a.ctypes.data_as(f64), n, t.ctypes.data_as(f64), t.shape[0], o.ctypes.data_as(f64), o.shape[0], delta, every,
                                v1.ctypes.data_as(i32), w1.ctypes.data_as(f64), idx.ctypes.data_as(i32), sdf.ctypes.data_as(f64),
                                v2.ctypes.data_as(i32), w2.ctypes.data_as(f64))
    if rc:
        raise ValueError(lib.psm_geometry_last_error().decode())
    X0, Y0 = create_uniform_grid(bd[0], bd[1], bd[2], bd[3], delta)
    return GeometryTables(ny.value, nx.value, v1, w1, idx, sdf.reshape(ny.value, nx.value), v2, w2, None, float(X0.min()), float(Y0.min()), delta)
