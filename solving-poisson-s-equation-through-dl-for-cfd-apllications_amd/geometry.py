"""One-time geometry set-up of the solver boundary (``init_func``, python_module.py:172-247;
``Evaluation.computeOnlyOnce``, SM_call.py:89-180): uniform grid, Delaunay barycentric weights
in both directions, signed-distance image and the grid-point -> image-cell index map.

This is the host (Python) side of the boundary, like in the reference; it uses the same
third-party routine the reference uses for the triangulation (SciPy's qhull ``Delaunay``).
shapely / matplotlib.path (convex hull + point-in-polygon, python_module.py:83-90) are replaced
by ``scipy.spatial.ConvexHull`` half-plane tests.  The tables go to the GPU library through
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


def create_uniform_grid(x_min, x_max, y_min, y_max, delta):
    """python_module.py:42-48 / utils.py:111-125."""
    X0 = np.linspace(x_min + delta / 2, x_max - delta / 2, num=int(round((x_max - x_min) / delta)))
    Y0 = np.linspace(y_min + delta / 2, y_max - delta / 2, num=int(round((y_max - y_min) / delta)))
    XX0, YY0 = np.meshgrid(X0, Y0)
    return XX0.flatten(), YY0.flatten()


def interp_weights(xyz, uvw):
    """python_module.py:52-62: simplex vertices and barycentric weights of ``uvw`` in the
    Delaunay triangulation of ``xyz`` (points outside the hull: simplex -1, i.e. the last
    simplex, with at least one negative weight)."""
    from scipy.spatial import Delaunay
    tri = Delaunay(xyz)
    simplex = tri.find_simplex(uvw)
    vertices = np.take(tri.simplices, simplex, axis=0)
    temp = np.take(tri.transform, simplex, axis=0)
    d = 2
    delta = uvw - temp[:, d]
    bary = np.einsum("njk,nk->nj", temp[:, :d, :], delta)
    return vertices.astype(np.int32), np.hstack((bary, 1 - bary.sum(axis=1, keepdims=True)))


def domain_dist(top, obst, xy0, every: int = 10):
    """python_module.py:72-99: inside the bounding box of the ``top`` patch and outside the convex
    hull of the obstacle; SDF = distance to the nearest of every ``every``-th boundary point."""
    from scipy.spatial import ConvexHull
    from scipy.spatial.distance import cdist
    max_x, max_y, min_x, min_y = np.max(top[:, 0]), np.max(top[:, 1]), np.min(top[:, 0]), np.min(top[:, 1])
    inside_box = (xy0[:, 0] <= max_x) & (xy0[:, 0] >= min_x) & (xy0[:, 1] <= max_y) & (xy0[:, 1] >= min_y)
    hull = ConvexHull(obst)
    # hull.equations: [normal_x, normal_y, offset], normal.x + offset <= 0 inside
    inside_obst = np.all(xy0 @ hull.equations[:, :2].T + hull.equations[:, 2] < 0.0, axis=1)
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
                             box: str = "mixed") -> GeometryTables:
    """``Evaluation.computeOnlyOnce`` (pressureSM_deltas/SM_call.py:89-180): like ``build_geometry`` with the
    evaluator's differences -- bounds rounded to 3 digits (:103-107), the box test mixes the ``top`` patch with
    the data bounds (:119-122), every 5th boundary point for the SDF (:139-140), ``p`` decides which grid
    points are interpolable (:165-169), and there is no grid -> mesh direction."""
    from scipy.spatial import ConvexHull
    from scipy.spatial.distance import cdist
    points = np.asarray(points, np.float64)
    top, obst = np.asarray(top, np.float64), np.asarray(obst, np.float64)
    x_min, x_max = round(float(np.min(points[:, 0])), round_digits), round(float(np.max(points[:, 0])), round_digits)
    y_min, y_max = round(float(np.min(points[:, 1])), round_digits), round(float(np.max(points[:, 1])), round_digits)
    X0, Y0 = create_uniform_grid(x_min, x_max, y_min, y_max, delta)
    xy0 = np.stack([X0, Y0], axis=-1)
    vtx, wts = interp_weights(points, xy0)
    if box == "mixed":
        max_x, max_y = max(top[:, 0].max(), x_max), min(top[:, 1].max(), y_max)      # SM_call.py:119
        min_x, min_y = max(top[:, 0].min(), x_min), min(top[:, 1].min(), y_min)      # SM_call.py:120
    else:
        max_x, max_y, min_x, min_y = top[:, 0].max(), top[:, 1].max(), top[:, 0].min(), top[:, 1].min()
    inside_box = (xy0[:, 0] <= max_x) & (xy0[:, 0] >= min_x) & (xy0[:, 1] <= max_y) & (xy0[:, 1] >= min_y)
    hull = ConvexHull(obst)
    inside_obst = np.all(xy0 @ hull.equations[:, :2].T + hull.equations[:, 2] < 0.0, axis=1)
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
