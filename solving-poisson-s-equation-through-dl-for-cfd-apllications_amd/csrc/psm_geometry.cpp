// psm_geometry.cpp -- the one-time geometry set-up of the solver boundary in C++ (no interpreter, no SciPy):
// what init_func builds at Thesis_Work/Chapter5/parallelized/test_case/python_module.py:195-243 from the solver's
// arrays (PythonComm_init.H:53-94): uniform grid, Delaunay interpolation tables in both directions, convex-hull /
// point-in-polygon domain mask, signed-distance image, grid-point -> image-cell index map.
//
// Third-party routines of the reference and what stands here instead:
//   scipy.spatial.qhull.Delaunay (mesh -> grid, PM:210)  -> incremental Bowyer-Watson triangulation of the cell centres.
//       The Delaunay triangulation of points in general position is unique, so simplices and barycentric weights equal
//       qhull's up to the vertex order inside a simplex; cocircular quadruples (structured cell patches) get one of the
//       valid diagonals.  Targets outside the hull take the LAST simplex of the list with its (partly negative) weights,
//       like `np.take(tri.simplices, -1)` -- which simplex is last is an accident of the triangulator in both codes.
//   qhull.Delaunay of the lattice (grid -> mesh, PM:211) -> closed form: every lattice square is cut by its
//       (lower-left, upper-right) diagonal.  All lattice squares are cocircular, qhull's choice is not reproducible;
//       callers that need qhull's very tables hand them to psm_set_geometry instead (the Python host does).
//   shapely convex_hull + matplotlib Path.contains_points (PM:83-90) -> monotone-chain hull handed over as GEOS's
//       clockwise closed ring + the crossings test of matplotlib's _path.h (pinned by tests/golden/domain_dist_case.npz).
//   scipy cdist(...).min (PM:97) -> brute-force minimum distance.
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <limits>
#include <string>
#include <vector>

#include "../../include/psm.h"

namespace {

// np.round(x, d): rint(x * 10^d) / 10^d  (what round(np.float64, d) evaluates, PM:197-201)
double np_round(double x, int digits) {
  const double s = std::pow(10.0, digits);
  return std::nearbyint(x * s) / s;
}

// np.linspace(start, stop, num): arange(num) * step + start with the last element forced to `stop`
void np_linspace(double start, double stop, int num, std::vector<double>& out) {
  out.resize(num);
  if (num == 1) { out[0] = start; return; }
  const double step = (stop - start) / (double)(num - 1);
  for (int i = 0; i < num; ++i) out[i] = (double)i * step + start;
  out[num - 1] = stop;
}

// ---- Delaunay triangulation (Bowyer-Watson, walking point location) ----------------------------------------------
struct Tri { int v[3]; int n[3]; };      // counter-clockwise vertices; n[i] = neighbour across the edge opposite v[i]

class Delaunay {
 public:
  // (x, y): the points; triangulates a copy displaced by a deterministic 1e-9 * diagonal jitter so that cocircular /
  // collinear input has a unique answer under plain double predicates (the jitter exceeds their rounding error by
  // > 6 orders of magnitude and is 1e-6 of a typical cell spacing).  The outside of the hull is covered by GHOST
  // triangles (hull edge + one vertex at infinity, index n), so the result is the Delaunay triangulation of the whole
  // convex hull, shallow pockets of a nearly straight boundary included.
  Delaunay(const double* xy, int64_t n, int stride) : n_(n), G_((int)n) {
    double xmin = 1e300, xmax = -1e300, ymin = 1e300, ymax = -1e300;
    for (int64_t i = 0; i < n; ++i) {
      const double x = xy[i * stride], y = xy[i * stride + 1];
      xmin = std::min(xmin, x); xmax = std::max(xmax, x); ymin = std::min(ymin, y); ymax = std::max(ymax, y);
    }
    const double ext = std::max(std::max(xmax - xmin, ymax - ymin), 1e-300), jit = 1e-9 * ext;
    px_.resize(n + 1); py_.resize(n + 1);
    for (int64_t i = 0; i < n; ++i) {
      uint64_t h = (uint64_t)i * 0x9E3779B97F4A7C15ull + 0x632BE59BD9B4E019ull;
      h ^= h >> 29; h *= 0xBF58476D1CE4E5B9ull; h ^= h >> 32;
      px_[i] = xy[i * stride] + jit * ((double)(h & 0xFFFFF) / 524288.0 - 1.0);
      py_[i] = xy[i * stride + 1] + jit * ((double)((h >> 20) & 0xFFFFF) / 524288.0 - 1.0);
    }
    px_[n] = py_[n] = 0.0;              // the ghost vertex has no coordinates (never read)
    // insertion order: Morton order of the quantised coordinates (keeps the walk short)
    std::vector<std::pair<uint32_t, int>> order(n);
    for (int64_t i = 0; i < n; ++i) {
      const uint32_t qx = (uint32_t)((px_[i] - xmin) / ext * 65535.0), qy = (uint32_t)((py_[i] - ymin) / ext * 65535.0);
      order[i] = {interleave(qx) | (interleave(qy) << 1), (int)i};
    }
    std::sort(order.begin(), order.end());
    by_start_.assign(n + 1, -1);
    // first triangle: the first two points and the first later point that is clearly off their line
    const int a = order[0].second, b = order[1].second;
    size_t k3 = 2;
    const double len2 = (px_[b] - px_[a]) * (px_[b] - px_[a]) + (py_[b] - py_[a]) * (py_[b] - py_[a]);
    while (k3 < order.size() && std::fabs(orient(px_[a], py_[a], px_[b], py_[b], px_[order[k3].second], py_[order[k3].second])) < 1e-3 * len2) ++k3;
    if (k3 == order.size()) return;     // all collinear: no simplex
    int c = order[k3].second, aa = a, bb = b;
    if (orient(px_[aa], py_[aa], px_[bb], py_[bb], px_[c], py_[c]) < 0.0) std::swap(aa, bb);
    // real triangle 0 = (aa, bb, c); ghosts 1..3 across its edges: opposite aa -> edge bb-c, opposite bb -> edge c-aa, opposite c -> edge aa-bb
    tri_ = {Tri{{aa, bb, c}, {1, 2, 3}}, Tri{{c, bb, G_}, {3, 2, 0}}, Tri{{aa, c, G_}, {1, 3, 0}}, Tri{{bb, aa, G_}, {2, 1, 0}}};
    alive_.assign(4, 1); mark_.assign(4, 0);
    last_ = 0;
    for (size_t k = 0; k < order.size(); ++k) {
      const int p = order[k].second;
      if (p == a || p == b || p == c) continue;
      insert(p);
    }
    for (size_t t = 0; t < tri_.size(); ++t)
      if (alive_[t] && !ghost(tri_[t])) real_.push_back((int)t);
  }
  int64_t n_simplices() const { return (int64_t)real_.size(); }
  const Tri& simplex(int64_t k) const { return tri_[real_[k]]; }
  // triangle (internal index) that contains (x, y), walking from `hint`; is_real = inside the hull (else: the ghost
  // triangle across whose hull edge the point lies)
  int locate(double x, double y, int hint, bool* is_real) const {
    int t = (hint >= 0 && hint < (int)tri_.size() && alive_[hint]) ? hint : last_;
    for (size_t steps = 0; steps < 4 * tri_.size() + 16; ++steps) {
      const Tri& T = tri_[t];
      if (ghost(T)) {
        int u, w;
        ghost_edge(T, &u, &w);
        if (orient(px_[u], py_[u], px_[w], py_[w], x, y) > 0.0) { *is_real = false; return t; }   // beyond this hull edge
        t = T.n[ghost_pos(T)];                                                                       // back inside
        continue;
      }
      int go = -1;
      for (int i = 0; i < 3; ++i) {
        const int a = T.v[(i + 1) % 3], b = T.v[(i + 2) % 3];
        if (orient(px_[a], py_[a], px_[b], py_[b], x, y) < 0.0) { go = T.n[i]; break; }
      }
      if (go < 0) { *is_real = true; return t; }
      if (ghost(tri_[go])) { *is_real = false; return go; }
      t = go;
    }
    *is_real = !ghost(tri_[t]);
    return t;
  }
  const Tri& triangle(int t) const { return tri_[t]; }

 private:
  static uint32_t interleave(uint32_t v) {
    v &= 0xFFFF; v = (v | (v << 8)) & 0x00FF00FF; v = (v | (v << 4)) & 0x0F0F0F0F; v = (v | (v << 2)) & 0x33333333; v = (v | (v << 1)) & 0x55555555;
    return v;
  }
  static double orient(double ax, double ay, double bx, double by, double cx, double cy) {
    return (bx - ax) * (cy - ay) - (by - ay) * (cx - ax);
  }
  // > 0: d inside the circumcircle of the counter-clockwise triangle a b c
  static double incircle(double ax, double ay, double bx, double by, double cx, double cy, double dx, double dy) {
    const double adx = ax - dx, ady = ay - dy, bdx = bx - dx, bdy = by - dy, cdx = cx - dx, cdy = cy - dy;
    const double ad = adx * adx + ady * ady, bd = bdx * bdx + bdy * bdy, cd = cdx * cdx + cdy * cdy;
    return adx * (bdy * cd - bd * cdy) - ady * (bdx * cd - bd * cdx) + ad * (bdx * cdy - bdy * cdx);
  }
  bool ghost(const Tri& T) const { return T.v[0] == G_ || T.v[1] == G_ || T.v[2] == G_; }
  int ghost_pos(const Tri& T) const { return T.v[0] == G_ ? 0 : (T.v[1] == G_ ? 1 : 2); }
  void ghost_edge(const Tri& T, int* u, int* w) const { const int k = ghost_pos(T); *u = T.v[(k + 1) % 3]; *w = T.v[(k + 2) % 3]; }
  // "circumcircle" of a ghost triangle = the open half-plane beyond its hull edge (+ the edge's own open segment)
  bool in_circle(int t, int p) const {
    const Tri& T = tri_[t];
    if (ghost(T)) {
      int u, w;
      ghost_edge(T, &u, &w);
      const double o = orient(px_[u], py_[u], px_[w], py_[w], px_[p], py_[p]);
      if (o != 0.0) return o > 0.0;
      return (px_[p] - px_[u]) * (px_[p] - px_[w]) + (py_[p] - py_[u]) * (py_[p] - py_[w]) < 0.0;
    }
    return incircle(px_[T.v[0]], py_[T.v[0]], px_[T.v[1]], py_[T.v[1]], px_[T.v[2]], py_[T.v[2]], px_[p], py_[p]) > 0.0;
  }
  int new_tri(const Tri& T) {
    int id;
    if (!free_.empty()) { id = free_.back(); free_.pop_back(); tri_[id] = T; alive_[id] = 1; }
    else { id = (int)tri_.size(); tri_.push_back(T); alive_.push_back(1); mark_.push_back(0); }
    return id;
  }
  void insert(int p) {
    bool real;
    const int t0 = locate(px_[p], py_[p], last_, &real);
    // cavity: connected set of triangles whose circumcircle holds p
    ++stamp_;
    cavity_.clear(); stack_.clear();
    stack_.push_back(t0); mark_[t0] = stamp_;
    while (!stack_.empty()) {
      const int t = stack_.back(); stack_.pop_back();
      cavity_.push_back(t);
      for (int i = 0; i < 3; ++i) {
        const int nb = tri_[t].n[i];
        if (nb >= 0 && mark_[nb] != stamp_ && in_circle(nb, p)) { mark_[nb] = stamp_; stack_.push_back(nb); }
      }
    }
    // boundary edges (a -> b counter-clockwise around the cavity) with the triangle outside
    edges_.clear();
    for (int t : cavity_)
      for (int i = 0; i < 3; ++i) {
        const int nb = tri_[t].n[i];
        if (nb < 0 || mark_[nb] != stamp_) edges_.push_back({tri_[t].v[(i + 1) % 3], tri_[t].v[(i + 2) % 3], nb, t});
      }
    created_.clear();                    // (the cavity's slots are recycled only after the new fan is linked: an outer
                                         //  triangle may touch two cavity triangles, whose ids must stay distinct from the new ones)
    for (auto& e : edges_) {
      Tri T{{e.a, e.b, p}, {-1, -1, e.out}};
      const int id = new_tri(T);
      if (e.out >= 0)
        for (int i = 0; i < 3; ++i)
          if (tri_[e.out].n[i] == e.in) tri_[e.out].n[i] = id;
      by_start_[e.a] = id;
      created_.push_back(id);
    }
    for (int id : created_) {            // fan links: edge (b, p) is shared with the new triangle that starts at b
      Tri& T = tri_[id];
      const int nxt = by_start_[T.v[1]];
      T.n[0] = nxt;                      // opposite a: edge b -> p
      tri_[nxt].n[1] = id;               // opposite its b: edge p -> its a (= our b)
    }
    for (int t : cavity_) { alive_[t] = 0; free_.push_back(t); }
    for (int id : created_) if (!ghost(tri_[id])) { last_ = id; break; }
  }

  struct Edge { int a, b, out, in; };
  int64_t n_;
  int G_;
  std::vector<double> px_, py_;
  std::vector<Tri> tri_;
  std::vector<char> alive_;
  std::vector<int> mark_, free_, cavity_, stack_, created_, by_start_, real_;
  std::vector<Edge> edges_;
  int stamp_ = 0, last_ = 0;
};

// qhull's barycentric transform (PM:56-62): b = T (p - r), T = inverse of [v0 - v2, v1 - v2], r = v2; weights (b0, b1, 1 - b0 - b1)
void bary_weights(const double* v0, const double* v1, const double* v2, double x, double y, double* w) {
  const double a = v0[0] - v2[0], b = v1[0] - v2[0], c = v0[1] - v2[1], d = v1[1] - v2[1];
  const double det = a * d - b * c;
  const double i00 = d / det, i01 = -b / det, i10 = -c / det, i11 = a / det;
  const double dx = x - v2[0], dy = y - v2[1];
  w[0] = i00 * dx + i01 * dy;
  w[1] = i10 * dx + i11 * dy;
  w[2] = 1.0 - (w[0] + w[1]);
}

// convex hull (monotone chain) -> GEOS order: clockwise, closed (first vertex repeated)
void convex_hull_ring(const double* pts, int64_t n, std::vector<double>& ring) {
  std::vector<std::pair<double, double>> p(n);
  for (int64_t i = 0; i < n; ++i) p[i] = {pts[2 * i], pts[2 * i + 1]};
  std::sort(p.begin(), p.end());
  p.erase(std::unique(p.begin(), p.end()), p.end());
  std::vector<std::pair<double, double>> h(2 * p.size() + 2);
  size_t k = 0;
  auto cross = [](const std::pair<double, double>& o, const std::pair<double, double>& a, const std::pair<double, double>& b) {
    return (a.first - o.first) * (b.second - o.second) - (a.second - o.second) * (b.first - o.first);
  };
  for (size_t i = 0; i < p.size(); ++i) { while (k >= 2 && cross(h[k - 2], h[k - 1], p[i]) <= 0) --k; h[k++] = p[i]; }
  for (size_t i = p.size() - 1, t = k + 1; i > 0; --i) { while (k >= t && cross(h[k - 2], h[k - 1], p[i - 1]) <= 0) --k; h[k++] = p[i - 1]; }
  if (k > 1) --k;                                   // counter-clockwise, open
  ring.clear();
  for (size_t i = 0; i < k; ++i) { ring.push_back(h[k - 1 - i].first); ring.push_back(h[k - 1 - i].second); }   // clockwise
  if (k) { ring.push_back(ring[0]); ring.push_back(ring[1]); }
}

// matplotlib Path(ring).contains_points(p), radius 0 (src/_path.h point_in_path_impl)
bool point_in_ring(const std::vector<double>& ring, double tx, double ty) {
  if (!(std::isfinite(tx) && std::isfinite(ty))) return false;
  const size_t n = ring.size() / 2;
  bool inside = false;
  for (size_t k = 0; k < n; ++k) {
    const double x0 = ring[2 * k], y0 = ring[2 * k + 1], x1 = ring[2 * ((k + 1) % n)], y1 = ring[2 * ((k + 1) % n) + 1];
    const bool f0 = y0 >= ty, f1 = y1 >= ty;
    if (f0 != f1 && (((y1 - ty) * (x0 - x1) >= (x1 - tx) * (y0 - y1)) == f1)) inside = !inside;
  }
  return inside;
}

double min_dist(const double* pts, int64_t n, int every, double x, double y) {
  double best = std::numeric_limits<double>::infinity();
  for (int64_t i = 0; i < n; i += every) {
    const double dx = x - pts[2 * i], dy = y - pts[2 * i + 1];
    best = std::min(best, dx * dx + dy * dy);
  }
  return std::sqrt(best);            // cdist: sqrt of the squared differences' sum; min and sqrt commute (monotonic, correctly rounded)
}

thread_local std::string g_geo_error;
int geo_fail(int code, const char* msg) { g_geo_error = msg; return code; }

}  // namespace

extern "C" {

const char* psm_geometry_last_error(void) { return g_geo_error.c_str(); }

int psm_geometry_shape(const double* cells, int64_t n, double delta, int32_t* ny, int32_t* nx, double* bounds) {
  if (!cells || n < 3 || !(delta > 0.0) || !ny || !nx) return geo_fail(PSM_ERR_ARG, "bad arguments");
  double xmin = 1e300, xmax = -1e300, ymin = 1e300, ymax = -1e300;
  for (int64_t i = 0; i < n; ++i) {
    xmin = std::min(xmin, cells[i * 5 + 2]); xmax = std::max(xmax, cells[i * 5 + 2]);
    ymin = std::min(ymin, cells[i * 5 + 3]); ymax = std::max(ymax, cells[i * 5 + 3]);
  }
  const double x_min = np_round(xmin, 2), x_max = np_round(xmax, 2), y_min = np_round(ymin, 2), y_max = np_round(ymax, 2);   // PM:197-201
  const double fx = std::nearbyint((x_max - x_min) / delta), fy = std::nearbyint((y_max - y_min) / delta);                   // PM:44-45, 216-217
  if (!(fx >= 2 && fy >= 2 && fx * fy < 2.0e8)) return geo_fail(PSM_ERR_ARG, "cell centres do not span a usable grid at this delta");
  *nx = (int32_t)fx; *ny = (int32_t)fy;
  if (bounds) { bounds[0] = x_min; bounds[1] = x_max; bounds[2] = y_min; bounds[3] = y_max; }
  return PSM_OK;
}

int psm_geometry_build(const double* cells, int64_t n, const double* top, int64_t n_top, const double* obst, int64_t n_obst,
                       double delta, int32_t every, int32_t* vtx_m2g, double* wts_m2g, int32_t* indices, double* sdfunct,
                       int32_t* vtx_g2m, double* wts_g2m) {
  if (!top || !obst || n_top < 1 || n_obst < 3 || every < 1 || !vtx_m2g || !wts_m2g || !indices || !sdfunct || !vtx_g2m || !wts_g2m)
    return geo_fail(PSM_ERR_ARG, "bad arguments");
  int32_t ny, nx;
  double bd[4];
  int rc = psm_geometry_shape(cells, n, delta, &ny, &nx, bd);
  if (rc) return rc;
  // create_uniform_grid (PM:42-48): cell-centred lattice, meshgrid flattened row-major (y outer)
  std::vector<double> X, Y;
  np_linspace(bd[0] + delta / 2, bd[1] - delta / 2, nx, X);
  np_linspace(bd[2] + delta / 2, bd[3] - delta / 2, ny, Y);
  const int64_t ng = (int64_t)ny * nx;

  // ---- mesh -> grid (PM:210): Delaunay of the cell centres, simplex + weights of every lattice point
  Delaunay tri(cells + 2, n, 5);
  if (tri.n_simplices() < 1) return geo_fail(PSM_ERR_ARG, "the cell centres are degenerate (no triangle)");
  // A simplex that is a sliver in the TRUE coordinates (collinear wall cells that only the jitter separates) has no
  // barycentric transform (qhull reports NaN there): such simplices count as "outside", and the stand-in for
  // `simplices[-1]` is the last simplex with a usable transform.
  auto usable = [&](const Tri& S) {
    const double* a = cells + (int64_t)S.v[0] * 5 + 2; const double* b = cells + (int64_t)S.v[1] * 5 + 2; const double* c = cells + (int64_t)S.v[2] * 5 + 2;
    const double area2 = std::fabs((b[0] - a[0]) * (c[1] - a[1]) - (b[1] - a[1]) * (c[0] - a[0]));
    const double e2 = std::max(std::max((b[0] - a[0]) * (b[0] - a[0]) + (b[1] - a[1]) * (b[1] - a[1]), (c[0] - a[0]) * (c[0] - a[0]) + (c[1] - a[1]) * (c[1] - a[1])),
                               (c[0] - b[0]) * (c[0] - b[0]) + (c[1] - b[1]) * (c[1] - b[1]));
    return area2 > 1e-7 * e2;
  };
  int64_t k_last = tri.n_simplices() - 1;
  while (k_last >= 0 && !usable(tri.simplex(k_last))) --k_last;
  if (k_last < 0) return geo_fail(PSM_ERR_ARG, "the cell centres are degenerate (no triangle with an area)");
  const Tri lastS = tri.simplex(k_last);
  std::vector<uint8_t> outside(ng, 0);
  int hint = -1;
  for (int32_t i = 0; i < ny; ++i) {
    // serpentine scan keeps consecutive targets adjacent for the walk; results are stored at the row-major index
    for (int32_t jj = 0; jj < nx; ++jj) {
      const int32_t j = (i & 1) ? nx - 1 - jj : jj;
      const int64_t t = (int64_t)i * nx + j;
      bool real;
      hint = tri.locate(X[j], Y[i], hint, &real);
      if (real && !usable(tri.triangle(hint))) real = false;
      const Tri& S = real ? tri.triangle(hint) : lastS;     // simplex -1 -> np.take(..., -1): the last simplex
      outside[t] = real ? 0 : 1;
      for (int q = 0; q < 3; ++q) vtx_m2g[t * 3 + q] = S.v[q];
      bary_weights(cells + (int64_t)S.v[0] * 5 + 2, cells + (int64_t)S.v[1] * 5 + 2, cells + (int64_t)S.v[2] * 5 + 2, X[j], Y[i], wts_m2g + t * 3);
      // A target ON an edge of its simplex has a weight of 0 +- rounding; `wts < 0` (interpolate_fill, PM:69) then drops
      // the point or not by the sign of that noise (in the reference too).  Here such a point counts as inside.
      if (real) for (int q = 0; q < 3; ++q) if (wts_m2g[t * 3 + q] < 0.0 && wts_m2g[t * 3 + q] > -1e-12) wts_m2g[t * 3 + q] = 0.0;
    }
  }

  // ---- grid -> mesh (PM:211): the lattice cut along the (lower-left, upper-right) diagonals, closed form
  const double gx0 = X[0], gy0 = Y[0];
  const double sx = nx > 1 ? (X[nx - 1] - X[0]) / (nx - 1) : delta, sy = ny > 1 ? (Y[ny - 1] - Y[0]) / (ny - 1) : delta;
  for (int64_t c = 0; c < n; ++c) {
    const double x = cells[c * 5 + 2], y = cells[c * 5 + 3];
    double fx = (x - gx0) / sx, fy = (y - gy0) / sy;
    const bool out = fx < 0.0 || fy < 0.0 || fx > (double)(nx - 1) || fy > (double)(ny - 1);
    int j = (int)std::floor(fx), i = (int)std::floor(fy);
    j = std::min(std::max(j, 0), nx - 2); i = std::min(std::max(i, 0), ny - 2);
    const double u = fx - j, v = fy - i;
    int a, b, d;                                   // triangle (a, b, d) of lattice indices
    const int p00 = i * nx + j, p10 = p00 + 1, p01 = p00 + nx, p11 = p01 + 1;
    if (out) { a = (ny - 2) * nx + nx - 2; b = a + 1; d = a + nx + 1; }    // "last simplex": weights below come out partly negative
    else if (v <= u) { a = p00; b = p10; d = p11; }                         // lower-right triangle
    else { a = p00; b = p11; d = p01; }                                     // upper-left triangle
    const int idx[3] = {a, b, d};
    double P[3][2];
    for (int q = 0; q < 3; ++q) { P[q][0] = X[idx[q] % nx]; P[q][1] = Y[idx[q] / nx]; }
    for (int q = 0; q < 3; ++q) vtx_g2m[c * 3 + q] = idx[q];
    bary_weights(P[0], P[1], P[2], x, y, wts_g2m + c * 3);
    if (out) {                                     // make sure the NaN fallback of interpolate_fill triggers (PM:495-496)
      double* w = wts_g2m + c * 3;
      if (w[0] >= 0 && w[1] >= 0 && w[2] >= 0) { w[0] = -1.0; w[1] = 1.0; w[2] = 1.0; }
    }
  }

  // ---- domain_dist (PM:72-99)
  double tmaxx = -1e300, tmaxy = -1e300, tminx = 1e300, tminy = 1e300;
  for (int64_t i = 0; i < n_top; ++i) {
    tmaxx = std::max(tmaxx, top[2 * i]); tminx = std::min(tminx, top[2 * i]);
    tmaxy = std::max(tmaxy, top[2 * i + 1]); tminy = std::min(tminy, top[2 * i + 1]);
  }
  std::vector<double> ring;
  convex_hull_ring(obst, n_obst, ring);
  if (ring.size() < 8) return geo_fail(PSM_ERR_ARG, "the obstacle points have no convex hull with an interior");
  double rxmin = 1e300, rxmax = -1e300, rymin = 1e300, rymax = -1e300;
  for (size_t k = 0; k < ring.size() / 2; ++k) {
    rxmin = std::min(rxmin, ring[2 * k]); rxmax = std::max(rxmax, ring[2 * k]);
    rymin = std::min(rymin, ring[2 * k + 1]); rymax = std::max(rymax, ring[2 * k + 1]);
  }
  // ---- index map + SDF image (PM:225-243; indices zero-initialised like SM_call.py:161)
  std::memset(indices, 0, (size_t)ng * 2 * sizeof(int32_t));
  std::memset(sdfunct, 0, (size_t)ng * sizeof(double));
  const double x0 = X[0], y0 = Y[0];               // np.min(X0), np.min(Y0)
  for (int32_t i = 0; i < ny; ++i)
    for (int32_t j = 0; j < nx; ++j) {
      const int64_t t = (int64_t)i * nx + j;
      const double x = X[j], y = Y[i];
      const bool in_box = x <= tmaxx && x >= tminx && y <= tmaxy && y >= tminy;
      if (!in_box) continue;
      const bool maybe_in = x >= rxmin && x <= rxmax && y >= rymin && y <= rymax;
      if (maybe_in && point_in_ring(ring, x, y)) continue;
      // interpolate_fill(Ux): NaN where a weight is negative (PM:231) -- a NaN in Ux itself also disqualifies the point
      const double* w = wts_m2g + t * 3;
      if (w[0] < 0.0 || w[1] < 0.0 || w[2] < 0.0) continue;
      double ux = 0.0;
      for (int q = 0; q < 3; ++q) ux += cells[(int64_t)vtx_m2g[t * 3 + q] * 5] * w[q];
      if (ux != ux) continue;
      const int32_t jj = (int32_t)std::nearbyint((x - x0) / delta), ii = (int32_t)std::nearbyint((y - y0) / delta);   // PM:235-236
      if (ii < 0 || ii >= ny || jj < 0 || jj >= nx) continue;
      indices[t * 2] = ii; indices[t * 2 + 1] = jj;
      const double sdf = std::min(min_dist(obst, n_obst, every, x, y), min_dist(top, n_top, every, x, y));
      sdfunct[(int64_t)ii * nx + jj] = sdf;
    }
  (void)outside;
  return PSM_OK;
}

}  // extern "C"
