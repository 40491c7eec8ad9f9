/* psm_cpu.c -- C99 + OpenMP restatement of the surrogate hot path on the CPU.  TEST INFRASTRUCTURE ONLY.
 *
 * "CPU back-end A" of BASELINE.md section 2: the same algorithm as oracle/psm_oracle.py (block extraction -> PCA encode ->
 * Dense stack -> PCA decode -> serial block-offset reassembly -> global shift) with the reference's precision split
 * (float64 PCA and reassembly, float32 network), written for speed so that the CPU baseline timed by bench.py is a
 * fair one (threaded, cache-blocked contractions instead of NumPy temporaries).  Only tests/ and the cpu_baseline leg
 * of bench.py may load it (through oracle/psm_cpu.py); the product path (libpsm_hip.so) never does and has no CPU path.
 *
 * Reference lines followed (paths relative to /root/reference):
 *   PM  = Thesis_Work/Chapter5/parallelized/test_case/python_module.py
 *   SMD = Improved_SM/deltaU_to_deltaP/source/pressureSM_deltas/SM_call.py
 *   UGP = Improved_SM/U_to_gradP/evaluation/Eval_dual_Dense_onlycil.py
 * Parity: pinned like the NumPy oracle -- tests/test_cpu_port.py runs it on every golden case of tests/golden/ (outputs
 * of the reference's own statements) and against psm_oracle.py.
 *
 * Build: gcc -O3 -mavx2 -mfma -fopenmp -shared -fPIC oracle/psm_cpu.c -o oracle/_build/libpsm_cpu.so  (oracle/Makefile)
 */
#ifndef _POSIX_C_SOURCE
#define _POSIX_C_SOURCE 200809L   /* clock_gettime under -std=c99 */
#endif
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <stdio.h>
#include <time.h>
#ifdef _OPENMP
#include <omp.h>
#else  /* a host compiler without OpenMP: one thread */
static void omp_set_num_threads(int n) { (void)n; }
static int omp_get_max_threads(void) { return 1; }
static int omp_get_thread_num(void) { return 0; }
static int omp_get_num_threads(void) { return 1; }
#endif

static double now_ms(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec * 1e3 + t.tv_nsec * 1e-6; }
typedef double v4d __attribute__((vector_size(32)));
static inline v4d ld4(const double* p) { v4d v; __builtin_memcpy(&v, p, 32); return v; }

enum { V_CHAPTER5 = 0, V_DELTAS = 1, V_GRADP = 2 };
enum { SC_MAX_ABS = 0, SC_STD = 1, SC_MIN_MAX = 2 };

typedef struct {
  int32_t variant, S, ov, c_in, c_out, p_in, p_out, n_dense, scaler, sdf_ch, strict;
  const double *comp_in, *mean_in, *comp_out, *mean_out;   /* sklearn components_[:p] [p][S*S*c], mean_ [S*S*c] */
  const double *in_a, *in_b, *out_a, *out_b;               /* scaler arrays [p_in] / [p_out] (max_abs: element 0) */
  const float* const* W;                                   /* Keras Dense kernels [n_in][n_out] */
  const float* const* b;
  const int32_t* dims;                                     /* [n_dense + 1] layer widths, dims[0] = p_in */
  double out_scale;                                        /* SMD:551 max_abs_p * U_max^2 */
} psm_cpu_model;

/* ---- a7: block layout (PM:306-329, SMD:461-479, UGP:479-500) ------------------------------------------------ */
typedef struct { int y0, x0, ti, tj; } blk_t;

static int layout(int variant, int Ny, int Nx, int S, int ov, blk_t* out, int cap, int* n_x, int* n_y) {
  const int st = S - ov;
  int n = 0;
  if (Ny < S || Nx < S || st < 1) return -1;
  if (variant == V_CHAPTER5) {
    *n_x = (Nx - S) / st; *n_y = (Ny - S) / st;                       /* int(): floor for non-negative values */
    for (int i = 0; i < *n_y + 2; ++i) {
      const int y0 = i == *n_y + 1 ? Ny - S : i * st;
      for (int j = 0; j <= *n_x; ++j) {
        if (n + 2 > cap) return -1;
        out[n++] = (blk_t){y0, Nx - S - j * st, i, *n_x - j};
        if (j == *n_x) out[n++] = (blk_t){y0, 0, i, -1};              /* the extra left-edge block, PM:323-329 */
      }
    }
  } else {
    *n_x = (Nx - S + st - 1) / st; *n_y = (Ny - S) / st;              /* ceil in x (SMD:461, UGP:479), floor in y */
    for (int i = 0; i < *n_y + 2; ++i) {
      const int y0 = i == *n_y + 1 ? Ny - S : i * st;
      for (int j = 0; j <= *n_x; ++j) {
        if (n + 1 > cap) return -1;
        if (variant == V_DELTAS) out[n++] = (blk_t){y0, j == *n_x ? 0 : Nx - S - j * st, i, *n_x - j};
        else out[n++] = (blk_t){y0, j == *n_x ? Nx - S : j * st, i, j};
      }
    }
  }
  return n;
}

int psm_cpu_num_blocks(int variant, int Ny, int Nx, int S, int ov) {
  int nx, ny;
  blk_t* tmp = (blk_t*)malloc(sizeof(blk_t) * 65536);
  const int n = layout(variant, Ny, Nx, S, ov, tmp, 65536, &nx, &ny);
  free(tmp);
  return n;
}

/* masked mean; NaN for an empty selection, like np.mean of an empty array (SMD:252, UGP:315, PM:417 test it) */
typedef struct { const double* cur; const uint8_t* m; int S; } view_t;
static double mmean(view_t v, int r0, int r1, int c0, int c1) {
  double s = 0.0; long n = 0;
  for (int r = r0; r < r1; ++r)
    for (int c = c0; c < c1; ++c)
      if (v.m[r * v.S + c]) { s += v.cur[r * v.S + c]; ++n; }
  return n ? s / (double)n : NAN;
}
/* data of `data` under the mask of `mask` (SMD:235: the current block's mask on the previous block's strip) */
static double mmean2(const double* data, const uint8_t* m, int S, int r0, int r1, int c0, int c1) {
  view_t v = {data, m, S};
  return mmean(v, r0, r1, c0, c1);
}
static void sub(double* cur, int n, double c) { for (int i = 0; i < n; ++i) cur[i] -= c; }

/* ---- a12: reassembly, one function per reference variant.  pred [B][S*S] (one output channel, modified in place),
 *      masks [B][S*S] (flow cells of the blocks), out [Ny][Nx].  Returns 0, or -1 where the reference itself raises. */
static int assemble_deltas(double* pred, const uint8_t* masks, const blk_t* blk, int B, int S, int ov, int n_x, int n_y,
                           int Ny, int Nx, int strict, double* out) {
  const int st = S - ov, SS = S * S;
  if (n_x < 1) return -1;                                   /* SMD:237-240 reads the previous block */
  const int p_i = Ny - (st * n_y + S), p_j = Nx - (st * n_x + S), lim = ov - p_j;     /* SMD:213,216,238 */
  if (p_i == 0 && strict) return -1;                        /* broadcast error at SMD:335 */
  double* up = (double*)calloc((size_t)n_x + 1, sizeof(double));                      /* BC_ups, SMD:210 */
  const double* prev = NULL;
  memset(out, 0, sizeof(double) * (size_t)Ny * Nx);
  for (int b = 0; b < B; ++b) {
    const int ti = blk[b].ti, tj = blk[b].tj;
    if (ti == n_y + 1 && p_i == 0) continue;
    double* cur = pred + (size_t)b * SS;
    const uint8_t* m = masks + (size_t)b * SS;
    view_t v = {cur, m, S};
    double c;
#define SIDE(w) (mmean(v, 0, S, S - (w), S) - mmean2(prev, m, S, 0, S, 0, (w)))
    if (ti == 0) {                                          /* SMD:228-246 */
      c = b == 0 ? mmean(v, 0, S, S - 1, S) - 0.0 : SIDE(ov);
      if (tj == 0) c = SIDE(lim);
      sub(cur, SS, c);
      up[tj] = mmean(v, S - ov, S, 0, S);
    } else if (ti != n_y + 1) {                             /* SMD:249-283 */
      if (isnan(up[tj])) c = tj == 0 ? SIDE(lim) : (tj == n_x ? mmean(v, 0, ov, 0, S) - up[tj] : SIDE(ov));
      else c = mmean(v, 0, ov, 0, S) - up[tj];
      sub(cur, SS, c);
      up[tj] = mmean(v, S - ov, S, 0, S);
      if (ti == n_y) up[tj] = mmean(v, p_i, S, 0, S);
    } else {                                                /* SMD:286-328 */
      const int a0 = S - p_i - ov, a1 = S - p_i;
      if (tj == n_x) c = mmean(v, a0, a1, 0, S) - up[tj];
      else {
        long n_up = 0;
        for (int r = a0; r < a1; ++r) for (int q = 0; q < S; ++q) n_up += m[r * S + q] != 0;
        if ((double)n_up / (128.0 * 128.0) > 0.9) c = tj == 0 ? SIDE(lim) : SIDE(ov);   /* SMD:307 */
        else c = mmean(v, 0, S - p_i, 0, S) - up[tj];
      }
      sub(cur, SS, c);
    }
#undef SIDE
    prev = cur;
    const int jr = n_x - tj, xs = tj == 0 ? 0 : Nx - S - jr * st;                       /* paste, SMD:334-348 */
    if (ti == n_y + 1) { for (int r = 0; r < p_i; ++r) memcpy(out + (size_t)(Ny - p_i + r) * Nx + xs, cur + (size_t)(S - p_i + r) * S, sizeof(double) * S); }
    else for (int r = 0; r < S; ++r) memcpy(out + (size_t)(st * ti + r) * Nx + xs, cur + (size_t)r * S, sizeof(double) * S);
  }
  double acc = 0.0;                                         /* SMD:350 */
  for (int r = 0; r < Ny; ++r) acc += 3.0 * out[(size_t)r * Nx + Nx - 1] - out[(size_t)r * Nx + Nx - 2];
  const double shift = acc / Ny / 3.0;
  for (size_t i = 0; i < (size_t)Ny * Nx; ++i) out[i] -= shift;
  free(up);
  return 0;
}

static int assemble_gradp(int field, double* pred, const uint8_t* masks, const blk_t* blk, int B, int S, int ov, int n_x, int n_y,
                          int Ny, int Nx, int strict, double* out) {
  const int st = S - ov, SS = S * S;
  if (n_x < 1) return -1;                                   /* UGP:307-310 */
  const int p_i = Ny - (S * (n_y + 1) - n_y * ov), p_j = (Nx - S) - n_x * st, lim = ov - p_j;   /* UGP:277,278,308 */
  const int skip_last = p_i == 0 && !strict;
  double* up = (double*)calloc((size_t)n_x + 1, sizeof(double));
  const double* prev = NULL;
  memset(out, 0, sizeof(double) * (size_t)Ny * Nx);
  for (int b = 0; b < B; ++b) {
    const int ti = blk[b].ti, tj = blk[b].tj;
    if (ti == n_y + 1 && skip_last) continue;
    double* cur = pred + (size_t)b * SS;
    const uint8_t* m = masks + (size_t)b * SS;
    view_t v = {cur, m, S};
    double c;
#define SIDE(w) (mmean(v, 0, S, 0, (w)) - mmean2(prev, m, S, 0, S, S - (w), S))
    if (ti == 0) {                                          /* UGP:288-312 */
      if (b == 0) {
        if (field == 0) {                                   /* dp/dx: first column with a flow cell, UGP:294-300 */
          int col = 0;
          for (;; ++col) { if (col >= S) { free(up); return -1; } int any = 0; for (int r = 0; r < S; ++r) any |= m[r * S + col]; if (any) break; }
          c = mmean(v, 0, S, col, col + 1) - 0.0;
        } else c = mmean(v, 1, 2, 0, S) - 0.0;              /* dp/dy: row 1, UGP:302-303 */
      } else c = SIDE(ov);
      if (tj == n_x) c = SIDE(lim);
      sub(cur, SS, c);
      up[tj] = mmean(v, S - ov, S, 0, S);
    } else if (ti != n_y + 1) {                             /* UGP:314-328 */
      c = isnan(up[tj]) ? (tj == n_x ? SIDE(lim) : SIDE(ov)) : mmean(v, 0, ov, 0, S) - up[tj];
      sub(cur, SS, c);
      up[tj] = mmean(v, S - ov, S, 0, S);
      if (ti == n_y) up[tj] = mmean(v, p_i, S, 0, S);
    } else {                                                /* UGP:330-341 */
      if (isnan(up[tj])) c = tj == n_x ? SIDE(lim) : SIDE(ov);
      else c = (p_i != 0 ? mmean(v, S - p_i - ov, S - p_i, 0, S) : NAN) - up[tj];     /* [-p_i-ov:-p_i] is empty when p_i == 0 */
      sub(cur, SS, c);
    }
#undef SIDE
    prev = cur;
    const int ys = ti == n_y + 1 ? Ny - st : ti * st, r0 = ti == n_y + 1 ? ov : 0;     /* paste, UGP:345-356 */
    for (int r = r0; r < S; ++r) {
      double* dst = out + (size_t)(ys + r - r0) * Nx;
      if (tj == n_x) memcpy(dst + Nx - lim, cur + (size_t)r * S + S - lim, sizeof(double) * lim);
      else memcpy(dst + tj * st, cur + (size_t)r * S, sizeof(double) * S);
    }
  }
  double acc = 0.0;
  if (field == 0) { for (int r = 0; r < Ny; ++r) acc += 3.0 * out[(size_t)r * Nx] - out[(size_t)r * Nx + 1]; acc /= Ny; }     /* UGP:359 */
  else { for (int q = 0; q < Nx; ++q) acc += 3.0 * out[(size_t)Nx + q] - out[(size_t)2 * Nx + q]; acc /= Nx; }                /* UGP:361 */
  const double shift = acc / 3.0;
  for (size_t i = 0; i < (size_t)Ny * Nx; ++i) out[i] -= shift;
  free(up);
  return 0;
}

static int assemble_chapter5(double* pred, const uint8_t* masks, const blk_t* blk, int B, int S, int av, int n_x, int n_y,
                             int Ny, int Nx, double* out) {
  const int st = S - av, SS = S * S;
  const int p = Ny - (S * (n_y + 1) - n_y * av), p_j = (Nx - S) - n_x * S + n_x * av;      /* PM:410, 397 */
  double* up = (double*)calloc((size_t)n_x + 1, sizeof(double));                            /* BC_ups */
  double up_m1 = NAN, ant0 = NAN, alter = 0.0;                                              /* BC_up_, BC_ant_0, BC_alter */
  memset(out, 0, sizeof(double) * (size_t)Ny * Nx);
  for (int b = 0; b < B; ++b) {
    const int ti = blk[b].ti, tj = blk[b].tj;
    double* cur = pred + (size_t)b * SS;
    const uint8_t* m = masks + (size_t)b * SS;
    view_t v = {cur, m, S};
    const int R0 = S - av, C0 = p_j, C1 = p_j + av;
    double c;
    if (ti == 0) {                                          /* PM:388-405 */
      if (tj == n_x) { c = mmean(v, 0, S, R0, S) - 0.0; sub(cur, SS, c); up[tj] = mmean(v, R0, S, R0, S); }
      else if (tj == -1) { c = mmean(v, 0, S, C0, C1) - ant0; sub(cur, SS, c); up_m1 = mmean(v, R0, S, C0, C1); }
      else { c = mmean(v, 0, S, R0, S) - ant0; sub(cur, SS, c); up[tj] = mmean(v, R0, S, 0, S); }
      ant0 = mmean(v, 0, S, 0, av);
    } else if (ti == n_y + 1) {                             /* PM:407-423 */
      const int T0 = S - p - av, T1 = S - p;
      if (tj == -1) c = mmean(v, T0, T1, C0, C1) - up_m1;
      else if (isnan(up[tj])) c = mmean(v, 0, S, R0, S) - alter;
      else c = mmean(v, T0, T1, 0, S) - up[tj];
      sub(cur, SS, c);
    } else {                                                /* PM:425-441 */
      if (tj == -1) {
        c = mmean(v, 0, av, C0, C1) - up_m1;
        sub(cur, SS, c);
        double s = 0.0;                                     /* PM:432: np.mean WITHOUT the mask */
        for (int r = R0; r < S; ++r) for (int q = C0; q < C1; ++q) s += cur[r * S + q];
        up_m1 = s / (double)((S - R0) * (C1 - C0));
      } else {
        c = isnan(up[tj]) ? mmean(v, 0, S, R0, S) - alter : mmean(v, 0, av, 0, S) - up[tj];
        sub(cur, SS, c);
        up[tj] = mmean(v, R0, S, 0, S);
      }
    }
    alter = mmean(v, 0, S, 0, av);                          /* PM:445 */
    if (ti == n_y + 1 && tj == -1) {                        /* paste, PM:449-467 */
      const int w = Nx - (n_x + 1) * st - av;
      for (int r = av; r < S; ++r) if (w > 0) memcpy(out + (size_t)(Ny - st + r - av) * Nx, cur + (size_t)r * S, sizeof(double) * w);
    } else if (tj == -1) {
      for (int r = 0; r < S; ++r) memcpy(out + (size_t)(ti * st + r) * Nx, cur + (size_t)r * S, sizeof(double) * S);
    } else {
      const int xs = Nx - S - (n_x - tj) * st;
      if (ti == n_y + 1) for (int r = av; r < S; ++r) memcpy(out + (size_t)(Ny - st + r - av) * Nx + xs, cur + (size_t)r * S, sizeof(double) * S);
      else for (int r = 0; r < S; ++r) memcpy(out + (size_t)(ti * st + r) * Nx + xs, cur + (size_t)r * S, sizeof(double) * S);
    }
  }
  double acc = 0.0;                                         /* PM:472 */
  for (int r = 0; r < Ny; ++r) acc += 3.0 * out[(size_t)r * Nx + Nx - 1] - out[(size_t)r * Nx + Nx - 2];
  const double shift = acc / Ny / 3.0;
  for (size_t i = 0; i < (size_t)Ny * Nx; ++i) out[i] -= shift;
  free(up);
  return 0;
}

/* ---- one grid-native solve (PM:299-473, SMD:452-575, UGP:470-547).  grid [Ny][Nx][grid_c] float64 (normalised image,
 *      first c_in channels used), fields [Ny][Nx][c_out] float64, x_input [B][p_in] (optional).  Returns the number of
 *      blocks, or < 0. */
int psm_cpu_solve_grid(const psm_cpu_model* M, const double* grid, int Ny, int Nx, int grid_c, double* fields, double* x_input, int threads) {
  const int S = M->S, SS = S * S, Kin = SS * M->c_in, Kout = SS * M->c_out, Pi = M->p_in, Po = M->p_out;
  blk_t* blk = (blk_t*)malloc(sizeof(blk_t) * 65536);
  int n_x = 0, n_y = 0;
  const int B = layout(M->variant, Ny, Nx, S, M->ov, blk, 65536, &n_x, &n_y);
  if (B < 1) { free(blk); return -1; }
#ifdef _OPENMP
  if (threads > 0) omp_set_num_threads(threads);
#endif
  const int prof = getenv("PSM_CPU_PROFILE") != NULL;
  double tp[6]; tp[0] = now_ms();
  double* X = (double*)malloc(sizeof(double) * (size_t)B * Kin);          /* centred blocks */
  uint8_t* masks = (uint8_t*)malloc((size_t)B * SS);
  double* coef = (double*)calloc((size_t)B * Pi, sizeof(double));
  /* a7 + centring (PM:341-349 / sklearn transform: (X - mean_) @ components_.T) */
#pragma omp parallel for schedule(static)
  for (int b = 0; b < B; ++b)
    for (int r = 0; r < S; ++r)
      for (int q = 0; q < S; ++q) {
        const double* px = grid + ((size_t)(blk[b].y0 + r) * Nx + blk[b].x0 + q) * grid_c;
        const size_t k = ((size_t)r * S + q) * M->c_in;
        for (int ch = 0; ch < M->c_in; ++ch) X[(size_t)b * Kin + k + ch] = px[ch] - M->mean_in[k + ch];
        masks[(size_t)b * SS + r * S + q] = px[M->sdf_ch] != 0.0;
      }
  tp[1] = now_ms();
  /* a9: encode, float64, split over K like a blocked GEMM: every thread owns a contiguous range of 256-wide K chunks
   * (X chunk 60 KB + comp_in chunk 256 KB stay in its cache, each element of comp_in is used B times) and a private
   * [B][p_in] accumulator; the accumulators are added in thread order (deterministic for a given thread count) */
  {
    const int KC = 256, nchunk = (Kin + KC - 1) / KC;
    int nt = 1;
#ifdef _OPENMP
    nt = omp_get_max_threads();
#endif
    double* part = (double*)calloc((size_t)nt * B * Pi, sizeof(double));
#pragma omp parallel
    {
      int t = 0, tn = 1;
#ifdef _OPENMP
      t = omp_get_thread_num(); tn = omp_get_num_threads();
#endif
      double* acc = part + (size_t)t * B * Pi;
      const int c0 = (int)((long)nchunk * t / tn), c1 = (int)((long)nchunk * (t + 1) / tn);
      for (int c = c0; c < c1; ++c) {
        const int k0 = c * KC, k1 = k0 + KC < Kin ? k0 + KC : Kin;
        for (int p = 0; p < Pi; ++p) {
          const double* cp = M->comp_in + (size_t)p * Kin;
          int b = 0;
          for (; b + 4 <= B; b += 4) {                      /* 4 blocks share every load of the component row */
            const double *x0 = X + (size_t)b * Kin, *x1 = x0 + Kin, *x2 = x1 + Kin, *x3 = x2 + Kin;
            v4d z = {0, 0, 0, 0}, a0 = z, a1 = z, b0 = z, b1 = z, c0 = z, c1 = z, d0 = z, d1 = z;
            int k = k0;
            for (; k + 8 <= k1; k += 8) {
              const v4d w0 = ld4(cp + k), w1 = ld4(cp + k + 4);
              a0 += ld4(x0 + k) * w0; a1 += ld4(x0 + k + 4) * w1;
              b0 += ld4(x1 + k) * w0; b1 += ld4(x1 + k + 4) * w1;
              c0 += ld4(x2 + k) * w0; c1 += ld4(x2 + k + 4) * w1;
              d0 += ld4(x3 + k) * w0; d1 += ld4(x3 + k + 4) * w1;
            }
            const v4d ta = a0 + a1, tb = b0 + b1, tc = c0 + c1, td = d0 + d1;
            double s[4] = {(ta[0] + ta[1]) + (ta[2] + ta[3]), (tb[0] + tb[1]) + (tb[2] + tb[3]), (tc[0] + tc[1]) + (tc[2] + tc[3]), (td[0] + td[1]) + (td[2] + td[3])};
            for (; k < k1; ++k) { s[0] += x0[k] * cp[k]; s[1] += x1[k] * cp[k]; s[2] += x2[k] * cp[k]; s[3] += x3[k] * cp[k]; }
            for (int q = 0; q < 4; ++q) acc[(size_t)(b + q) * Pi + p] += s[q];
          }
          for (; b < B; ++b) {
            const double* xb = X + (size_t)b * Kin;
            v4d a0 = {0, 0, 0, 0}, a1 = a0;
            int k = k0;
            for (; k + 8 <= k1; k += 8) { a0 += ld4(xb + k) * ld4(cp + k); a1 += ld4(xb + k + 4) * ld4(cp + k + 4); }
            const v4d t = a0 + a1;
            double s0 = (t[0] + t[1]) + (t[2] + t[3]);
            for (; k < k1; ++k) s0 += xb[k] * cp[k];
            acc[(size_t)b * Pi + p] += s0;
          }
        }
      }
    }
    for (int t = 0; t < nt; ++t)
      for (size_t i = 0; i < (size_t)B * Pi; ++i) coef[i] += part[(size_t)t * B * Pi + i];
    free(part);
  }
  tp[2] = now_ms();
  /* scaler (PM:351; SMD:505-523) -> float32 network (Keras Dense, PM:121-134) -> inverse scaler (SMD:532-539) */
  int wmax = Pi > Po ? Pi : Po;
  for (int l = 0; l <= M->n_dense; ++l) if (M->dims[l] > wmax) wmax = M->dims[l];
  float* h0 = (float*)malloc(sizeof(float) * (size_t)B * wmax), *h1 = (float*)malloc(sizeof(float) * (size_t)B * wmax);
  for (int b = 0; b < B; ++b)
    for (int p = 0; p < Pi; ++p) {
      const double t = coef[(size_t)b * Pi + p];
      double x;
      if (M->scaler == SC_MAX_ABS) x = t / M->in_a[0];
      else if (M->scaler == SC_STD) x = (t - M->in_a[p]) / M->in_b[p];
      else x = (t - M->in_a[p]) / (M->in_b[p] - M->in_a[p]);
      if (x_input) x_input[(size_t)b * Pi + p] = x;
      h0[(size_t)b * wmax + p] = (float)x;
    }
  for (int l = 0; l < M->n_dense; ++l) {
    const int ni = M->dims[l], no = M->dims[l + 1];
    const float* W = M->W[l]; const float* bias = M->b[l];
#pragma omp parallel for schedule(static)
    for (int b = 0; b < B; ++b) {
      float* o = h1 + (size_t)b * wmax;
      for (int j = 0; j < no; ++j) o[j] = 0.f;
      for (int k = 0; k < ni; ++k) { const float a = h0[(size_t)b * wmax + k]; const float* w = W + (size_t)k * no; for (int j = 0; j < no; ++j) o[j] += a * w[j]; }
      for (int j = 0; j < no; ++j) { const float t = o[j] + bias[j]; o[j] = (l + 1 < M->n_dense && t < 0.f) ? 0.f : t; }
    }
    float* t = h0; h0 = h1; h1 = t;
  }
  double* res = (double*)malloc(sizeof(double) * (size_t)B * Po);
  for (int b = 0; b < B; ++b)
    for (int p = 0; p < Po; ++p) {
      const double r = (double)h0[(size_t)b * wmax + p];
      if (M->scaler == SC_MAX_ABS) res[(size_t)b * Po + p] = r * M->out_a[0];
      else if (M->scaler == SC_STD) res[(size_t)b * Po + p] = r * M->out_b[p] + M->out_a[p];
      else res[(size_t)b * Po + p] = r * (M->out_b[p] - M->out_a[p]) + M->out_a[p];
    }
  tp[3] = now_ms();
  /* a11: decode (PM:365-366, SMD:541-551), float64, channel-interleaved like UGP:534; column chunks per thread so that
   * comp_out is streamed once and every chunk of it serves all blocks */
  double* pred = (double*)malloc(sizeof(double) * (size_t)M->c_out * B * SS);   /* [c_out][B][SS]: one plane per field */
  const int CH = 512;
#pragma omp parallel for schedule(dynamic, 1)
  for (int k0 = 0; k0 < Kout; k0 += CH) {
    const int k1 = k0 + CH < Kout ? k0 + CH : Kout;
    double acc[512];
    for (int b = 0; b < B; ++b) {
      for (int k = k0; k < k1; ++k) acc[k - k0] = 0.0;
      for (int p = 0; p < Po; ++p) {
        const double r = res[(size_t)b * Po + p];
        const double* cp = M->comp_out + (size_t)p * Kout;
        for (int k = k0; k < k1; ++k) acc[k - k0] += r * cp[k];
      }
      for (int k = k0; k < k1; ++k) {
        const int px = k / M->c_out, f = k - px * M->c_out;
        pred[((size_t)f * B + b) * SS + px] = (acc[k - k0] + M->mean_out[k]) * M->out_scale;
      }
    }
  }
  tp[4] = now_ms();
  /* a12: reassembly per output field */
  double* plane = (double*)malloc(sizeof(double) * (size_t)Ny * Nx);
  int rc = 0;
  for (int f = 0; f < M->c_out && rc == 0; ++f) {
    double* pf = pred + (size_t)f * B * SS;
    if (M->variant == V_DELTAS) rc = assemble_deltas(pf, masks, blk, B, S, M->ov, n_x, n_y, Ny, Nx, M->strict, plane);
    else if (M->variant == V_GRADP) rc = assemble_gradp(f, pf, masks, blk, B, S, M->ov, n_x, n_y, Ny, Nx, M->strict, plane);
    else rc = assemble_chapter5(pf, masks, blk, B, S, M->ov, n_x, n_y, Ny, Nx, plane);
    if (rc == 0) for (size_t i = 0; i < (size_t)Ny * Nx; ++i) fields[i * M->c_out + f] = plane[i];
  }
  tp[5] = now_ms();
  if (prof) fprintf(stderr, "psm_cpu: blocks %.2f ms, encode %.2f, network %.2f, decode %.2f, reassembly %.2f\n", tp[1] - tp[0], tp[2] - tp[1], tp[3] - tp[2], tp[4] - tp[3], tp[5] - tp[4]);
  free(plane); free(pred); free(res); free(h0); free(h1); free(coef); free(masks); free(X); free(blk);
  return rc == 0 ? B : -2;
}
