// ensemble_host.cpp -- a C++ host driving the surrogate through the C-ABI only (include/psm.h), the way a PISO
// solver that owns several independent cases would: the model is installed once, then grid images are packed into the
// slots of the pinned ring (H2D / kernels / D2H of neighbouring tickets overlap) and the results read from the slots.
//
//   g++ -std=c++17 -O2 -I include examples/ensemble_host.cpp -L <dir of libpsm_hip.so> -lpsm_hip -o ensemble_host
//   ./ensemble_host model.bin grids.bin n_cases fields_out.bin
//
// model.bin (written by tests/test_cpp_host.py): int32 header {variant, c_in, c_out, p_in, p_out, n_dense, scaler,
// ny, nx} followed by float64 comp_in, mean_in, comp_out, mean_out, scaler arrays, then per layer int32 {n_in,
// n_out}, float32 kernel, float32 bias.  grids.bin: float32 [n_cases][ny][nx][c_in].
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "psm.h"

#define CHECK(call)                                                                                     \
  do { const int rc_ = (call); if (rc_ != PSM_OK) { std::fprintf(stderr, "%s -> %d: %s\n", #call, rc_, psm_last_error(sm)); return 2; } } while (0)

template <typename T>
static bool read_vec(FILE* f, std::vector<T>& v, size_t n) { v.resize(n); return std::fread(v.data(), sizeof(T), n, f) == n; }

int main(int argc, char** argv) {
  if (argc != 5) { std::fprintf(stderr, "usage: %s model.bin grids.bin n_cases fields_out.bin\n", argv[0]); return 1; }
  FILE* fm = std::fopen(argv[1], "rb");
  if (!fm) { std::perror(argv[1]); return 1; }
  int32_t hd[9];
  if (std::fread(hd, sizeof(int32_t), 9, fm) != 9) return 1;
  const int variant = hd[0], c_in = hd[1], c_out = hd[2], p_in = hd[3], p_out = hd[4], n_dense = hd[5], scaler = hd[6], ny = hd[7], nx = hd[8];
  const size_t K_in = 128u * 128u * c_in, K_out = 128u * 128u * c_out;
  std::vector<double> comp_in, mean_in, comp_out, mean_out, in_a, in_b, out_a, out_b;
  const size_t ns_in = scaler == PSM_SCALER_MAX_ABS ? 1 : p_in, ns_out = scaler == PSM_SCALER_MAX_ABS ? 1 : p_out;
  if (!read_vec(fm, comp_in, p_in * K_in) || !read_vec(fm, mean_in, K_in) || !read_vec(fm, comp_out, p_out * K_out) ||
      !read_vec(fm, mean_out, K_out) || !read_vec(fm, in_a, ns_in) || !read_vec(fm, in_b, ns_in) || !read_vec(fm, out_a, ns_out) ||
      !read_vec(fm, out_b, ns_out)) return 1;

  psm_handle* sm = nullptr;
  psm_config cfg = {PSM_ABI_VERSION, variant, 128, 0, c_in, c_out, p_in, p_out, n_dense, scaler, c_in - 1, 0, 1, 0, PSM_PRECISION_F32};
  if (psm_create(&cfg, &sm) != PSM_OK) { std::fprintf(stderr, "psm_create: %s\n", psm_last_error(nullptr)); return 2; }
  CHECK(psm_set_pca(sm, comp_in.data(), mean_in.data(), comp_out.data(), mean_out.data()));
  CHECK(psm_set_scaler(sm, in_a.data(), in_b.data(), out_a.data(), out_b.data()));
  for (int l = 0; l < n_dense; ++l) {
    int32_t sh[2];
    std::vector<float> W, b;
    if (std::fread(sh, sizeof(int32_t), 2, fm) != 2 || !read_vec(fm, W, (size_t)sh[0] * sh[1]) || !read_vec(fm, b, sh[1])) return 1;
    CHECK(psm_set_dense(sm, l, sh[0], sh[1], W.data(), b.data()));
  }
  std::fclose(fm);
  CHECK(psm_plan_grid(sm, ny, nx));

  const int n_cases = std::atoi(argv[3]);
  const size_t gin = (size_t)ny * nx * c_in, gout = (size_t)ny * nx * c_out;
  std::vector<float> grids, fields((size_t)n_cases * gout);
  FILE* fg = std::fopen(argv[2], "rb");
  if (!fg || !read_vec(fg, grids, (size_t)n_cases * gin)) { std::fprintf(stderr, "cannot read %s\n", argv[2]); return 1; }
  std::fclose(fg);

  // Zero-copy ring (include/psm.h): the grid of case k is packed straight into the pinned buffer of its slot
  // (here a memcpy from the file image; a solver writes its fields there directly), the ticket is ONE graph replay
  // (H2D -> kernels -> D2H on the slot's own stream), and the result is read from the slot's pinned output buffer.
  const int depth = 3;                                   // tickets in flight (< PSM_RING_SLOTS)
  std::vector<int64_t> ticket(n_cases);
  std::vector<float*> out_of(n_cases, nullptr);
  const auto t0 = std::chrono::steady_clock::now();
  for (int k = 0; k < n_cases + depth; ++k) {
    if (k >= depth) {                                    // retire the oldest ticket: its slot comes free
      CHECK(psm_ring_wait(sm, ticket[k - depth]));
      std::memcpy(&fields[(size_t)(k - depth) * gout], out_of[k - depth], gout * sizeof(float));
    }
    if (k < n_cases) {
      float *gi = nullptr, *fo = nullptr;
      CHECK(psm_ring_acquire(sm, &ticket[k], &gi, &fo));
      std::memcpy(gi, &grids[(size_t)k * gin], gin * sizeof(float));
      out_of[k] = fo;
      CHECK(psm_ring_submit(sm, ticket[k], 1, nullptr));
    }
  }
  const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
  std::printf("blocks per case %d, %d cases in %.1f us: %.0f solves/s (host buffers, zero-copy ring, %d tickets in flight)\n",
              psm_num_blocks(sm), n_cases, us, n_cases / us * 1e6, depth);
  FILE* fo = std::fopen(argv[4], "wb");
  if (!fo || std::fwrite(fields.data(), sizeof(float), fields.size(), fo) != fields.size()) return 1;
  std::fclose(fo);
  psm_destroy(sm);
  return 0;
}
