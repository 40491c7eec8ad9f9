// ensemble_multi.cpp -- a C++ host that spreads an ensemble of independent PISO cases over the GPUs of one node, through the C-ABI
// only (include/psm.h).  The reference has no multi-device path (its MPI ranks funnel every cell to rank 0,
// Thesis_Work/Chapter5/parallelized/test_case/python_module.py:179-185, 258-260, 501-511); here the case batch -- BASELINE
// configs[3]: independent geometries / time steps -- is partitioned contiguously over the devices, ONE handle and ONE host thread per
// device, and no data-path collective: every device solves its own shard (dist.shard_cases's rule: first = r * base + min(r, extra))
// in case batches of --batch cases per call and stores the fields straight into the shared result array (registered host memory:
// the D2H copy of a shard is a DMA into its final place).
//
//   g++ -std=c++17 -O2 -pthread -I include examples/ensemble_multi.cpp -L <dir of libpsm_hip.so> -lpsm_hip -o ensemble_multi
//   ./ensemble_multi model.bin grids.bin n_cases fields_out.bin --devices 0,1,2,3,4,5,6,7 [--batch 8] [--repeat 1]
//
// --devices takes a list of HIP ordinals; an ordinal may repeat (two handles sharing a card: what the one-GPU test box runs).
// With -DPSM_WITH_RCCL (link -lrccl -lamdhip64) and --rccl-broadcast the model file is read by the first device's thread only and its
// bytes travel to the other devices with ncclBroadcast over xGMI before they are installed -- the C++ counterpart of
// dist.broadcast_model; it needs distinct devices.
//
// model.bin / grids.bin: the formats of examples/ensemble_host.cpp (written by tests/test_cpp_host.py).
#include <atomic>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "psm.h"

#ifdef PSM_WITH_RCCL
#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>
#endif

namespace {

struct ModelFile {
  int32_t hd[9];
  std::vector<char> bytes;          // everything behind the header, as read
};

bool read_model(const char* path, ModelFile& m) {
  FILE* f = std::fopen(path, "rb");
  if (!f) { std::perror(path); return false; }
  if (std::fread(m.hd, sizeof(int32_t), 9, f) != 9) { std::fclose(f); return false; }
  std::fseek(f, 0, SEEK_END);
  const long end = std::ftell(f);
  std::fseek(f, 9 * sizeof(int32_t), SEEK_SET);
  m.bytes.resize((size_t)end - 9 * sizeof(int32_t));
  const bool ok = std::fread(m.bytes.data(), 1, m.bytes.size(), f) == m.bytes.size();
  std::fclose(f);
  return ok;
}

// install the model of `m` on a fresh handle for `device`; -> nullptr + message on failure
psm_handle* make_handle(const ModelFile& m, int device, int max_cases, std::string& err) {
  const int variant = m.hd[0], c_in = m.hd[1], c_out = m.hd[2], p_in = m.hd[3], p_out = m.hd[4], n_dense = m.hd[5], scaler = m.hd[6], ny = m.hd[7], nx = m.hd[8];
  const size_t K_in = 128u * 128u * c_in, K_out = 128u * 128u * c_out;
  const size_t ns_in = scaler == PSM_SCALER_MAX_ABS ? 1 : p_in, ns_out = scaler == PSM_SCALER_MAX_ABS ? 1 : p_out;
  const char* p = m.bytes.data();
  const char* end = p + m.bytes.size();
  auto take = [&](size_t n_bytes) -> const char* { if (p + n_bytes > end) return nullptr; const char* q = p; p += n_bytes; return q; };
  const double* comp_in = (const double*)take(p_in * K_in * 8);
  const double* mean_in = (const double*)take(K_in * 8);
  const double* comp_out = (const double*)take(p_out * K_out * 8);
  const double* mean_out = (const double*)take(K_out * 8);
  const double* in_a = (const double*)take(ns_in * 8);
  const double* in_b = (const double*)take(ns_in * 8);
  const double* out_a = (const double*)take(ns_out * 8);
  const double* out_b = (const double*)take(ns_out * 8);
  if (!out_b) { err = "model file too short"; return nullptr; }
  psm_handle* h = nullptr;
  psm_config cfg = {PSM_ABI_VERSION, variant, 128, 0, c_in, c_out, p_in, p_out, n_dense, scaler, c_in - 1, device, max_cases, 0, PSM_PRECISION_F32};
  if (psm_create(&cfg, &h) != PSM_OK) { err = std::string("psm_create: ") + psm_last_error(nullptr); return nullptr; }
  auto bad = [&](const char* what) { err = std::string(what) + ": " + psm_last_error(h); psm_destroy(h); return (psm_handle*)nullptr; };
  if (psm_set_pca(h, comp_in, mean_in, comp_out, mean_out) != PSM_OK) return bad("psm_set_pca");
  if (psm_set_scaler(h, in_a, in_b, out_a, out_b) != PSM_OK) return bad("psm_set_scaler");
  for (int l = 0; l < n_dense; ++l) {
    const int32_t* sh = (const int32_t*)take(8);
    if (!sh) { err = "model file too short"; psm_destroy(h); return nullptr; }
    const float* W = (const float*)take((size_t)sh[0] * sh[1] * 4);
    const float* b = (const float*)take((size_t)sh[1] * 4);
    if (!b) { err = "model file too short"; psm_destroy(h); return nullptr; }
    if (psm_set_dense(h, l, sh[0], sh[1], W, b) != PSM_OK) return bad("psm_set_dense");
  }
  if (psm_plan_grid(h, ny, nx) != PSM_OK) return bad("psm_plan_grid");
  return h;
}

// contiguous balanced partition (solving-..._amd/dist.py, shard_cases)
void shard(int n_cases, int world, int rank, int* first, int* count) {
  const int base = n_cases / world, extra = n_cases % world;
  *first = rank * base + (rank < extra ? rank : extra);
  *count = base + (rank < extra ? 1 : 0);
}

}  // namespace

int main(int argc, char** argv) {
  if (argc < 5) {
    std::fprintf(stderr, "usage: %s model.bin grids.bin n_cases fields_out.bin [--devices 0,1,...] [--batch 8] [--repeat 1] [--rccl-broadcast]\n", argv[0]);
    return 1;
  }
  std::vector<int> devices = {0};
  int batch = 8, repeat = 1;
  bool rccl = false;
  for (int i = 5; i < argc; ++i) {
    const std::string a = argv[i];
    if (a == "--devices" && i + 1 < argc) {
      devices.clear();
      for (const char* q = argv[++i]; *q;) { devices.push_back((int)std::strtol(q, const_cast<char**>(&q), 10)); if (*q == ',') ++q; }
    } else if (a == "--batch" && i + 1 < argc) batch = std::atoi(argv[++i]);
    else if (a == "--repeat" && i + 1 < argc) repeat = std::atoi(argv[++i]);
    else if (a == "--rccl-broadcast") rccl = true;
    else { std::fprintf(stderr, "unknown argument %s\n", a.c_str()); return 1; }
  }
  const int world = (int)devices.size();
  if (world < 1 || batch < 1 || repeat < 1) { std::fprintf(stderr, "bad --devices / --batch / --repeat\n"); return 1; }
  const int n_cases = std::atoi(argv[3]);

  ModelFile model;
  if (!read_model(argv[1], model)) { std::fprintf(stderr, "cannot read %s\n", argv[1]); return 1; }
  const int c_in = model.hd[1], c_out = model.hd[2], ny = model.hd[7], nx = model.hd[8];
  const size_t gin = (size_t)ny * nx * c_in, gout = (size_t)ny * nx * c_out;
  std::vector<float> grids((size_t)n_cases * gin), fields((size_t)n_cases * gout);
  {
    FILE* fg = std::fopen(argv[2], "rb");
    if (!fg || std::fread(grids.data(), sizeof(float), grids.size(), fg) != grids.size()) { std::fprintf(stderr, "cannot read %s\n", argv[2]); return 1; }
    std::fclose(fg);
  }

  std::vector<ModelFile> per_dev(world);        // what each device installs: the file image, or the bytes it received over RCCL
  for (auto& m : per_dev) std::memcpy(m.hd, model.hd, sizeof(model.hd));
  if (rccl) {
#ifdef PSM_WITH_RCCL
    // the 36 bytes of header are known to every thread (command line); the payload travels device 0 -> every device
    for (int r = 0; r < world; ++r)
      for (int q = 0; q < r; ++q)
        if (devices[r] == devices[q]) { std::fprintf(stderr, "--rccl-broadcast needs distinct devices\n"); return 1; }
    std::vector<ncclComm_t> comms(world);
    if (ncclCommInitAll(comms.data(), world, devices.data()) != ncclSuccess) { std::fprintf(stderr, "ncclCommInitAll failed\n"); return 2; }
    std::vector<void*> dbuf(world, nullptr);
    std::vector<hipStream_t> st(world);
    const size_t nb = model.bytes.size();
    for (int r = 0; r < world; ++r) {
      if (hipSetDevice(devices[r]) != hipSuccess || hipMalloc(&dbuf[r], nb) != hipSuccess || hipStreamCreate(&st[r]) != hipSuccess) { std::fprintf(stderr, "hip setup failed\n"); return 2; }
      if (r == 0 && hipMemcpy(dbuf[0], model.bytes.data(), nb, hipMemcpyHostToDevice) != hipSuccess) return 2;
    }
    ncclGroupStart();
    for (int r = 0; r < world; ++r) ncclBroadcast(dbuf[r], dbuf[r], nb, ncclChar, 0, comms[r], st[r]);
    if (ncclGroupEnd() != ncclSuccess) { std::fprintf(stderr, "ncclBroadcast failed\n"); return 2; }
    for (int r = 0; r < world; ++r) {
      (void)hipSetDevice(devices[r]);
      if (hipStreamSynchronize(st[r]) != hipSuccess) return 2;
      per_dev[r].bytes.resize(nb);
      if (hipMemcpy(per_dev[r].bytes.data(), dbuf[r], nb, hipMemcpyDeviceToHost) != hipSuccess) return 2;
      (void)hipFree(dbuf[r]); (void)hipStreamDestroy(st[r]); ncclCommDestroy(comms[r]);
    }
    std::printf("model payload (%zu bytes) broadcast from device %d to %d device(s) with ncclBroadcast\n", nb, devices[0], world);
#else
    std::fprintf(stderr, "built without -DPSM_WITH_RCCL\n");
    return 1;
#endif
  } else {
    for (auto& m : per_dev) m.bytes = model.bytes;
  }

  std::vector<std::string> errs(world);
  std::vector<double> us(world, 0.0);
  std::vector<int> blocks(world, 0);
  std::atomic<int> ready{0}, failed{0};
  std::vector<std::thread> th;
  for (int r = 0; r < world; ++r)
    th.emplace_back([&, r]() {
      int first = 0, count = 0;
      shard(n_cases, world, r, &first, &count);
      psm_handle* h = make_handle(per_dev[r], devices[r], count < batch ? (count > 0 ? count : 1) : batch, errs[r]);
      if (!h) failed.fetch_add(1);
      else {
        blocks[r] = psm_num_blocks(h);
        // the shard's ranges of the shared arrays, registered: both copies of every call are DMAs from / into their final place
        if (count > 0 && (psm_host_register(h, grids.data() + (size_t)first * gin, (size_t)count * gin * sizeof(float)) != PSM_OK ||
                          psm_host_register(h, fields.data() + (size_t)first * gout, (size_t)count * gout * sizeof(float)) != PSM_OK)) {
          errs[r] = std::string("psm_host_register: ") + psm_last_error(h); failed.fetch_add(1);
        }
      }
      ready.fetch_add(1);
      while (ready.load() < world) std::this_thread::yield();            // every device set up: start together
      if (failed.load() == 0) {
        const auto t0 = std::chrono::steady_clock::now();
        for (int rep = 0; rep < repeat && errs[r].empty(); ++rep)
          for (int k = 0; k < count; k += batch) {
            const int n = count - k < batch ? count - k : batch;
            if (psm_solve_grid(h, grids.data() + (size_t)(first + k) * gin, n, nullptr, fields.data() + (size_t)(first + k) * gout) != PSM_OK) {
              errs[r] = std::string("psm_solve_grid: ") + psm_last_error(h); failed.fetch_add(1); break;
            }
          }
        us[r] = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
      }
      if (h) psm_destroy(h);
    });
  for (auto& t : th) t.join();
  if (failed.load()) {
    for (int r = 0; r < world; ++r) if (!errs[r].empty()) std::fprintf(stderr, "device slot %d (HIP device %d): %s\n", r, devices[r], errs[r].c_str());
    return 2;
  }
  double slowest = 0.0;
  for (int r = 0; r < world; ++r) {
    int first, count;
    shard(n_cases, world, r, &first, &count);
    std::printf("slot %d: HIP device %d, cases [%d, %d), blocks per case %d, %.1f us\n", r, devices[r], first, first + count, blocks[r], us[r]);
    if (us[r] > slowest) slowest = us[r];
  }
  std::printf("%d cases x %d pass(es) on %d device slot(s), case batches of %d: %.1f us = %.0f solves/s (host grids in, host fields out)\n",
              n_cases, repeat, world, batch, slowest, (double)n_cases * repeat / slowest * 1e6);
  FILE* fo = std::fopen(argv[4], "wb");
  if (!fo || std::fwrite(fields.data(), sizeof(float), fields.size(), fo) != fields.size()) return 1;
  std::fclose(fo);
  return 0;
}
