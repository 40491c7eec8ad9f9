// A stand-in for DLPoissonFoam's Python side: the CPython C-API call sequence of the reference solver, restated call for call,
// against whatever `python_module.py` sits in the working directory (tests/test_embed_host.py puts the drop-in shim there).
//
//   start-up   Thesis_Work/Chapter5/parallelized/DLPoissonSolver/PythonComm_init.H:3-21   Py_Initialize, sys.path.append("."),
//              import_array1, PyImport_Import("python_module"), GetAttrString py_func / init_func, the two argument tuples
//   geometry   PythonComm_init.H:53-55, 80-94   new double[N][5] (+ top, obstacle) wrapped WITHOUT a copy by
//              PyArray_SimpleNewFromData, stolen into init_args (rank_val too), init_func called, its result dropped
//   each step  PythonComm.H:3-36   the SAME input_vals rewritten in place, a fresh ndarray view over it stolen into slot 0 of
//              the ONE py_args tuple (PyTuple_SetItem releases the previous step's view), rank_val stolen AGAIN into slot 1 (the
//              solver never owned a second reference: every step takes one off the small-int object), py_func called, the
//              result read with PyArray_GETPTR2(pValue, id, 0) and never released
//   serial     singleCore/DLPoissonSolver_1/PythonComm_init.H:16,19: py_args of 1, init_args of 3, no rank_val
//
// Differences from the solver, all outside the sequence above: the mesh comes from files instead of an fvMesh; U of step s is
// the start field times (1 + 0.01 (s % 7)) and column 4 is the pressure the previous step returned (PythonComm.H:9, 34); a
// failed import / missing attribute ends with exit code 3 and the Python traceback where the solver dereferences NULL
// (test_case/log.DL:34-42: SIGSEGV inside PyObject_GetAttrString); the reference counts the shim's persistent arrays are
// printed so the test can see that nothing accumulates.
//
// usage: embed_host serial|parallel STEPS cells.f64 top.f64 obst.f64 out.f64 [RANK]      (run from the case directory; RANK = what
//        Pstream::myProcNo() would return in this solver process, default 0)
#define NPY_NO_DEPRECATED_API NPY_1_7_API_VERSION
#include <Python.h>
#include <numpy/arrayobject.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

static std::vector<double> read_f64(const char* path, long cols, long* rows) {
  FILE* f = std::fopen(path, "rb");
  if (!f) { std::fprintf(stderr, "embed_host: cannot open %s\n", path); std::exit(2); }
  std::fseek(f, 0, SEEK_END);
  long bytes = std::ftell(f);
  std::fseek(f, 0, SEEK_SET);
  *rows = bytes / (long)(sizeof(double) * cols);
  std::vector<double> v((size_t)*rows * cols);
  if (std::fread(v.data(), sizeof(double), v.size(), f) != v.size()) { std::fprintf(stderr, "embed_host: short read %s\n", path); std::exit(2); }
  std::fclose(f);
  return v;
}

// refcount of python_module.<dict_name>[key] (or -1 when absent / None): borrowed lookups only, nothing is retained
static long shim_refcnt(const char* dict_name, const char* key) {
  PyObject* mods = PyImport_GetModuleDict();
  PyObject* mod = PyDict_GetItemString(mods, "python_module");
  if (!mod) return -1;
  PyObject* d = PyObject_GetAttrString(mod, dict_name);
  if (!d) { PyErr_Clear(); return -1; }
  PyObject* o = PyDict_Check(d) ? PyDict_GetItemString(d, key) : nullptr;
  long r = (o && o != Py_None) ? (long)Py_REFCNT(o) : -1;
  Py_DECREF(d);
  return r;
}

static int run(int argc, char** argv) {
  if (argc < 7) { std::fprintf(stderr, "usage: embed_host serial|parallel STEPS cells top obst out\n"); return 2; }
  const bool parallel = std::strcmp(argv[1], "parallel") == 0;
  const int steps = std::atoi(argv[2]);

  Py_Initialize();                                                 // PythonComm_init.H:3
  PyRun_SimpleString("import sys");                                // :4
  PyRun_SimpleString("sys.path.append(\".\")");                    // :5
  import_array1(-1);                                               // :8

  PyObject* pName = PyUnicode_DecodeFSDefault("python_module");    // :11
  PyObject* pModule = PyImport_Import(pName);                      // :12
  Py_DECREF(pName);                                                // :13
  if (!pModule) {                                                  // the solver goes on and crashes in :15 (log.DL:34-42)
    PyErr_Print();
    std::fprintf(stderr, "embed_host: import python_module failed\n");
    return 3;
  }
  PyObject* py_func = PyObject_GetAttrString(pModule, "py_func");  // :15
  PyObject* py_args = PyTuple_New(parallel ? 2 : 1);               // :16 (serial: singleCore :16)
  PyObject* init_func = PyObject_GetAttrString(pModule, "init_func");   // :18
  PyObject* init_args = PyTuple_New(parallel ? 4 : 3);             // :19 (serial: singleCore :19)
  Py_DECREF(pModule);                                              // :21
  if (!py_func || !init_func) { PyErr_Print(); std::fprintf(stderr, "embed_host: python_module lacks py_func / init_func\n"); return 3; }

  PyObject* rank_val = PyLong_FromLong(argc > 7 ? std::atol(argv[7]) : 0);   // :28 (Pstream::myProcNo())

  long n_cells = 0, n_top = 0, n_obst = 0;
  std::vector<double> cells0 = read_f64(argv[3], 5, &n_cells), top0 = read_f64(argv[4], 2, &n_top), obst0 = read_f64(argv[5], 2, &n_obst);
  const int col = 5;
  double(*input_vals)[col]{new double[n_cells][col]};              // :53  one allocation for the whole run, never freed
  double(*input_vals_top)[2]{new double[n_top][2]};                // :54
  double(*input_vals_obst)[2]{new double[n_obst][2]};              // :55
  std::memcpy(input_vals, cells0.data(), cells0.size() * sizeof(double));          // :58-65
  std::memcpy(input_vals_obst, obst0.data(), obst0.size() * sizeof(double));       // :67-71
  std::memcpy(input_vals_top, top0.data(), top0.size() * sizeof(double));          // :73-77

  npy_intp dim[] = {n_cells, 5};                                   // :80-82
  npy_intp dim_top[] = {n_top, 2};
  npy_intp dim_obstacle[] = {n_obst, 2};
  PyObject* array_2d = PyArray_SimpleNewFromData(2, dim, NPY_DOUBLE, reinterpret_cast<void*>(input_vals));            // :85
  PyObject* array_2d_top = PyArray_SimpleNewFromData(2, dim_top, NPY_DOUBLE, reinterpret_cast<void*>(input_vals_top)); // :86
  PyObject* array_2d_obst = PyArray_SimpleNewFromData(2, dim_obstacle, NPY_DOUBLE, reinterpret_cast<void*>(input_vals_obst));   // :87
  PyTuple_SetItem(init_args, 0, array_2d);                         // :89
  PyTuple_SetItem(init_args, 1, array_2d_top);                     // :90
  PyTuple_SetItem(init_args, 2, array_2d_obst);                    // :91
  if (parallel) PyTuple_SetItem(init_args, 3, rank_val);           // :92  (the tuple now owns the solver's only reference)
  PyObject* init_ret = PyObject_CallObject(init_func, init_args);  // :94  "(void)": the result is dropped
  if (!init_ret) { PyErr_Print(); std::fprintf(stderr, "embed_host: init_func raised\n"); return 4; }

  FILE* out = std::fopen(argv[6], "wb");
  if (!out) { std::fprintf(stderr, "embed_host: cannot write %s\n", argv[6]); return 2; }
  std::vector<double> p((size_t)n_cells);
  for (long id = 0; id < n_cells; ++id) p[id] = cells0[id * 5 + 4];

  for (int s = 0; s < steps; ++s) {
    const double scale = 1.0 + 0.01 * (double)(s % 7);
    for (long id = 0; id < n_cells; ++id) {                        // PythonComm.H:3-10
      input_vals[id][0] = cells0[id * 5 + 0] * scale;
      input_vals[id][1] = cells0[id * 5 + 1] * scale;
      input_vals[id][2] = cells0[id * 5 + 2];
      input_vals[id][3] = cells0[id * 5 + 3];
      input_vals[id][4] = p[id];                                   // the pressure of the previous step
    }
    array_2d = PyArray_SimpleNewFromData(2, dim, NPY_DOUBLE, &input_vals[0]);       // :17  a new view over the same memory
    PyTuple_SetItem(py_args, 0, array_2d);                         // :19  releases last step's view
    if (parallel) PyTuple_SetItem(py_args, 1, rank_val);           // :20  re-stolen every step
    PyArrayObject* pValue = reinterpret_cast<PyArrayObject*>(PyObject_CallObject(py_func, py_args));                   // :24-27
    if (!pValue) { PyErr_Print(); std::fprintf(stderr, "embed_host: py_func raised at step %d\n", s); return 4; }
    if (!PyArray_Check((PyObject*)pValue) || PyArray_TYPE(pValue) != NPY_DOUBLE || PyArray_NDIM(pValue) < 1 || PyArray_DIM(pValue, 0) != n_cells) {
      std::fprintf(stderr, "embed_host: py_func returned something PythonComm.H:34 cannot read at step %d\n", s);
      return 5;
    }
    for (long id = 0; id < n_cells; ++id)                          // :31-36  (pValue is never released by the solver)
      p[id] = *((double*)PyArray_GETPTR2(pValue, id, 0));
    std::fwrite(p.data(), sizeof(double), p.size(), out);
    if (s == 2 || s == steps - 1)
      std::printf("REFCNT step %d pin_array %ld pin_out %ld cat_buf %ld cat_out %ld rank_val %ld py_args %ld\n", s, shim_refcnt("_pin_state", "array"),
                  shim_refcnt("_pin_state", "out"), shim_refcnt("_cat", "buf"), shim_refcnt("_cat", "out"), (long)Py_REFCNT(rank_val), (long)Py_REFCNT(py_args));
  }
  std::fclose(out);
  std::printf("embed_host: %d steps, %ld cells, %s binding OK\n", steps, n_cells, parallel ? "parallel (4 / 2 arguments)" : "serial (3 / 1 arguments)");
  std::fflush(stdout);
  // The solver never calls Py_Finalize (DLPoissonFoam.C ends with "End"): neither does this host.
  return 0;
}

int main(int argc, char** argv) { return run(argc, argv); }
