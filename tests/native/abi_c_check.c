/* The public headers are plain C: this file is compiled with `gcc -std=c99 -pedantic -Wall -Werror` by
 * tests/test_abi.py (no linking, no GPU).  It also pins the struct layout the ctypes / cgo / JNI side relies on. */
#include <stddef.h>

#include "psm.h"
#include "psm_unet.h"

typedef char psm_config_is_15_int32[(sizeof(psm_config) == 15 * sizeof(int32_t)) ? 1 : -1];
typedef char psm_config_precision_is_last[(offsetof(psm_config, precision) == 14 * sizeof(int32_t)) ? 1 : -1];

int use_every_entry_point(psm_handle* h, psm_unet* u) {
  int (*f1)(const psm_config*, psm_handle**) = psm_create;
  int (*f2)(psm_handle*, const double*, int64_t, int32_t, double*) = psm_solve;
  int (*f3)(psm_handle*, const float*, int32_t, const float*, int64_t*) = psm_submit_grid;
  int (*f4)(psm_unet*, const float*, int32_t, float*) = psm_unet_forward;
  (void)h; (void)u;
  return (f1 != 0) + (f2 != 0) + (f3 != 0) + (f4 != 0) + PSM_RING_SLOTS + PSM_ABI_VERSION;
}
