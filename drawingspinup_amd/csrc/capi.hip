// Error strings / ABI version of libdsu_hip.so.
#include "common.h"

extern "C" {

const char* dsu_strerror(int code) {
  switch (code) {
    case DSU_OK: return "ok";
    case DSU_EINVAL: return "invalid argument";
    case DSU_ELAUNCH: return "HIP launch/runtime error";
    case DSU_EUNSUP: return "configuration not supported by the gfx950 kernels";
    default: return "unknown dsu error";
  }
}

int dsu_abi_version(void) { return 1; }

int32_t dsu_onewave_grid_cap_value = 0;     // 0 = one workgroup per CU (256)
int32_t dsu_scatter_grid_cap_value = 0;     // 0 = the resident count (three 256-thread workgroups per CU: 768)
int dsu_set_scatter_grid_cap(int32_t workgroups) {
  if (workgroups < 0 || workgroups > 4096) return DSU_EINVAL;
  dsu_scatter_grid_cap_value = workgroups;
  return DSU_OK;
}
int32_t dsu_nsr_side_priority_value = DSU_NSR_SIDE_PRIO_DEFAULT;   // 1 high, 2 normal, 0 low (nsr_driver.hip)
int dsu_set_nsr_side_stream_priority(int32_t level) {
  if (level < 0 || level > 2) return DSU_EINVAL;
  dsu_nsr_side_priority_value = level;
  return DSU_OK;
}
int32_t dsu_nsr_side_pool_value = 0;   // 1: side streams handed from one step driver to the next (nsr_driver.hip)
int dsu_set_nsr_side_stream_pooling(int32_t on) {
  if (on < 0 || on > 1) return DSU_EINVAL;
  dsu_nsr_side_pool_value = on;
  return DSU_OK;
}
int dsu_set_onewave_grid_cap(int32_t workgroups) {
  if (workgroups < 0 || workgroups > 256) return DSU_EINVAL;
  dsu_onewave_grid_cap_value = workgroups;
  return DSU_OK;
}

// 1 in a variant build with the A/B environment switches compiled in (-DDSU_AB_SWITCHES), 0 in the
// product library
int dsu_ab_switches(void) {
#ifdef DSU_AB_SWITCHES
  return 1;
#else
  return 0;
#endif
}

}  // extern "C"
