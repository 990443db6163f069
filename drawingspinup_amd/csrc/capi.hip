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

}  // extern "C"
