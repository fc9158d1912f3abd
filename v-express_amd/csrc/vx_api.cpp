// Error plumbing and device query of libvexpress_hip.so (see include/vexpress_hip.h).
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdio.h>

#include "../../include/vexpress_hip.h"

static thread_local char g_err[512] = "";

extern "C" void vx_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

int vx_check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    vx_set_error("%s: launch failed: %s", what, hipGetErrorString(e));
    return -3;
  }
  return 0;
}

extern "C" const char* vx_last_error_string(void) { return g_err; }
extern "C" int vx_abi_version(void) { return VX_ABI_VERSION; }

// Identity of THIS binary: the Makefile stamps the hash of the kernel sources it compiled (tools/lib_id.py =
// v_express_amd.lib.source_id) and the extra -D flags of the build into this translation unit, which depends on every
// source; a library built any other way (tools/build_*_variants.sh) says "unstamped".  lib.py compares it with the
// sources on disk, so a stale or variant .so can never be paired with rocprofv3 / PMC files of another build.
#ifndef VX_BUILD_SRC_ID
#define VX_BUILD_SRC_ID "unstamped"
#endif
#ifndef VX_BUILD_DEFS
#define VX_BUILD_DEFS ""
#endif
extern "C" const char* vx_build_id(void) { return VX_BUILD_SRC_ID "|" VX_BUILD_DEFS; }

// the 16-bit element type this binary computes on (vx_common.h: one set of sources, two libraries)
#ifdef VX_ELEM_F16
extern "C" const char* vx_element_type(void) { return "f16"; }
#else
extern "C" const char* vx_element_type(void) { return "bf16"; }
#endif

extern "C" int vx_device_info(int device, int* out4) {
  hipDeviceProp_t prop;
  hipError_t e = hipGetDeviceProperties(&prop, device);
  if (e != hipSuccess) {
    vx_set_error("vx_device_info: %s", hipGetErrorString(e));
    return -3;
  }
  out4[0] = prop.multiProcessorCount;
  out4[1] = (int)prop.sharedMemPerBlock;
  out4[2] = prop.warpSize;
  out4[3] = prop.clockRate;
  return 0;
}
