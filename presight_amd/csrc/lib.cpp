// libpresight_hip.so: error reporting + version of the C ABI declared in include/presight_hip.h
#include <string.h>
#include <hip/hip_runtime.h>
#include "common.hpp"

static thread_local char g_err[512] = "";

void ps_set_error(const char* msg) {
  strncpy(g_err, msg ? msg : "", sizeof(g_err) - 1);
  g_err[sizeof(g_err) - 1] = 0;
}

extern "C" const char* ps_last_error(void) { return g_err; }
extern "C" int ps_abi_version(void) { return 1; }

extern "C" int ps_device_info(int* cu_count, int* wave_size, char* arch, int arch_len) {
  hipDeviceProp_t p;
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e == hipSuccess) e = hipGetDeviceProperties(&p, dev);
  if (e != hipSuccess) {
    ps_set_error(hipGetErrorString(e));
    return (int)e;
  }
  if (cu_count) *cu_count = p.multiProcessorCount;
  if (wave_size) *wave_size = p.warpSize;
  if (arch && arch_len > 0) {
    strncpy(arch, p.gcnArchName, arch_len - 1);
    arch[arch_len - 1] = 0;
  }
  return 0;
}
