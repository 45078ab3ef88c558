// Library identification and status strings of libanirec (host only).
#include <hip/hip_runtime.h>
#include <string.h>

#include "../../include/anirec.h"

extern "C" {

int anirec_abi_version(void) { return ANIREC_ABI_VERSION; }

const char *anirec_status_string(int status) {
  switch (status) {
    case ANIREC_OK:
      return "ok";
    case ANIREC_EINVAL:
      return "invalid argument";
    case ANIREC_ENODEVICE:
      return "no gfx950 HIP device (or RCCL not loadable)";
    case ANIREC_EWORKSPACE:
      return "workspace too small";
    case ANIREC_ECAPTURE:
      return "hipGraph capture/instantiate failed";
    case ANIREC_ECOMM:
      return "RCCL call failed";
    default:
      break;
  }
  if (status > 0) return hipGetErrorString((hipError_t)status);
  return "unknown anirec status";
}

int anirec_device_name(char *buf, size_t len) {
  if (!buf || len == 0) return ANIREC_EINVAL;
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n < 1) return ANIREC_ENODEVICE;
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return (int)e;
  hipDeviceProp_t p;
  e = hipGetDeviceProperties(&p, dev);
  if (e != hipSuccess) return (int)e;
  snprintf(buf, len, "%s (%s)", p.name, p.gcnArchName);
  return ANIREC_OK;
}

}  // extern "C"
