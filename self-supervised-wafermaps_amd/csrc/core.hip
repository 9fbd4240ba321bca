// Library identity + error strings for the wafer_hip C ABI.
#include "common.h"

extern "C" int wm_version(void) { return WM_ABI_VERSION; }

extern "C" const char* wm_error_string(int code) {
  switch (code) {
    case WM_OK: return "ok";
    case WM_EINVAL: return "invalid argument (null pointer or non-positive size)";
    case WM_EUNSUPPORTED: return "unsupported shape/dtype combination";
    case WM_EWORKSPACE: return "workspace too small";
    case WM_EALIGN: return "pointer or pitch alignment requirement not met";
    default: break;
  }
  if (code > 0) return hipGetErrorString(static_cast<hipError_t>(code));
  return "unknown wafer_hip error";
}
