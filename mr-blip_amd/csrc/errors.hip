// Error plumbing for the C ABI: every entry returns 0 or a negative code; the message is thread-local.
#include "common.h"
#include <stdarg.h>
#include <stdio.h>

static thread_local char g_err[512] = "";

void mrblip_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

int mrblip_check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    mrblip_set_error("%s: launch failed: %s", what, hipGetErrorString(e));
    return MRBLIP_ELAUNCH;
  }
  return MRBLIP_OK;
}

static thread_local char* g_reduce_ws = nullptr;
char* mrblip_reduce_workspace() { return g_reduce_ws; }
extern "C" int mrblip_set_reduce_workspace(void* ws, long long bytes) {
  MRB_REQUIRE(!ws || (bytes >= MRB_RWS_BYTES && ((uintptr_t)ws % 16) == 0), "reduce workspace: need >= %lld bytes of 16-B aligned, ZEROED device memory", (long long)MRB_RWS_BYTES);
  g_reduce_ws = (char*)ws;
  return MRBLIP_OK;
}
extern "C" long long mrblip_reduce_workspace_bytes(void) { return MRB_RWS_BYTES; }

extern "C" const char* mrblip_last_error(void) { return g_err; }
extern "C" int mrblip_abi_version(void) { return 1; }
