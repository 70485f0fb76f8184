// Device buffers that another process on this machine can map (hipIpc): the carrier of the
// device-to-device KV hand-over between a prefill engine and a decode engine.  The owner allocates a
// buffer and publishes its 64-byte handle; a peer process opens the handle, copies out of the mapping
// with ordinary device copies (over xGMI when the two sit on different GPUs), and closes it.
#include <string.h>

#include "common.h"

VDAMD_API int vdamd_ipc_handle_bytes() { return (int)sizeof(hipIpcMemHandle_t); }

// Allocates `bytes` of device memory on the current device and writes its IPC handle; nullptr on failure.
VDAMD_API void* vdamd_ipc_alloc(size_t bytes, void* handle_out) {
  void* p = nullptr;
  if (bytes == 0 || handle_out == nullptr) return nullptr;
  if (hipMalloc(&p, bytes) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
  if (hipIpcGetMemHandle((hipIpcMemHandle_t*)handle_out, p) != hipSuccess) {
    (void)hipGetLastError();
    (void)hipFree(p);
    return nullptr;
  }
  return p;
}

VDAMD_API int vdamd_ipc_free(void* p) { return p == nullptr ? 0 : (int)hipFree(p); }

// Maps a buffer another process exported; *err receives the HIP status when the result is nullptr.
VDAMD_API void* vdamd_ipc_open(const void* handle, int* err) {
  void* p = nullptr;
  hipIpcMemHandle_t h;
  memcpy(&h, handle, sizeof(h));
  hipError_t e = hipIpcOpenMemHandle(&p, h, hipIpcMemLazyEnablePeerAccess);
  if (e != hipSuccess) (void)hipGetLastError();
  if (err != nullptr) *err = (int)e;
  return e == hipSuccess ? p : nullptr;
}

VDAMD_API int vdamd_ipc_close(void* p) { return p == nullptr ? 0 : (int)hipIpcCloseMemHandle(p); }
