// common.hip -- error plumbing + ABI version for libvd3d_hip.so
#include "common.h"
#include <stdio.h>
#include <string.h>

static thread_local char g_err[512] = "";

void vd3d_set_error(const char* msg) {
    strncpy(g_err, msg, sizeof(g_err) - 1);
    g_err[sizeof(g_err) - 1] = 0;
}

int vd3d_check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) return VD3D_OK;
    snprintf(g_err, sizeof(g_err), "%s: %s", what, hipGetErrorString(e));
    return VD3D_ELAUNCH;
}

extern "C" int vd3d_abi_version(void) { return 1; }
extern "C" const char* vd3d_last_error(void) { return g_err; }
