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

int vd3d_current_device() {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kVd3dMaxDevices) {
        vd3d_set_error("hipGetDevice failed (or device ordinal >= 64)");
        return -1;
    }
    return dev;
}

int vd3d_device_cu_count() {
    static int cus[kVd3dMaxDevices] = {};
    const int dev = vd3d_current_device();
    if (dev < 0) return -1;
    if (!cus[dev]) {
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, dev) != hipSuccess) { vd3d_check_launch("hipGetDeviceProperties"); return -1; }
        cus[dev] = prop.multiProcessorCount;
    }
    return cus[dev];
}

int vd3d_raise_lds_limit(const void* kern, int bytes, Vd3dLdsLimit& state, const char* what) {
    const int dev = vd3d_current_device();
    if (dev < 0) return VD3D_ELAUNCH;
    if (bytes > state.bytes[dev]) {
        if (hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, bytes) != hipSuccess) return vd3d_check_launch(what);
        state.bytes[dev] = bytes;
    }
    return VD3D_OK;
}

extern "C" int vd3d_abi_version(void) { return 2; }
extern "C" const char* vd3d_last_error(void) { return g_err; }
