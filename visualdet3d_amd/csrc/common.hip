// common.hip -- error plumbing + ABI version for libvd3d_hip.so
#include "common.h"
#include <stdio.h>
#include <string.h>
#include <stdlib.h>
#include <atomic>
#include <mutex>
#include "test_hooks.h"

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

static const char* const kSwitchNames[VD3D_SW_COUNT] = {
    "VD3D_CONV_DEBUG", "VD3D_FORCE_GROUP_M", "VD3D_NO_GROUP_M", "VD3D_NO_LINE_STORE", "VD3D_DCN_GENERIC",
    "VD3D_DCN_COLUMNS_GENERIC", "VD3D_CONV3D_VALU", "VD3D_PSM_VALU", "VD3D_NO_NARROW", "VD3D_DWCONVT_GENERIC", "VD3D_HEAD_PARKED", "VD3D_STEM_WG4", "VD3D_HEAD_NO_STAGGER", "VD3D_CONV_NO_STAGGER", "VD3D_DCN_NO_LSTAGE", "VD3D_NO_STRIP_SPLIT", "VD3D_PLAIN_TILE_WALK", "VD3D_DWCONV_PLAIN"};
static std::atomic<int> g_switch[VD3D_SW_COUNT];
static std::once_flag g_switch_once;

static void switches_from_env() {
    for (int i = 0; i < VD3D_SW_COUNT; ++i) g_switch[i].store(getenv(kSwitchNames[i]) != nullptr, std::memory_order_relaxed);
}

bool vd3d_switch(Vd3dSwitch s) {
    std::call_once(g_switch_once, switches_from_env);
    return g_switch[s].load(std::memory_order_relaxed) != 0;
}

extern "C" int vd3d_test_set_switch(const char* name, int on) {
    std::call_once(g_switch_once, switches_from_env);
    for (int i = 0; i < VD3D_SW_COUNT; ++i)
        if (name && !strcmp(name, kSwitchNames[i])) { g_switch[i].store(on != 0, std::memory_order_relaxed); return VD3D_OK; }
    vd3d_set_error("vd3d_test_set_switch: unknown switch");
    return VD3D_EINVAL;
}

extern "C" int vd3d_test_get_switch(const char* name) {
    std::call_once(g_switch_once, switches_from_env);
    for (int i = 0; i < VD3D_SW_COUNT; ++i)
        if (name && !strcmp(name, kSwitchNames[i])) return g_switch[i].load(std::memory_order_relaxed) != 0;
    vd3d_set_error("vd3d_test_get_switch: unknown switch");
    return -1;
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

extern "C" int vd3d_abi_version(void) { return 6; }
extern "C" const char* vd3d_last_error(void) { return g_err; }
#ifndef VD3D_SRC_HASH
#define VD3D_SRC_HASH "unstamped"
#endif
extern "C" const char* vd3d_source_hash(void) { return VD3D_SRC_HASH; }
