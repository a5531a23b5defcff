// test_hooks.h -- TEST hooks exported by libvd3d_hip.so.  NOT part of the drop-in C-ABI (include/vd3d.h does not declare them and a
// reference-side binding never calls them); tests/ and tools/ reach them through visualdet3d_amd/_lib.py `TEST_HOOKS`.
#pragma once
#ifdef __cplusplus
extern "C" {
#endif
/* Force a conv tile configuration (0 = built-in heuristic) for the following vd3d_conv2d_igemm calls OF THE CALLING THREAD
 * (thread-local).  The product library only accepts the ids vd3d_conv2d_production_tiles lists; a forced tile that cannot run a
 * given convolution makes that call return VD3D_EINVAL -- never a silent fallback. */
int vd3d_test_force_conv_tile(int cfg);
/* Override one of the A/B environment switches of DESIGN 3.4 ("VD3D_NO_LINE_STORE", ...), which the library otherwise reads
 * once per process.  Both settings of every switch are correct implementations. */
int vd3d_test_set_switch(const char* name, int on);
/* Current state of a switch (0 | 1), -1 for an unknown name: lets a test restore what it found (a switch may have been enabled
 * through the environment for the whole run). */
int vd3d_test_get_switch(const char* name);
#ifdef __cplusplus
}
#endif
