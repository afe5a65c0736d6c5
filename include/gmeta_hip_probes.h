/* Probe entry points of libgmeta_hip_probes.so -- the SAME sources as libgmeta_hip.so compiled with -DGM_PROBES
 * (`python g-meta_amd/build.py --probes`).  Nothing a caller of the hot path needs; the product library (include/gmeta_hip.h) exports none of
 * them and carries no probe state.  Users: tools/coreside_probe.py, tools/head_loss_probe.py (they build the probe library themselves and
 * load it through GMETA_HIP_LIB).  The probe build also reads two launch-time experiment variables, GM_AGG_STREAM_PRIO (s_setprio of the stream
 * aggregate's waves) and GM_HEAD_TWICE (every k_head_loss launch issued twice -- it is idempotent -- to see a warm relaunch). */
#ifndef GMETA_HIP_PROBES_H
#define GMETA_HIP_PROBES_H
#include "gmeta_hip.h"
#ifdef __cplusplus
extern "C" {
#endif
/* A one-thread kernel on `stream` writes the device's 100 MHz constant clock to *out (device uint64) -- stamps taken on different streams are
 * comparable, which HIP event times are not. */
int gm_debug_stamp(void* out, void* stream);
/* Stream aggregate (agg_stream.hip): enable != 0 makes every later launch stamp the start / end clock of each of its workgroups (2 x uint64 per
 * workgroup) into a device buffer of n entries; out != NULL copies that buffer to the host; enable == 0 releases it. */
int gm_stream_debug(int32_t enable, unsigned long long* out, int32_t n);
/* k_head_loss: 64 stamps of block 0 of the LAST launch (phases 0-6, shader cycles in 7, per-wave start / end of the logits phase in 8-23 / 24-39). */
int gm_head_loss_debug(int32_t enable, unsigned long long* out);
#ifdef __cplusplus
}
#endif
#endif /* GMETA_HIP_PROBES_H */
