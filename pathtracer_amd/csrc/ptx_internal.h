// ptx_internal.h -- what the translation units of libptx_hip.so share beyond the public headers
#ifndef PTX_INTERNAL_H
#define PTX_INTERNAL_H

#include <hip/hip_runtime.h>

#include <cstddef>
#include <cstdint>
#include <string>

#include "ptx_hip.h"

// Device counters of a context: [0..3] paths, closest-hit queries, shadow queries, out-of-range refraction-stack
// accesses of the render kernels since ptx_render_begin; [PTX_STATS_WALK ..] the ptx_walk_stats fields of the last
// ptx_trace_paths call.
// [PTX_STATS_DBG ..] cycle and lane counts of the instrumented build (-DPTX_PHASE_TIMERS) since the context was created,
// read with ptx_debug_counters() (tools/phase_timers.py): of the render kernel for hierarchy scenes (22 values), or of
// k_trace_samples on a scene without a hierarchy (cycles, runs and lane visits of its twelve parts).
// [PTX_STATS_WORK] the work-tile counter of the render kernel for hierarchy scenes (its low 32 bits), cleared before every launch.
enum { PTX_STATS_WALK = 4, PTX_N_WALK = 11, PTX_STATS_DBG = PTX_STATS_WALK + PTX_N_WALK, PTX_N_DBG = 36, PTX_STATS_WORK = PTX_STATS_DBG + PTX_N_DBG,
       PTX_N_STATS = PTX_STATS_WORK + 1 };
extern "C" int ptx_debug_counters(ptx_context *c, unsigned long long *out /* PTX_N_DBG values */, int clear);

// Scratch buffers of the tone mapping, owned by a context: allocated at the first image (the byte image again when a
// larger one is asked for), released with the context.
struct ptx_tonemap_scratch {
    double *d_part = nullptr;              // block minima and maxima
    double *d_prm = nullptr;               // min, max, range, low cut, high cut, cut range
    unsigned long long *d_hist = nullptr;  // 1000 bins
    unsigned long long *d_unsure = nullptr;
    uint8_t *d_bits = nullptr;
    size_t bits_bytes = 0;
};
// Tone-maps n accumulator values on the device (d_im) into n bytes on the host; returns a PTX_* code and, on
// failure, a message in err.  Defined in ptx_tonemap.hip.
int ptx_tonemap_run(hipStream_t stream, ptx_tonemap_scratch &ws, const double *d_im, size_t n, uint8_t *h_bits, ptx_tonemap_info *info,
                    std::string &err);
void ptx_tonemap_release(ptx_tonemap_scratch &ws);

#endif
