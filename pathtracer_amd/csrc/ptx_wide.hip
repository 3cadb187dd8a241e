// ptx_wide.hip -- the tracing kernel for hierarchy scenes instantiated for the four-wide hierarchy (PtxWideNode): the
// variants selected for scenes whose node arrays are far larger than the L2 caches.  A translation unit of its own so
// that it compiles beside ptx_hip.hip, which holds the variants for the binary hierarchy.
#include <hip/hip_runtime.h>

#include "ptx_trace_bvh.h"

template <int F>
static void launch_wide_flags(bool pinhole, dim3 grid, hipStream_t stream, const TraceBvhArgs &a) {
    if (pinhole) launch_trace_bvh<F, true, true>(grid, stream, a);
    else launch_trace_bvh<F, false, true>(grid, stream, a);
}

void ptx_launch_trace_bvh_wide(int flags, bool pinhole, dim3 grid, hipStream_t stream, const TraceBvhArgs &a) {
    switch (flags & 7) {
#ifndef PTX_ONLY_FLAGS7
    case 0: launch_wide_flags<0>(pinhole, grid, stream, a); break;
    case 1: launch_wide_flags<1>(pinhole, grid, stream, a); break;
    case 2: launch_wide_flags<2>(pinhole, grid, stream, a); break;
    case 3: launch_wide_flags<3>(pinhole, grid, stream, a); break;
    case 4: launch_wide_flags<4>(pinhole, grid, stream, a); break;
    case 5: launch_wide_flags<5>(pinhole, grid, stream, a); break;
    case 6: launch_wide_flags<6>(pinhole, grid, stream, a); break;
#endif
    default: launch_wide_flags<7>(pinhole, grid, stream, a); break;
    }
}

#ifdef PTX_WIDE_DIAG
extern "C" int ptx_debug_wide_diag(unsigned long long *out) {
    if (hipDeviceSynchronize() != hipSuccess) return -1;
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_wide_diag), sizeof(g_wide_diag)) != hipSuccess) return -1;
    unsigned long long zero[16] = {};
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_wide_diag), zero, sizeof(zero)) != hipSuccess) return -1;
    return 0;
}
#endif
