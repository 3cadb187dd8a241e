// ptx_internal.h -- what the translation units of libptx_hip.so share beyond the public headers
#ifndef PTX_INTERNAL_H
#define PTX_INTERNAL_H

#include <hip/hip_runtime.h>

#include <cstddef>
#include <cstdint>
#include <string>

#include "ptx_hip.h"

// Tone-maps n accumulator values on the device (d_im) into n bytes on the host; returns a PTX_* code and, on
// failure, a message in err.  Defined in ptx_tonemap.hip.
int ptx_tonemap_run(hipStream_t stream, const double *d_im, size_t n, uint8_t *h_bits, ptx_tonemap_info *info, std::string &err);

#endif
