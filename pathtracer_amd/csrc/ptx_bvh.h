// ptx_bvh.h -- node layout of the GPU-built bounding-volume hierarchy and its build entry point (ptx_bvh.hip).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <string>

// Nodes are stored in depth-first order.  Walking the array needs no stack: after a node whose box the ray
// enters comes the next entry; after a node it misses, or a leaf, comes entry `skip`.
// The box is held in single precision, every bound moved outward by a millionth of the scene's size and then
// rounded outward, so it contains the double-precision box of the scene description with room to spare: a ray
// that enters the latter enters the former however the walk rounds its slab test, and the primitive test, not
// the box, decides about a hit.  32 bytes a node: two 16-byte loads per step, and the eight
// arrays of a scene of 8000 primitives (4 MB) stay within one L2.
struct PtxBvhNode {
    float box[6];  // xl xu yl yu zl zu, world space; in the array of an octant of ray directions (eight-array build) the two
                   // bounds of an axis are stored in the order a ray of that octant reaches them: the upper one first
                   // on an axis along which the octant's rays travel downward
    int32_t skip;  // index of the first node after this node's subtree
    int32_t obj;   // leaf: what the caller gave for the primitive (see PTX_LEAF_*); inner node: -1
};
// What the path tracer stores for a leaf: the primitive's index in the scene's object array, whether it is a
// uniformly scaled sphere, and its type, so that a ray can pick the intersection routine before the record of the
// primitive has arrived.
#define PTX_LEAF_INDEX 0x07ffffff
#define PTX_LEAF_FAST 0x08000000
#define PTX_LEAF_TYPE_SHIFT 28
static_assert(sizeof(PtxBvhNode) == 32, "PtxBvhNode layout");

// Builds the hierarchy over n primitives given their world boxes (host memory, 6 doubles each) and the scene
// leaf value of each (non-negative).  n_orders is 1 (one array, left child first) or 8 (one array per octant of ray directions,
// octant o = (d.x < 0) | (d.y < 0) << 1 | (d.z < 0) << 2, nearer child first).  On success *d_nodes_out points to
// n_orders arrays of *n_nodes_out = 2n-1 nodes each, back to back in device memory (caller frees with hipFree);
// links are relative to the start of their own array.
// use_sah: topology by the surface-area heuristic, built on the host, instead of the Morton-code build on the GPU.
// *coord_bound (if not null): no bound of any node exceeds it in magnitude.
int ptx_bvh_build(const double *h_boxes, const int *h_prim_ids, int n, int n_orders, bool use_sah, hipStream_t stream,
                  PtxBvhNode **d_nodes_out, int *n_nodes_out, float *coord_bound, std::string &err);
