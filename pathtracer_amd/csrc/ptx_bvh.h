// ptx_bvh.h -- node layout of the GPU-built bounding-volume hierarchy and its build entry point (ptx_bvh.hip).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <string>

// Nodes are stored in depth-first order.  Walking the array needs no stack: after a node whose box the ray
// enters comes the next entry; after a node it misses, or a leaf, comes entry `skip`.
struct PtxBvhNode {
    double box[6];  // xl xu yl yu zl zu, world space
    int32_t skip;   // index of the first node after this node's subtree
    int32_t obj;    // leaf: index of the primitive in the scene's object array; inner node: -1
    double pad;
};
static_assert(sizeof(PtxBvhNode) == 64, "PtxBvhNode layout");

// Builds the hierarchy over n primitives given their world boxes (host memory, 6 doubles each) and the scene
// object index of each.  n_orders is 1 (one array, left child first) or 8 (one array per octant of ray directions,
// octant o = (d.x < 0) | (d.y < 0) << 1 | (d.z < 0) << 2, nearer child first).  On success *d_nodes_out points to
// n_orders arrays of *n_nodes_out = 2n-1 nodes each, back to back in device memory (caller frees with hipFree);
// links are relative to the start of their own array.
int ptx_bvh_build(const double *h_boxes, const int *h_prim_ids, int n, int n_orders, hipStream_t stream,
                  PtxBvhNode **d_nodes_out, int *n_nodes_out, std::string &err);
