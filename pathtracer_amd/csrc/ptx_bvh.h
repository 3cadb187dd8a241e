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

// The same hierarchy four children to a node, for scenes whose node arrays are far larger than the L2 caches: there a
// step of the walk waits for device memory, and what it waits for is a whole 128-byte line whatever it uses of it.  A
// wide node is one such line: the boxes of the four grandchildren of a node of the binary tree (of a child itself where
// that child is a leaf: its second slot is then empty) and what lies below each.  The walk keeps the children it has
// yet to visit on a short stack per lane, so one array serves all ray directions; half as many dependent fetches per
// ray as the binary walk, every fetched byte used.
struct PtxWideNode {
    float lo_x[4], hi_x[4], lo_y[4], hi_y[4], lo_z[4], hi_z[4];  // child boxes, rounded outward like PtxBvhNode::box; an empty
                                                                // slot has lo = +infinity, hi = -infinity (never entered)
    int32_t child[4];  // >= 0: a leaf, the caller's value for the primitive (PTX_LEAF_*); < 0: ~index of a wide node
    uint8_t order[8];  // per octant of ray directions: the four slots, two bits each from bit 0, in the order in which a
                       // ray of that octant should visit them (nearer first, by the split axes of the binary nodes)
    int32_t pad[2];
};
static_assert(sizeof(PtxWideNode) == 128, "PtxWideNode layout");

struct PtxBvhBuilt {
    PtxBvhNode *d_nodes = nullptr;  // n_orders arrays of n_nodes = 2n-1 nodes each, back to back (caller frees with hipFree)
    int n_nodes = 0;
    float coord_bound = 0;          // no bound of any node exceeds this in magnitude
    PtxWideNode *d_wide = nullptr;  // n_wide wide nodes, the root first (caller frees), or null
    int n_wide = 0;
    int wide_depth = 0;             // wide nodes on the longest path from the root to a primitive
};

// Builds the hierarchy over n primitives given their world boxes (host memory, 6 doubles each) and the scene
// leaf value of each (non-negative).  n_orders is 1 (one array, left child first) or 8 (one array per octant of ray directions,
// octant o = (d.x < 0) | (d.y < 0) << 1 | (d.z < 0) << 2, nearer child first).  On success out->d_nodes points to
// n_orders arrays of out->n_nodes = 2n-1 nodes each, back to back in device memory;
// links are relative to the start of their own array.
// use_sah: topology by the surface-area heuristic, built on the host, instead of the Morton-code build on the GPU.
// want_wide: also emit the four-wide form of the same tree (PtxWideNode).
int ptx_bvh_build(const double *h_boxes, const int *h_prim_ids, int n, int n_orders, bool use_sah, bool want_wide, hipStream_t stream,
                  PtxBvhBuilt *out, std::string &err);
