// ptx_bvh.hip -- bounding-volume hierarchy over a scene's small primitives, built on the GPU.
//
// The reference accelerates findFirstHit with its BoxTree (magic.c:142-247): a CPU-built binary tree that keeps
// objects in inner nodes and is walked depth first with no distance culling (PathTracer.c:198-251).  The closest
// hit does not depend on the structure used to find it, so the GPU builds its own: a linear BVH (Karras 2012)
// over the world boxes the scene description carries for every object --
//   1. k_morton     63-bit Morton code of each box centre inside the bounds of all centres
//   2. radix sort   (rocPRIM) of (code, primitive) pairs
//   3. k_hierarchy  one thread per inner node finds its range and split from the sorted codes
//   4. k_refit      boxes bottom-up, the second thread to reach a node continues upward
//   5. k_emit       nodes written in depth-first order with a "skip" link each
// The emitted array is walked without a stack: a node that is hit is followed by the next array entry, a node
// that is missed (or a leaf that has been tested) by its skip link.
#include <cstring>

#include <hip/hip_runtime.h>
#include <rocprim/rocprim.hpp>

#include <string>
#include <vector>

#include "ptx_bvh.h"

namespace {

__device__ inline unsigned long long spread21(unsigned long long x) {  // 21 bits -> every third bit
    x &= 0x1fffffULL;
    x = (x | x << 32) & 0x1f00000000ffffULL;
    x = (x | x << 16) & 0x1f0000ff0000ffULL;
    x = (x | x << 8) & 0x100f00f00f00f00fULL;
    x = (x | x << 4) & 0x10c30c30c30c30c3ULL;
    x = (x | x << 2) & 0x1249249249249249ULL;
    return x;
}

__global__ void k_morton(const double *__restrict__ boxes, int n, double lox, double loy, double loz, double sx, double sy,
                         double sz, unsigned long long *__restrict__ codes, int *__restrict__ ids) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double *b = boxes + 6 * (size_t)i;
    double cx = 0.5 * (b[0] + b[1]), cy = 0.5 * (b[2] + b[3]), cz = 0.5 * (b[4] + b[5]);
    double fx = fmin(fmax((cx - lox) * sx, 0.0), 2097151.0);
    double fy = fmin(fmax((cy - loy) * sy, 0.0), 2097151.0);
    double fz = fmin(fmax((cz - loz) * sz, 0.0), 2097151.0);
    codes[i] = spread21((unsigned long long)fx) << 2 | spread21((unsigned long long)fy) << 1 | spread21((unsigned long long)fz);
    ids[i] = i;
}

// length of the common prefix of the keys at sorted positions i and j; equal codes are told apart by position
__device__ inline int prefix_len(const unsigned long long *codes, int n, int i, int j) {
    if (j < 0 || j >= n) return -1;
    unsigned long long a = codes[i], b = codes[j];
    if (a != b) return __clzll((long long)(a ^ b));
    return 64 + __clz(i ^ j);
}

// children: >= 0 inner node, < 0 leaf ~child
__global__ void k_hierarchy(const unsigned long long *__restrict__ codes, int n, int *__restrict__ left, int *__restrict__ right,
                            int *__restrict__ parent_inner, int *__restrict__ parent_leaf, int *__restrict__ first,
                            int *__restrict__ last) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n - 1) return;
    int d = prefix_len(codes, n, i, i + 1) - prefix_len(codes, n, i, i - 1) > 0 ? 1 : -1;
    int dmin = prefix_len(codes, n, i, i - d);
    int lmax = 2;
    while (prefix_len(codes, n, i, i + lmax * d) > dmin) lmax *= 2;
    int l = 0;
    for (int t = lmax / 2; t >= 1; t /= 2)
        if (prefix_len(codes, n, i, i + (l + t) * d) > dmin) l += t;
    int j = i + l * d;
    int dnode = prefix_len(codes, n, i, j);
    int s = 0;
    for (int t = (l + 1) / 2;; t = (t + 1) / 2) {
        if (prefix_len(codes, n, i, i + (s + t) * d) > dnode) s += t;
        if (t == 1) break;
    }
    int gamma = i + s * d + (d < 0 ? -1 : 0);
    int lo = i < j ? i : j, hi = i < j ? j : i;
    int lc = lo == gamma ? ~gamma : gamma;
    int rc = hi == gamma + 1 ? ~(gamma + 1) : gamma + 1;
    left[i] = lc;
    right[i] = rc;
    first[i] = lo;
    last[i] = hi;
    if (lc >= 0) parent_inner[lc] = i; else parent_leaf[~lc] = i;
    if (rc >= 0) parent_inner[rc] = i; else parent_leaf[~rc] = i;
    if (i == 0) parent_inner[0] = -1;
}

__device__ inline void box_union(const double *a, const double *b, double *o) {
    o[0] = fmin(a[0], b[0]); o[1] = fmax(a[1], b[1]);
    o[2] = fmin(a[2], b[2]); o[3] = fmax(a[3], b[3]);
    o[4] = fmin(a[4], b[4]); o[5] = fmax(a[5], b[5]);
}

__global__ void k_refit(const double *__restrict__ boxes, const int *__restrict__ ids, int n, const int *__restrict__ left,
                        const int *__restrict__ right, const int *__restrict__ parent_inner, const int *__restrict__ parent_leaf,
                        double *__restrict__ inner_box, int *__restrict__ arrived) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    int node = parent_leaf[i];
    while (node >= 0) {
        __threadfence();
        if (atomicAdd(&arrived[node], 1) == 0) return;  // the sibling subtree is not finished: its thread goes on
        __threadfence();
        const int lc = left[node], rc = right[node];
        const double *a = lc >= 0 ? inner_box + 6 * (size_t)lc : boxes + 6 * (size_t)ids[~lc];
        const double *b = rc >= 0 ? inner_box + 6 * (size_t)rc : boxes + 6 * (size_t)ids[~rc];
        double u[6];
        box_union(a, b, u);
        for (int q = 0; q < 6; q++) inner_box[6 * (size_t)node + q] = u[q];
        node = parent_inner[node];
    }
}

// depth-first position of a node whose leaf range starts at `lo`: 2*lo + (number of ancestors that hold it in their
// left subtree)
__device__ inline int preorder_of(int lo, int node_is_leaf, int idx, const int *left, const int *parent_inner,
                                  const int *parent_leaf) {
    int count = 0;
    int child = node_is_leaf ? ~idx : idx;
    int p = node_is_leaf ? parent_leaf[idx] : parent_inner[idx];
    while (p >= 0) {
        if (left[p] == child) count++;
        child = p;
        p = parent_inner[p];
    }
    return 2 * lo + count;
}

__global__ void k_emit(const double *__restrict__ boxes, const int *__restrict__ ids, const int *__restrict__ prim_ids, int n,
                       const int *__restrict__ left, const int *__restrict__ parent_inner, const int *__restrict__ parent_leaf,
                       const int *__restrict__ first, const int *__restrict__ last, const double *__restrict__ inner_box,
                       PtxBvhNode *__restrict__ out) {
    int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= 2 * n - 1) return;
    PtxBvhNode node;
    int pre;
    if (t < n - 1) {  // inner node t
        pre = preorder_of(first[t], 0, t, left, parent_inner, parent_leaf);
        for (int q = 0; q < 6; q++) node.box[q] = inner_box[6 * (size_t)t + q];
        node.obj = -1;
        node.skip = pre + 2 * (last[t] - first[t] + 1) - 1;
    } else {  // leaf at sorted position i
        int i = t - (n - 1);
        pre = n == 1 ? 0 : preorder_of(i, 1, i, left, parent_inner, parent_leaf);
        const double *b = boxes + 6 * (size_t)ids[i];
        for (int q = 0; q < 6; q++) node.box[q] = b[q];
        node.obj = prim_ids[ids[i]];
        node.skip = pre + 1;
    }
    node.pad = 0;
    out[pre] = node;
}

#define BVHCHK(expr)                                                       \
    do {                                                                   \
        hipError_t e_ = (expr);                                            \
        if (e_ != hipSuccess) {                                            \
            err = std::string(#expr) + ": " + hipGetErrorString(e_);       \
            goto done;                                                     \
        }                                                                  \
    } while (0)

}  // namespace

int ptx_bvh_build(const double *h_boxes, const int *h_prim_ids, int n, hipStream_t stream, PtxBvhNode **d_nodes_out,
                  int *n_nodes_out, std::string &err) {
    *d_nodes_out = nullptr;
    *n_nodes_out = 0;
    if (n <= 0) return 0;
    double lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (int i = 0; i < n; i++)
        for (int a = 0; a < 3; a++) {
            double c = 0.5 * (h_boxes[6 * (size_t)i + 2 * a] + h_boxes[6 * (size_t)i + 2 * a + 1]);
            if (c < lo[a]) lo[a] = c;
            if (c > hi[a]) hi[a] = c;
        }
    double sc[3];
    for (int a = 0; a < 3; a++) sc[a] = hi[a] > lo[a] ? 2097151.0 / (hi[a] - lo[a]) : 0.0;

    double *d_boxes = nullptr, *d_inner_box = nullptr;
    int *d_prim = nullptr, *d_ids = nullptr, *d_ids2 = nullptr, *d_left = nullptr, *d_right = nullptr, *d_pin = nullptr,
        *d_pleaf = nullptr, *d_first = nullptr, *d_last = nullptr, *d_arrived = nullptr;
    unsigned long long *d_codes = nullptr, *d_codes2 = nullptr;
    void *d_tmp = nullptr;
    PtxBvhNode *d_nodes = nullptr;
    size_t tmp_bytes = 0;
    const int n_inner = n > 1 ? n - 1 : 1;
    const unsigned gb = (unsigned)((n + 255) / 256), gb2 = (unsigned)((2 * n - 1 + 255) / 256);
    int rc = -1;

    BVHCHK(hipMalloc((void **)&d_boxes, sizeof(double) * 6 * (size_t)n));
    BVHCHK(hipMalloc((void **)&d_prim, sizeof(int) * (size_t)n));
    BVHCHK(hipMalloc((void **)&d_ids, sizeof(int) * (size_t)n));
    BVHCHK(hipMalloc((void **)&d_ids2, sizeof(int) * (size_t)n));
    BVHCHK(hipMalloc((void **)&d_codes, sizeof(unsigned long long) * (size_t)n));
    BVHCHK(hipMalloc((void **)&d_codes2, sizeof(unsigned long long) * (size_t)n));
    BVHCHK(hipMalloc((void **)&d_left, sizeof(int) * (size_t)n_inner));
    BVHCHK(hipMalloc((void **)&d_right, sizeof(int) * (size_t)n_inner));
    BVHCHK(hipMalloc((void **)&d_pin, sizeof(int) * (size_t)n_inner));
    BVHCHK(hipMalloc((void **)&d_first, sizeof(int) * (size_t)n_inner));
    BVHCHK(hipMalloc((void **)&d_last, sizeof(int) * (size_t)n_inner));
    BVHCHK(hipMalloc((void **)&d_arrived, sizeof(int) * (size_t)n_inner));
    BVHCHK(hipMalloc((void **)&d_pleaf, sizeof(int) * (size_t)n));
    BVHCHK(hipMalloc((void **)&d_inner_box, sizeof(double) * 6 * (size_t)n_inner));
    BVHCHK(hipMalloc((void **)&d_nodes, sizeof(PtxBvhNode) * (size_t)(2 * n - 1)));
    BVHCHK(hipMemcpyAsync(d_boxes, h_boxes, sizeof(double) * 6 * (size_t)n, hipMemcpyHostToDevice, stream));
    BVHCHK(hipMemcpyAsync(d_prim, h_prim_ids, sizeof(int) * (size_t)n, hipMemcpyHostToDevice, stream));
    BVHCHK(hipMemsetAsync(d_arrived, 0, sizeof(int) * (size_t)n_inner, stream));
    BVHCHK(hipMemsetAsync(d_pleaf, 0xff, sizeof(int) * (size_t)n, stream));  // -1: a lone leaf has no parent

    hipLaunchKernelGGL(k_morton, dim3(gb), dim3(256), 0, stream, d_boxes, n, lo[0], lo[1], lo[2], sc[0], sc[1], sc[2], d_codes, d_ids);
    BVHCHK(hipGetLastError());
    BVHCHK(rocprim::radix_sort_pairs(nullptr, tmp_bytes, d_codes, d_codes2, d_ids, d_ids2, (size_t)n, 0, 64, stream));
    BVHCHK(hipMalloc(&d_tmp, tmp_bytes ? tmp_bytes : 16));
    BVHCHK(rocprim::radix_sort_pairs(d_tmp, tmp_bytes, d_codes, d_codes2, d_ids, d_ids2, (size_t)n, 0, 64, stream));
    if (n > 1) {
        hipLaunchKernelGGL(k_hierarchy, dim3(gb), dim3(256), 0, stream, d_codes2, n, d_left, d_right, d_pin, d_pleaf, d_first, d_last);
        BVHCHK(hipGetLastError());
        hipLaunchKernelGGL(k_refit, dim3(gb), dim3(256), 0, stream, d_boxes, d_ids2, n, d_left, d_right, d_pin, d_pleaf, d_inner_box,
                           d_arrived);
        BVHCHK(hipGetLastError());
    }
    hipLaunchKernelGGL(k_emit, dim3(gb2), dim3(256), 0, stream, d_boxes, d_ids2, d_prim, n, d_left, d_pin, d_pleaf, d_first, d_last,
                       d_inner_box, d_nodes);
    BVHCHK(hipGetLastError());
    BVHCHK(hipStreamSynchronize(stream));
    *d_nodes_out = d_nodes;
    *n_nodes_out = 2 * n - 1;
    d_nodes = nullptr;
    rc = 0;
done:
    for (void *p : {(void *)d_boxes, (void *)d_inner_box, (void *)d_prim, (void *)d_ids, (void *)d_ids2, (void *)d_left,
                    (void *)d_right, (void *)d_pin, (void *)d_pleaf, (void *)d_first, (void *)d_last, (void *)d_arrived,
                    (void *)d_codes, (void *)d_codes2, d_tmp, (void *)d_nodes})
        (void)hipFree(p);
    return rc;
}
