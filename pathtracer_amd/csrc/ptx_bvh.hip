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
//   5. k_order      per inner node and per octant of ray directions: which child lies nearer
//   6. k_emit       nodes written in depth-first order with a "skip" link each, once per octant
// Steps 1-3 have an alternative on the host, build_sah: the same arrays from a top-down build by the surface-area
// heuristic, slower to make (a second for a million triangles against 0.2 s) and cheaper to walk.
// The emitted array is walked without a stack: a node that is hit is followed by the next array entry, a node
// that is missed (or a leaf that has been tested) by its skip link.  A fixed order meets the primitives of a ray
// that travels against it last, after every box along the ray; with one array per octant -- the child that is
// nearer along the axis separating the two children comes first -- a ray meets its near primitives early and
// the distance of the best hit culls the rest.
#include <cstring>

#include <hip/hip_runtime.h>
#include <rocprim/rocprim.hpp>

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdlib>
#include <mutex>
#include <string>
#include <thread>
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

// Bit o of order[i]: for rays of octant o (bit 0: d.x < 0, bit 1: d.y < 0, bit 2: d.z < 0) the right child of inner
// node i is walked before the left one.  The children are compared along the axis on which their box centres lie
// furthest apart.
__global__ void k_order(const double *__restrict__ boxes, const int *__restrict__ ids, int n, const int *__restrict__ left,
                        const int *__restrict__ right, const double *__restrict__ inner_box, unsigned char *__restrict__ order) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n - 1) return;
    const int lc = left[i], rc = right[i];
    const double *a = lc >= 0 ? inner_box + 6 * (size_t)lc : boxes + 6 * (size_t)ids[~lc];
    const double *b = rc >= 0 ? inner_box + 6 * (size_t)rc : boxes + 6 * (size_t)ids[~rc];
    int axis = 0;
    double sep = 0, best = -1;
    for (int q = 0; q < 3; q++) {
        const double dq = (b[2 * q] + b[2 * q + 1]) - (a[2 * q] + a[2 * q + 1]);  // twice the distance of the centres
        if (fabs(dq) > best) {
            best = fabs(dq);
            sep = dq;
            axis = q;
        }
    }
    const bool left_is_low = sep >= 0;
    unsigned bits = 0;
    for (int o = 0; o < 8; o++) {
        const bool negative = (o >> axis) & 1;
        if (left_is_low == negative) bits |= 1u << o;  // the ray comes from the right child's side
    }
    order[i] = (unsigned char)bits;
}

__device__ inline int subtree_nodes(int child, const int *first, const int *last) {
    return child < 0 ? 1 : 2 * (last[child] - first[child] + 1) - 1;
}

// depth-first position of a node in the order of octant o: one for every ancestor, plus the subtree of the sibling
// wherever the sibling is walked first
__device__ inline int preorder_of(int node_is_leaf, int idx, int o, const int *left, const int *right, const int *parent_inner,
                                  const int *parent_leaf, const int *first, const int *last, const unsigned char *order) {
    int pos = 0;
    int child = node_is_leaf ? ~idx : idx;
    int p = node_is_leaf ? parent_leaf[idx] : parent_inner[idx];
    while (p >= 0) {
        const bool is_left = left[p] == child;
        const bool right_first = (order[p] >> o) & 1;
        pos += 1;
        if (is_left == right_first) pos += subtree_nodes(is_left ? right[p] : left[p], first, last);
        child = p;
        p = parent_inner[p];
    }
    return pos;
}

// x moved outward by `margin` as a float that is not above it (a lower bound) or not below it (an upper bound)
__device__ inline float outward(double x, int upper, double margin) {
    return upper ? __double2float_ru(x + margin) : __double2float_rd(x - margin);
}

__global__ void k_emit(const double *__restrict__ boxes, const int *__restrict__ ids, const int *__restrict__ prim_ids, int n,
                       int o, int near_first, double margin, const int *__restrict__ left, const int *__restrict__ right, const int *__restrict__ parent_inner,
                       const int *__restrict__ parent_leaf, const int *__restrict__ first, const int *__restrict__ last,
                       const unsigned char *__restrict__ order, const double *__restrict__ inner_box,
                       PtxBvhNode *__restrict__ out) {
    int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= 2 * n - 1) return;
    PtxBvhNode node;
    int pre;
    if (t < n - 1) {  // inner node t
        pre = preorder_of(0, t, o, left, right, parent_inner, parent_leaf, first, last, order);
        for (int q = 0; q < 6; q++) node.box[q] = outward(inner_box[6 * (size_t)t + q], q & 1, margin);
        node.obj = -1;
        node.skip = pre + 2 * (last[t] - first[t] + 1) - 1;
    } else {  // leaf at sorted position i
        int i = t - (n - 1);
        pre = n == 1 ? 0 : preorder_of(1, i, o, left, right, parent_inner, parent_leaf, first, last, order);
        const double *b = boxes + 6 * (size_t)ids[i];
        for (int q = 0; q < 6; q++) node.box[q] = outward(b[q], q & 1, margin);
        node.obj = prim_ids[ids[i]];
        node.skip = pre + 1;
    }
    if (near_first)  // the array of an octant: on an axis along which its rays travel downward the upper bound comes first
        for (int a = 0; a < 3; a++)
            if ((o >> a) & 1) {
                const float lower = node.box[2 * a];
                node.box[2 * a] = node.box[2 * a + 1];
                node.box[2 * a + 1] = lower;
            }
    out[pre] = node;
}

// ---- the same tree, four children to a node (PtxWideNode)
// Depth of every inner node of the binary tree (the root has depth 0), and the largest depth of any leaf.
__global__ void k_depth(int n, const int *__restrict__ parent_inner, const int *__restrict__ parent_leaf, int *__restrict__ depth,
                        int *__restrict__ even_flag, int *__restrict__ max_leaf_depth) {
    int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= 2 * n - 1) return;
    if (t < n - 1) {
        int d = 0;
        for (int p = parent_inner[t]; p >= 0; p = parent_inner[p]) d++;
        depth[t] = d;
        even_flag[t] = (d & 1) == 0;
    } else {
        int d = 0;
        for (int p = parent_leaf[t - (n - 1)]; p >= 0; p = parent_inner[p]) d++;
        atomicMax(max_leaf_depth, d);
    }
}

// One wide node per inner node of even depth: the four grandchildren (or a child itself where it is a leaf) with their
// boxes, and for each octant of ray directions the order in which to visit them -- the nearer child of the node first
// and, within each child, its nearer child first (k_order's choice for the three binary nodes involved).
__global__ void k_emit_wide(const double *__restrict__ boxes, const int *__restrict__ ids, const int *__restrict__ prim_ids, int n,
                            double margin, const int *__restrict__ left, const int *__restrict__ right, const int *__restrict__ depth,
                            const int *__restrict__ wide_index, const unsigned char *__restrict__ order,
                            const double *__restrict__ inner_box, PtxWideNode *__restrict__ out) {
    int t = blockIdx.x * blockDim.x + threadIdx.x;
    PtxWideNode w;
    for (int s = 0; s < 4; s++) {
        w.lo_x[s] = w.lo_y[s] = w.lo_z[s] = INFINITY;
        w.hi_x[s] = w.hi_y[s] = w.hi_z[s] = -INFINITY;
        w.child[s] = 0;
    }
    w.pad[0] = w.pad[1] = 0;
    auto put = [&](int slot, int child) {  // child: >= 0 inner node (of even depth), < 0 leaf ~sorted position
        const double *b = child >= 0 ? inner_box + 6 * (size_t)child : boxes + 6 * (size_t)ids[~child];
        w.lo_x[slot] = outward(b[0], 0, margin); w.hi_x[slot] = outward(b[1], 1, margin);
        w.lo_y[slot] = outward(b[2], 0, margin); w.hi_y[slot] = outward(b[3], 1, margin);
        w.lo_z[slot] = outward(b[4], 0, margin); w.hi_z[slot] = outward(b[5], 1, margin);
        w.child[slot] = child >= 0 ? ~wide_index[child] : prim_ids[ids[~child]];
    };
    if (n == 1) {  // a lone primitive: no inner node at all
        if (t != 0) return;
        put(0, ~0);
        for (int o = 0; o < 8; o++) w.order[o] = 0xE4;  // slots 0, 1, 2, 3
        out[0] = w;
        return;
    }
    if (t >= n - 1 || (depth[t] & 1)) return;
    const int side[2] = {left[t], right[t]};
    for (int h = 0; h < 2; h++) {
        const int c = side[h];
        if (c < 0) {
            put(2 * h, c);
        } else {
            put(2 * h, left[c]);
            put(2 * h + 1, right[c]);
        }
    }
    for (int o = 0; o < 8; o++) {
        const int first_half = (order[t] >> o) & 1;  // 1: the right child's pair comes first
        unsigned bits = 0;
        for (int q = 0; q < 2; q++) {
            const int h = q ^ first_half;
            const int c = side[h];
            const int swap = c >= 0 ? (order[c] >> o) & 1 : 0;
            bits |= (unsigned)(2 * h + swap) << (4 * q);
            bits |= (unsigned)(2 * h + (swap ^ 1)) << (4 * q + 2);
        }
        w.order[o] = (unsigned char)bits;
    }
    out[wide_index[t]] = w;
}

// Topology by the surface-area heuristic, on the host: the same arrays k_hierarchy fills (a permutation of the
// primitives in which every inner node covers a run first..last, its two children, the parents), so that refit,
// child order and emission are shared with the Morton build.  Top-down; every node is split where
// area(left) * count(left) + area(right) * count(right) is smallest among the planes between 16 bins of the
// centroids on each axis; a node whose centroids coincide is cut in the middle.  One primitive per leaf.
// The work is shared out over host threads.  A node of many primitives is split by all threads together (each bins
// and partitions a slice of the node's run); below a cut -- a node small enough that there are several such nodes per
// thread -- a subtree is one thread's job, with a block of inner-node indices of its own, one per primitive less one,
// reserved when its root is made.  The splits do not depend on the number of threads; the numbering of the inner
// nodes does, which no result depends on.
struct SahTopology {
    std::vector<int> ids, left, right, parent_inner, parent_leaf, first, last;
};

namespace sah {

constexpr int K = 16;
struct Bin {
    float b[6];
    int count;
};
const float kEmpty[6] = {INFINITY, -INFINITY, INFINITY, -INFINITY, INFINITY, -INFINITY};

inline void grow(float *b, const double *x) {
    for (int a = 0; a < 3; a++) {
        b[2 * a] = std::min(b[2 * a], (float)x[2 * a]);
        b[2 * a + 1] = std::max(b[2 * a + 1], (float)x[2 * a + 1]);
    }
}
inline void merge(float *b, const float *x) {
    for (int a = 0; a < 3; a++) {
        b[2 * a] = std::min(b[2 * a], x[2 * a]);
        b[2 * a + 1] = std::max(b[2 * a + 1], x[2 * a + 1]);
    }
}
inline float area(const float *b) {
    const float dx = b[1] - b[0], dy = b[3] - b[2], dz = b[5] - b[4];
    return dx * dy + dy * dz + dz * dx;
}

// runs f(t, lo, hi) for slices [lo, hi) of [0, m) on n_threads threads (the caller's among them)
template <typename F>
void slices(int n_threads, int m, F &&f) {
    if (n_threads <= 1 || m < 4096) {
        f(0, 0, m);
        return;
    }
    std::vector<std::thread> pool;
    const int per = (m + n_threads - 1) / n_threads;
    for (int t = 1; t < n_threads; t++) {
        const int lo = std::min(m, t * per), hi = std::min(m, lo + per);
        if (lo < hi) pool.emplace_back([&f, t, lo, hi] { f(t, lo, hi); });
    }
    f(0, 0, std::min(m, per));
    for (std::thread &th : pool) th.join();
}

struct Builder {
    const double *boxes;
    int n;
    SahTopology &t;
    std::vector<float> cen;
    std::vector<int> tmp;  // scratch of the partitions done by several threads

    struct Split {
        int axis;   // -1: no plane separates the centroids
        int k;      // plane between bins k-1 and k
        float cmin[3], scale[3];
    };

    // centroid bounds, bins and the best plane of the run lo..hi, on n_threads threads
    Split choose(int lo, int hi, int n_threads) const {
        const int m = hi - lo + 1;
        const int T = std::max(1, n_threads);
        std::vector<float> tmin(3 * T, INFINITY), tmax(3 * T, -INFINITY);
        slices(T, m, [&](int th, int a0, int a1) {
            float mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
            for (int q = lo + a0; q < lo + a1; q++)
                for (int a = 0; a < 3; a++) {
                    const float c = cen[3 * (size_t)t.ids[q] + a];
                    mn[a] = std::min(mn[a], c);
                    mx[a] = std::max(mx[a], c);
                }
            for (int a = 0; a < 3; a++) {
                tmin[3 * th + a] = mn[a];
                tmax[3 * th + a] = mx[a];
            }
        });
        Split sp;
        sp.axis = -1;
        sp.k = 0;
        float cmax[3];
        for (int a = 0; a < 3; a++) {
            sp.cmin[a] = INFINITY;
            cmax[a] = -INFINITY;
            for (int th = 0; th < T; th++) {
                sp.cmin[a] = std::min(sp.cmin[a], tmin[3 * th + a]);
                cmax[a] = std::max(cmax[a], tmax[3 * th + a]);
            }
            sp.scale[a] = cmax[a] > sp.cmin[a] ? (float)K * (1.0f - 1e-6f) / (cmax[a] - sp.cmin[a]) : 0.0f;
        }
        if (m <= 2) return sp;
        std::vector<Bin> all((size_t)T * 3 * K);
        for (Bin &bn : all) {
            memcpy(bn.b, kEmpty, sizeof(kEmpty));
            bn.count = 0;
        }
        slices(T, m, [&](int th, int a0, int a1) {
            Bin *bins = all.data() + (size_t)th * 3 * K;
            for (int q = lo + a0; q < lo + a1; q++) {
                const int id = t.ids[q];
                for (int a = 0; a < 3; a++) {
                    if (sp.scale[a] == 0.0f) continue;
                    Bin &bn = bins[a * K + (int)((cen[3 * (size_t)id + a] - sp.cmin[a]) * sp.scale[a])];
                    grow(bn.b, boxes + 6 * (size_t)id);
                    bn.count++;
                }
            }
        });
        Bin *bins = all.data();  // (min, max and sums of counts: the result is the same however the run was sliced)
        for (int th = 1; th < T; th++)
            for (int q = 0; q < 3 * K; q++) {
                const Bin &o = all[(size_t)th * 3 * K + q];
                if (o.count) merge(bins[q].b, o.b);
                bins[q].count += o.count;
            }
        float best = INFINITY;
        for (int a = 0; a < 3; a++) {
            if (sp.scale[a] == 0.0f) continue;
            float right_area[K];
            float acc[6];
            memcpy(acc, kEmpty, sizeof(kEmpty));
            for (int k = K - 1; k >= 1; k--) {
                if (bins[a * K + k].count) merge(acc, bins[a * K + k].b);
                right_area[k] = area(acc);
            }
            memcpy(acc, kEmpty, sizeof(kEmpty));
            int left_count = 0;
            for (int k = 1; k < K; k++) {  // plane between bins k-1 and k
                if (bins[a * K + k - 1].count) merge(acc, bins[a * K + k - 1].b);
                left_count += bins[a * K + k - 1].count;
                if (left_count == 0 || left_count == m) continue;
                const float cost = area(acc) * (float)left_count + right_area[k] * (float)(m - left_count);
                if (cost < best) {
                    best = cost;
                    sp.axis = a;
                    sp.k = k;
                }
            }
        }
        return sp;
    }

    // the run lo..hi rearranged so that the primitives on the low side of the plane come first; returns where the high
    // side starts.  Within each side the primitives keep their order (any order would do: the subtrees sort them again).
    int partition(int lo, int hi, const Split &sp, int n_threads) {
        const int m = hi - lo + 1, a = sp.axis;
        auto low_side = [&](int id) { return (int)((cen[3 * (size_t)id + a] - sp.cmin[a]) * sp.scale[a]) < sp.k; };
        if (n_threads <= 1 || m < 4096) {
            int *b = t.ids.data() + lo, *e = t.ids.data() + hi + 1;
            return lo + (int)(std::stable_partition(b, e, low_side) - b);
        }
        const int T = n_threads;
        std::vector<int> n_low(T + 1, 0), from(T + 1, 0);
        slices(T, m, [&](int th, int a0, int a1) {
            int c = 0;
            for (int q = lo + a0; q < lo + a1; q++) c += low_side(t.ids[q]);
            n_low[th + 1] = c;
            from[th + 1] = a1;
        });
        for (int th = 0; th < T; th++) {
            n_low[th + 1] += n_low[th];
            from[th + 1] = std::max(from[th + 1], from[th]);
        }
        const int total_low = n_low[T];
        slices(T, m, [&](int th, int a0, int a1) {
            int wl = lo + n_low[th], wh = lo + total_low + (a0 - n_low[th]);
            for (int q = lo + a0; q < lo + a1; q++) {
                const int id = t.ids[q];
                if (low_side(id)) tmp[wl++] = id; else tmp[wh++] = id;
            }
        });
        slices(T, m, [&](int, int a0, int a1) { memcpy(t.ids.data() + lo + a0, tmp.data() + lo + a0, sizeof(int) * (size_t)(a1 - a0)); });
        return lo + total_low;
    }

    // where to cut lo..hi (> lo, <= hi)
    int split(int lo, int hi, int n_threads) {
        const int m = hi - lo + 1;
        int mid = -1;
        const Split sp = choose(lo, hi, n_threads);
        if (sp.axis >= 0) {
            mid = partition(lo, hi, sp, n_threads);
            if (mid == lo || mid > hi) mid = -1;
        }
        if (mid < 0) {  // two primitives, or no plane separates the centroids: the middle along the longest axis
            float mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
            for (int q = lo; q <= hi; q++)
                for (int a = 0; a < 3; a++) {
                    const float c = cen[3 * (size_t)t.ids[q] + a];
                    mn[a] = std::min(mn[a], c);
                    mx[a] = std::max(mx[a], c);
                }
            int a = 0;
            for (int q = 1; q < 3; q++)
                if (mx[q] - mn[q] > mx[a] - mn[a]) a = q;
            mid = lo + m / 2;
            std::nth_element(t.ids.begin() + lo, t.ids.begin() + mid, t.ids.begin() + hi + 1, [&](int x, int y) {
                const float cx = cen[3 * (size_t)x + a], cy = cen[3 * (size_t)y + a];
                return cx < cy || (cx == cy && x < y);
            });
        }
        return mid;
    }

    struct Job {
        int lo, hi, node;
    };
    // node covers lo..hi and owns the inner-node indices node+1 .. node+(hi-lo)-1 for its descendants
    template <typename Push>
    void make_node(const Job &job, int mid, Push &&push) {
        const int lo = job.lo, hi = job.hi, node = job.node;
        t.first[node] = lo;
        t.last[node] = hi;
        int next = node + 1;
        for (int side = 0; side < 2; side++) {
            const int clo = side ? mid : lo, chi = side ? hi : mid - 1;
            int child;
            if (clo == chi) {
                child = ~clo;
                t.parent_leaf[clo] = node;
            } else {
                child = next;
                next += chi - clo;  // a subtree over c primitives has c - 1 inner nodes
                t.parent_inner[child] = node;
                push(Job{clo, chi, child});
            }
            (side ? t.right : t.left)[node] = child;
        }
    }

    void subtree(Job root) {  // one thread, everything below root
        std::vector<Job> stack{root};
        while (!stack.empty()) {
            const Job job = stack.back();
            stack.pop_back();
            make_node(job, split(job.lo, job.hi, 1), [&](const Job &j) { stack.push_back(j); });
        }
    }
};

}  // namespace sah

void build_sah(const double *boxes, int n, SahTopology &t) {
    t.ids.resize(n);
    for (int i = 0; i < n; i++) t.ids[i] = i;
    const int n_inner = n > 1 ? n - 1 : 1;
    t.left.assign(n_inner, 0);
    t.right.assign(n_inner, 0);
    t.parent_inner.assign(n_inner, -1);
    t.first.assign(n_inner, 0);
    t.last.assign(n_inner, 0);
    t.parent_leaf.assign(n, -1);
    if (n < 2) return;
    // (eight threads: a million triangles take 0.38 s on one thread, 0.1 s on six to eight, and more again on sixteen)
    int n_threads = (int)std::min(8u, std::max(1u, std::thread::hardware_concurrency()));
    if (const char *e = getenv("PTX_BUILD_THREADS")) n_threads = std::max(1, std::min(64, atoi(e)));
    if (n < 32768) n_threads = 1;
    sah::Builder bd{boxes, n, t, std::vector<float>(3 * (size_t)n), std::vector<int>(n_threads > 1 ? n : 0)};
    sah::slices(n_threads, n, [&](int, int a0, int a1) {
        for (int i = a0; i < a1; i++)
            for (int a = 0; a < 3; a++) bd.cen[3 * (size_t)i + a] = (float)(0.5 * (boxes[6 * (size_t)i + 2 * a] + boxes[6 * (size_t)i + 2 * a + 1]));
    });
    // the top of the tree, node by node with all threads on each; what falls below the cut is collected
    const int cut = n_threads > 1 ? std::max(4096, n / (8 * n_threads)) : n + 1;
    std::vector<sah::Builder::Job> top{{0, n - 1, 0}}, small;
    while (!top.empty()) {
        const sah::Builder::Job job = top.back();
        top.pop_back();
        if (job.hi - job.lo + 1 <= cut && n_threads > 1) {
            small.push_back(job);
            continue;
        }
        bd.make_node(job, bd.split(job.lo, job.hi, n_threads), [&](const sah::Builder::Job &j) { top.push_back(j); });
    }
    if (n_threads <= 1) return;  // (cut > n: everything was done above)
    // the subtrees below the cut, largest first, taken by the threads from a counter
    std::sort(small.begin(), small.end(), [](const sah::Builder::Job &x, const sah::Builder::Job &y) { return x.hi - x.lo > y.hi - y.lo; });
    std::atomic<size_t> next{0};
    std::vector<std::thread> pool;
    auto work = [&] {
        for (size_t q = next++; q < small.size(); q = next++) bd.subtree(small[q]);
    };
    for (int th = 1; th < n_threads; th++) pool.emplace_back(work);
    work();
    for (std::thread &th : pool) th.join();
}

// The topology of the last scene built with the surface-area heuristic, kept for the next call with the same boxes:
// a program that renders on several GPUs uploads one scene once per GPU, and the topology is the part of the build
// that is made on the host.
struct SahCache {
    std::mutex lock;
    int n = 0;
    uint64_t key = 0;
    SahTopology topo;
};
SahCache g_sah_cache;

uint64_t boxes_key(const double *boxes, int n) {  // FNV-1a over the bytes of the boxes, eight at a time
    uint64_t h = 1469598103934665603ULL;
    const uint64_t *w = (const uint64_t *)boxes;
    for (size_t q = 0; q < 6 * (size_t)n; q++) {
        h ^= w[q];
        h *= 1099511628211ULL;
    }
    return h;
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

int ptx_bvh_build(const double *h_boxes, const int *h_prim_ids, int n, int n_orders, bool use_sah, bool want_wide, hipStream_t stream,
                  PtxBvhBuilt *out, std::string &err) {
    *out = PtxBvhBuilt();
    if (n <= 0) return 0;
    if (n_orders != 1 && n_orders != 8) {
        err = "ptx_bvh_build: one order or eight";
        return -1;
    }
    double lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (int i = 0; i < n; i++)
        for (int a = 0; a < 3; a++) {
            double c = 0.5 * (h_boxes[6 * (size_t)i + 2 * a] + h_boxes[6 * (size_t)i + 2 * a + 1]);
            if (c < lo[a]) lo[a] = c;
            if (c > hi[a]) hi[a] = c;
        }
    double sc[3];
    for (int a = 0; a < 3; a++) sc[a] = hi[a] > lo[a] ? 2097151.0 / (hi[a] - lo[a]) : 0.0;
    // Node boxes are pushed outward by a millionth of the scene's size (the largest coordinate or extent): the
    // walk tests them with a shorter, less exact form of the slab test and without a tolerance of its own, and
    // a ray that starts on a primitive's surface must still find itself inside that primitive's box.
    double scale = 0;
    for (size_t q = 0; q < 6 * (size_t)n; q++)
        if (std::isfinite(h_boxes[q])) scale = std::max(scale, std::fabs(h_boxes[q]));
    for (int a = 0; a < 3; a++) scale = std::max(scale, hi[a] - lo[a]);
    const double margin = scale * 0x1p-20;

    double *d_boxes = nullptr, *d_inner_box = nullptr;
    int *d_prim = nullptr, *d_ids = nullptr, *d_ids2 = nullptr, *d_left = nullptr, *d_right = nullptr, *d_pin = nullptr,
        *d_pleaf = nullptr, *d_first = nullptr, *d_last = nullptr, *d_arrived = nullptr;
    unsigned long long *d_codes = nullptr, *d_codes2 = nullptr;
    unsigned char *d_order = nullptr;
    void *d_tmp = nullptr, *d_scan_tmp = nullptr;
    PtxBvhNode *d_nodes = nullptr;
    PtxWideNode *d_wide = nullptr;
    int *d_depth = nullptr, *d_even = nullptr, *d_widx = nullptr, *d_maxdepth = nullptr;
    int n_wide = 0, wide_depth = 0;
    size_t tmp_bytes = 0, scan_bytes = 0;
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
    BVHCHK(hipMalloc((void **)&d_order, (size_t)n_inner));
    BVHCHK(hipMalloc((void **)&d_nodes, sizeof(PtxBvhNode) * (size_t)(2 * n - 1) * n_orders));
    BVHCHK(hipMemsetAsync(d_order, 0, (size_t)n_inner, stream));
    BVHCHK(hipMemcpyAsync(d_boxes, h_boxes, sizeof(double) * 6 * (size_t)n, hipMemcpyHostToDevice, stream));
    BVHCHK(hipMemcpyAsync(d_prim, h_prim_ids, sizeof(int) * (size_t)n, hipMemcpyHostToDevice, stream));
    BVHCHK(hipMemsetAsync(d_arrived, 0, sizeof(int) * (size_t)n_inner, stream));
    BVHCHK(hipMemsetAsync(d_pleaf, 0xff, sizeof(int) * (size_t)n, stream));  // -1: a lone leaf has no parent

    if (use_sah) {
        std::lock_guard<std::mutex> hold(g_sah_cache.lock);
        const uint64_t key = boxes_key(h_boxes, n);
        if (g_sah_cache.n != n || g_sah_cache.key != key || getenv("PTX_BVH_NO_CACHE")) {
            build_sah(h_boxes, n, g_sah_cache.topo);
            g_sah_cache.n = n;
            g_sah_cache.key = key;
        }
        const SahTopology &t = g_sah_cache.topo;
        BVHCHK(hipMemcpyAsync(d_ids2, t.ids.data(), sizeof(int) * (size_t)n, hipMemcpyHostToDevice, stream));
        BVHCHK(hipMemcpyAsync(d_pleaf, t.parent_leaf.data(), sizeof(int) * (size_t)n, hipMemcpyHostToDevice, stream));
        BVHCHK(hipMemcpyAsync(d_left, t.left.data(), sizeof(int) * (size_t)n_inner, hipMemcpyHostToDevice, stream));
        BVHCHK(hipMemcpyAsync(d_right, t.right.data(), sizeof(int) * (size_t)n_inner, hipMemcpyHostToDevice, stream));
        BVHCHK(hipMemcpyAsync(d_pin, t.parent_inner.data(), sizeof(int) * (size_t)n_inner, hipMemcpyHostToDevice, stream));
        BVHCHK(hipMemcpyAsync(d_first, t.first.data(), sizeof(int) * (size_t)n_inner, hipMemcpyHostToDevice, stream));
        BVHCHK(hipMemcpyAsync(d_last, t.last.data(), sizeof(int) * (size_t)n_inner, hipMemcpyHostToDevice, stream));
        BVHCHK(hipStreamSynchronize(stream));  // the copies are done before the cache is let go of
    } else {
        hipLaunchKernelGGL(k_morton, dim3(gb), dim3(256), 0, stream, d_boxes, n, lo[0], lo[1], lo[2], sc[0], sc[1], sc[2], d_codes, d_ids);
        BVHCHK(hipGetLastError());
        BVHCHK(rocprim::radix_sort_pairs(nullptr, tmp_bytes, d_codes, d_codes2, d_ids, d_ids2, (size_t)n, 0, 64, stream));
        BVHCHK(hipMalloc(&d_tmp, tmp_bytes ? tmp_bytes : 16));
        BVHCHK(rocprim::radix_sort_pairs(d_tmp, tmp_bytes, d_codes, d_codes2, d_ids, d_ids2, (size_t)n, 0, 64, stream));
        if (n > 1) {
            hipLaunchKernelGGL(k_hierarchy, dim3(gb), dim3(256), 0, stream, d_codes2, n, d_left, d_right, d_pin, d_pleaf, d_first, d_last);
            BVHCHK(hipGetLastError());
        }
    }
    if (n > 1) {
        hipLaunchKernelGGL(k_refit, dim3(gb), dim3(256), 0, stream, d_boxes, d_ids2, n, d_left, d_right, d_pin, d_pleaf, d_inner_box,
                           d_arrived);
        BVHCHK(hipGetLastError());
    }
    if (n > 1 && (n_orders == 8 || want_wide)) {
        hipLaunchKernelGGL(k_order, dim3(gb), dim3(256), 0, stream, d_boxes, d_ids2, n, d_left, d_right, d_inner_box, d_order);
        BVHCHK(hipGetLastError());
    }
    for (int o = 0; o < n_orders; o++) {
        hipLaunchKernelGGL(k_emit, dim3(gb2), dim3(256), 0, stream, d_boxes, d_ids2, d_prim, n, o, n_orders == 8 ? 1 : 0, margin, d_left, d_right, d_pin, d_pleaf,
                           d_first, d_last, d_order, d_inner_box, d_nodes + (size_t)o * (2 * n - 1));
        BVHCHK(hipGetLastError());
    }
    if (want_wide) {
        // wide nodes: the inner nodes of even depth, numbered in index order (the root, inner node 0, first)
        int max_leaf_depth = 0;
        BVHCHK(hipMalloc((void **)&d_depth, sizeof(int) * (size_t)n_inner));
        BVHCHK(hipMalloc((void **)&d_even, sizeof(int) * (size_t)n_inner));
        BVHCHK(hipMalloc((void **)&d_widx, sizeof(int) * (size_t)n_inner));
        BVHCHK(hipMalloc((void **)&d_maxdepth, sizeof(int)));
        BVHCHK(hipMemsetAsync(d_maxdepth, 0, sizeof(int), stream));
        BVHCHK(hipMemsetAsync(d_even, 0, sizeof(int) * (size_t)n_inner, stream));
        if (n > 1) {
            hipLaunchKernelGGL(k_depth, dim3(gb2), dim3(256), 0, stream, n, d_pin, d_pleaf, d_depth, d_even, d_maxdepth);
            BVHCHK(hipGetLastError());
            BVHCHK(rocprim::exclusive_scan(nullptr, scan_bytes, d_even, d_widx, 0, (size_t)n_inner, rocprim::plus<int>(), stream));
            BVHCHK(hipMalloc(&d_scan_tmp, scan_bytes ? scan_bytes : 16));
            BVHCHK(rocprim::exclusive_scan(d_scan_tmp, scan_bytes, d_even, d_widx, 0, (size_t)n_inner, rocprim::plus<int>(), stream));
            int last_idx = 0, last_flag = 0;
            BVHCHK(hipMemcpyAsync(&last_idx, d_widx + (n_inner - 1), sizeof(int), hipMemcpyDeviceToHost, stream));
            BVHCHK(hipMemcpyAsync(&last_flag, d_even + (n_inner - 1), sizeof(int), hipMemcpyDeviceToHost, stream));
            BVHCHK(hipMemcpyAsync(&max_leaf_depth, d_maxdepth, sizeof(int), hipMemcpyDeviceToHost, stream));
            BVHCHK(hipStreamSynchronize(stream));
            n_wide = last_idx + last_flag;
        } else {
            n_wide = 1;
            max_leaf_depth = 1;
        }
        // a leaf at depth d of the binary tree (the root's children have depth 1) is a child of the wide node made from
        // its ancestor of depth d-1 or d-2, whichever is even
        wide_depth = (max_leaf_depth - 1) / 2 + 1;
        BVHCHK(hipMalloc((void **)&d_wide, sizeof(PtxWideNode) * (size_t)n_wide));
        hipLaunchKernelGGL(k_emit_wide, dim3(gb), dim3(256), 0, stream, d_boxes, d_ids2, d_prim, n, margin, d_left, d_right, d_depth, d_widx,
                           d_order, d_inner_box, d_wide);
        BVHCHK(hipGetLastError());
    }
    BVHCHK(hipStreamSynchronize(stream));
    out->d_nodes = d_nodes;
    out->n_nodes = 2 * n - 1;
    out->coord_bound = (float)((scale + margin) * (1.0 + 1e-6));
    out->d_wide = d_wide;
    out->n_wide = n_wide;
    out->wide_depth = wide_depth;
    d_nodes = nullptr;
    d_wide = nullptr;
    rc = 0;
done:
    for (void *p : {(void *)d_boxes, (void *)d_inner_box, (void *)d_prim, (void *)d_ids, (void *)d_ids2, (void *)d_left,
                    (void *)d_right, (void *)d_pin, (void *)d_pleaf, (void *)d_first, (void *)d_last, (void *)d_arrived,
                    (void *)d_codes, (void *)d_codes2, (void *)d_order, d_tmp, (void *)d_nodes, (void *)d_wide, (void *)d_depth,
                    (void *)d_even, (void *)d_widx, (void *)d_maxdepth, d_scan_tmp})
        (void)hipFree(p);
    return rc;
}
