// ptx_device.h -- device-side data layout and per-lane routines of the
// path-tracing core (gfx950).  Included only by ptx_hip.hip.
//
// Each routine names the reference routine whose result it reproduces.  The
// arithmetic is IEEE binary64 like the reference's, but it is not a
// transcription: transforms use the 3x4 affine part only, uniformly scaled
// spheres are intersected in world space, trigonometric round trips
// (atan2/acos followed by sin/cos of the same angle) are replaced by the
// algebraic sine and cosine, and hit attributes are computed once for the
// winning primitive instead of once per candidate.  Results agree with the
// reference to rounding (about 1e-11 relative on lambda for the radius-500
// wall spheres of the Cornell scene, far inside the 1e-5 first-hit bar).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "ptx_bvh.h"
#include "ptx_rng.h"
#include "ptx_scene.h"

#define PT_PI 3.14159265354  // reference PathTracer.h:27 (not pi: the truncated value is part of the behaviour)
#define PT_TOL 0.000000001   // reference PathTracer.h:28
// 2*PT_PI/pi rounded to binary64: sin(2*PT_PI*u) = sinpi(PT_TWO_PI_OVER_PI*u)
#define PT_TWO_PI_OVER_PI 0x1.ffffffffdd257p+0

// Threads per workgroup of every kernel that uses LaneStore: one wave.  A workgroup's place on its compute unit
// is free again only when all its waves are done, and waves of the render kernel run for different times.
#define PTX_BLOCK 64

struct V3 {
    double x, y, z;
};
__device__ __forceinline__ V3 mk(double x, double y, double z) { return V3{x, y, z}; }
__device__ __forceinline__ V3 operator+(V3 a, V3 b) { return mk(a.x + b.x, a.y + b.y, a.z + b.z); }
__device__ __forceinline__ V3 operator-(V3 a, V3 b) { return mk(a.x - b.x, a.y - b.y, a.z - b.z); }
__device__ __forceinline__ V3 operator*(V3 a, double s) { return mk(a.x * s, a.y * s, a.z * s); }
__device__ __forceinline__ double dot3(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
__device__ __forceinline__ V3 cross3(V3 u, V3 v) {
    return mk(u.y * v.z - v.y * u.z, v.x * u.z - u.x * v.z, u.x * v.y - v.x * u.y);
}
// 1/x for finite non-zero x: hardware estimate r (relative error e = 1 - x r of 2^-24) times 1 + e + e^2, i.e. two
// Newton steps folded into three instructions; equal to the correctly rounded quotient on every argument of
// tools/rcp_accuracy.hip.  Not for divisors that may be zero: the reference's divisions by zero yield infinities that
// later tests rely on.
__device__ __forceinline__ double fast_rcp(double x) {
    double r = __builtin_amdgcn_rcp(x);
    double e = fma(-x, r, 1.0);
    return fma(r, fma(e, e, e), r);
}
// sqrt(|x| + 1e-300), from the hardware reciprocal square root plus Goldschmidt steps: sqrt(x) for x >= 1e-284,
// 1e-150 for zero, and the root of the magnitude for the negative rounding residue of an expression that is zero
// in exact arithmetic.  The estimate (relative error 2^-24) gets one Goldschmidt step and one correction by the exact
// residual x - g^2; the half reciprocal root h that scales the residual stays the estimate's (its error of 2^-24
// enters the result times the residual's 2^-48): largest error measured 0.71 ulp (tools/rcp_accuracy.hip), against
// 0.5 ulp for one more instruction that refines h too.
__device__ __forceinline__ double fast_sqrt(double x) {
    x = fabs(x) + 1e-300;
    double y = __builtin_amdgcn_rsq(x);
    double g = x * y, h = 0.5 * y;
    double r = fma(-h, g, 0.5);
    g = fma(g, r, g);
#ifdef PTX_SQRT_REFINE_H
    h = fma(h, r, h);
#endif
    double d = fma(-g, g, x);
    return fma(d, h, g);
}
// 1/sqrt(x) for finite x > 0: hardware estimate plus one third-order correction (error about 1 ulp)
__device__ __forceinline__ double fast_rsqrt(double x) {
    double y = __builtin_amdgcn_rsq(x);
    double e = fma(-x * y, y, 1.0);
    return fma(y * e, fma(0.375, e, 0.5), y);
}
// 1/x as the box test wants it.  For a direction component that is zero (or too small for its reciprocal to be
// finite) the reference's (bound - p0) / d is -infinity or +infinity by the sign of bound - p0; a large finite
// stand-in of the component's sign gives bound * r - p0 * r that sign too (an infinite one would give
// inf - inf = NaN wherever bound and p0 have the same sign).  2^500 leaves room for coordinates up to 2^500.
__device__ __forceinline__ double rcp_for_slab(double x) { return fabs(x) < 0x1p-500 ? copysign(0x1p500, x) : fast_rcp(x); }
__device__ __forceinline__ V3 norm3(V3 v) {  // normalize(), reference utils_path.h:103-114
    return v * fast_rsqrt(dot3(v, v));
}
// Natural logarithm, error below 1 ulp for positive normal x: the argument reduction and polynomial of the
// classic fdlibm log (x = 2^k * m, sqrt(1/2) <= m < sqrt(2); s = f/(2+f), f = m-1), with the division done by
// fast_rcp.  About half the instructions of the library routine.  Zero, negative, huge and NaN arguments (which
// the callers meet only in corrupt paths) take the library routine.
__device__ __forceinline__ double lean_log(double x) {
    if (!(x > 2.3e-308 && x < 1e300)) return log(x);
    double m = __builtin_amdgcn_frexp_mant(x);  // [0.5, 1)
    int k = __builtin_amdgcn_frexp_exp(x);
    const bool low = m < 0.70710678118654752440;
    m = low ? m + m : m;
    k = low ? k - 1 : k;
    const double f = m - 1.0;
    const double s = f * fast_rcp(2.0 + f);
    const double z = s * s, w = z * z;
    const double t1 = w * fma(w, fma(w, 1.531383769920937332e-01, 2.222219843214978396e-01), 3.999999999940941908e-01);
    const double t2 = z * fma(w, fma(w, fma(w, 1.479819860511658591e-01, 1.818357216161805012e-01), 2.857142874366239149e-01),
                              6.666666666666735130e-01);
    const double R = t2 + t1;
    const double hfsq = 0.5 * f * f;
    const double dk = (double)k;
    return dk * 6.93147180369123816490e-01 - ((hfsq - fma(s, hfsq + R, dk * 1.90821492927058770002e-10)) - f);
}
__device__ __forceinline__ double comp(V3 v, int a) { return a == 0 ? v.x : (a == 1 ? v.y : v.z); }

// ---------------------------------------------------------------- scene
// Geometry record, 160 bytes, read either with a wave-uniform index (scalar
// loads) or per lane for the winning primitive.
//   analytic primitives: m[0..11] = rows 0..2 of Tinv (3x4)
//   triangles:           m[0..8]  = v1 v2 v3 (world space), m[9..11] = unit normal
//   fast = 1: a sphere whose transform is a uniform scale plus a translation;
//             c = centre, e[0] = radius^2, e[1] = 1/radius
//   planes:   c = world-space unit normal for a ray that starts on the canonical +z side, e = the same for the
//             -z side (normalTransform of (0,0,1) and (0,0,-1); they differ in more than sign where a component
//             is a signed zero, which the sampling rotation tells apart)
struct DevGeom {
    int32_t type;
    int32_t fast;
    double m[12];
    double c[3];
    double e[3];
    double pad;
};
static_assert(sizeof(DevGeom) == 160, "DevGeom layout");
// The always-tested list is read through the constant address space: the scene is never written while a kernel
// runs, and a load from that address space with a wave-uniform address is a scalar load whatever else the kernel
// does (a barrier, an atomic or an opaque asm in a loop makes the compiler turn such loads from global memory
// into per-lane loads).
#define PTX_CONSTANT_AS __attribute__((address_space(4)))
typedef const PTX_CONSTANT_AS DevGeom *ConstGeomPtr;

// Shading record, 96 bytes, read per lane for the primitive that was hit.
struct DevMat {
    double col[3];
    double diffPct, reflPct, tranPct;
    double refl_sig, r_index;
    double surfaceArea;
    int32_t is_light, tex_img, normal_img, alpha_img;
    double pad;
};
static_assert(sizeof(DevMat) == 96, "DevMat layout");

// Light record: what the light-choice walk and the point samplers need
// (reference PathTracer.c:399-415, utils_path.c:1149-1209).
struct DevLight {
    int32_t obj;
    int32_t type;
    double weight;  // normalised LSweight
    double T[12];   // rows 0..2 of T
};

// Texels are the bytes of the image file, row by row as in the file (tiles_x = 0).  With PTX_TEXEL_LAYOUT=tiled at upload
// the three-channel images of a scene with a hierarchy are stored in tiles of 8 x 4 texels, a tile to a 128-byte line
// (96 bytes used), tiles row by row (tiles_x > 0): the four texels of a bilinear look-up are two rows apart in the
// file's order (3 sx bytes, two cache lines for every look-up) and lie in one line for two look-ups in three in the
// tiled one.  Measured, that saves L2 requests and no time, so the file's order is the default.
struct DevImage {
    int32_t sx, sy, channels, tiles_x;
    double divisor;
    uint64_t offset;
};
// byte offset of channel 0 of texel (x, y) of a tiled image, relative to DevImage::offset
__device__ __forceinline__ long tiled_texel(const DevImage &im, int x, int y) {
    return ((long)(y >> 2) * im.tiles_x + (x >> 3)) * 128 + ((y & 3) * 8 + (x & 7)) * 3;
}

struct DevCamera {
    double e[3], w[3];
    double C2W[12];
    double f, wl, wt, wsize;
    double fp[3];  // C2W * (0,0,focus_distance,1), reference PathTracer.c:589-594
};

// Groups of the always-tested primitives by intersection routine
enum { LC_FAST_SPHERE = 0, LC_SPHERE, LC_PLANE, LC_TRIANGLE, LC_CYL, LC_BOX, LC_N };
// Closest-hit queries test two sets of primitives: a short list that every ray tests in turn (all primitives of a
// small scene; in a large scene the few whose boxes are far bigger than the rest, which would spoil a hierarchy),
// and the remaining primitives through the bounding-volume hierarchy.
struct DevScene {
    const DevGeom *geom;      // every primitive, scene order (= the reference's visiting order)
    const DevMat *mat;
    const DevGeom *list_geom; // the always-tested primitives, ascending scene index
    const int32_t *list_ids;  // their scene indices
    const PtxBvhNode *bvh;    // hierarchy over the others, or nullptr
    const DevLight *lights;
    const DevImage *images;
    const uint8_t *texels;
    int32_t n_objects, n_list, n_bvh_nodes;  // n_bvh_nodes: per array
    int32_t bvh_per_octant;                  // 1: bvh holds eight arrays, one per octant of ray directions
    int32_t n_lights, n_images, has_refraction;
    DevCamera cam;
    const DevGeom *class_geom;  // the always-tested primitives grouped by intersection routine (LC_*), ascending scene
    const int32_t *class_ids;   // index within a group, and their scene indices
    int32_t class_end[6];       // where each group ends in class_geom
    float bvh_bound;            // no bound of any hierarchy node exceeds this in magnitude
    const PtxWideNode *wide;    // the hierarchy four children to a node (scenes whose node arrays exceed the L2 caches), or nullptr
    int32_t n_wide, wide_stack_bound;  // wide_stack_bound: no walk of `wide` ever holds more entries on its stack
};

struct DevRender {
    int32_t sx, max_depth;
    int32_t rank, n_ranks, band_rows, rows_owned;
    double aperture;
    uint64_t seed;
    double du, dv;
    int32_t clean, pad;  // PTX_CLEAN asked for (ptx_scene.h): not the reference's behaviour
};

// ------------------------------------------------------------------ RNG
// Counter stream of ptx_rng.h kept incrementally: ctr = key + GOLDEN*(draws so far).
struct Rng {
    uint64_t ctr;
    __device__ __forceinline__ double next() {
        ctr += PTX_GOLDEN;
        return ptx_bits_to_unit(ptx_mix64(ctr));
    }
    __device__ __forceinline__ void skip(int n) { ctr += PTX_GOLDEN * (uint64_t)n; }
};

// --------------------------------------------------- per-lane LDS store
// Cold per-lane state lives in LDS instead of registers: slot s of lane t is
// word [s][t], so a wave touches 64 consecutive words (no bank conflicts)
// and a lane can index its slots dynamically without going to scratch.
// Double slots:
//   0..2    ray3D.srcN x,y,z (reference PathTracer.h:85)
//   3..5    direction of the pixel's primary ray (the same for every sample when the aperture is zero: the
//           reference samples pixel centres, PathTracer.c:625-628)
// Int slots:
//   0..9    ray3D.r_index[0..9] (reference PathTracer.h:88), held as the index of the object whose r_index
//           was pushed (RI_ONE: the initial 1.0 of initRay; RI_ZERO: never written)
// 6 double + 10 int slots = 5.5 KiB per wave.
enum { LS_SRCN = 0, LS_PRIMARY_D = 3, LS_DSLOTS = 6 };
// further slots of the render kernels: LS_PARKED, colours of up to three finished samples waiting for their line's
// fourth; LS_PEND (hierarchy kernel only), the part of an explicit light sample that waits in LDS for its shadow query
enum { LS_PARKED = LS_DSLOTS, LS_PEND = LS_PARKED + 9, LS_PEND_N = 5 };
enum { LS_RINDEX = 0, LS_ISLOTS = 10 };
enum { RI_ONE = -1, RI_ZERO = -2 };
struct LaneStore {
    double *base;  // &lds_d[0][threadIdx.x]
    int *ibase;    // &lds_i[0][threadIdx.x]
    int pend;      // first of the LS_PEND_N double slots of the pending light sample (hierarchy kernels; LS_PEND unless the
                   // kernel parks no samples)
    __device__ __forceinline__ double get(int slot) const { return base[slot * PTX_BLOCK]; }
    __device__ __forceinline__ void put(int slot, double v) const { base[slot * PTX_BLOCK] = v; }
    __device__ __forceinline__ int geti(int slot) const { return ibase[slot * PTX_BLOCK]; }
    __device__ __forceinline__ void puti(int slot, int v) const { ibase[slot * PTX_BLOCK] = v; }
};
#define PTX_LANE_STORE(ls)                                   \
    __shared__ double lds_d_[LS_DSLOTS][PTX_BLOCK];          \
    __shared__ int lds_i_[LS_ISLOTS][PTX_BLOCK];             \
    const LaneStore ls{&lds_d_[0][threadIdx.x], &lds_i_[0][threadIdx.x], LS_PEND}

// ------------------------------------------------------- transforms
template <typename MP>
__device__ __forceinline__ V3 xform_point(MP m, V3 p) {
    return mk(m[0] * p.x + m[1] * p.y + m[2] * p.z + m[3], m[4] * p.x + m[5] * p.y + m[6] * p.z + m[7],
              m[8] * p.x + m[9] * p.y + m[10] * p.z + m[11]);
}
template <typename MP>
__device__ __forceinline__ V3 xform_dir(MP m, V3 d) {
    return mk(m[0] * d.x + m[1] * d.y + m[2] * d.z, m[4] * d.x + m[5] * d.y + m[6] * d.z,
              m[8] * d.x + m[9] * d.y + m[10] * d.z);
}
// normalTransform (reference utils_path.c:58-76): direction of Tinv^T n.
template <typename MP>
__device__ __forceinline__ V3 xform_normal(MP m, V3 n) {
    return norm3(mk(m[0] * n.x + m[4] * n.y + m[8] * n.z, m[1] * n.x + m[5] * n.y + m[9] * n.z,
                    m[2] * n.x + m[6] * n.y + m[10] * n.z));
}

// ------------------------------------------------- primitive intersectors
// Each returns the lambda the reference intersect function would report
// (a value <= PT_TOL means "no usable hit") and, in `part`, which face or
// section produced it.

// Smallest root > TOL of the sphere quadratic (reference utils_path.c:564-587), given l2 <= l1: l2 if it
// qualifies, else l1; the caller rejects a result that is not > TOL.
__device__ __forceinline__ double sphere_roots(double l1, double l2) { return l2 > PT_TOL ? l2 : l1; }

// sphereIntersect, reference utils_path.c:546-587, in object space
template <typename MP>
__device__ __forceinline__ double isect_sphere(MP m, V3 p0, V3 d) {
    V3 tp = xform_point(m, p0), td = xform_dir(m, d);
    double A = dot3(td, td), B = dot3(tp, td), C = dot3(tp, tp) - 1.0;
    double disc = B * B - A * C;
    if (disc < 0.0) return -1.0;
    double sq = fast_sqrt(disc);
    const double inv_a = fast_rcp(A);  // A = |d'|^2 > 0
    double mb = -B * inv_a, sa = sq * inv_a;
    return sphere_roots(mb + sa, mb - sa);
}

// The same quadratic for a uniformly scaled sphere, in world space: with
// p' = (p0-c)/r and d' = d/r the reference's A, B, C are (d.d, oc.d,
// oc.oc - r^2)/r^2 and the roots are unchanged.  inv_dd = 1/(d.d) is shared
// by all spheres of one ray.
template <typename GP>
__device__ __forceinline__ double isect_sphere_uniform(GP g, V3 p0, V3 d, double dd, double inv_dd) {
    V3 oc = mk(p0.x - g->c[0], p0.y - g->c[1], p0.z - g->c[2]);
    double B = dot3(oc, d), C = dot3(oc, oc) - g->e[0];
    double disc = B * B - dd * C;
    if (disc < 0.0) return -1.0;
    double sq = fast_sqrt(disc);
    // the smaller root if it exceeds TOL, else the larger (sphere_roots), chosen before the division by d.d
    double t2 = -B - sq;
    return (t2 > PT_TOL * dd ? t2 : sq - B) * inv_dd;
}

// planeIntersect + planeHelper, reference utils_path.c:439-544
template <typename MP>
__device__ __forceinline__ double isect_plane(MP m, V3 p0, V3 d) {
    V3 tp = xform_point(m, p0), td = xform_dir(m, d);
    if (fabs(td.z) < PT_TOL) return -1.0;
    double lam = -tp.z * fast_rcp(td.z);  // |td.z| >= TOL here
    if (!(fabs(tp.x + lam * td.x) <= 1.0 - PT_TOL)) return -1.0;
    if (!(fabs(tp.y + lam * td.y) <= 1.0 - PT_TOL)) return -1.0;
    return lam;
}

// cubeIntersect / roomIntersect, reference utils_path.c:832-995, :997-1143:
// faces tried in the order +z -z +y -y +x -x, strictly smaller lambda wins.
template <typename MP>
__device__ __forceinline__ double isect_box(MP m, V3 p0, V3 d, int &part) {
    V3 tp = xform_point(m, p0), td = xform_dir(m, d);
    double best = INFINITY;
    part = -1;
#pragma unroll
    for (int f = 0; f < 6; f++) {
        const int a = 2 - (f >> 1);
        const double s = (f & 1) ? -1.0 : 1.0;
        double da = comp(td, a), pa = comp(tp, a);
        if (fabs(da) < PT_TOL) continue;
        double lam = (s - pa) * fast_rcp(da);  // |da| >= TOL here
        const int b = (a + 1) % 3, c = (a + 2) % 3;
        if (!(fabs(comp(tp, b) + lam * comp(td, b)) <= 1.0 - PT_TOL)) continue;
        if (!(fabs(comp(tp, c) + lam * comp(td, c)) <= 1.0 - PT_TOL)) continue;
        if (lam > PT_TOL && lam < best) {
            best = lam;
            part = f;
        }
    }
    return part < 0 ? -1.0 : best;
}

// cylIntersect, reference utils_path.c:625-779.  part: 0 wall, 1 z=-1 cap, 2 z=+1 cap.
// The reference's four divisions are multiplications by reciprocals here.  Where it divides by zero (a ray along
// the axis for the wall, a ray across the axis for the caps) its quotients are infinite or NaN and every test on
// them fails; fast_rcp(0) is NaN, and the tests fail the same way.
template <typename MP>
__device__ __forceinline__ double isect_cyl(MP m, V3 p0, V3 d, int &part) {
    V3 tp = xform_point(m, p0), td = xform_dir(m, d);
    double best = INFINITY;
    part = -1;
    double A = td.x * td.x + td.y * td.y;
    double B = 2 * tp.x * td.x + 2 * tp.y * td.y;
    double C = tp.x * tp.x + tp.y * tp.y - 1;
    double det = B * B - 4 * A * C;
    if (!(det < 0)) {
        const double sq = fast_sqrt(det), inv_2a = fast_rcp(2 * A);
        double l1 = (-1 * B + sq) * inv_2a;
        double l2 = (-1 * B - sq) * inv_2a;
        if (fabs(tp.z + l1 * td.z) <= 1 && l1 > PT_TOL && best > l1) { best = l1; part = 0; }
        if (fabs(tp.z + l2 * td.z) <= 1 && l2 > PT_TOL && best > l2) { best = l2; part = 0; }
    }
    const double inv_dz = fast_rcp(td.z);
    double c1 = (-1 - tp.z) * inv_dz;
    double c2 = (1 - tp.z) * inv_dz;
    {
        double x = tp.x + c1 * td.x, y = tp.y + c1 * td.y;
        if (x * x + y * y <= 1 && c1 > PT_TOL && best > c1) { best = c1; part = 1; }
    }
    {
        double x = tp.x + c2 * td.x, y = tp.y + c2 * td.y;
        if (x * x + y * y <= 1 && c2 > PT_TOL && best > c2) { best = c2; part = 2; }
    }
    return part < 0 ? -1.0 : best;
}

// triangleIntersect, reference utils_path.c:781-830.  part: 1 if the stored
// normal must be flipped to face the ray.
template <typename MP>
__device__ __forceinline__ double isect_triangle(MP m, V3 p0, V3 d, int &part) {
    V3 p1 = mk(m[0], m[1], m[2]), p2 = mk(m[3], m[4], m[5]), p3 = mk(m[6], m[7], m[8]);
    V3 n = mk(m[9], m[10], m[11]);
    double ddn = dot3(n, d);
    part = ddn > 0 ? 1 : 0;
    if (-PT_TOL < ddn && ddn < PT_TOL) return -1.0;
    double lam = dot3(p1 - p0, n) * fast_rcp(ddn);  // |ddn| >= TOL here
    if (!(lam > 0)) return -1.0;
    V3 poi = p0 + d * lam;
    V3 e12 = p2 - p1, e23 = p3 - p2, e31 = p1 - p3, e13 = p3 - p1, e32 = p2 - p3, e21 = p1 - p2;
    double v1 = dot3(cross3(poi - p1, e12), cross3(e13, e12));
    double v2 = dot3(cross3(poi - p2, e23), cross3(e21, e23));
    double v3 = dot3(cross3(poi - p3, e31), cross3(e32, e31));
    if (v1 >= 0 && v2 >= 0 && v3 >= 0) return lam;
    return -1.0;
}

// ---- the same intersectors without branches, for the always-tested list of the hierarchy kernel
// There a primitive is tested by (nearly) all lanes of a wave at once and its record arrives by scalar loads.  The
// early returns above save such a wave nothing (some lane always goes on), and they cut the routine into blocks, each
// of which asks for its part of the record only when it is reached and then waits for it: three waits in a row for a
// plane.  Written as one block, the record is asked for once, at the top.  Every lane computes what the routines above
// compute on the path it would take there, so a hit has the same bits; where they return early these compute on
// (possibly with infinities or NaNs, which no comparison accepts) and discard the result.
// `hit` (or `missed`): whether the value returned means anything; the caller folds it into its own test instead of having
// a -1 selected here.  The box and cylinder routines return infinity for "nothing", which no test accepts.
template <typename MP>
__device__ __forceinline__ double isect_sphere_nb(MP m, V3 p0, V3 d, bool &hit) {
    V3 tp = xform_point(m, p0), td = xform_dir(m, d);
    double A = dot3(td, td), B = dot3(tp, td), C = dot3(tp, tp) - 1.0;
    double disc = B * B - A * C;
    double sq = fast_sqrt(disc);
    const double inv_a = fast_rcp(A);
    double mb = -B * inv_a, sa = sq * inv_a;
    hit = !(disc < 0.0);
    return sphere_roots(mb + sa, mb - sa);
}
// missed: the ray's line does not meet the sphere; the value returned is then meaningless (the caller tests `missed`
// instead of having a -1 selected here)
template <typename GP>
__device__ __forceinline__ double isect_sphere_uniform_nb(GP g, V3 p0, V3 d, double dd, double inv_dd, bool &missed) {
    V3 oc = mk(p0.x - g->c[0], p0.y - g->c[1], p0.z - g->c[2]);
    double B = dot3(oc, d), C = dot3(oc, oc) - g->e[0];
    double disc = B * B - dd * C;
    double sq = fast_sqrt(disc);
    double t2 = -B - sq;
    missed = disc < 0.0;
    return (t2 > PT_TOL * dd ? t2 : sq - B) * inv_dd;
}
template <typename MP>
__device__ __forceinline__ double isect_plane_nb(MP m, V3 p0, V3 d, bool &hit) {
    V3 tp = xform_point(m, p0), td = xform_dir(m, d);
    double lam = -tp.z * fast_rcp(td.z);
    hit = !(fabs(td.z) < PT_TOL) & (fabs(tp.x + lam * td.x) <= 1.0 - PT_TOL) & (fabs(tp.y + lam * td.y) <= 1.0 - PT_TOL);
    return lam;
}
template <typename MP>
__device__ __forceinline__ double isect_box_nb(MP m, V3 p0, V3 d, int &part) {
    V3 tp = xform_point(m, p0), td = xform_dir(m, d);
    double best = INFINITY;
    part = -1;
#pragma unroll
    for (int f = 0; f < 6; f++) {
        const int a = 2 - (f >> 1);
        const double s = (f & 1) ? -1.0 : 1.0;
        double da = comp(td, a), pa = comp(tp, a);
        double lam = (s - pa) * fast_rcp(da);
        const int b = (a + 1) % 3, c = (a + 2) % 3;
        const bool ok = !(fabs(da) < PT_TOL) & (fabs(comp(tp, b) + lam * comp(td, b)) <= 1.0 - PT_TOL) &
                        (fabs(comp(tp, c) + lam * comp(td, c)) <= 1.0 - PT_TOL) & (lam > PT_TOL) & (lam < best);
        best = ok ? lam : best;
        part = ok ? f : part;
    }
    return best;
}
template <typename MP>
__device__ __forceinline__ double isect_cyl_nb(MP m, V3 p0, V3 d, int &part) {
    V3 tp = xform_point(m, p0), td = xform_dir(m, d);
    double best = INFINITY;
    part = -1;
    double A = td.x * td.x + td.y * td.y;
    double B = 2 * tp.x * td.x + 2 * tp.y * td.y;
    double C = tp.x * tp.x + tp.y * tp.y - 1;
    double det = B * B - 4 * A * C;
    const bool wall = !(det < 0);
    const double sq = fast_sqrt(det), inv_2a = fast_rcp(2 * A);
    double l1 = (-1 * B + sq) * inv_2a;
    double l2 = (-1 * B - sq) * inv_2a;
    const bool ok1 = wall & (fabs(tp.z + l1 * td.z) <= 1) & (l1 > PT_TOL) & (best > l1);
    best = ok1 ? l1 : best;
    part = ok1 ? 0 : part;
    const bool ok2 = wall & (fabs(tp.z + l2 * td.z) <= 1) & (l2 > PT_TOL) & (best > l2);
    best = ok2 ? l2 : best;
    part = ok2 ? 0 : part;
    const double inv_dz = fast_rcp(td.z);
    double c1 = (-1 - tp.z) * inv_dz;
    double c2 = (1 - tp.z) * inv_dz;
    {
        double x = tp.x + c1 * td.x, y = tp.y + c1 * td.y;
        const bool ok = (x * x + y * y <= 1) & (c1 > PT_TOL) & (best > c1);
        best = ok ? c1 : best;
        part = ok ? 1 : part;
    }
    {
        double x = tp.x + c2 * td.x, y = tp.y + c2 * td.y;
        const bool ok = (x * x + y * y <= 1) & (c2 > PT_TOL) & (best > c2);
        best = ok ? c2 : best;
        part = ok ? 2 : part;
    }
    return best;
}
template <typename MP>
__device__ __forceinline__ double isect_triangle_nb(MP m, V3 p0, V3 d, int &part, bool &hit) {
    V3 p1 = mk(m[0], m[1], m[2]), p2 = mk(m[3], m[4], m[5]), p3 = mk(m[6], m[7], m[8]);
    V3 n = mk(m[9], m[10], m[11]);
    double ddn = dot3(n, d);
    part = ddn > 0 ? 1 : 0;
    double lam = dot3(p1 - p0, n) * fast_rcp(ddn);
    V3 poi = p0 + d * lam;
    V3 e12 = p2 - p1, e23 = p3 - p2, e31 = p1 - p3, e13 = p3 - p1, e32 = p2 - p3, e21 = p1 - p2;
    double v1 = dot3(cross3(poi - p1, e12), cross3(e13, e12));
    double v2 = dot3(cross3(poi - p2, e23), cross3(e21, e23));
    double v3 = dot3(cross3(poi - p3, e31), cross3(e32, e31));
    hit = !((-PT_TOL < ddn) & (ddn < PT_TOL)) & (lam > 0) & (v1 >= 0) & (v2 >= 0) & (v3 >= 0);
    return lam;
}

struct Hit {
    double lambda;
    int obj;
    int part;
};

// lambda of one primitive for one ray (dd = d.d, inv_dd = 1/dd)
// g: pointer to the geometry record, in global memory (per lane) or, for the always-tested list, in the constant
// address space (ConstGeomPtr)
// type, fast: the record's own fields, passed in where the caller knows them without reading the record (a leaf of
// the hierarchy carries them), so that the loads of the record do not wait for its type
template <typename GP>
__device__ __forceinline__ double isect_typed(GP g, int type, int fast, V3 p0, V3 d, double dd, double inv_dd, int &part) {
    part = 0;
    if (fast) return isect_sphere_uniform(g, p0, d, dd, inv_dd);
    switch (type) {
    case PTX_SPHERE: return isect_sphere(&g->m[0], p0, d);
    case PTX_PLANE: return isect_plane(&g->m[0], p0, d);
    case PTX_TRIANGLE: return isect_triangle(&g->m[0], p0, d, part);
    case PTX_CYL: return isect_cyl(&g->m[0], p0, d, part);
    default: return isect_box(&g->m[0], p0, d, part);  // PTX_CUBE, PTX_ROOM
    }
}
template <typename GP>
__device__ __forceinline__ double isect_any(GP g, V3 p0, V3 d, double dd, double inv_dd, int &part) {
    return isect_typed(g, g->type, g->fast, p0, d, dd, inv_dd, part);
}

#ifndef PTX_LIST_UNROLL
#define PTX_LIST_UNROLL 2  // trips of the list loops unrolled (measured: 2 is 3 % better than 1 on BubbleTea)
#endif
// The always-tested primitives against one ray, group by group (DevScene::class_geom): h receives the closest hit
// among them if it beats the one h holds.
__device__ __forceinline__ void list_hits_nb(const DevScene &sc, V3 p0, V3 d, int os, double dd, double inv_dd, Hit &h) {
    // The always-tested primitives, group by group: a loop that knows its intersection routine asks for the part
    // of the record it needs without first waiting for the record's type (the records arrive by scalar loads, and
    // a wave has little else to do while it waits for one).  Equal lambdas go to the lower scene index, as in
    // closest_hit_all, where the primitives are tested in that order.
    const ConstGeomPtr cg = (ConstGeomPtr)sc.class_geom;
    const PTX_CONSTANT_AS int32_t *ids = (const PTX_CONSTANT_AS int32_t *)sc.class_ids;
    auto take = [&](int id, double lam, int part, bool hit) {
        const bool win = hit & (id != os) & (lam > PT_TOL) & ((lam < h.lambda) | ((lam == h.lambda) & (id < h.obj)));
        h.lambda = win ? lam : h.lambda;
        h.obj = win ? id : h.obj;
        h.part = win ? part : h.part;
    };
    int k = 0;
#ifdef PTX_CULL32_PROBE
    // Measurement only (not exact: no error bounds, ties ignored): what a single-precision first pass over the uniformly
    // scaled spheres could save at best.  The nearest sphere is picked in single precision (centre and squared radius as
    // floats in the unused transform words of the record) and only that one is intersected in double precision, its
    // record fetched per lane.  An exact version needs error bounds on top of this pass; this is its lower bound in time.
    {
        const float px = (float)p0.x, py = (float)p0.y, pz = (float)p0.z, dx = (float)d.x, dy = (float)d.y, dz = (float)d.z;
        const float ddf = (float)dd, tol = 1e-9f * ddf;
        float best = INFINITY;
        int arg = -1;
#pragma unroll PTX_LIST_UNROLL
        for (; k < sc.class_end[LC_FAST_SPHERE]; k++) {
            const PTX_CONSTANT_AS float *s = (const PTX_CONSTANT_AS float *)&cg[k].m[0];
            const float ox = px - s[0], oy = py - s[1], oz = pz - s[2];
            const float B = ox * dx + oy * dy + oz * dz, C = ox * ox + oy * oy + oz * oz - s[3];
            const float disc = B * B - ddf * C;
            const float sq = __builtin_amdgcn_sqrtf(fabsf(disc));
            const float t2 = -B - sq;
            const float t = t2 > tol ? t2 : sq - B;
            const bool win = !(disc < 0) & (ids[k] != os) & (t > tol) & (t < best);
            best = win ? t : best;
            arg = win ? k : arg;
        }
        if (arg >= 0) {
            bool missed;
            const double lam = isect_sphere_uniform_nb(sc.class_geom + arg, p0, d, dd, inv_dd, missed);
            const int id = sc.class_ids[arg];
            const bool win = !missed & (lam > PT_TOL) & (lam < h.lambda);
            h.lambda = win ? lam : h.lambda;
            h.obj = win ? id : h.obj;
        }
    }
#endif
    // The first group meets an empty h and its members come in ascending scene index, so a strictly smaller lambda is
    // the whole rule there: of two equal ones the first tested has the lower index.
#pragma unroll PTX_LIST_UNROLL
    for (; k < sc.class_end[LC_FAST_SPHERE]; k++) {
        bool missed;
        const double lam = isect_sphere_uniform_nb(cg + k, p0, d, dd, inv_dd, missed);
        const int id = ids[k];
        const bool win = !missed & (id != os) & (lam > PT_TOL) & (lam < h.lambda);
        h.lambda = win ? lam : h.lambda;
        h.obj = win ? id : h.obj;
    }
#pragma unroll PTX_LIST_UNROLL
    for (; k < sc.class_end[LC_SPHERE]; k++) {
        bool hit;
        const double lam = isect_sphere_nb(&cg[k].m[0], p0, d, hit);
        take(ids[k], lam, 0, hit);
    }
#pragma unroll PTX_LIST_UNROLL
    for (; k < sc.class_end[LC_PLANE]; k++) {
        bool hit;
        const double lam = isect_plane_nb(&cg[k].m[0], p0, d, hit);
        take(ids[k], lam, 0, hit);
    }
#pragma unroll PTX_LIST_UNROLL
    for (; k < sc.class_end[LC_TRIANGLE]; k++) {
        int part;
        bool hit;
        const double lam = isect_triangle_nb(&cg[k].m[0], p0, d, part, hit);
        take(ids[k], lam, part, hit);
    }
#pragma unroll PTX_LIST_UNROLL
    for (; k < sc.class_end[LC_CYL]; k++) {
        int part;
        const double lam = isect_cyl_nb(&cg[k].m[0], p0, d, part);
        take(ids[k], lam, part, true);
    }
#pragma unroll PTX_LIST_UNROLL
    for (; k < sc.class_end[LC_BOX]; k++) {
        int part;
        const double lam = isect_box_nb(&cg[k].m[0], p0, d, part);
        take(ids[k], lam, part, true);
    }
}

// findFirstHit (reference PathTracer.c:198-251): the smallest lambda > TOL wins and, between equal lambdas, the
// primitive the reference visits first (the lower scene index).
//  - The always-tested list is walked with a wave-uniform index, so its records arrive through scalar loads and
//    no lane diverges on the primitive type.
//  - The hierarchy is walked per lane without a stack (PtxBvhNode::skip).  A node is entered if the ray's slab
//    interval is not empty, does not end behind the origin and does not start beyond the best hit so far; the
//    reference's box_intersect (magic.c:457-474) is the same slab test without the last condition.  The node
//    boxes are a little larger than the reference's, which keeps rounding from rejecting one it would enter.
// COUNT: the instrumented build -- wc receives how many list primitives, hierarchy nodes and leaf primitives (by
// type) the query tested.
#define PTX_N_PHASES 12
struct WalkCount {
    uint32_t queries, hits, list_tests, node_visits, leaf_tests[6], texel_lookups;
#ifdef PTX_PHASE_TIMERS
    long long cyc_list, cyc_walk, cyc_leaf;
    double steps_wave, leaf_passes_wave;  // loop trips as the wave sees them: every active lane adds 1/(active lanes)
    // the parts of bounce() and of the loop of k_trace_samples around it (PH_*): cycles, runs and lanes of each, kept
    // per wave in LDS (three counters a part; null: not counted) and updated by the first lane that is in the part
    unsigned long long *ph;
#endif
};
enum { PH_QUERY = 0, PH_SURFACE, PH_DIFFUSE, PH_LIGHT, PH_SHADOW_QUERY, PH_SHADOW_APPLY, PH_MIRROR, PH_REFRACT, PH_NEW_PATH, PH_GLOSSY,
       PH_SAMPLE_END, PH_LOOP };
#ifdef PTX_PHASE_TIMERS
// A part's cycles are what the wave's clock has advanced since the last part ended, whichever lanes ran either: the
// time of the last PH_ADD is kept per wave (ph[PH_LAST]) beside the counters, so every cycle of the loop is counted
// once, under the part whose end comes next.  (The scheduling barriers keep the compiler's scheduler from moving
// arithmetic across the clock reads; what earlier passes sink from one part of the source into a later one -- the new
// direction of a diffuse bounce, which nothing needs before the shadow query is over -- is counted where it runs.)
#define PH_LAST (3 * PTX_N_PHASES - 1)
__device__ __forceinline__ long long ph_clock() {
    __builtin_amdgcn_sched_barrier(0);
    const long long t = clock64();
    __builtin_amdgcn_sched_barrier(0);
    return t;
}
#define PH_T() 0LL
#define PH_ADD(wc, i, t0)                                                                  \
    do {                                                                                   \
        if (COUNT && (wc)->ph) {                                                           \
            const long long now_ = ph_clock();                                             \
            const unsigned long long m_ = __ballot(1);                                     \
            if ((int)(threadIdx.x & 63) == __ffsll((long long)m_) - 1) {                   \
                /* (a clock read the compiler has hoisted above an earlier part's end counts nothing) */ \
                const long long last_ = (long long)atomicMax(&(wc)->ph[PH_LAST], (unsigned long long)now_); \
                const long long dt_ = now_ > last_ ? now_ - last_ : 0;                     \
                atomicAdd(&(wc)->ph[3 * (i)], (unsigned long long)dt_);                    \
                atomicAdd(&(wc)->ph[3 * (i) + 1], 1ULL);                                   \
                atomicAdd(&(wc)->ph[3 * (i) + 2], (unsigned long long)__popcll(m_));       \
            }                                                                              \
            (void)(t0);                                                                    \
        }                                                                                  \
    } while (0)
#else
#define PH_T() 0LL
#define PH_ADD(wc, i, t0) ((void)0)
#endif
// UNIT: d is a unit vector to within rounding (every ray of the render kernels: a bounce starts by normalising its
// direction, a shadow ray's direction is a difference times the reciprocal of its length).  The world-space sphere
// test then takes d.d = 1 instead of computing it, dividing by it and scaling its tolerance with it -- the same
// lambda to within two units in the last place, for a dot product, a reciprocal and two multiplications per sphere
// less.  Rays handed in from outside (ptx_first_hit) may have any length: UNIT = false.
template <bool USE_BVH, bool COUNT = false, bool UNIT = false>
__device__ __forceinline__ Hit closest_hit_all(const DevScene &sc, V3 p0, V3 d, int os, WalkCount *wc = nullptr) {
    Hit h;
    h.lambda = INFINITY;
    h.obj = -1;
    h.part = 0;
    const double dd = UNIT ? 1.0 : dot3(d, d);
    const double inv_dd = UNIT ? 1.0 : fast_rcp(dd);
    const ConstGeomPtr list_geom = (ConstGeomPtr)sc.list_geom;
    const PTX_CONSTANT_AS int32_t *list_ids = (const PTX_CONSTANT_AS int32_t *)sc.list_ids;
#ifdef PTX_PHASE_TIMERS
    long long tm0 = COUNT ? clock64() : 0;
#endif
    if (!USE_BVH) {
        // a scene without a hierarchy: all its primitives, group by group (0.6 % faster on the Cornell box than the
        // loop below, which has to wait for a record's type before it can ask for the rest)
        list_hits_nb(sc, p0, d, os, dd, inv_dd, h);
        if (COUNT) wc->list_tests += sc.n_list;
    } else
    for (int k = 0; k < sc.n_list; k++) {
        int part;
        const int id = USE_BVH ? list_ids[k] : k;
        double lam = isect_any(list_geom + k, p0, d, dd, inv_dd, part);
        if (COUNT) wc->list_tests++;
        if (id != os && lam > PT_TOL && lam < h.lambda) {
            h.lambda = lam;
            h.obj = id;
            h.part = part;
        }
    }
#ifdef PTX_PHASE_TIMERS
    long long tm1 = COUNT ? clock64() : 0;
    if (COUNT) wc->cyc_list += tm1 - tm0;
#endif
    if (USE_BVH) {
        // slab test as b * (1/d) - p0 * (1/d): one fused multiply-add per plane.  It is less exact than
        // (b - p0) / d near b = p0, by far less than the margin the node boxes carry (ptx_bvh.h).  For a direction
        // component of zero the two planes come out hugely negative and hugely positive if p0 lies between them
        // (the slab then never rejects) and both on one side otherwise (the box is missed), as in the reference.
        const double ix = rcp_for_slab(d.x), iy = rcp_for_slab(d.y), iz = rcp_for_slab(d.z);
        const double ox = p0.x * ix, oy = p0.y * iy, oz = p0.z * iz;
        int node = 0;
        const int n_nodes = sc.n_bvh_nodes;
        // the array whose order suits this ray: nearer children first (ptx_bvh.h)
        const int octant = sc.bvh_per_octant ? ((d.x < 0) | (d.y < 0) << 1 | (d.z < 0) << 2) : 0;
        const PtxBvhNode *nodes = sc.bvh + (size_t)octant * (size_t)n_nodes;
        while (node < n_nodes) {
            const PtxBvhNode nd = nodes[node];
            if (COUNT) wc->node_visits++;
#ifdef PTX_PHASE_TIMERS
            if (COUNT) wc->steps_wave += 1.0 / __popcll(__ballot(1));
#endif
            double t0 = fma((double)nd.box[0], ix, -ox), t1 = fma((double)nd.box[1], ix, -ox);
            double t2 = fma((double)nd.box[2], iy, -oy), t3 = fma((double)nd.box[3], iy, -oy);
            double t4 = fma((double)nd.box[4], iz, -oz), t5 = fma((double)nd.box[5], iz, -oz);
            double lmin = fmax(fmax(fmin(t0, t1), fmin(t2, t3)), fmin(t4, t5));
            double lmax = fmin(fmin(fmax(t0, t1), fmax(t2, t3)), fmax(t4, t5));
            int next = nd.skip;
            if (!(lmin > lmax) && !(lmax < 0) && !(lmin > h.lambda)) {
                const int leaf = nd.obj;
                const int id = leaf & PTX_LEAF_INDEX;
                next = node + 1;
                if (leaf >= 0 && id != os) {
                    int part;
                    if (COUNT) wc->leaf_tests[(leaf >> PTX_LEAF_TYPE_SHIFT) & 7]++;
#ifdef PTX_PHASE_TIMERS
                    if (COUNT) wc->leaf_passes_wave += 1.0 / __popcll(__ballot(1));
                    long long tl0 = COUNT ? clock64() : 0;
#endif
                    double lam = isect_typed(&sc.geom[id], leaf >> PTX_LEAF_TYPE_SHIFT, leaf & PTX_LEAF_FAST, p0, d, dd, inv_dd, part);
#ifdef PTX_PHASE_TIMERS
                    if (COUNT) wc->cyc_leaf += clock64() - tl0;
#endif
                    if (lam > PT_TOL && (lam < h.lambda || (lam == h.lambda && id < h.obj))) {
                        h.lambda = lam;
                        h.obj = id;
                        h.part = part;
                    }
                }
            }
            node = next;
        }
#ifdef PTX_PHASE_TIMERS
        if (COUNT) wc->cyc_walk += clock64() - tm1;
#endif
    }
    if (COUNT) {
        wc->queries++;
        if (h.obj >= 0) wc->hits++;
    }
    return h;
}

// Point, normal and (if WANT_UV) texture coordinates of a hit, as the
// reference intersect functions leave them after findFirstHit's final
// normalize (PathTracer.c:242).
template <bool WANT_UV>
__device__ __forceinline__ void finish_hit(const DevScene &sc, const Hit &h, V3 p0, V3 d, V3 &p, V3 &n, double &a,
                                           double &b) {
    const DevGeom &g = sc.geom[h.obj];
    const int type = g.type;
    p = p0 + d * h.lambda;
    a = 0;
    b = 0;
    if (type == PTX_TRIANGLE) {
        V3 nn = mk(g.m[9], g.m[10], g.m[11]);
        n = h.part ? nn * -1.0 : nn;
        return;
    }
    if (g.fast) {
        // canonical point (p-c)/r; Tinv^T is a multiple of the identity, so the normal is its direction
        V3 pc = mk(p.x - g.c[0], p.y - g.c[1], p.z - g.c[2]) * g.e[1];
        n = norm3(pc);
        if (WANT_UV) {
            a = atan2(pc.y, pc.x) / (2 * PT_PI);
            b = acos(pc.z) / PT_PI;
        }
        return;
    }
    if (!WANT_UV && type == PTX_PLANE) {
        // planeIntersect picks the side the ray starts on (reference utils_path.c:505-519)
        double tz = g.m[8] * p0.x + g.m[9] * p0.y + g.m[10] * p0.z + g.m[11];
        const double *nn = tz < 0 ? g.e : g.c;
        n = mk(nn[0], nn[1], nn[2]);
        return;
    }
    double m[12];
#pragma unroll
    for (int q = 0; q < 12; q++) m[q] = g.m[q];
    V3 tp = xform_point(m, p0), td = xform_dir(m, d);
    V3 pc = tp + td * h.lambda;
    V3 nc;
    if (type == PTX_SPHERE) {
        nc = norm3(pc);
        if (WANT_UV) {
            a = atan2(pc.y, pc.x) / (2 * PT_PI);
            b = acos(pc.z) / PT_PI;
        }
    } else if (type == PTX_PLANE) {
        nc = mk(0, 0, tp.z < 0 ? -1.0 : 1.0);
        if (WANT_UV) {
            a = (1 + pc.x) / 2;
            b = (1 + pc.y) / 2;
        }
    } else if (type == PTX_CYL) {
        if (h.part == 0) {
            nc = mk(pc.x, pc.y, 0);
            if (WANT_UV) {
                a = (atan2(pc.y, pc.x) + PT_PI) / (2 * PT_PI);
                b = (pc.z + 1) / 2;
            }
        } else {
            nc = mk(0, 0, h.part == 1 ? -1.0 : 1.0);
            if (WANT_UV) {
                a = pc.x / 2 + 0.5;
                b = pc.y / 2 + 0.5;
            }
        }
    } else {  // cube, room
        const int f = h.part;
        const int ax = 2 - (f >> 1);
        double s = (f & 1) ? -1.0 : 1.0;
        if (type == PTX_ROOM) s = -s;
        nc = mk(ax == 0 ? s : 0.0, ax == 1 ? s : 0.0, ax == 2 ? s : 0.0);
        if (WANT_UV && type == PTX_CUBE) {
            if (ax == 2) { a = (1 + pc.x) / 2; b = (1 + pc.y) / 2; }
            else if (ax == 1) { a = (1 + pc.x) / 2; b = (1 + pc.z) / 2; }
            else { a = (1 + pc.y) / 2; b = (1 + pc.z) / 2; }
        }
    }
    n = xform_normal(m, nc);
}

// The same with the head of the primitive's record fetched beforehand.  finish_hit asks for the record piece by piece as
// its branches are reached: the type first, then what that type needs -- two or three requests one after the other, each
// a trip to the cache that the lane waits for.  The kernels for scenes with a hierarchy wait for memory more than for
// anything else, so shade_hit and apply_shadow request, in one go and together with the material record, everything
// that settles the common hits (triangles, uniformly scaled spheres, planes without texture coordinates); only the
// general case fetches the rest of the transform afterwards.  Same operations on the same values as finish_hit.
struct GeomHead {
    int type, fast;
    double m8, m9, m10, m11;  // the last row of Tinv; a triangle's unit normal in m9..m11
    double c[3], e[3];
};
// Keeps a value that has been requested from memory from being requested later instead: the compiler moves a load that
// has a single use into the branch that uses it, which is where the wait for it would then be.
template <typename T>
__device__ __forceinline__ void keep_here(T &v) {
    asm volatile("" : "+v"(v));
}
__device__ __forceinline__ GeomHead geom_head(const DevGeom &g) {
    GeomHead gh;
    gh.type = g.type;
    gh.fast = g.fast;
    gh.m8 = g.m[8];
    gh.m9 = g.m[9];
    gh.m10 = g.m[10];
    gh.m11 = g.m[11];
#pragma unroll
    for (int q = 0; q < 3; q++) {
        gh.c[q] = g.c[q];
        gh.e[q] = g.e[q];
    }
    return gh;
}
__device__ __forceinline__ void keep_here(GeomHead &gh) {
    keep_here(gh.type);
    keep_here(gh.fast);
    keep_here(gh.m8);
    keep_here(gh.m9);
    keep_here(gh.m10);
    keep_here(gh.m11);
#pragma unroll
    for (int q = 0; q < 3; q++) {
        keep_here(gh.c[q]);
        keep_here(gh.e[q]);
    }
}
template <bool WANT_UV>
__device__ __forceinline__ void finish_hit_pre(const DevScene &sc, const Hit &h, const GeomHead &gh, V3 p0, V3 d, V3 &p, V3 &n,
                                               double &a, double &b) {
    const int type = gh.type;
    p = p0 + d * h.lambda;
    a = 0;
    b = 0;
    if (type == PTX_TRIANGLE) {
        V3 nn = mk(gh.m9, gh.m10, gh.m11);
        n = h.part ? nn * -1.0 : nn;
        return;
    }
    if (gh.fast) {
        V3 pc = mk(p.x - gh.c[0], p.y - gh.c[1], p.z - gh.c[2]) * gh.e[1];
        n = norm3(pc);
        if (WANT_UV) {
            a = atan2(pc.y, pc.x) / (2 * PT_PI);
            b = acos(pc.z) / PT_PI;
        }
        return;
    }
    if (!WANT_UV && type == PTX_PLANE) {
        double tz = gh.m8 * p0.x + gh.m9 * p0.y + gh.m10 * p0.z + gh.m11;
        n = tz < 0 ? mk(gh.e[0], gh.e[1], gh.e[2]) : mk(gh.c[0], gh.c[1], gh.c[2]);
        return;
    }
    finish_hit<WANT_UV>(sc, h, p0, d, p, n, a, b);  // the general case, with the whole record
}

// ------------------------------------------------------ texture look-ups
// byte / divisor, rounded as the IEEE division the reference performs when it loads an image
// (utils_path.c:1849, :1921): quotient from the reciprocal (inv = fast_rcp(divisor)), then one correction with
// the exact remainder.  Equal to the division for every byte and every integer divisor 1..255 (checked
// exhaustively by ptx_selftest).
__device__ __forceinline__ double texel_value(double byte, double divisor, double inv) {
    const double q = byte * inv;
    return fma(fma(-q, divisor, byte), inv, q);
}
// texel (flat index idx of the file's order, clamped to the image buffer where the reference reads out of bounds)
// TILES: the kernel may meet tiled images (the kernels for scenes with a hierarchy; ptx_scene_upload tiles for those only)
template <bool TILES>
__device__ __forceinline__ double texel_at(const DevScene &sc, const DevImage &im, double inv, long idx, long n) {
    idx = idx < 0 ? 0 : (idx >= n ? n - 1 : idx);
    long at = idx;
    if (TILES && im.tiles_x > 0) {  // tiled: from the flat index back to texel and channel
        const long t = idx / 3;
        const int ch = (int)(idx - 3 * t), y = (int)(t / im.sx), x = (int)(t - (long)y * im.sx);
        at = tiled_texel(im, x, y) + ch;
    }
    return texel_value((double)sc.texels[im.offset + at], im.divisor, inv);
}
// texMap, reference utils_path.c:1322-1374 (flat index clamped to the image
// buffer where the reference reads out of bounds).
// WIDE: the kernel is one for scenes with a hierarchy: images may be tiled, and the two texels of a row are fetched
// with one load where all four lie inside the image.  The kernels for
// hierarchy scenes, which spend much of their time waiting for memory, gain 3 % by it; in the kernels for small
// scenes the extra code only costs registers (1.5 % on the Cornell box, which has no textures at all).
template <bool WIDE>
__device__ __forceinline__ void tex_map(const DevScene &sc, int img, double a, double b, double &R, double &G, double &B) {
    const DevImage im = sc.images[img];
    double x = (im.sx - 1) * a, y = (im.sy - 1) * b;
    int x1 = (int)x, y1 = (int)y;
    int x2 = x1 + 1, y2 = y1 + 1;
    long n = (long)im.sx * im.sy * 3;
    const double inv = fast_rcp(im.divisor);
    double out[3];
    if (WIDE && x1 >= 0 && y1 >= 0 && x2 < im.sx && y2 < im.sy) {
        // all four texels inside the image; each arrives as the low three bytes of a word (the texel buffer ends in
        // spare bytes).  File order: the two texels of a row are six adjacent bytes, one 8-byte load.  Tiled: a
        // 4-byte load per texel (the second texel of a row may be the first of the next tile), the four mostly in
        // one cache line.
        const uint8_t *base = sc.texels + im.offset;
        uint32_t t11, t21, t12, t22;
        if (im.tiles_x > 0) {
            __builtin_memcpy(&t11, base + tiled_texel(im, x1, y1), 4);
            __builtin_memcpy(&t21, base + tiled_texel(im, x2, y1), 4);
            __builtin_memcpy(&t12, base + tiled_texel(im, x1, y2), 4);
            __builtin_memcpy(&t22, base + tiled_texel(im, x2, y2), 4);
        } else {
            uint64_t row1, row2;
            __builtin_memcpy(&row1, base + 3L * ((long)y1 * im.sx + x1), 8);
            __builtin_memcpy(&row2, base + 3L * ((long)y2 * im.sx + x1), 8);
            t11 = (uint32_t)row1;
            t21 = (uint32_t)(row1 >> 24);
            t12 = (uint32_t)row2;
            t22 = (uint32_t)(row2 >> 24);
        }
#pragma unroll
        for (int c = 0; c < 3; c++) {
            double c11 = texel_value((double)((t11 >> (8 * c)) & 255u), im.divisor, inv);
            double c21 = texel_value((double)((t21 >> (8 * c)) & 255u), im.divisor, inv);
            double c12 = texel_value((double)((t12 >> (8 * c)) & 255u), im.divisor, inv);
            double c22 = texel_value((double)((t22 >> (8 * c)) & 255u), im.divisor, inv);
            double avg1 = c11 * (x2 - x) + c21 * (x - x1);
            double avg2 = c12 * (x2 - x) + c22 * (x - x1);
            out[c] = avg1 * (y2 - y) + avg2 * (y - y1);
        }
    } else {
#pragma unroll
        for (int c = 0; c < 3; c++) {
            double c11 = texel_at<WIDE>(sc, im, inv, 3L * ((long)y1 * im.sx + x1) + c, n);
            double c12 = texel_at<WIDE>(sc, im, inv, 3L * ((long)y2 * im.sx + x1) + c, n);
            double c21 = texel_at<WIDE>(sc, im, inv, 3L * ((long)y1 * im.sx + x2) + c, n);
            double c22 = texel_at<WIDE>(sc, im, inv, 3L * ((long)y2 * im.sx + x2) + c, n);
            double avg1 = c11 * (x2 - x) + c21 * (x - x1);
            double avg2 = c12 * (x2 - x) + c22 * (x - x1);
            out[c] = avg1 * (y2 - y) + avg2 * (y - y1);
        }
    }
    R = out[0];
    G = out[1];
    B = out[2];
}
// alphaMap, reference utils_path.c:1376-1406
__device__ __forceinline__ double alpha_map(const DevScene &sc, int img, double a, double b) {
    const DevImage im = sc.images[img];
    double x = (im.sx - 1) * a, y = (im.sy - 1) * b;
    int x1 = (int)x, y1 = (int)y;
    int x2 = x1 + 1, y2 = y1 + 1;
    long n = (long)im.sx * im.sy;
    const double inv = fast_rcp(im.divisor);
    double a11 = texel_at<false>(sc, im, inv, (long)y1 * im.sx + x1, n);
    double a12 = texel_at<false>(sc, im, inv, (long)y2 * im.sx + x1, n);
    double a21 = texel_at<false>(sc, im, inv, (long)y1 * im.sx + x2, n);
    double a22 = texel_at<false>(sc, im, inv, (long)y2 * im.sx + x2, n);
    double avg1 = a11 * (x2 - x) + a21 * (x - x1);
    double avg2 = a12 * (x2 - x) + a22 * (x - x1);
    return avg1 * (y2 - y) + avg2 * (y - y1);
}

// Rotation that takes +z to the unit vector n: Rz(theta) * Ry(phi) with
// theta = atan2(n.y, n.x), phi = acos(n.z), as built by cosWeightedSample
// (reference utils_path.c:1253-1264) and vectorReorient (:98-114).  The
// sines and cosines of those angles are taken from n directly.
__device__ __forceinline__ V3 rotate_z_to(V3 n, V3 v) {
    double h2 = n.x * n.x + n.y * n.y;
    double ct = 1.0, st = 0.0, sp = 0.0;  // atan2(0,0) = 0
    if (h2 > 0) {
        double ih = fast_rsqrt(h2);
        ct = n.x * ih;
        st = n.y * ih;
        sp = h2 * ih;  // sin(acos(n.z)) of a unit vector
    } else if (n.x < 0 || (n.x == 0 && signbit(n.x))) {
        ct = -1.0;  // atan2(+-0, -0) = +-pi
    }
    // cos(acos(n.z)): a unit vector whose z exceeds 1 by a rounding error (where the reference's acos would
    // give NaN) is taken as it is
    const double cp = n.z;
    double x1 = v.x * cp + v.z * sp;
    double z1 = -v.x * sp + v.z * cp;
    return mk(x1 * ct - v.y * st, x1 * st + v.y * ct, z1);
}

// normalDeform, reference utils_path.c:1408-1428
template <bool WIDE>
__device__ __forceinline__ V3 normal_deform(const DevScene &sc, int img, double a, double b, V3 n) {
    double r, g, bb;
    tex_map<WIDE>(sc, img, a, b, r, g, bb);
    V3 nd = norm3(mk(r * 2 - 1, g * 2 - 1, bb * 2 - 1));
    return rotate_z_to(norm3(n), nd);
}

// sin and cos of 2*PT_PI*u for u in [0,1)
__device__ __forceinline__ void sincos_turn(double u, double &s, double &c) { sincospi(PT_TWO_PI_OVER_PI * u, &s, &c); }

// cosWeightedSample, reference utils_path.c:1224-1265 (two draws)
__device__ __forceinline__ V3 cos_weighted_sample(V3 n, Rng &rng) {
    double u1 = rng.next();
    double r = fast_sqrt(u1);
    double u2 = rng.next();
    double s, c;
    sincos_turn(u2, s, c);
    double x = r * c, y = r * s;
    double z = fast_sqrt(1.0 - x * x - y * y);
    return rotate_z_to(n, mk(x, y, z));
}

// The uniform-hemisphere branch of PathTrace (reference PathTracer.c:366-387)
// with hemisphereReorient/reorient (utils_path.c:79-96): four draws, two
// discarded.
__device__ __forceinline__ V3 uniform_hemisphere_sample(V3 n, Rng &rng) {
    double c2 = 2 * rng.next() - 1;  // cos(angle_2)
    rng.skip(2);
    double u = rng.next();
    double s2 = sqrt(fmax(0.0, 1.0 - c2 * c2));  // sin(acos(c2))
    double s1, c1;
    sincos_turn(u, s1, c1);
    V3 v = norm3(mk(s1 * s2, fabs(c2), -c1 * s2));
    // Rodrigues rotation taking (0,1,0) to n
    V3 up = mk(0, 1, 0);
    V3 axis = norm3(cross3(up, n));
    double ca = n.y;  // cos(acos(up.n))
    double sa = sqrt(fmax(0.0, 1.0 - ca * ca));
    return v * ca + cross3(axis, v) * sa + axis * (dot3(axis, v) * (1 - ca));
}

// random_normal, reference PathTracer.c:46-51 (first draw feeds log, second cos)
__device__ __forceinline__ double random_normal(Rng &rng) {
    double ul = rng.next();
    double uc = rng.next();
    return fast_sqrt(-2 * lean_log(ul)) * cospi(PT_TWO_PI_OVER_PI * uc);
}

// ------------------------------------------------------------ path state
// What survives from one bounce to the next: the live fields of the
// reference's struct ray3D (PathTracer.h:76-90) plus PathTrace's Os and CEL
// arguments.  The refraction-index stack and the normal of the previous
// bounce live in the lane's LDS slots.
struct Path {
    V3 p0, d, col, I;
    Rng rng;
    int os, cel, depth;
    int ridx;  // r_index_index
};

struct Counters {
    uint32_t closest, shadow, oob;
};

// Loads/stores of ray->r_index[k] as the reference performs them, including
// the out-of-range indices refractRay can produce (PathTracer.c:101, :131),
// which land on neighbouring fields of struct ray3D: k=-1 srcN.pw (always 1),
// -2..-4 srcN.pz..px, -5..-7 I.B..R, -8..-10 col.B..R, 10 = r_index_index.
__device__ inline double rindex_load(const DevScene &sc, Path &ps, const LaneStore &ls, int k, uint32_t &oob) {
    if (k >= 0 && k <= 9) {
        int id = ls.geti(LS_RINDEX + k);
        return id >= 0 ? sc.mat[id].r_index : (id == RI_ONE ? 1.0 : 0.0);
    }
    oob++;
    switch (k) {
    case -1: return 1.0;
    case -2: return ls.get(LS_SRCN + 2);
    case -3: return ls.get(LS_SRCN + 1);
    case -4: return ls.get(LS_SRCN + 0);
    case -5: return ps.I.z;
    case -6: return ps.I.y;
    case -7: return ps.I.x;
    case -8: return ps.col.z;
    case -9: return ps.col.y;
    case -10: return ps.col.x;
    case 10: return __longlong_as_double((long long)(uint32_t)ps.ridx);
    default: return 1.0;
    }
}
// stores obj's r_index (value v) at r_index[k]
__device__ inline void rindex_store(Path &ps, const LaneStore &ls, int k, int obj, double v, uint32_t &oob) {
    if (k >= 0 && k <= 9) {
        ls.puti(LS_RINDEX + k, obj);
        return;
    }
    oob++;
    switch (k) {
    case -5: ps.I.z = v; break;
    case -6: ps.I.y = v; break;
    case -7: ps.I.x = v; break;
    case -8: ps.col.z = v; break;
    case -9: ps.col.y = v; break;
    case -10: ps.col.x = v; break;
    case 10: ps.ridx = (int)(uint32_t)(__double_as_longlong(v) & 0xFFFFFFFFLL); break;
    default: break;  // -1..-4 land on srcN, which PathTrace overwrites right after (:482)
    }
}

// reflectRay, reference PathTracer.c:53-72, for a unit direction and a unit normal.  The reference normalises both on
// the way in and the result on the way out; the callers' d has been normalised at the top of the bounce and their
// normals come normalised from finish_hit, the mirror image of a unit vector in a unit normal is a unit vector, and the
// next bounce begins by normalising its direction (PathTracer.c:279) -- so the three normalisations would each
// multiply by one to within rounding, and are left out (a third of the instructions of a mirror bounce).
__device__ __forceinline__ V3 reflect_dir(V3 d, V3 n) {
    double dn = dot3(d, n);
    return mk(d.x - 2 * dn * n.x, d.y - 2 * dn * n.y, d.z - 2 * dn * n.z);
}

// refractRay, reference PathTracer.c:74-140.  The reference normalises the direction and the normal on the way in and
// the new direction on the way out; here the direction has been normalised at the top of the bounce, the normal comes
// normalised from finish_hit, the refracted direction r d + (r c - sqrt(1 - r^2 (1 - c^2))) n of unit d and n is a unit
// vector, and the next bounce normalises its direction first thing: those normalisations are left out (see reflect_dir).
__device__ inline void refract_ray(const DevScene &sc, Path &ps, const LaneStore &ls, V3 p, V3 &n, int obj, double obj_r_index,
                                   Counters &cnt) {
    double n1 = rindex_load(sc, ps, ls, ps.ridx, cnt.oob);
    double n2 = obj_r_index;
    bool leaving = false;
    V3 rn = n;
    if (!(dot3(ps.d, n) < 0)) {
        n1 = obj_r_index;
        n2 = rindex_load(sc, ps, ls, ps.ridx - 1, cnt.oob);
        rn = n * -1.0;
        leaving = true;
    }
    // (the two quotients as products with reciprocals: a division is three times the instructions, and few lanes of
    // a wave are here at a time)
    const double r = n1 * fast_rcp(n2);
    double c = -dot3(rn, ps.d);
    double disc = 1 - (r * r) * (1 - (c * c));
    double q = (n1 - n2) * fast_rcp(n1 + n2);
    double r0 = q * q;
    double omc = 1 - c;
    double reflectance = r0 + ((1 - r0) * (omc * omc * omc * omc * omc));
    double dice = ps.rng.next();
    ps.p0 = p;
    if (disc < 0 || dice <= reflectance) {
        ps.d = reflect_dir(ps.d, rn);
        return;
    }
    if (leaving) {
        ps.ridx = ps.ridx - 1;
    } else {
        ps.ridx = ps.ridx + 1;
        rindex_store(ps, ls, ps.ridx, obj, obj_r_index, cnt.oob);
    }
    ps.d = ps.d * r + rn * (r * c - fast_sqrt(disc));

}

// Everything of a new path except its ray: initRay (reference utils_path.h:211-230) and PathTrace's initial
// arguments (PathTracer.c:644).
__device__ __forceinline__ void path_init(const DevScene &sc, const LaneStore &ls, Path &ps) {
    ps.col = mk(1, 1, 1);
    ps.I = mk(0, 0, 0);
    ps.os = -1;
    ps.cel = 0;
    ps.depth = 1;
    ps.ridx = 0;
    if (sc.has_refraction) {
        ls.puti(LS_RINDEX + 0, RI_ONE);
#pragma unroll
        for (int q = 1; q < 10; q++) ls.puti(LS_RINDEX + q, RI_ZERO);
        ls.put(LS_SRCN + 0, 0.0);
        ls.put(LS_SRCN + 1, 0.0);
        ls.put(LS_SRCN + 2, 1.0);
    }
}

// Primary ray of pixel (i,j), sample k: reference PathTracer.c:625-641 with
// sampleLensPoint (:142-158) and focusRay (:160-195).  Two draws are consumed
// whatever the aperture; with a zero aperture they do not enter the result.
__device__ inline void primary_ray(const DevScene &sc, const DevRender &rp, const LaneStore &ls, int i, int j, int k,
                                   Path &ps) {
    ps.rng.ctr = ptx_path_key(rp.seed, (uint64_t)j * (uint64_t)rp.sx + (uint64_t)i, (uint64_t)k);
    const DevCamera &cam = sc.cam;
    V3 pc = mk(cam.wl + (i * rp.du), cam.wt + (j * rp.dv), cam.f);
    if (rp.clean) {  // PTX_CLEAN: a random point of the pixel, not its centre
        const double ju = ps.rng.next(), jv = ps.rng.next();
        pc.x = cam.wl + ((i + (ju - 0.5)) * rp.du);
        pc.y = cam.wt + ((j + (jv - 0.5)) * rp.dv);
    }
    V3 pl = pc;
    if (rp.aperture != 0) {
        double r = rp.aperture * sqrt(ps.rng.next());
        double u = ps.rng.next();
        double s, c;
        sincos_turn(u, s, c);
        pl.x = pc.x + r * c;
        pl.y = pc.y + r * s;
    } else {
        ps.rng.skip(2);
    }
    pl = xform_point(cam.C2W, pl);
    pc = xform_point(cam.C2W, pc);
    V3 e = mk(cam.e[0], cam.e[1], cam.e[2]), w = mk(cam.w[0], cam.w[1], cam.w[2]);
    V3 fp = mk(cam.fp[0], cam.fp[1], cam.fp[2]);
    V3 rayd = norm3(pc - e);
    double lambda = dot3(fp - pc, w) / dot3(rayd, w);
    V3 hit = pc + rayd * lambda;
    ps.p0 = pl;
    ps.d = norm3(hit - pl);
    path_init(sc, ls, ps);
}

// The same when the direction has been computed before and parked in the lane's LDS slots (zero aperture only).
__device__ __forceinline__ void primary_ray_cached(const DevScene &sc, const DevRender &rp, const LaneStore &ls, int i, int j,
                                                   int k, Path &ps) {
    ps.rng.ctr = ptx_path_key(rp.seed, (uint64_t)j * (uint64_t)rp.sx + (uint64_t)i, (uint64_t)k);
    ps.rng.skip(2);
    const DevCamera &cam = sc.cam;
    ps.p0 = xform_point(cam.C2W, mk(cam.wl + (i * rp.du), cam.wt + (j * rp.dv), cam.f));
    ps.d = mk(ls.get(LS_PRIMARY_D + 0), ls.get(LS_PRIMARY_D + 1), ls.get(LS_PRIMARY_D + 2));
    path_init(sc, ls, ps);
}

// Light-point samplers: planeSample / sphereSample (reference
// utils_path.c:1189-1209).  A cylinder light draws twice and yields no point
// in the reference (cylCoordinates, :1179-1187, returns nothing); (0,0,0) is
// used.  Other light types have no sampler there; (0,0,0) without a draw.
__device__ __forceinline__ V3 light_sample(const DevLight &ls, Rng &rng) {
    if (ls.type == PTX_PLANE) {
        double v1 = rng.next(), v2 = rng.next();
        return xform_point(ls.T, mk(v1 * 2 - 1, v2 * 2 - 1, 0));
    }
    if (ls.type == PTX_SPHERE) {
        double v1 = rng.next(), v2 = rng.next();
        double cphi = 2 * v2 - 1;  // cos(acos(.))
        double sphi = sqrt(fmax(0.0, 1.0 - cphi * cphi));
        double s, c;
        sincos_turn(v1, s, c);
        return xform_point(ls.T, mk(c * sphi, s * sphi, cphi));
    }
    if (ls.type == PTX_CYL) rng.skip(2);
    return mk(0, 0, 0);
}

// One invocation of PathTrace (reference PathTracer.c:253-484) for a live
// path whose depth does not exceed MAX_DEPTH: returns true when the path has
// ended, its colour then being ps.I.  The reference's depth test at the top
// of the next invocation (:273) is applied at the end of this one, so a path
// that has used up its depth ends without another trip through the loop.
// With DEFER_GLOSSY the three normal deviates of a glossy mirror bounce (:460-464) are not drawn here: the
// lane reports its refl_sig in `glossy` (0 otherwise) and the caller adds them with glossy_service().
// ALLOW_CLEAN: the kernel honours rp.clean (PTX_CLEAN); the kernel specialised for a zero aperture is never asked
// to, because a jittered primary ray differs from sample to sample like one through a lens.
template <int FLAGS, bool USE_BVH, bool DEFER_GLOSSY, bool COUNT = false, bool ALLOW_CLEAN = true>
__device__ inline bool bounce(const DevScene &sc, const DevRender &rp, const LaneStore &ls, Path &ps, Counters &cnt,
                              double &glossy, WalkCount *wc = nullptr) {
    const bool clean = ALLOW_CLEAN && rp.clean != 0;
    glossy = 0;
    long long ph_t = PH_T();
    ps.d = norm3(ps.d);
    cnt.closest++;
    Hit h = closest_hit_all<USE_BVH, COUNT, true>(sc, ps.p0, ps.d, ps.os, wc);
    PH_ADD(wc, PH_QUERY, ph_t);
    if (h.obj < 0) return true;
    const DevMat &mt = sc.mat[h.obj];
    const int tex_img = mt.tex_img, normal_img = mt.normal_img, alpha_img = mt.alpha_img;
    V3 p, n;
    double a, b;
    if ((tex_img & normal_img & alpha_img) >= 0) finish_hit<true>(sc, h, ps.p0, ps.d, p, n, a, b);  // any index >= 0
    else finish_hit<false>(sc, h, ps.p0, ps.d, p, n, a, b);
    double R = mt.col[0], G = mt.col[1], B = mt.col[2];
    if (tex_img >= 0) tex_map<USE_BVH>(sc, tex_img, a, b, R, G, B);
    if (normal_img >= 0) n = normal_deform<USE_BVH>(sc, normal_img, a, b, n);
    if (COUNT) wc->texel_lookups += (tex_img >= 0) + (normal_img >= 0) + (alpha_img >= 0 && !mt.is_light);
    if (mt.is_light) {
        if (!ps.cel) ps.I = mk(ps.I.x + ps.col.x * R, ps.I.y + ps.col.y * G, ps.I.z + ps.col.z * B);
        return true;
    }
    double rr = 0;
    if (FLAGS & PTX_USE_RR) {
        double mx = ps.col.x < ps.col.z ? ps.col.z : ps.col.x;  // max(max(R,B),G), reference PathTracer.c:324
        mx = mx < ps.col.y ? ps.col.y : mx;
        rr = 1 - mx;
        if (rr >= .98) rr = .98;
    }
    if (rr > ps.rng.next()) return true;
    double diffusivity = mt.diffPct, reflectivity = mt.reflPct, transparency = mt.tranPct;
    if (alpha_img >= 0) {
        transparency = alpha_map(sc, alpha_img, a, b);
        double both = reflectivity + diffusivity;
        reflectivity = (1 - transparency) * mt.reflPct / both;
        diffusivity = (1 - transparency) * mt.diffPct / both;
    }
    const double inv_keep = (FLAGS & PTX_USE_RR) ? fast_rcp(1 - rr) : 1.0;
    double dice = ps.rng.next();
    int next_os = h.obj;
    PH_ADD(wc, PH_SURFACE, ph_t);
    if (dice <= diffusivity) {
        V3 dir;
        double scale;
        if (FLAGS & PTX_USE_IS) {
            dir = cos_weighted_sample(n, ps.rng);  // a unit vector (rotation of (x, y, sqrt(1 - x^2 - y^2))); normalised at the next bounce's top
            scale = inv_keep;  // (n.dir) / probability with probability = n.dir, reference PathTracer.c:364, :390
        } else {
            dir = uniform_hemisphere_sample(n, ps.rng);
            scale = dot3(n, dir) * inv_keep;
        }
        ps.col = mk(ps.col.x * (R * scale), ps.col.y * (G * scale), ps.col.z * (B * scale));
        ps.p0 = p;
        ps.d = dir;
        PH_ADD(wc, PH_DIFFUSE, ph_t);
        if (FLAGS & PTX_USE_ES) {
                double u = ps.rng.next();
            int li = 0;
            double acc = sc.lights[0].weight;
            while (acc < u && li + 1 < sc.n_lights) {
                li++;
                acc += sc.lights[li].weight;
            }
            V3 lp = light_sample(sc.lights[li], ps.rng);
            V3 dl = lp - p;
            double d2 = dot3(dl, dl);
            const double inv_len = fast_rsqrt(d2);
            V3 ed = dl * inv_len;
            cnt.shadow++;
            PH_ADD(wc, PH_LIGHT, ph_t);
                Hit sh = closest_hit_all<USE_BVH, COUNT, true>(sc, p, ed, h.obj, wc);
            PH_ADD(wc, PH_SHADOW_QUERY, ph_t);
                int lit = 0;
            if (sh.obj >= 0) {
                const DevMat &lm = sc.mat[sh.obj];
                if (lm.is_light) {
                    V3 lpnt, ln;
                    double la, lb;
                    finish_hit<false>(sc, sh, p, ed, lpnt, ln, la, lb);
                    double weight2 = dot3(n, ed) * -dot3(ln, ed);
                    double wgt = (lm.surfaceArea * weight2) * (inv_len * inv_len);  // 1 / d2
                    wgt = 1 > wgt ? wgt : 1;  // min(1, .), reference PathTracer.c:437
                    if (clean && wgt < 0) wgt = 0;
                    wgt *= inv_keep;
                    ps.I = mk(ps.I.x + ps.col.x * R * lm.col[0] * wgt, ps.I.y + ps.col.y * G * lm.col[1] * wgt,
                              ps.I.z + ps.col.z * B * lm.col[2] * wgt);
                    lit = 1;
                }
            }
            ps.cel = clean ? 1 : lit;
            PH_ADD(wc, PH_SHADOW_APPLY, ph_t);
        }
    } else if (dice <= diffusivity + reflectivity) {
        ph_t = PH_T();  // (a branch of its own: the lanes in it have waited for the diffuse branch)
        if (clean) ps.cel = 0;
        ps.col = mk(ps.col.x * (R * inv_keep), ps.col.y * (G * inv_keep), ps.col.z * (B * inv_keep));
        ps.d = reflect_dir(ps.d, n);
        ps.p0 = p;
        const double sig = mt.refl_sig;
        if (sig > 0) {
            if (DEFER_GLOSSY) {
                glossy = sig;
            } else {
                ps.d.x += sig * random_normal(ps.rng);
                ps.d.y += sig * random_normal(ps.rng);
                ps.d.z += sig * random_normal(ps.rng);
            }
        }
        PH_ADD(wc, PH_MIRROR, ph_t);
    } else if (dice <= diffusivity + reflectivity + transparency) {
        ph_t = PH_T();
        if (clean) ps.cel = 0;
        ps.col = mk(ps.col.x * (R * inv_keep), ps.col.y * (G * inv_keep), ps.col.z * (B * inv_keep));
        refract_ray(sc, ps, ls, p, n, h.obj, mt.r_index, cnt);
        next_os = -1;
        PH_ADD(wc, PH_REFRACT, ph_t);
    }
    ph_t = PH_T();  // (the lanes meet again here)
    if (sc.has_refraction) {
        ls.put(LS_SRCN + 0, n.x);
        ls.put(LS_SRCN + 1, n.y);
        ls.put(LS_SRCN + 2, n.z);
    }
    ps.os = next_os;
    ps.depth++;
    PH_ADD(wc, PH_DIFFUSE, ph_t);  // (the end of the bounce: work the compiler moved behind the last query, the new direction mostly)
    return ps.depth > rp.max_depth;
}

// ---------------------------------------------------- PathTrace in two halves, for the kernels of large scenes
// In a scene with a hierarchy a closest-hit query takes a different number of steps for every ray, so the render
// kernel for such scenes does not run queries inside bounce(): every lane carries one query at a time through
// slices of the walk (k_trace_samples_bvh), and PathTrace is cut where its queries are:
//   shade_hit()    everything bounce() does after its first query, up to the point where the explicit light
//                  sample needs the second one (reference PathTracer.c:285-428);
//   apply_shadow() what follows that second query (:428-445).
// Same operations in the same order as bounce(), so both give the same bits.
enum { SH_END = 0, SH_NEXT = 1, SH_SHADOW = 2 };
// What the explicit light sample keeps while its shadow query runs.  shade_hit leaves n_ed, rcp_d2, inv_keep, cr and cg
// in the lane's LDS slots LS_PEND.. at the point where it computes them (shadow_pending_load fetches them for
// apply_shadow) and returns only ed and cb in registers: the walk and the primitive tests of the query need the
// registers, and values that the compiler moves to scratch memory instead cost a store to device memory each.
struct ShadowPending {
    V3 ed;          // unit direction towards the sampled light point (the shadow ray starts at the path's new p0)
    double n_ed;    // n . ed
    double rcp_d2;  // 1 / squared distance to the sampled point
    double cr, cg, cb;  // throughput * surface colour, per channel
    double inv_keep;    // 1 / (1 - RR_prob)
};

// h: result of the closest-hit query along ps.p0 + lambda * ps.d (ps.d normalised before the query, os = ps.os).
// SH_END: the path is over, its colour is ps.I.  SH_NEXT: ps holds the next ray.  SH_SHADOW: ps holds the next
// ray and `pend` the explicit light sample whose shadow ray (from ps.p0 along pend.ed, excluding ps.os) must be
// resolved with apply_shadow() before the path goes on; *over is set if the path then ends (depth used up).
template <int FLAGS>
__device__ inline int shade_hit(const DevScene &sc, const DevRender &rp, const LaneStore &ls, Path &ps, Counters &cnt, const Hit &h,
                                double &glossy, ShadowPending &pend, bool &over) {
    glossy = 0;
    over = false;
    if (h.obj < 0) return SH_END;
    const DevMat &mt = sc.mat[h.obj];
    // the material and the head of the primitive's record in one request (see finish_hit_pre)
    GeomHead gh = geom_head(sc.geom[h.obj]);
    int tex_img = mt.tex_img, normal_img = mt.normal_img, alpha_img = mt.alpha_img, is_light = mt.is_light;
    double R = mt.col[0], G = mt.col[1], B = mt.col[2];
    double mt_diff = mt.diffPct, mt_refl = mt.reflPct, mt_tran = mt.tranPct;
    keep_here(gh);
    keep_here(tex_img); keep_here(normal_img); keep_here(alpha_img); keep_here(is_light);
    keep_here(R); keep_here(G); keep_here(B);
    keep_here(mt_diff); keep_here(mt_refl); keep_here(mt_tran);
    V3 p, n;
    double a, b;
    if ((tex_img & normal_img & alpha_img) >= 0) finish_hit_pre<true>(sc, h, gh, ps.p0, ps.d, p, n, a, b);  // any index >= 0
    else finish_hit_pre<false>(sc, h, gh, ps.p0, ps.d, p, n, a, b);
    if (tex_img >= 0) tex_map<true>(sc, tex_img, a, b, R, G, B);
    if (normal_img >= 0) n = normal_deform<true>(sc, normal_img, a, b, n);
    if (is_light) {
        if (!ps.cel) ps.I = mk(ps.I.x + ps.col.x * R, ps.I.y + ps.col.y * G, ps.I.z + ps.col.z * B);
        return SH_END;
    }
    double rr = 0;
    if (FLAGS & PTX_USE_RR) {
        double mx = ps.col.x < ps.col.z ? ps.col.z : ps.col.x;  // max(max(R,B),G), reference PathTracer.c:324
        mx = mx < ps.col.y ? ps.col.y : mx;
        rr = 1 - mx;
        if (rr >= .98) rr = .98;
    }
    if (rr > ps.rng.next()) return SH_END;
    double diffusivity = mt_diff, reflectivity = mt_refl, transparency = mt_tran;
    if (alpha_img >= 0) {
        transparency = alpha_map(sc, alpha_img, a, b);
        double both = reflectivity + diffusivity;
        reflectivity = (1 - transparency) * mt_refl / both;
        diffusivity = (1 - transparency) * mt_diff / both;
    }
    const double inv_keep = (FLAGS & PTX_USE_RR) ? fast_rcp(1 - rr) : 1.0;
    double dice = ps.rng.next();
    int next_os = h.obj;
    int result = SH_NEXT;
    // one of three kinds of bounce, or none (reference PathTracer.c:336-428).  The code of the diffuse bounce comes
    // last: what it leaves for the shadow query then does not have to survive the code of the other two.
    const bool is_diffuse = dice <= diffusivity;
    const bool is_mirror = !is_diffuse && dice <= diffusivity + reflectivity;
    const bool is_refraction = !is_diffuse && !is_mirror && dice <= diffusivity + reflectivity + transparency;
    if (is_mirror) {
        if (rp.clean) ps.cel = 0;
        ps.col = mk(ps.col.x * (R * inv_keep), ps.col.y * (G * inv_keep), ps.col.z * (B * inv_keep));
        ps.d = reflect_dir(ps.d, n);
        ps.p0 = p;
        const double sig = mt.refl_sig;
        if (sig > 0) glossy = sig;
    }
    if (is_refraction) {
        if (rp.clean) ps.cel = 0;
        ps.col = mk(ps.col.x * (R * inv_keep), ps.col.y * (G * inv_keep), ps.col.z * (B * inv_keep));
        refract_ray(sc, ps, ls, p, n, h.obj, mt.r_index, cnt);
        next_os = -1;
    }
    if (sc.has_refraction) {
        ls.put(LS_SRCN + 0, n.x);
        ls.put(LS_SRCN + 1, n.y);
        ls.put(LS_SRCN + 2, n.z);
    }
    if (is_diffuse) {
        V3 dir;
        double scale;
        if (FLAGS & PTX_USE_IS) {
            dir = cos_weighted_sample(n, ps.rng);  // a unit vector (rotation of (x, y, sqrt(1 - x^2 - y^2))); normalised at the next bounce's top
            scale = inv_keep;
        } else {
            dir = uniform_hemisphere_sample(n, ps.rng);
            scale = dot3(n, dir) * inv_keep;
        }
        ps.col = mk(ps.col.x * (R * scale), ps.col.y * (G * scale), ps.col.z * (B * scale));
        ps.p0 = p;
        ps.d = dir;
        if (FLAGS & PTX_USE_ES) {
            double u = ps.rng.next();
            int li = 0;
            if (sc.n_lights > 1) {  // (with a single light the walk below never moves and its weight is not needed)
                double acc = sc.lights[0].weight;
                while (acc < u && li + 1 < sc.n_lights) {
                    li++;
                    acc += sc.lights[li].weight;
                }
            }
            // the light's record in one request, type and transform together (light_sample asks for the type first)
            DevLight lt;
            lt.type = sc.lights[li].type;
#pragma unroll
            for (int q = 0; q < 12; q++) lt.T[q] = sc.lights[li].T[q];
            keep_here(lt.type);
#pragma unroll
            for (int q = 0; q < 12; q++) keep_here(lt.T[q]);
            V3 lp = light_sample(lt, ps.rng);
            V3 dl = lp - p;
            double d2 = dot3(dl, dl);
            const double inv_len = fast_rsqrt(d2);
            pend.ed = dl * inv_len;
            ls.put(ls.pend + 0, dot3(n, pend.ed));
            ls.put(ls.pend + 1, inv_len * inv_len);  // 1 / d2
            ls.put(ls.pend + 2, inv_keep);
            ls.put(ls.pend + 3, ps.col.x * R);
            ls.put(ls.pend + 4, ps.col.y * G);
            pend.cb = ps.col.z * B;
            result = SH_SHADOW;
        }
    }
    ps.os = next_os;
    ps.depth++;
    over = ps.depth > rp.max_depth;
    return (result == SH_NEXT && over) ? SH_END : result;
}

__device__ __forceinline__ void shadow_pending_load(const LaneStore &ls, ShadowPending &pend) {
    pend.n_ed = ls.get(ls.pend + 0);
    pend.rcp_d2 = ls.get(ls.pend + 1);
    pend.inv_keep = ls.get(ls.pend + 2);
    pend.cr = ls.get(ls.pend + 3);
    pend.cg = ls.get(ls.pend + 4);
}

// sh: result of the shadow query from ps.p0 along pend.ed (os = ps.os).  Reference PathTracer.c:428-445.
__device__ inline void apply_shadow(const DevScene &sc, const DevRender &rp, Path &ps, const Hit &sh, const ShadowPending &pend) {
    int lit = 0;
    if (sh.obj >= 0) {
        const DevMat &lm = sc.mat[sh.obj];
        // what a light's contribution needs, all of it requested before the first of it is looked at: one wait instead of
        // four in a row (the material's flag, the primitive's type, its plane, its normal)
        GeomHead gh = geom_head(sc.geom[sh.obj]);
        int is_light = lm.is_light;
        double area = lm.surfaceArea, lr = lm.col[0], lg = lm.col[1], lb_ = lm.col[2];
        keep_here(gh);
        keep_here(is_light);
        keep_here(area); keep_here(lr); keep_here(lg); keep_here(lb_);
        if (is_light) {
            V3 lpnt, ln;
            double la, lb;
            finish_hit_pre<false>(sc, sh, gh, ps.p0, pend.ed, lpnt, ln, la, lb);
            double weight2 = pend.n_ed * -dot3(ln, pend.ed);
            double wgt = (area * weight2) * pend.rcp_d2;
            wgt = 1 > wgt ? wgt : 1;  // min(1, .), reference PathTracer.c:437
            if (rp.clean && wgt < 0) wgt = 0;
            wgt *= pend.inv_keep;
            ps.I = mk(ps.I.x + pend.cr * lr * wgt, ps.I.y + pend.cg * lg * wgt, ps.I.z + pend.cb * lb_ * wgt);
            lit = 1;
        }
    }
    ps.cel = rp.clean ? 1 : lit;
}

// ------------------------------------------------ a closest-hit query carried in slices
// The same query as closest_hit_all<true>, cut into pieces a kernel can interleave with other work: the lanes of a
// wave need very different numbers of steps for their rays, and a wave that runs whole queries waits for its
// slowest ray twice per bounce.
//   query_begin()  the always-tested list and the constants of the walk
//   query_step()   one node of the hierarchy; a leaf whose box the ray enters is not tested at once but noted in
//                  Query::leaf (the lane then pauses), so that a wave can test the leaves of many lanes together
//   query_leaf()   the primitive test of the noted leaf
// The query is over when node >= sc.n_bvh_nodes and leaf < 0.  The closest hit, ties included, does not depend on
// the order of the tests, so the result has the bits closest_hit_all gives.
// The box test of the walk is done in single precision, made safe by slack: with i = 1/d and o = p0/d rounded to
// single precision, bound * i - o differs from the double-precision value closest_hit_all tests by at most
// 2^-23 (|bound * i| + |o|), so the entry plane is tested with o + e and the exit plane with o - e,
// e = 2^-21 (B |i| + |o|), B the largest magnitude of any bound (DevScene::bvh_bound), and the best hit's distance
// is rounded upward.  The test then enters every node the double-precision test enters (and a few more, whose
// primitives are tested in double precision like all others), at a third of the instructions.
struct Query {
    float ix, iy, iz;     // 1/d (a huge value of the octant's sign for a component that is zero)
    float nx, ny, nz;     // o + e: subtracted on the plane a ray reaches first
    float fx, fy, fz;     // o - e: subtracted on the plane it reaches second
    float best;           // h.lambda rounded upward
    Hit h;
    int node;  // next node to visit, relative to the start of the octant's array
    int base;  // where that array starts in sc.bvh
    int leaf;  // >= 0: PtxBvhNode::obj of a leaf waiting for its primitive test
#ifdef PTX_LDS_TOP
    int oct;   // which octant's array (the experiment with the arrays' first entries in LDS)
#endif
};
// 1/x in single precision for the walk: of the same sign as the octant bit computed from x < 0, finite, and huge
// (2^100) for a component too small for its reciprocal (the double-precision walk uses 2^500 there; both put the
// two planes hugely far on either side of a ray that runs between them and both on one side otherwise)
__device__ __forceinline__ float rcp_for_slab32(double x) {
    const double r = fabs(x) < 0x1p-100 ? (x < 0 ? -0x1p100 : 0x1p100) : fast_rcp(x);
    return (float)r;
}
__device__ __forceinline__ void query_axis32(double p, float i, float bound, float &near_o, float &far_o) {
    const double o = p * (double)i;
    const double e = 0x1p-21 * ((double)bound * fabs((double)i) + fabs(o));
    near_o = __double2float_ru(o + e);
    far_o = __double2float_rd(o - e);
}
__device__ __forceinline__ void query_begin(const DevScene &sc, V3 p0, V3 d, int os, Query &q) {
    q.h.lambda = INFINITY;
    q.h.obj = -1;
    q.h.part = 0;
    const double dd = 1.0, inv_dd = 1.0;  // d is a unit vector (see closest_hit_all, UNIT)
    list_hits_nb(sc, p0, d, os, dd, inv_dd, q.h);
    q.best = __double2float_ru(q.h.lambda);
    q.ix = rcp_for_slab32(d.x);
    q.iy = rcp_for_slab32(d.y);
    q.iz = rcp_for_slab32(d.z);
    query_axis32(p0.x, q.ix, sc.bvh_bound, q.nx, q.fx);
    query_axis32(p0.y, q.iy, sc.bvh_bound, q.ny, q.fy);
    query_axis32(p0.z, q.iz, sc.bvh_bound, q.nz, q.fz);
    const int octant = sc.bvh_per_octant ? ((d.x < 0) | (d.y < 0) << 1 | (d.z < 0) << 2) : 0;
    q.base = octant * sc.n_bvh_nodes;
    q.node = 0;
    q.leaf = -1;
#ifdef PTX_LDS_TOP
    q.oct = octant;
#endif
}
// Free of branches, and the node fetched as two 16-byte words whatever the outcome of the box test.
// top, top_nodes: a copy in LDS of the first top_nodes entries of each octant's array (or null, 0): an experiment,
// see k_trace_samples_bvh.
#define PTX_LDS_TOP_MAX 12
__device__ __forceinline__ void query_step(const DevScene &sc, int os, Query &q, const float4 *top = nullptr, int top_nodes = 0) {
    float4 lo, hi;  // box[0..3]; box[4], box[5], skip, obj
#ifdef PTX_LDS_TOP
    if (q.node < top_nodes) {
        const float4 *np = top + 2 * (q.oct * PTX_LDS_TOP_MAX + q.node);
        lo = np[0];
        hi = np[1];
    } else
#endif
    {
        const float4 *np = (const float4 *)(sc.bvh + (size_t)(uint32_t)(q.base + q.node));
        lo = np[0];
        hi = np[1];
    }
    float x0 = lo.x, x1 = lo.y, y0 = lo.z, y1 = lo.w, z0 = hi.x, z1 = hi.y;
    if (!sc.bvh_per_octant) {  // one array for all directions: the plane reached first is the lower one if i >= 0
        if (q.ix < 0) { x0 = lo.y; x1 = lo.x; }
        if (q.iy < 0) { y0 = lo.w; y1 = lo.z; }
        if (q.iz < 0) { z0 = hi.y; z1 = hi.x; }
    }
    const float lmin = fmaxf(fmaxf(fmaf(x0, q.ix, -q.nx), fmaf(y0, q.iy, -q.ny)), fmaf(z0, q.iz, -q.nz));
    const float lmax = fminf(fminf(fmaf(x1, q.ix, -q.fx), fmaf(y1, q.iy, -q.fy)), fmaf(z1, q.iz, -q.fz));
    const int skip = __float_as_int(hi.z), obj = __float_as_int(hi.w);
    const bool enter = !(lmin > lmax) && !(lmax < 0) && !(lmin > q.best);
    q.leaf = (enter && obj >= 0 && (obj & PTX_LEAF_INDEX) != os) ? obj : -1;
    q.node = enter ? q.node + 1 : skip;
}
__device__ __forceinline__ void query_leaf(const DevScene &sc, V3 p0, V3 d, Query &q) {
    const int leaf = q.leaf, id = leaf & PTX_LEAF_INDEX;
    int part;
    double lam;
    if ((leaf >> PTX_LEAF_TYPE_SHIFT) == PTX_TRIANGLE) {
        // a triangle's record in one request: the test asks for the normal and a vertex, and for the other two vertices
        // only behind its early returns, which is a second wait for memory in a phase that does little else
        double m[12];
#pragma unroll
        for (int c = 0; c < 12; c++) m[c] = sc.geom[id].m[c];
#pragma unroll
        for (int c = 0; c < 12; c++) keep_here(m[c]);
        lam = isect_triangle(m, p0, d, part);
    } else {
        lam = isect_typed(&sc.geom[id], leaf >> PTX_LEAF_TYPE_SHIFT, leaf & PTX_LEAF_FAST, p0, d, 1.0, 1.0, part);  // (unit d)
    }
    if (lam > PT_TOL && (lam < q.h.lambda || (lam == q.h.lambda && id < q.h.obj))) {
        q.h.lambda = lam;
        q.h.obj = id;
        q.h.part = part;
        q.best = __double2float_ru(lam);
    }
    q.leaf = -1;
}

// ------------------------------------------------ the same query over the four-wide hierarchy (PtxWideNode)
// For scenes whose node arrays are far larger than the L2 caches a step of the walk waits for device memory.  The wide
// walk fetches one whole 128-byte line per step -- the boxes of four children -- and keeps the children it has yet to
// visit on a stack of its own: the first WIDE_STACK_LDS entries in the lane's LDS slots, the rest (deep trees only) in a
// region of device memory set aside per lane.  Query::node counts the entries on the stack; the query is over when the
// stack is empty and no leaf waits.  Children are pushed in the order PtxWideNode::order gives for the ray's octant,
// farthest first, so the nearest is popped first and its hits cull the others when their turn comes.  The box test is
// query_step's (single precision with slack, never rejecting a box the double-precision test enters); which bound of
// an axis a ray reaches first is picked by the sign of its direction, since one array serves all directions.  The
// ray's own surface (os) is dropped when its leaf comes off the stack.
// The closest hit, ties included, does not depend on the order of the visits, so the result has closest_hit_all's bits.
#define WIDE_STACK_LDS 20
#define PTX_WIDE_SPILL 44  // entries of device memory per lane behind the LDS part
// (the LDS part through a pointer that says so: a pointer that may be either LDS or device memory makes every access a
// flat one)
typedef __attribute__((address_space(3))) int *LdsIntPtr;
struct WideStack {
    LdsIntPtr lds;  // &stack[0][lane] of the wave's LDS stack: entry e of this lane is lds[e * PTX_BLOCK]
    int *spill;   // entry e >= lds_n of this lane is spill[(e - lds_n) * spill_stride]
    size_t spill_stride;
    int lds_n;    // entries kept in LDS: WIDE_STACK_LDS (fewer only in tests of the spill path)
    __device__ __forceinline__ void put(int e, int w) const {
        if (e < lds_n) lds[e * PTX_BLOCK] = w;
        else spill[(size_t)(e - lds_n) * spill_stride] = w;
    }
    __device__ __forceinline__ int get(int e) const {
        return e < lds_n ? lds[e * PTX_BLOCK] : spill[(size_t)(e - lds_n) * spill_stride];
    }
};
__device__ __forceinline__ void query_begin_wide(const DevScene &sc, V3 p0, V3 d, int os, Query &q, const WideStack &st) {
    q.h.lambda = INFINITY;
    q.h.obj = -1;
    q.h.part = 0;
    const double dd = 1.0, inv_dd = 1.0;  // d is a unit vector (see closest_hit_all, UNIT)
    list_hits_nb(sc, p0, d, os, dd, inv_dd, q.h);
    q.best = __double2float_ru(q.h.lambda);
    q.ix = rcp_for_slab32(d.x);
    q.iy = rcp_for_slab32(d.y);
    q.iz = rcp_for_slab32(d.z);
    query_axis32(p0.x, q.ix, sc.bvh_bound, q.nx, q.fx);
    query_axis32(p0.y, q.iy, sc.bvh_bound, q.ny, q.fy);
    query_axis32(p0.z, q.iz, sc.bvh_bound, q.nz, q.fz);
    q.base = (d.x < 0) | (d.y < 0) << 1 | (d.z < 0) << 2;  // the ray's octant
    st.lds[0] = ~0;  // the root
    q.node = 1;
    q.leaf = -1;
}
// One entry off the stack: a leaf is noted in Query::leaf (the lane then pauses, as in query_step; the ray's own
// surface is dropped); a wide node is fetched, its four child boxes tested, and the children the ray enters pushed,
// farthest first.
// Which bound of an axis the ray reaches first is chosen by address: the lower and upper bounds of an axis are adjacent
// 16-byte words of the node, and the octant's sign bit picks which of the two is loaded as "near" (a select per value
// would cost 24 instructions a step).  While every lane of the wave has its whole stack in LDS -- always, but for trees
// of unusual depth -- popping is one LDS read and pushing one LDS write per child without a branch (a child the ray
// does not enter is written too and overwritten by the next push).
#ifdef PTX_WIDE_DIAG
// Diagnosis only: what the four-wide walk does, summed over a launch (ptx_debug_wide_diag reads and clears).
static __device__ unsigned long long g_wide_diag[16];
#define WIDE_DIAG(i, cond)                                                                     \
    do {                                                                                       \
        const unsigned long long m_ = __ballot(cond);                                          \
        if (m_ && (int)(threadIdx.x & 63) == __ffsll((long long)__ballot(1)) - 1) atomicAdd(&g_wide_diag[i], (unsigned long long)__popcll(m_)); \
    } while (0)
#else
#define WIDE_DIAG(i, cond) ((void)0)
#endif
__device__ __forceinline__ void query_step_wide(const DevScene &sc, int os, Query &q, const WideStack &st) {
    const int top = --q.node;
    int e;
    WIDE_DIAG(4, top >= st.lds_n);
    WIDE_DIAG(8 + (0), top < 4);
    WIDE_DIAG(8 + (1), top >= 4 && top < 8);
    WIDE_DIAG(8 + (2), top >= 8 && top < 12);
    WIDE_DIAG(8 + (3), top >= 12 && top < 16);
    WIDE_DIAG(8 + (4), top >= 16 && top < 20);
    WIDE_DIAG(8 + (5), top >= 20 && top < 28);
    WIDE_DIAG(8 + (6), top >= 28 && top < 40);
    WIDE_DIAG(8 + (7), top >= 40);
    if (__ballot(top >= st.lds_n) == 0ULL) e = st.lds[top * PTX_BLOCK];
    else e = st.get(top);
    WIDE_DIAG(5, e >= 0);
    WIDE_DIAG(6, e >= 0 && (e & PTX_LEAF_INDEX) == os);
    if (e >= 0) {
        q.leaf = (e & PTX_LEAF_INDEX) != os ? e : -1;
        return;
    }
    const char *nodes = (const char *)sc.wide;
    const uint32_t at = (uint32_t)~e << 7;  // byte offset of the node (arrays beyond 4 GB are not built)
    const uint32_t sx = (q.base & 1) << 4, sy = (q.base & 2) << 3, sz = (q.base & 4) << 2;  // 16 where the ray travels downward
    const float4 x0 = *(const float4 *)(nodes + (at | sx)), x1 = *(const float4 *)(nodes + (at | (sx ^ 16u)));
    const float4 y0 = *(const float4 *)(nodes + (at | 32u | sy)), y1 = *(const float4 *)(nodes + (at | 32u | (sy ^ 16u)));
    const float4 z0 = *(const float4 *)(nodes + (at | 64u | sz)), z1 = *(const float4 *)(nodes + (at | 64u | (sz ^ 16u)));
    const int4 ch = *(const int4 *)(nodes + (at | 96u));
    const uint2 ord = *(const uint2 *)(nodes + (at | 112u));
    int hits = 0;
#define PTX_WIDE_CHILD(c, bit)                                                                                              \
    {                                                                                                                       \
        const float lmin = fmaxf(fmaxf(fmaf(x0.c, q.ix, -q.nx), fmaf(y0.c, q.iy, -q.ny)), fmaf(z0.c, q.iz, -q.nz));         \
        const float lmax = fminf(fminf(fmaf(x1.c, q.ix, -q.fx), fmaf(y1.c, q.iy, -q.fy)), fmaf(z1.c, q.iz, -q.fz));         \
        hits |= (!(lmin > lmax) && !(lmax < 0) && !(lmin > q.best)) ? (1 << bit) : 0;                                       \
    }
    PTX_WIDE_CHILD(x, 0)
    PTX_WIDE_CHILD(y, 1)
    PTX_WIDE_CHILD(z, 2)
    PTX_WIDE_CHILD(w, 3)
#undef PTX_WIDE_CHILD
    const unsigned order = ((q.base & 4 ? ord.y : ord.x) >> (8 * (q.base & 3))) & 255u;
    WIDE_DIAG(0, true);
    WIDE_DIAG(1, hits == 0);
    WIDE_DIAG(2, hits & 1); WIDE_DIAG(2, hits & 2); WIDE_DIAG(2, hits & 4); WIDE_DIAG(2, hits & 8);
    WIDE_DIAG(3, q.node + __popc(hits) > st.lds_n);
    WIDE_DIAG(7, __popc(hits) >= 3);
    if (__ballot(q.node + 4 > st.lds_n) == 0ULL) {
#pragma unroll
        for (int t = 3; t >= 0; t--) {
            const int s = (order >> (2 * t)) & 3;
            const int w = s & 2 ? (s & 1 ? ch.w : ch.z) : (s & 1 ? ch.y : ch.x);
            st.lds[q.node * PTX_BLOCK] = w;
            q.node += (hits >> s) & 1;
        }
    } else {
#pragma unroll
        for (int t = 3; t >= 0; t--) {
            const int s = (order >> (2 * t)) & 3;
            if ((hits >> s) & 1) {
                const int w = s & 2 ? (s & 1 ? ch.w : ch.z) : (s & 1 ? ch.y : ch.x);
                st.put(q.node, w);
                q.node++;
            }
        }
    }
}

// The glossy perturbation d += refl_sig * (N1, N2, N3) (reference PathTracer.c:460-464) for every lane of the
// wave that asked for it, to be called by all 64 lanes together.  Few lanes reflect off a glossy surface in any
// one bounce, and each needs three independent deviates (a log, a cos and a square root apiece), so the deviates
// are handed out to the lanes of the wave, one per lane, and sent back with lane permutes: lane 3q+c computes
// deviate c of the q-th requesting lane from that lane's counter, which gives the bits the lane itself would get.
__device__ __forceinline__ void glossy_service(Path &ps, double glossy, int lane) {
    const unsigned long long req = __ballot(glossy > 0);
    if (req == 0ULL) return;
    const int n_req = __popcll(req);
    const int rank = __popcll(req & ((1ULL << lane) - 1ULL));
    for (int base = 0; base < n_req; base += 21) {
        const int q = base + lane / 3, c = lane - 3 * (lane / 3);
        const bool work = lane < 63 && q < n_req;
        unsigned long long m = req;
        for (int t = 0; t < q && work; t++) m &= m - 1;  // drop the q lowest requesters
        const int src = work ? __ffsll((long long)m) - 1 : lane;
        const unsigned long long ctr = __shfl(ps.rng.ctr, src);
        double dev = 0;
        if (work) {
            Rng r;
            r.ctr = ctr + PTX_GOLDEN * (unsigned long long)(2 * c);
            dev = random_normal(r);
        }
        const int mine = rank - base;
        const bool served = glossy > 0 && mine >= 0 && mine < 21;
        const int from = served ? 3 * mine : 0;
        double n0 = __shfl(dev, from), n1 = __shfl(dev, from + 1), n2 = __shfl(dev, from + 2);
        if (served) {
            ps.d.x += glossy * n0;
            ps.d.y += glossy * n1;
            ps.d.z += glossy * n2;
            ps.rng.skip(6);
        }
    }
}
