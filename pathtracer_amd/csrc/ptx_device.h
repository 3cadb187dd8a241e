// ptx_device.h -- device-side data layout and per-lane routines of the
// path-tracing core (gfx950).  Included only by ptx_hip.hip.
//
// Each routine names the reference routine whose result it reproduces.  The
// arithmetic is IEEE binary64 like the reference's, but it is not a
// transcription: transforms use the 3x4 affine part only, trigonometric
// round trips (atan2/acos followed by sin/cos of the same angle) are replaced
// by the algebraic sine and cosine, and hit attributes are computed once for
// the winning primitive instead of once per candidate.  Results agree with
// the reference to rounding (about 1e-13 relative on lambda).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "ptx_rng.h"
#include "ptx_scene.h"

#define PT_PI 3.14159265354  // reference PathTracer.h:27 (not pi: the truncated value is part of the behaviour)
#define PT_TOL 0.000000001   // reference PathTracer.h:28

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
__device__ __forceinline__ V3 norm3(V3 v) {  // normalize(), reference utils_path.h:103-114
    double l = 1.0 / sqrt(dot3(v, v));
    return v * l;
}
__device__ __forceinline__ double comp(V3 v, int a) { return a == 0 ? v.x : (a == 1 ? v.y : v.z); }

// ---------------------------------------------------------------- scene
// Geometry record, 128 bytes, read either with a wave-uniform index (scalar
// loads) or per lane for the winning primitive.
//   analytic primitives: m[0..11] = rows 0..2 of Tinv (3x4)
//   triangles:           m[0..8]  = v1 v2 v3 (world space), m[9..11] = unit normal
struct DevGeom {
    int32_t type;
    int32_t pad;
    double m[12];
    double spare[3];
};
static_assert(sizeof(DevGeom) == 128, "DevGeom layout");

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

struct DevImage {
    int32_t sx, sy, channels, pad;
    double inv_div_unused;
    double divisor;
    uint64_t offset;
};

struct DevCamera {
    double e[3], w[3];
    double C2W[12];
    double f, wl, wt, wsize;
    double fp[3];  // C2W * (0,0,focus_distance,1), reference PathTracer.c:589-594
};

struct DevScene {
    const DevGeom *geom;
    const DevMat *mat;
    const DevLight *lights;
    const DevImage *images;
    const uint8_t *texels;
    int32_t n_objects, n_lights, n_images, has_refraction;
    DevCamera cam;
};

struct DevRender {
    int32_t sx, max_depth;
    int32_t rank, n_ranks, band_rows, rows_owned;
    double aperture;
    uint64_t seed;
    double du, dv;
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

// ------------------------------------------------------- transforms
__device__ __forceinline__ V3 xform_point(const double *m, V3 p) {
    return mk(m[0] * p.x + m[1] * p.y + m[2] * p.z + m[3], m[4] * p.x + m[5] * p.y + m[6] * p.z + m[7],
              m[8] * p.x + m[9] * p.y + m[10] * p.z + m[11]);
}
__device__ __forceinline__ V3 xform_dir(const double *m, V3 d) {
    return mk(m[0] * d.x + m[1] * d.y + m[2] * d.z, m[4] * d.x + m[5] * d.y + m[6] * d.z,
              m[8] * d.x + m[9] * d.y + m[10] * d.z);
}
// normalTransform (reference utils_path.c:58-76): direction of Tinv^T n.
__device__ __forceinline__ V3 xform_normal(const double *m, V3 n) {
    return norm3(mk(m[0] * n.x + m[4] * n.y + m[8] * n.z, m[1] * n.x + m[5] * n.y + m[9] * n.z,
                    m[2] * n.x + m[6] * n.y + m[10] * n.z));
}

// ------------------------------------------------- primitive intersectors
// Each returns the lambda the reference intersect function would report
// (a value <= PT_TOL means "no usable hit") and, in `part`, which face or
// section produced it.

// sphereIntersect, reference utils_path.c:546-587
__device__ __forceinline__ double isect_sphere(const double *m, V3 p0, V3 d) {
    V3 tp = xform_point(m, p0), td = xform_dir(m, d);
    double A = dot3(td, td), B = dot3(tp, td), C = dot3(tp, tp) - 1.0;
    double disc = B * B - A * C;
    if (disc < 0.0) return -1.0;
    double sq = sqrt(disc);
    double mb = -B / A, sa = sq / A;
    double l1 = mb + sa, l2 = mb - sa;
    double lam = -1.0;
    if (l1 > PT_TOL) lam = l1;
    if (l2 > PT_TOL && (l1 <= PT_TOL || l2 < l1)) lam = l2;
    return lam;
}

// planeIntersect + planeHelper, reference utils_path.c:439-544
__device__ __forceinline__ double isect_plane(const double *m, V3 p0, V3 d) {
    V3 tp = xform_point(m, p0), td = xform_dir(m, d);
    if (fabs(td.z) < PT_TOL) return -1.0;
    double lam = -tp.z / td.z;
    if (!(fabs(tp.x + lam * td.x) <= 1.0 - PT_TOL)) return -1.0;
    if (!(fabs(tp.y + lam * td.y) <= 1.0 - PT_TOL)) return -1.0;
    return lam;
}

// cubeIntersect / roomIntersect, reference utils_path.c:832-995, :997-1143:
// faces tried in the order +z -z +y -y +x -x, strictly smaller lambda wins.
__device__ __forceinline__ double isect_box(const double *m, V3 p0, V3 d, int &part) {
    V3 tp = xform_point(m, p0), td = xform_dir(m, d);
    double best = INFINITY;
    part = -1;
#pragma unroll
    for (int f = 0; f < 6; f++) {
        const int a = 2 - (f >> 1);
        const double s = (f & 1) ? -1.0 : 1.0;
        double da = comp(td, a), pa = comp(tp, a);
        if (fabs(da) < PT_TOL) continue;
        double lam = (s - pa) / da;
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
__device__ __forceinline__ double isect_cyl(const double *m, V3 p0, V3 d, int &part) {
    V3 tp = xform_point(m, p0), td = xform_dir(m, d);
    double best = INFINITY;
    part = -1;
    double A = td.x * td.x + td.y * td.y;
    double B = 2 * tp.x * td.x + 2 * tp.y * td.y;
    double C = tp.x * tp.x + tp.y * tp.y - 1;
    double det = B * B - 4 * A * C;
    if (!(det < 0)) {
        double sq = sqrt(det);
        double l1 = (-1 * B + sq) / (2 * A);
        double l2 = (-1 * B - sq) / (2 * A);
        if (fabs(tp.z + l1 * td.z) <= 1 && l1 > PT_TOL && best > l1) { best = l1; part = 0; }
        if (fabs(tp.z + l2 * td.z) <= 1 && l2 > PT_TOL && best > l2) { best = l2; part = 0; }
    }
    double c1 = (-1 - tp.z) / td.z;
    double c2 = (1 - tp.z) / td.z;
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
__device__ __forceinline__ double isect_triangle(const double *m, V3 p0, V3 d, int &part) {
    V3 p1 = mk(m[0], m[1], m[2]), p2 = mk(m[3], m[4], m[5]), p3 = mk(m[6], m[7], m[8]);
    V3 n = mk(m[9], m[10], m[11]);
    double ddn = dot3(n, d);
    part = ddn > 0 ? 1 : 0;
    if (-PT_TOL < ddn && ddn < PT_TOL) return -1.0;
    double lam = dot3(p1 - p0, n) / ddn;
    if (!(lam > 0)) return -1.0;
    V3 poi = p0 + d * lam;
    V3 e12 = p2 - p1, e23 = p3 - p2, e31 = p1 - p3, e13 = p3 - p1, e32 = p2 - p3, e21 = p1 - p2;
    double v1 = dot3(cross3(poi - p1, e12), cross3(e13, e12));
    double v2 = dot3(cross3(poi - p2, e23), cross3(e21, e23));
    double v3 = dot3(cross3(poi - p3, e31), cross3(e32, e31));
    if (v1 >= 0 && v2 >= 0 && v3 >= 0) return lam;
    return -1.0;
}

__device__ __forceinline__ double isect_any(const DevGeom &g, V3 p0, V3 d, int &part) {
    part = 0;
    switch (g.type) {
    case PTX_SPHERE: return isect_sphere(g.m, p0, d);
    case PTX_PLANE: return isect_plane(g.m, p0, d);
    case PTX_TRIANGLE: return isect_triangle(g.m, p0, d, part);
    case PTX_CYL: return isect_cyl(g.m, p0, d, part);
    default: return isect_box(g.m, p0, d, part);  // PTX_CUBE, PTX_ROOM
    }
}

struct Hit {
    double lambda;
    int obj;
    int part;
};

// findFirstHit (reference PathTracer.c:198-251) by exhaustive search: every
// primitive is tested, the smallest lambda > TOL wins, the earlier primitive
// wins ties (primitives are stored in the reference's visiting order).  The
// loop index is wave-uniform, so the geometry records arrive through scalar
// loads and no lane diverges on the primitive type.
__device__ __forceinline__ Hit closest_hit_all(const DevScene &sc, V3 p0, V3 d, int os) {
    Hit h;
    h.lambda = INFINITY;
    h.obj = -1;
    h.part = 0;
    for (int k = 0; k < sc.n_objects; k++) {
        int part;
        double lam = isect_any(sc.geom[k], p0, d, part);
        if (k != os && lam > PT_TOL && lam < h.lambda) {
            h.lambda = lam;
            h.obj = k;
            h.part = part;
        }
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
    p = p0 + d * h.lambda;
    a = 0;
    b = 0;
    if (g.type == PTX_TRIANGLE) {
        V3 nn = mk(g.m[9], g.m[10], g.m[11]);
        n = h.part ? nn * -1.0 : nn;
        return;
    }
    double m[12];
#pragma unroll
    for (int q = 0; q < 12; q++) m[q] = g.m[q];
    V3 tp = xform_point(m, p0), td = xform_dir(m, d);
    V3 pc = tp + td * h.lambda;
    V3 nc;
    if (g.type == PTX_SPHERE) {
        nc = norm3(pc);
        if (WANT_UV) {
            a = atan2(pc.y, pc.x) / (2 * PT_PI);
            b = acos(pc.z) / PT_PI;
        }
    } else if (g.type == PTX_PLANE) {
        nc = mk(0, 0, tp.z < 0 ? -1.0 : 1.0);
        if (WANT_UV) {
            a = (1 + pc.x) / 2;
            b = (1 + pc.y) / 2;
        }
    } else if (g.type == PTX_CYL) {
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
        if (g.type == PTX_ROOM) s = -s;
        nc = mk(ax == 0 ? s : 0.0, ax == 1 ? s : 0.0, ax == 2 ? s : 0.0);
        if (WANT_UV && g.type == PTX_CUBE) {
            if (ax == 2) { a = (1 + pc.x) / 2; b = (1 + pc.y) / 2; }
            else if (ax == 1) { a = (1 + pc.x) / 2; b = (1 + pc.z) / 2; }
            else { a = (1 + pc.y) / 2; b = (1 + pc.z) / 2; }
        }
    }
    n = xform_normal(m, nc);
}

// ------------------------------------------------------ texture look-ups
__device__ __forceinline__ double texel_at(const DevScene &sc, const DevImage &im, long idx, long n) {
    idx = idx < 0 ? 0 : (idx >= n ? n - 1 : idx);
    return (double)sc.texels[im.offset + idx] / im.divisor;
}
// texMap, reference utils_path.c:1322-1374 (flat index clamped to the image
// buffer where the reference reads out of bounds).
__device__ __forceinline__ void tex_map(const DevScene &sc, int img, double a, double b, double &R, double &G, double &B) {
    const DevImage im = sc.images[img];
    double x = (im.sx - 1) * a, y = (im.sy - 1) * b;
    int x1 = (int)x, y1 = (int)y;
    int x2 = x1 + 1, y2 = y1 + 1;
    long n = (long)im.sx * im.sy * 3;
    double out[3];
#pragma unroll
    for (int c = 0; c < 3; c++) {
        double c11 = texel_at(sc, im, 3L * ((long)y1 * im.sx + x1) + c, n);
        double c12 = texel_at(sc, im, 3L * ((long)y2 * im.sx + x1) + c, n);
        double c21 = texel_at(sc, im, 3L * ((long)y1 * im.sx + x2) + c, n);
        double c22 = texel_at(sc, im, 3L * ((long)y2 * im.sx + x2) + c, n);
        double avg1 = c11 * (x2 - x) + c21 * (x - x1);
        double avg2 = c12 * (x2 - x) + c22 * (x - x1);
        out[c] = avg1 * (y2 - y) + avg2 * (y - y1);
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
    double a11 = texel_at(sc, im, (long)y1 * im.sx + x1, n);
    double a12 = texel_at(sc, im, (long)y2 * im.sx + x1, n);
    double a21 = texel_at(sc, im, (long)y1 * im.sx + x2, n);
    double a22 = texel_at(sc, im, (long)y2 * im.sx + x2, n);
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
    double ct = 1.0, st = 0.0;  // atan2(0,0) = 0
    if (h2 > 0) {
        double ih = 1.0 / sqrt(h2);
        ct = n.x * ih;
        st = n.y * ih;
    } else if (n.x < 0 || (n.x == 0 && signbit(n.x))) {
        ct = -1.0;  // atan2(+-0, -0) = +-pi
    }
    double cp = n.z > 1.0 ? NAN : (n.z < -1.0 ? NAN : n.z);  // acos outside [-1,1] is NaN in the reference
    double sp = sqrt(fmax(0.0, 1.0 - cp * cp));
    double x1 = v.x * cp + v.z * sp;
    double z1 = -v.x * sp + v.z * cp;
    return mk(x1 * ct - v.y * st, x1 * st + v.y * ct, z1);
}

// normalDeform, reference utils_path.c:1408-1428
__device__ __forceinline__ V3 normal_deform(const DevScene &sc, int img, double a, double b, V3 n) {
    double r, g, bb;
    tex_map(sc, img, a, b, r, g, bb);
    V3 nd = norm3(mk(r * 2 - 1, g * 2 - 1, bb * 2 - 1));
    return rotate_z_to(norm3(n), nd);
}

// cosWeightedSample, reference utils_path.c:1224-1265 (two draws)
__device__ __forceinline__ V3 cos_weighted_sample(V3 n, Rng &rng) {
    double u1 = rng.next();
    double r = sqrt(u1);
    double theta = 2 * PT_PI * rng.next();
    double s, c;
    sincos(theta, &s, &c);
    double x = r * c, y = r * s;
    double z = sqrt(1.0 - x * x - y * y);
    return rotate_z_to(n, mk(x, y, z));
}

// The uniform-hemisphere branch of PathTrace (reference PathTracer.c:366-387)
// with hemisphereReorient/reorient (utils_path.c:79-96): four draws, two
// discarded.
__device__ __forceinline__ V3 uniform_hemisphere_sample(V3 n, Rng &rng) {
    double c2 = 2 * rng.next() - 1;  // cos(angle_2)
    rng.skip(2);
    double a1 = 2 * PT_PI * rng.next();
    double s2 = sqrt(fmax(0.0, 1.0 - c2 * c2));  // sin(acos(c2))
    double s1, c1;
    sincos(a1, &s1, &c1);
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
    return sqrt(-2 * log(ul)) * cos(2 * PT_PI * uc);
}

// ------------------------------------------------------------ path state
// What survives from one bounce to the next: the live fields of the
// reference's struct ray3D (PathTracer.h:76-90) plus PathTrace's Os and CEL
// arguments.  The refraction-index stack and the normal of the previous
// bounce are kept only for scenes with transmissive materials (rs[]).
struct Path {
    V3 p0, d, col, I;
    Rng rng;
    int os, cel, depth;
    int ridx;        // r_index_index
    double rs[10];   // r_index[]
    V3 srcN;
};

// Loads/stores of ray->r_index[k] as the reference performs them, including
// the out-of-range indices refractRay can produce (PathTracer.c:101, :131),
// which land on neighbouring fields of struct ray3D: k=-1 srcN.pw (always 1),
// -2..-4 srcN.pz..px, -5..-7 I.B..R, -8..-10 col.B..R, 10 = r_index_index.
__device__ inline double rindex_load(Path &ps, int k, uint32_t &oob) {
    if (k >= 0 && k <= 9) return ps.rs[k];
    oob++;
    switch (k) {
    case -1: return 1.0;
    case -2: return ps.srcN.z;
    case -3: return ps.srcN.y;
    case -4: return ps.srcN.x;
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
__device__ inline void rindex_store(Path &ps, int k, double v, uint32_t &oob) {
    if (k >= 0 && k <= 9) { ps.rs[k] = v; return; }
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

// reflectRay, reference PathTracer.c:53-72
__device__ __forceinline__ V3 reflect_dir(V3 d, V3 n) {
    d = norm3(d);
    double dn = dot3(d, n);
    return norm3(mk(d.x - 2 * dn * n.x, d.y - 2 * dn * n.y, d.z - 2 * dn * n.z));
}

struct Counters {
    uint32_t closest, shadow, oob;
};

// refractRay, reference PathTracer.c:74-140.  n is normalised in place like
// the reference's *normal.
__device__ inline void refract_ray(Path &ps, V3 p, V3 &n, const DevMat &mt, Counters &cnt) {
    double n1 = rindex_load(ps, ps.ridx, cnt.oob);
    double n2 = mt.r_index;
    double r = n1 / n2;
    bool leaving = false;
    ps.d = norm3(ps.d);
    n = norm3(n);
    V3 rn = n;
    if (!(dot3(ps.d, n) < 0)) {
        n1 = mt.r_index;
        n2 = rindex_load(ps, ps.ridx - 1, cnt.oob);
        r = n1 / n2;
        rn = n * -1.0;
        leaving = true;
    }
    double c = -dot3(rn, ps.d);
    double disc = 1 - (r * r) * (1 - (c * c));
    double q = (n1 - n2) / (n1 + n2);
    double r0 = q * q;
    double omc = 1 - c;
    double reflectance = r0 + ((1 - r0) * (omc * omc * omc * omc * omc));
    double dice = ps.rng.next();
    ps.p0 = p;
    if (disc < 0 || dice <= reflectance) {
        ps.d = reflect_dir(ps.d, norm3(rn));
        return;
    }
    if (leaving) {
        ps.ridx = ps.ridx - 1;
    } else {
        ps.ridx = ps.ridx + 1;
        rindex_store(ps, ps.ridx, mt.r_index, cnt.oob);
    }
    ps.d = norm3(ps.d * r + rn * (r * c - sqrt(disc)));
}

// Primary ray of pixel (i,j), sample k: reference PathTracer.c:625-641 with
// sampleLensPoint (:142-158), focusRay (:160-195), initRay
// (utils_path.h:211-230).  Two draws are consumed whatever the aperture.
__device__ inline void primary_ray(const DevScene &sc, const DevRender &rp, int i, int j, int k, bool track_rindex,
                                   Path &ps) {
    ps.rng.ctr = ptx_path_key(rp.seed, (uint64_t)j * (uint64_t)rp.sx + (uint64_t)i, (uint64_t)k);
    const DevCamera &cam = sc.cam;
    V3 pc = mk(cam.wl + (i * rp.du), cam.wt + (j * rp.dv), cam.f);
    V3 pl = pc;
    if (rp.aperture != 0) {
        double r = rp.aperture * sqrt(ps.rng.next());
        double theta = ps.rng.next() * 2 * PT_PI;
        double s, c;
        sincos(theta, &s, &c);
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
    ps.col = mk(1, 1, 1);
    ps.I = mk(0, 0, 0);
    ps.os = -1;
    ps.cel = 0;
    ps.depth = 1;
    ps.ridx = 0;
    if (track_rindex) {
#pragma unroll
        for (int q = 0; q < 10; q++) ps.rs[q] = 0;
        ps.rs[0] = 1;
        ps.srcN = mk(0, 0, 1);
    }
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
        double theta = 2 * PT_PI * v1;
        double cphi = 2 * v2 - 1;  // cos(acos(.))
        double sphi = sqrt(fmax(0.0, 1.0 - cphi * cphi));
        double s, c;
        sincos(theta, &s, &c);
        return xform_point(ls.T, mk(c * sphi, s * sphi, cphi));
    }
    if (ls.type == PTX_CYL) rng.skip(2);
    return mk(0, 0, 0);
}

// One invocation of PathTrace (reference PathTracer.c:253-484) for a live
// path: returns true when the path has ended, its colour then being ps.I.
template <int FLAGS>
__device__ inline bool bounce(const DevScene &sc, const DevRender &rp, Path &ps, Counters &cnt) {
    if (ps.depth > rp.max_depth) return true;
    ps.d = norm3(ps.d);
    cnt.closest++;
    Hit h = closest_hit_all(sc, ps.p0, ps.d, ps.os);
    if (h.obj < 0) return true;
    const DevMat mt = sc.mat[h.obj];
    V3 p, n;
    double a, b;
    const bool want_uv = (mt.tex_img >= 0) | (mt.normal_img >= 0) | (mt.alpha_img >= 0);
    if (want_uv) finish_hit<true>(sc, h, ps.p0, ps.d, p, n, a, b);
    else finish_hit<false>(sc, h, ps.p0, ps.d, p, n, a, b);
    double R = mt.col[0], G = mt.col[1], B = mt.col[2];
    if (mt.tex_img >= 0) tex_map(sc, mt.tex_img, a, b, R, G, B);
    if (mt.normal_img >= 0) n = normal_deform(sc, mt.normal_img, a, b, n);
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
    if (mt.alpha_img >= 0) {
        transparency = alpha_map(sc, mt.alpha_img, a, b);
        reflectivity = (1 - transparency) * mt.reflPct / (mt.reflPct + mt.diffPct);
        diffusivity = (1 - transparency) * mt.diffPct / (mt.reflPct + mt.diffPct);
    }
    const double inv_keep = 1.0 / (1 - rr);
    double dice = ps.rng.next();
    int next_os = h.obj;
    if (dice <= diffusivity) {
        V3 dir;
        double scale;
        if (FLAGS & PTX_USE_IS) {
            dir = norm3(cos_weighted_sample(n, ps.rng));
            scale = 1.0;  // (n.dir) / probability with probability = n.dir, reference PathTracer.c:364, :390
        } else {
            dir = uniform_hemisphere_sample(n, ps.rng);
            scale = dot3(n, dir);
        }
        ps.col = mk(ps.col.x * (R * scale * inv_keep), ps.col.y * (G * scale * inv_keep), ps.col.z * (B * scale * inv_keep));
        ps.p0 = p;
        ps.d = dir;
        if (FLAGS & PTX_USE_ES) {
            double u = ps.rng.next();
            int li = 0;
            double acc = sc.lights[0].weight;
            while (acc < u && li + 1 < sc.n_lights) {
                li++;
                acc += sc.lights[li].weight;
            }
            V3 lp = light_sample(sc.lights[li], ps.rng);
            V3 ed = norm3(lp - p);
            cnt.shadow++;
            Hit sh = closest_hit_all(sc, p, ed, h.obj);
            bool lit = false;
            if (sh.obj >= 0) {
                const DevMat lm = sc.mat[sh.obj];
                if (lm.is_light) {
                    V3 lpnt, ln;
                    double la, lb;
                    finish_hit<false>(sc, sh, p, ed, lpnt, ln, la, lb);
                    V3 dl = lp - p;
                    double d2 = dl.x * dl.x + dl.y * dl.y + dl.z * dl.z;
                    double weight2 = dot3(n, ed) * -dot3(ln, ed);
                    double wgt = (lm.surfaceArea * weight2) / d2;
                    wgt = 1 > wgt ? wgt : 1;  // min(1, .), reference PathTracer.c:437
                    ps.I = mk(ps.I.x + ps.col.x * R * lm.col[0] * wgt * inv_keep,
                              ps.I.y + ps.col.y * G * lm.col[1] * wgt * inv_keep,
                              ps.I.z + ps.col.z * B * lm.col[2] * wgt * inv_keep);
                    lit = true;
                }
            }
            ps.cel = lit ? 1 : 0;
        }
    } else if (dice <= diffusivity + reflectivity) {
        ps.col = mk(ps.col.x * (R * inv_keep), ps.col.y * (G * inv_keep), ps.col.z * (B * inv_keep));
        n = norm3(n);
        ps.d = reflect_dir(ps.d, n);
        ps.p0 = p;
        if (mt.refl_sig > 0) {
            ps.d.x += mt.refl_sig * random_normal(ps.rng);
            ps.d.y += mt.refl_sig * random_normal(ps.rng);
            ps.d.z += mt.refl_sig * random_normal(ps.rng);
        }
    } else if (dice <= diffusivity + reflectivity + transparency) {
        ps.col = mk(ps.col.x * (R * inv_keep), ps.col.y * (G * inv_keep), ps.col.z * (B * inv_keep));
        refract_ray(ps, p, n, mt, cnt);
        next_os = -1;
    }
    if (sc.has_refraction) ps.srcN = n;
    ps.os = next_os;
    ps.depth++;
    return false;
}
