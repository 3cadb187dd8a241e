/*
 * ptx_oracle.c -- TEST INFRASTRUCTURE: CPU oracle for the path-tracing hot
 * path.  Never shipped, never called by the product (see ptx_oracle.h).
 *
 * Every routine follows the arithmetic of the reference routine cited above
 * it operation for operation (same expression trees, same order of
 * additions, homogeneous w handled as divideHCoords does), so that, built
 * with the reference's compiler flags and fed the same scene and the same
 * random numbers, it returns the same bits as the reference.  That claim is
 * pinned by tests/test_oracle_vs_ref.py against oracle/_ref (the compiled
 * reference) and by the golden vectors under tests/golden/.
 *
 * Reference behaviour that is undefined in C is modelled, not fixed:
 *  - triangle and room intersections leave the texture coordinates a,b
 *    unset (utils_path.c:781-830, :997-1143): the oracle reports 0,0;
 *  - texMap/alphaMap index outside the image for a,b outside [0,1)
 *    (utils_path.c:1337-1358): the flat index is clamped to the buffer;
 *  - ray3D.r_index[] is indexed below 0 and above 9 by refractRay
 *    (PathTracer.c:101, :131): the neighbouring fields of struct ray3D
 *    (col, I, srcN before; r_index_index after; PathTracer.h:83-89) are
 *    modelled as the x86-64 layout places them.
 */
#include <math.h>
#include <omp.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "ptx_oracle.h"
#include "ptx_rng.h"
#include "ptx_scene_io.h"

#define PI 3.14159265354 /* PathTracer.h:27 */
#define TOL 0.000000001  /* PathTracer.h:28 */
/* PathTracer.c:31, magic.c:90-91: not fmin/fmax (NaN handling differs). */
#define rmin(A, B) ((A) > (B) ? (B) : (A))
#define rmax(A, B) ((A) < (B) ? (B) : (A))

typedef struct { double x, y, z, w; } pt; /* struct point3D, PathTracer.h:60-65 */

struct orc_scene {
    ptx_scene_desc d;       /* owns copies of all arrays */
    double **img;           /* images as the reference holds them: double(u8)/divisor */
};

/* ------------------------------------------------------------------ RNG */
typedef struct {
    int mode;               /* 0 counter, 1 lcg48, 2 script */
    uint64_t key;
    uint32_t draw;
    uint64_t *lcg;
    const double *script;
} rng_t;

static inline double rnd(rng_t *r) {
    if (r->mode == 0) return ptx_counter_draw(r->key, r->draw++);
    if (r->mode == 1) return ptx_lcg48_draw(r->lcg);
    return r->script[r->draw++];
}

/* ----------------------------------------------- utils_path.h:58-230 kit */
static inline void hdiv(pt *v) { /* divideHCoords, utils_path.h:58-68 */
    v->w = fabs(v->w);
    if (v->w != 1) {
        v->x = v->x / v->w;
        v->y = v->y / v->w;
        v->z = v->z / v->w;
        v->w = 1;
    }
}
static inline void matvec(const double *A, pt *p) { /* matVecMult, utils_path.h:70-82 */
    pt r;
    r.x = (A[0] * p->x) + (A[1] * p->y) + (A[2] * p->z) + (A[3] * p->w);
    r.y = (A[4] * p->x) + (A[5] * p->y) + (A[6] * p->z) + (A[7] * p->w);
    r.z = (A[8] * p->x) + (A[9] * p->y) + (A[10] * p->z) + (A[11] * p->w);
    r.w = (A[12] * p->x) + (A[13] * p->y) + (A[14] * p->z) + (A[15] * p->w);
    *p = r;
    hdiv(p);
}
static inline void matmul(const double *A, double *B) { /* matMult B=A*B, utils_path.h:26-42 */
    double C[16];
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 4; j++)
            C[4 * i + j] = (A[4 * i + 0] * B[0 + j]) + (A[4 * i + 1] * B[4 + j]) + (A[4 * i + 2] * B[8 + j]) +
                           (A[4 * i + 3] * B[12 + j]);
    memcpy(B, C, sizeof(C));
}
static inline void normalize(pt *v) { /* utils_path.h:103-114 */
    double l;
    hdiv(v);
    l = v->x * v->x;
    l += (v->y * v->y);
    l += (v->z * v->z);
    l = 1.0 / sqrt(l);
    v->x *= l;
    v->y *= l;
    v->z *= l;
}
static inline double dot(pt *u, pt *v) { /* utils_path.h:116-121 (dot2 :173 is the by-value twin) */
    hdiv(u);
    hdiv(v);
    return (u->x * v->x) + (u->y * v->y) + (u->z * v->z);
}
static inline double dot2(pt a, pt b) { /* utils_path.h:173-177 */
    hdiv(&a);
    hdiv(&b);
    return a.x * b.x + a.y * b.y + a.z * b.z;
}
static inline pt cross2(pt u, pt v) { /* utils_path.h:162-171 */
    pt r;
    hdiv(&u);
    hdiv(&v);
    r.x = (u.y * v.z) - (v.y * u.z);
    r.y = (v.x * u.z) - (u.x * v.z);
    r.z = (u.x * v.y) - (v.x * u.y);
    r.w = 1;
    return r;
}
static inline pt addvec(pt a, pt b) { /* utils_path.h:179-188 */
    pt r;
    hdiv(&a);
    hdiv(&b);
    r.x = a.x + b.x;
    r.y = a.y + b.y;
    r.z = a.z + b.z;
    r.w = 1;
    return r;
}
static inline pt smult(pt v, double f) { /* utils_path.h:190-199 */
    pt r;
    hdiv(&v);
    r.x = v.x * f;
    r.y = v.y * f;
    r.z = v.z * f;
    r.w = v.w;
    return r;
}
static inline void subvec(pt *a, pt *b) { /* subVectors b=b-a, utils_path.h:146-154 */
    hdiv(a);
    hdiv(b);
    b->x = b->x - a->x;
    b->y = b->y - a->y;
    b->z = b->z - a->z;
    b->w = 1;
}
static inline pt raypos(const pt *p0, const pt *d, double lambda) { /* rayPosition, utils_path.h:202-209 */
    pt r;
    r.x = p0->x + (lambda * d->x);
    r.y = p0->y + (lambda * d->y);
    r.z = p0->z + (lambda * d->z);
    r.w = 1;
    return r;
}

static void rot_y(double *T, double th) { /* RotateYMat, utils_path.c:1505-1521 */
    double R[16];
    memset(R, 0, sizeof(R));
    R[0] = cos(th);
    R[2] = sin(th);
    R[5] = 1.0;
    R[8] = -sin(th);
    R[10] = cos(th);
    R[15] = 1.0;
    matmul(R, T);
}
static void rot_z(double *T, double th) { /* RotateZMat, utils_path.c:1541-1557 */
    double R[16];
    memset(R, 0, sizeof(R));
    R[0] = cos(th);
    R[1] = -sin(th);
    R[4] = sin(th);
    R[5] = cos(th);
    R[10] = 1.0;
    R[15] = 1.0;
    matmul(R, T);
}
static const double EYE[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};

/* ------------------------------------------ ray / normal transforms */
/* rayTransform, utils_path.c:29-56: p0' = Tinv p0, d' = Tinv (T 0 + d). */
static void ray_transform(const ptx_object *o, const pt *p0, const pt *d, pt *tp0, pt *td) {
    pt t = {0, 0, 0, 1};
    matvec(o->T, &t);
    *tp0 = *p0;
    matvec(o->Tinv, tp0);
    td->x = t.x + d->x;
    td->y = t.y + d->y;
    td->z = t.z + d->z;
    td->w = d->w;
    matvec(o->Tinv, td);
}
/* normalTransform, utils_path.c:58-76: n = normalize(Tinv^T normalize(n0)). */
static void normal_transform(const ptx_object *o, const pt *n0, pt *n) {
    double TinvT[16];
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 4; j++) TinvT[4 * j + i] = o->Tinv[4 * i + j];
    *n = *n0;
    normalize(n);
    matvec(TinvT, n);
    normalize(n);
}

/* ------------------------------------------------ primitive intersectors */
/* planeHelper, utils_path.c:439-486 (arguments by value there as here). */
static double plane_helper(pt p0, pt d, pt P, pt n, double limit) {
    double lambda = -1;
    double verified = 0;
    if (!(fabs(dot(&d, &n)) < TOL)) {
        subvec(&p0, &P);
        lambda = dot(&P, &n) / dot(&d, &n);
        if (n.x == 0) {
            if (fabs(p0.x + lambda * d.x) <= limit - TOL) verified += 1;
        }
        if (n.y == 0) {
            if (fabs(p0.y + lambda * d.y) <= limit - TOL) verified += 1;
        }
        if (n.z == 0) {
            if (fabs(p0.z + lambda * d.z) <= limit - TOL) verified += 1;
        }
        if (verified == 2) return lambda;
        return -1;
    }
    return -1;
}

/* All intersectors: *lambda is -1 on entry (findFirstHit's lambdac). */

/* planeIntersect, utils_path.c:488-544 */
static void plane_intersect(const ptx_object *o, const pt *rp0, const pt *rd, double *lambda, pt *p, pt *n, double *a,
                            double *b) {
    pt tp0, td, nc, P = {0, 0, 0, 1};
    ray_transform(o, rp0, rd, &tp0, &td);
    if (tp0.z < 0) { nc.x = 0; nc.y = 0; nc.z = -1; nc.w = 1; }
    else { nc.x = 0; nc.y = 0; nc.z = 1; nc.w = 1; }
    *lambda = plane_helper(tp0, td, P, nc, 1);
    if (*lambda > TOL) {
        *p = raypos(&tp0, &td, *lambda);
        *a = (1 + p->x) / 2;
        *b = (1 + p->y) / 2;
        *p = raypos(rp0, rd, *lambda);
        normal_transform(o, &nc, n);
    }
}

/* sphereIntersect, utils_path.c:546-623 */
static void sphere_intersect(const ptx_object *o, const pt *rp0, const pt *rd, double *lambda, pt *p, pt *n, double *a,
                             double *b) {
    pt p0, d;
    double disc, l1, l2;
    ray_transform(o, rp0, rd, &p0, &d);
    double A = dot(&d, &d);
    double B = dot(&p0, &d);
    double C = dot(&p0, &p0) - 1;
    disc = (B * B) - A * C;
    if (disc < 0) {
        *lambda = -1;
    } else {
        l1 = (-B / A) + sqrt(disc) / A;
        l2 = (-B / A) - sqrt(disc) / A;
        if (l1 > TOL) *lambda = l1;
        if (l2 > TOL) {
            if (l1 <= TOL) *lambda = l2;
            else if (l1 > TOL && l2 < l1) *lambda = l2;
        }
    }
    if (*lambda > 0) {
        *p = raypos(&p0, &d, *lambda);
        double alpha = atan2(p->y, p->x);
        double beta = acos(p->z);
        *a = alpha / (2 * PI);
        *b = beta / PI;
        pt nc = *p;
        normalize(&nc);
        normal_transform(o, &nc, n);
        *p = raypos(rp0, rd, *lambda);
    }
}

/* cylIntersect, utils_path.c:625-779 */
static void cyl_intersect(const ptx_object *o, const pt *rp0, const pt *rd, double *lambda, pt *p, pt *n, double *a,
                          double *b) {
    double det, A, B, C, l1, l2;
    pt tp0, td, nc;
    nc.x = 0;
    nc.y = 0;
    nc.z = 0;
    nc.w = 1; /* reference leaves z,w unset until a hit assigns them */
    *lambda = INFINITY;
    ray_transform(o, rp0, rd, &tp0, &td);
    double dx = td.x / td.w, dy = td.y / td.w, dz = td.z / td.w;
    double px = tp0.x / tp0.w, py = tp0.y / tp0.w, pz = tp0.z / tp0.w;
    A = dx * dx + dy * dy;
    B = 2 * px * dx + 2 * py * dy;
    C = px * px + py * py - 1;
    det = B * B - 4 * A * C;
    if (det < 0) {
        *lambda = -1;
    } else {
        l1 = (-1 * B + sqrt(det)) / (2 * A);
        l2 = (-1 * B - sqrt(det)) / (2 * A);
        if (fabs(pz + l1 * dz) <= 1 && l1 > TOL) {
            if (*lambda > l1) {
                *lambda = l1;
                *p = raypos(&tp0, &td, *lambda);
                nc.x = p->x; nc.y = p->y; nc.z = 0; nc.w = 1;
                double alpha = atan2(p->y, p->x);
                *a = (alpha + PI) / (2 * PI);
                *b = (p->z + 1) / 2;
            }
        }
        if (fabs(pz + l2 * dz) <= 1 && l2 > TOL) {
            if (*lambda > l2) {
                *lambda = l2;
                *p = raypos(&tp0, &td, *lambda);
                nc.x = p->x; nc.y = p->y; nc.z = 0; nc.w = 1;
                double alpha = atan2(p->y, p->x);
                *a = (alpha + PI) / (2 * PI);
                *b = (p->z + 1) / 2;
            }
        }
    }
    l1 = (-1 - pz) / dz;
    l2 = (1 - pz) / dz;
    if ((px + l1 * dx) * (px + l1 * dx) + (py + l1 * dy) * (py + l1 * dy) <= 1 && l1 > TOL) {
        if (*lambda > l1) {
            *lambda = l1;
            nc.x = 0; nc.y = 0; nc.z = -1; nc.w = 1;
            *p = raypos(&tp0, &td, *lambda);
            *a = p->x / 2 + 0.5;
            *b = p->y / 2 + 0.5;
        }
    }
    if ((px + l2 * dx) * (px + l2 * dx) + (py + l2 * dy) * (py + l2 * dy) <= 1 && l2 > TOL) {
        if (*lambda > l2) {
            *lambda = l2;
            nc.x = 0; nc.y = 0; nc.z = 1; nc.w = 1;
            *p = raypos(&tp0, &td, *lambda);
            *a = p->x / 2 + 0.5;
            *b = p->y / 2 + 0.5;
        }
    }
    if (*lambda == INFINITY) *lambda = -1;
    if (*lambda > TOL) {
        *p = raypos(rp0, rd, *lambda);
        normal_transform(o, &nc, n);
    }
}

/* triangleIntersect, utils_path.c:781-830.  a,b are never written there. */
static void triangle_intersect(const ptx_object *o, const pt *rp0, const pt *rd, double *lambda, pt *p, pt *n, double *a,
                               double *b) {
    pt p1 = {o->v[0][0], o->v[0][1], o->v[0][2], o->v[0][3]};
    pt p2 = {o->v[1][0], o->v[1][1], o->v[1][2], o->v[1][3]};
    pt p3 = {o->v[2][0], o->v[2][1], o->v[2][2], o->v[2][3]};
    pt e12 = addvec(p2, smult(p1, -1));
    pt e23 = addvec(p3, smult(p2, -1));
    pt e31 = addvec(p1, smult(p3, -1));
    pt e13 = addvec(p3, smult(p1, -1));
    pt e32 = addvec(p2, smult(p3, -1));
    pt e21 = addvec(p1, smult(p2, -1));
    pt normal = cross2(e12, e13);
    pt rdd = *rd, rpp = *rp0;
    normalize(&normal);
    double d_dot_n = dot(&normal, &rdd);
    (void)a;
    (void)b;
    if (-TOL < d_dot_n && d_dot_n < TOL) {
        *lambda = -1;
    } else {
        pt temp = addvec(p1, smult(rpp, -1));
        double l2 = dot(&temp, &normal) / dot(&rdd, &normal);
        if (l2 > 0) {
            pt poi = raypos(rp0, rd, l2);
            pt e1i = addvec(poi, smult(p1, -1));
            pt e2i = addvec(poi, smult(p2, -1));
            pt e3i = addvec(poi, smult(p3, -1));
            double v1 = dot2(cross2(e1i, e12), cross2(e13, e12));
            double v2 = dot2(cross2(e2i, e23), cross2(e21, e23));
            double v3 = dot2(cross2(e3i, e31), cross2(e32, e31));
            if (v1 >= 0 && v2 >= 0 && v3 >= 0) {
                *lambda = l2;
                *p = raypos(rp0, rd, *lambda);
                if (d_dot_n > 0) *n = smult(normal, -1);
                else *n = normal;
            } else {
                *lambda = -1;
            }
        } else {
            *lambda = -1;
        }
    }
}

/* cubeIntersect, utils_path.c:832-995, and roomIntersect, :997-1143: six
 * planeHelper calls in the order +z -z +y -y +x -x.  The room's normals
 * point inward and it writes no a,b. */
static void box_faces_intersect(const ptx_object *o, int room, const pt *rp0, const pt *rd, double *lambda, pt *p, pt *n,
                                double *a, double *b) {
    static const double face[6][3] = {{0, 0, 1}, {0, 0, -1}, {0, 1, 0}, {0, -1, 0}, {1, 0, 0}, {-1, 0, 0}};
    pt tp0, td;
    double lc;
    *lambda = INFINITY;
    ray_transform(o, rp0, rd, &tp0, &td);
    for (int f = 0; f < 6; f++) {
        pt P = {face[f][0], face[f][1], face[f][2], 1};
        pt nc = P;
        if (room) { nc.x = -P.x; nc.y = -P.y; nc.z = -P.z; }
        /* the reference writes the zero components as literal 0 */
        if (nc.x == 0) nc.x = 0;
        if (nc.y == 0) nc.y = 0;
        if (nc.z == 0) nc.z = 0;
        lc = plane_helper(tp0, td, P, nc, 1);
        if (lc > TOL && lc < *lambda) {
            *lambda = lc;
            if (!room) {
                pt q = raypos(&tp0, &td, *lambda);
                if (f < 2) { *a = (1 + q.x) / 2; *b = (1 + q.y) / 2; }
                else if (f < 4) { *a = (1 + q.x) / 2; *b = (1 + q.z) / 2; }
                else { *a = (1 + q.y) / 2; *b = (1 + q.z) / 2; }
            }
            *p = raypos(rp0, rd, *lambda);
            normal_transform(o, &nc, n);
        }
    }
    if (*lambda == INFINITY) *lambda = -1;
}

static void intersect(const ptx_object *o, const pt *p0, const pt *d, double *lambda, pt *p, pt *n, double *a, double *b) {
    switch (o->type) {
    case PTX_PLANE: plane_intersect(o, p0, d, lambda, p, n, a, b); break;
    case PTX_SPHERE: sphere_intersect(o, p0, d, lambda, p, n, a, b); break;
    case PTX_CYL: cyl_intersect(o, p0, d, lambda, p, n, a, b); break;
    case PTX_TRIANGLE: triangle_intersect(o, p0, d, lambda, p, n, a, b); break;
    case PTX_CUBE: box_faces_intersect(o, 0, p0, d, lambda, p, n, a, b); break;
    default: box_faces_intersect(o, 1, p0, d, lambda, p, n, a, b); break;
    }
}

/* box_intersect, magic.c:457-474 */
static int box_intersect(const pt *p0, const pt *d, const double *box) {
    double l1 = (box[0] - p0->x) / d->x;
    double l2 = (box[1] - p0->x) / d->x;
    double l3 = (box[2] - p0->y) / d->y;
    double l4 = (box[3] - p0->y) / d->y;
    double l5 = (box[4] - p0->z) / d->z;
    double l6 = (box[5] - p0->z) / d->z;
    double lmin = rmax(rmax(rmin(l1, l2), rmin(l3, l4)), rmin(l5, l6));
    double lmax = rmin(rmin(rmax(l1, l2), rmax(l3, l4)), rmax(l5, l6));
    return (lmin <= lmax && lmax >= 0);
}

/* findFirstHit, PathTracer.c:198-251.  os = index of the object to skip or
 * -1; *obj untouched on a miss; *lambda must be -1 on entry. */
static void first_hit(const ptx_scene_desc *s, int node, const pt *p0, const pt *d, double *lambda, int os, int *obj,
                      pt *p, pt *n, double *a, double *b) {
    if (node < 0 || node >= s->n_nodes) return;
    const ptx_node *bt = &s->nodes[node];
    if (!box_intersect(p0, d, bt->box)) return;
    double best = *lambda;
    for (int k = bt->first_obj; k < bt->first_obj + bt->n_objs; k++) {
        const ptx_object *o = &s->objects[k];
        if (k != os && box_intersect(p0, d, o->box)) {
            double lc = -1;
            pt ip = {0, 0, 0, 1}, in = {0, 0, 0, 1};
            double ia = 0, ib = 0;
            intersect(o, p0, d, &lc, &ip, &in, &ia, &ib);
            if ((best < TOL || lc < best) && lc > TOL) {
                best = lc;
                *lambda = best;
                *obj = k;
                *p = ip;
                *n = in;
                normalize(n);
                *a = ia;
                *b = ib;
            }
        }
    }
    first_hit(s, bt->child1, p0, d, lambda, os, obj, p, n, a, b);
    first_hit(s, bt->child2, p0, d, lambda, os, obj, p, n, a, b);
}

/* --------------------------------------------------- texture look-ups */
static inline double texel(const double *data, long n, long idx) {
    if (idx < 0) idx = 0;
    if (idx >= n) idx = n - 1;
    return data[idx];
}
/* texMap, utils_path.c:1322-1374 */
static void tex_map(const ptx_image *im, const double *data, double a, double b, double *R, double *G, double *B) {
    double x = (im->sx - 1) * a;
    double y = (im->sy - 1) * b;
    int x1 = (im->sx - 1) * a;
    int x2 = x1 + 1;
    int y1 = (im->sy - 1) * b;
    int y2 = y1 + 1;
    long n = (long)im->sx * im->sy * 3;
    double c11[3], c12[3], c21[3], c22[3], out[3];
    for (int c = 0; c < 3; c++) {
        c11[c] = texel(data, n, 3L * ((long)y1 * im->sx + x1) + c);
        c12[c] = texel(data, n, 3L * ((long)y2 * im->sx + x1) + c);
        c21[c] = texel(data, n, 3L * ((long)y1 * im->sx + x2) + c);
        c22[c] = texel(data, n, 3L * ((long)y2 * im->sx + x2) + c);
    }
    for (int c = 0; c < 3; c++) {
        double avg1 = c11[c] * (x2 - x) / (x2 - x1) + c21[c] * (x - x1) / (x2 - x1);
        double avg2 = c12[c] * (x2 - x) / (x2 - x1) + c22[c] * (x - x1) / (x2 - x1);
        out[c] = avg1 * (y2 - y) / (y2 - y1) + avg2 * (y - y1) / (y2 - y1);
    }
    *R = out[0];
    *G = out[1];
    *B = out[2];
}
/* alphaMap, utils_path.c:1376-1406 */
static void alpha_map(const ptx_image *im, const double *data, double a, double b, double *alpha) {
    double x = (im->sx - 1) * a;
    double y = (im->sy - 1) * b;
    int x1 = (im->sx - 1) * a;
    int x2 = x1 + 1;
    int y1 = (im->sy - 1) * b;
    int y2 = y1 + 1;
    long n = (long)im->sx * im->sy;
    double a11 = texel(data, n, (long)y1 * im->sx + x1);
    double a12 = texel(data, n, (long)y2 * im->sx + x1);
    double a21 = texel(data, n, (long)y1 * im->sx + x2);
    double a22 = texel(data, n, (long)y2 * im->sx + x2);
    double avg1 = a11 * (x2 - x) / (x2 - x1) + a21 * (x - x1) / (x2 - x1);
    double avg2 = a12 * (x2 - x) / (x2 - x1) + a22 * (x - x1) / (x2 - x1);
    *alpha = avg1 * (y2 - y) / (y2 - y1) + avg2 * (y - y1) / (y2 - y1);
}
/* vectorReorient, utils_path.c:98-114 */
static void vector_reorient(pt *d, const pt *n) {
    double R[16];
    memcpy(R, EYE, sizeof(R));
    double theta = atan2(n->y, n->x);
    double phi = acos(n->z);
    rot_y(R, phi);
    rot_z(R, theta);
    matvec(R, d);
}
/* normalDeform, utils_path.c:1408-1428 */
static void normal_deform(const orc_scene *s, const ptx_object *o, double a, double b, pt *n) {
    pt nd;
    tex_map(&s->d.images[o->normal_img], s->img[o->normal_img], a, b, &nd.x, &nd.y, &nd.z);
    nd.x = (nd.x * 2) - 1;
    nd.y = (nd.y * 2) - 1;
    nd.z = (nd.z * 2) - 1;
    nd.w = 1;
    normalize(&nd);
    normalize(n);
    vector_reorient(&nd, n);
    *n = nd;
}

/* ------------------------------------------------------------ sampling */
/* cosWeightedSample, utils_path.c:1224-1265 */
static void cos_weighted_sample(const pt *n, pt *d, rng_t *rng) {
    double u1, r, theta, phi, x, y, z;
    double R[16];
    u1 = rnd(rng);
    r = sqrt(u1);
    theta = 2 * PI * rnd(rng);
    x = r * cos(theta);
    y = r * sin(theta);
    z = sqrt(1.0 - (x * x) - (y * y));
    memcpy(R, EYE, sizeof(R));
    theta = atan2(n->y, n->x);
    phi = acos(n->z);
    rot_y(R, phi);
    rot_z(R, theta);
    d->x = x;
    d->y = y;
    d->z = z;
    d->w = 1.0;
    matvec(R, d);
}
/* reorient, utils_path.c:87-96 (Rodrigues) */
static pt reorient(pt from, pt to, pt v) {
    pt axis = cross2(from, to);
    normalize(&axis);
    double angle = acos(dot2(from, to));
    return addvec(addvec(smult(v, cos(angle)), smult(cross2(axis, v), sin(angle))),
                  smult(axis, dot2(axis, v) * (1 - cos(angle))));
}
/* hemisphereReorient, utils_path.c:79-82 */
static pt hemisphere_reorient(pt m, pt v) {
    pt up = {0, 1, 0, 1};
    return reorient(up, m, v);
}
/* random_normal, PathTracer.c:46-51.  The reference's compiled code calls
 * drand48() for the log() factor first and for the cos() factor second
 * (C leaves the order open; this order is pinned against oracle/_ref). */
static double random_normal(rng_t *rng) {
    double u_log = rnd(rng);
    double u_cos = rnd(rng);
    return sqrt(-2 * log(u_log)) * cos(2 * PI * u_cos);
}

/* Path state = the fields of struct ray3D that survive a bounce
 * (PathTracer.h:76-90).  mem[] models the neighbourhood of r_index[] so
 * that refractRay's out-of-range indices read and write what they do in
 * the reference on x86-64:
 *   mem[0..2]=col  mem[3..5]=I  mem[6..9]=srcN  mem[10..19]=r_index[0..9]
 *   mem[20] = the 8 bytes holding r_index_index (low 4) and padding. */
typedef struct {
    pt p0, d;
    double mem[21];
    int idx; /* r_index_index */
} path_t;
#define COL(ps, c) ((ps)->mem[c])
#define INT(ps, c) ((ps)->mem[3 + (c)])
#define RBASE 10

static double rindex_load(path_t *ps, int k, orc_stats *st) {
    if (k < 0 || k > 9) st->rindex_oob++;
    if (k == 10) { union { double d; int32_t i[2]; } u; u.d = 0; u.i[0] = ps->idx; return u.d; }
    if (k < -RBASE || k > 10) return 1.0;
    return ps->mem[RBASE + k];
}
static void rindex_store(path_t *ps, int k, double v, orc_stats *st) {
    if (k < 0 || k > 9) st->rindex_oob++;
    if (k == 10) { union { double d; int32_t i[2]; } u; u.d = v; ps->idx = u.i[0]; return; }
    if (k < -RBASE || k > 10) return;
    ps->mem[RBASE + k] = v;
}

/* reflectRay, PathTracer.c:53-72 (normalises *normal in place) */
static void reflect_ray(path_t *ps, const pt *p, pt *normal) {
    normalize(&ps->d);
    normalize(normal);
    pt r;
    r.x = ps->d.x - 2 * dot(&ps->d, normal) * normal->x;
    r.y = ps->d.y - 2 * dot(&ps->d, normal) * normal->y;
    r.z = ps->d.z - 2 * dot(&ps->d, normal) * normal->z;
    r.w = 1;
    normalize(&r);
    ps->p0 = *p;
    ps->d = r;
}
/* refractRay, PathTracer.c:74-140 */
static void refract_ray(path_t *ps, const pt *p, pt *normal, const ptx_object *obj, rng_t *rng, orc_stats *st) {
    pt dref;
    double n1 = rindex_load(ps, ps->idx, st);
    double n2 = obj->r_index;
    double r = n1 / n2;
    pt rn;
    int leaving = 0;
    normalize(&ps->d);
    normalize(normal);
    if (dot2(ps->d, *normal) < 0) {
        rn = *normal;
    } else {
        n1 = obj->r_index;
        n2 = rindex_load(ps, ps->idx - 1, st);
        r = n1 / n2;
        rn = smult(*normal, -1);
        leaving = 1;
    }
    double c = dot2(smult(rn, -1), ps->d);
    double disc = 1 - (r * r) * (1 - (c * c));
    double q = ((n1 - n2) / (n1 + n2));
    double r0 = q * q;
    double costheta = -dot2(rn, ps->d);
    double reflectance = r0 + ((1 - r0) * pow(1 - costheta, 5));
    double dice = rnd(rng);
    if (disc < 0 || dice <= reflectance) {
        reflect_ray(ps, p, &rn);
        return;
    } else {
        if (leaving) {
            ps->idx = ps->idx - 1;
        } else {
            ps->idx = ps->idx + 1;
            rindex_store(ps, ps->idx, obj->r_index, st);
        }
        dref = addvec(smult(ps->d, r), smult(rn, r * c - sqrt(disc)));
    }
    normalize(&dref);
    ps->p0 = *p;
    ps->d = dref;
}

/* Primary ray: PathTracer.c:625-641 with sampleLensPoint (:142-158),
 * focusRay (:160-195) and initRay (utils_path.h:211-230). */
/* jitter: PTX_CLEAN's random point of the pixel instead of its centre (not in the reference) */
static void primary_ray_j(const ptx_camera *cam, int sx, int i, int j, double aperture, const pt *fp, rng_t *rng, path_t *ps, int jitter) {
    double du = cam->wsize / (sx - 1);
    double dv = -cam->wsize / (sx - 1);
    pt pc = {cam->wl + (i * du), cam->wt + (j * dv), cam->f, 1};
    if (jitter) {
        double ju = rnd(rng), jv = rnd(rng);
        pc.x = cam->wl + ((i + (ju - 0.5)) * du);
        pc.y = cam->wt + ((j + (jv - 0.5)) * dv);
    }
    /* sampleLensPoint */
    double r = aperture * sqrt(rnd(rng));
    double theta = rnd(rng) * 2 * PI;
    pt pl;
    pl.x = pc.x + r * cos(theta);
    pl.y = pc.y + r * sin(theta);
    pl.z = cam->f;
    pl.w = 1;
    matvec(cam->C2W, &pl);
    matvec(cam->C2W, &pc);
    /* focusRay(lp=pl, c=pc, fp) */
    pt e = {cam->e[0], cam->e[1], cam->e[2], cam->e[3]};
    pt w = {cam->w[0], cam->w[1], cam->w[2], cam->w[3]};
    pt rayd = pc;
    subvec(&e, &rayd);
    normalize(&rayd);
    pt fpc = *fp;
    subvec(&pc, &fpc);
    double lambda = dot(&fpc, &w) / dot(&rayd, &w);
    pt hit = raypos(&pc, &rayd, lambda);
    pt d;
    d.x = hit.x - pl.x;
    d.y = hit.y - pl.y;
    d.z = hit.z - pl.z;
    d.w = 1;
    normalize(&d);
    /* initRay */
    memset(ps, 0, sizeof(*ps));
    ps->p0 = pl;
    ps->d = d;
    ps->idx = 0;
    ps->mem[RBASE + 0] = 1;
    COL(ps, 0) = COL(ps, 1) = COL(ps, 2) = 1.0;
    INT(ps, 0) = INT(ps, 1) = INT(ps, 2) = 0;
    ps->mem[6] = 0; ps->mem[7] = 0; ps->mem[8] = 1; ps->mem[9] = 1; /* srcN */
}
static void primary_ray(const ptx_camera *cam, int sx, int i, int j, double aperture, const pt *fp, rng_t *rng, path_t *ps) {
    primary_ray_j(cam, sx, i, j, aperture, fp, rng, ps, 0);
}

/* Light point samplers: planeSample/planeCoordinates (utils_path.c:1189,
 * :1149), sphereSample/sphereCoordinates (:1199, :1164).  cylSample
 * (:1211) draws two numbers and cylCoordinates (:1179) returns nothing:
 * the reference then uses uninitialised rx,ry,rz; the oracle uses 0,0,0.
 * Triangle, cube and room lights have randomPoint == NULL there (a crash);
 * here they draw nothing and sample 0,0,0. */
static void light_sample(const ptx_object *ls, rng_t *rng, double *x, double *y, double *z) {
    *x = *y = *z = 0;
    if (ls->type == PTX_PLANE) {
        double v1 = rnd(rng);
        double v2 = rnd(rng);
        pt p = {v1 * 2 - 1, v2 * 2 - 1, 0, 1};
        matvec(ls->T, &p);
        *x = p.x; *y = p.y; *z = p.z;
    } else if (ls->type == PTX_SPHERE) {
        double v1 = rnd(rng);
        double v2 = rnd(rng);
        double theta = 2 * PI * v1;
        double phi = acos(2 * v2 - 1);
        pt p = {cos(theta) * sin(phi), sin(theta) * sin(phi), cos(phi), 1};
        matvec(ls->T, &p);
        *x = p.x; *y = p.y; *z = p.z;
    } else if (ls->type == PTX_CYL) {
        (void)rnd(rng);
        (void)rnd(rng);
    }
}

static int g_debug = 0;
/* Prints the state after every bounce to stderr (debugging aid for the tests). */
void orc_set_debug(int on) { g_debug = on; }

/* PathTrace, PathTracer.c:253-484, with the tail recursion (:483) written
 * as a loop.  Returns the path's colour in col[3]. */
static void path_trace(const orc_scene *sc, path_t *ps, int max_depth, int flags, rng_t *rng, double col[3],
                       orc_stats *st) {
    const ptx_scene_desc *s = &sc->d;
    int os = -1, CEL = 0;
    for (int depth = 1;; depth++) {
        double lambda = -1, a = 0, b = 0, R, G, B, dice;
        int oi = -1;
        pt p = {0, 0, 0, 1}, n = {0, 0, 0, 1};
        if (depth > max_depth) { st->end_depth++; break; }
        normalize(&ps->d);
        st->closest_hits++;
        first_hit(s, 0, &ps->p0, &ps->d, &lambda, os, &oi, &p, &n, &a, &b);
        if (lambda < 0) { st->end_miss++; break; }
        const ptx_object *obj = &s->objects[oi];
        if (obj->tex_img >= 0) {
            tex_map(&s->images[obj->tex_img], sc->img[obj->tex_img], a, b, &R, &G, &B);
        } else {
            R = obj->col[0]; G = obj->col[1]; B = obj->col[2];
        }
        if (obj->normal_img >= 0) normal_deform(sc, obj, a, b, &n);
        if (obj->is_light) {
            if (!CEL) {
                INT(ps, 0) = INT(ps, 0) + (COL(ps, 0) * R);
                INT(ps, 1) = INT(ps, 1) + (COL(ps, 1) * G);
                INT(ps, 2) = INT(ps, 2) + (COL(ps, 2) * B);
            }
            st->end_light++;
            break;
        }
        double RR_prob = 0;
        if (flags & PTX_USE_RR) {
            RR_prob = 1 - rmax(rmax(COL(ps, 0), COL(ps, 2)), COL(ps, 1));
            if (RR_prob >= .98) RR_prob = .98;
        }
        if (RR_prob > rnd(rng)) { st->end_rr++; break; }
        double diffusivity, transparency, reflectivity;
        if (obj->alpha_img >= 0) {
            alpha_map(&s->images[obj->alpha_img], sc->img[obj->alpha_img], a, b, &transparency);
            reflectivity = (1 - transparency) * obj->reflPct / (obj->reflPct + obj->diffPct);
            diffusivity = (1 - transparency) * obj->diffPct / (obj->reflPct + obj->diffPct);
        } else {
            transparency = obj->tranPct;
            reflectivity = obj->reflPct;
            diffusivity = obj->diffPct;
        }
        dice = rnd(rng);
        if (dice <= diffusivity) {
            pt dir;
            double probability;
            if (flags & PTX_USE_IS) {
                cos_weighted_sample(&n, &dir, rng);
                normalize(&dir);
                probability = dot2(n, dir);
            } else {
                double angle_2 = acos(2 * rnd(rng) - 1);
                (void)rnd(rng);
                (void)rnd(rng);
                double angle_1 = 2 * PI * rnd(rng);
                dir.z = -cos(angle_1) * sin(angle_2);
                dir.x = sin(angle_1) * sin(angle_2);
                dir.y = fabs(cos(angle_2));
                dir.w = 1;
                normalize(&dir);
                dir = hemisphere_reorient(n, dir);
                probability = 1;
            }
            COL(ps, 0) *= R * dot2(n, dir) / probability / (1 - RR_prob);
            COL(ps, 1) *= G * dot2(n, dir) / probability / (1 - RR_prob);
            COL(ps, 2) *= B * dot2(n, dir) / probability / (1 - RR_prob);
            ps->p0 = p;
            ps->d = dir;
            if (flags & PTX_USE_ES) {
                /* light choice: walk the list until the running weight reaches u */
                double ls_num = rnd(rng);
                int li = 0;
                double acc = s->objects[s->lights[0]].LSweight;
                while (acc < ls_num) {
                    li++;
                    if (li >= s->n_lights) { li = s->n_lights - 1; break; } /* reference walks off the list (crash) */
                    acc += s->objects[s->lights[li]].LSweight;
                }
                const ptx_object *ls = &s->objects[s->lights[li]];
                double rx, ry, rz;
                light_sample(ls, rng, &rx, &ry, &rz);
                pt ed = {rx - p.x, ry - p.y, rz - p.z, 1};
                normalize(&ed);
                double lambda_es = -1, ea = 0, eb = 0;
                pt en = {0, 0, 0, 1}, ep = {0, 0, 0, 1};
                int hi = -1;
                st->shadow_rays++;
                first_hit(s, 0, &p, &ed, &lambda_es, oi, &hi, &ep, &en, &ea, &eb);
                if (lambda_es > 0 && s->objects[hi].is_light) {
                    const ptx_object *ho = &s->objects[hi];
                    CEL = 1;
                    double d2 = (rx - p.x) * (rx - p.x) + (ry - p.y) * (ry - p.y) + (rz - p.z) * (rz - p.z);
                    double weight2 = dot(&n, &ed) * -dot(&en, &ed);
                    double weight = rmin(1, (ho->surfaceArea * weight2) / d2);
                    if ((flags & PTX_CLEAN) && weight < 0) weight = 0;
                    INT(ps, 0) += COL(ps, 0) * R * ho->col[0] * weight / (1 - RR_prob);
                    INT(ps, 1) += COL(ps, 1) * G * ho->col[1] * weight / (1 - RR_prob);
                    INT(ps, 2) += COL(ps, 2) * B * ho->col[2] * weight / (1 - RR_prob);
                } else {
                    CEL = 0;
                }
                if (flags & PTX_CLEAN) CEL = 1;
            }
        } else if (dice <= (diffusivity + reflectivity)) {
            if (flags & PTX_CLEAN) CEL = 0;
            COL(ps, 0) *= R / (1 - RR_prob);
            COL(ps, 1) *= G / (1 - RR_prob);
            COL(ps, 2) *= B / (1 - RR_prob);
            reflect_ray(ps, &p, &n);
            if (obj->refl_sig > 0) {
                ps->d.x += obj->refl_sig * random_normal(rng);
                ps->d.y += obj->refl_sig * random_normal(rng);
                ps->d.z += obj->refl_sig * random_normal(rng);
            }
        } else if (dice <= (diffusivity + reflectivity + transparency)) {
            if (flags & PTX_CLEAN) CEL = 0;
            COL(ps, 0) *= R / (1 - RR_prob);
            COL(ps, 1) *= G / (1 - RR_prob);
            COL(ps, 2) *= B / (1 - RR_prob);
            refract_ray(ps, &p, &n, obj, rng, st);
            oi = -1;
        }
        ps->mem[6] = n.x; ps->mem[7] = n.y; ps->mem[8] = n.z; ps->mem[9] = n.w; /* ray->srcN = n, :482 */
        os = oi;
        if (g_debug)
            fprintf(stderr, "  orc depth %d os %d p0 %.17g %.17g %.17g d %.17g %.17g %.17g col %.6g %.6g %.6g I %.6g %.6g %.6g ridx %d\n",
                    depth + 1, os, ps->p0.x, ps->p0.y, ps->p0.z, ps->d.x, ps->d.y, ps->d.z, COL(ps, 0), COL(ps, 1), COL(ps, 2),
                    INT(ps, 0), INT(ps, 1), INT(ps, 2), ps->idx);
    }
    col[0] = INT(ps, 0);
    col[1] = INT(ps, 1);
    col[2] = INT(ps, 2);
}

/* ------------------------------------------------------------ public API */
orc_scene *orc_scene_create(const ptx_scene_desc *desc) {
    orc_scene *s = (orc_scene *)calloc(1, sizeof(*s));
    s->d = *desc;
    ptx_object *o = (ptx_object *)malloc(sizeof(ptx_object) * (size_t)(desc->n_objects + 1));
    ptx_node *nd = (ptx_node *)malloc(sizeof(ptx_node) * (size_t)(desc->n_nodes + 1));
    int32_t *l = (int32_t *)malloc(4 * (size_t)(desc->n_lights + 1));
    ptx_image *im = (ptx_image *)malloc(sizeof(ptx_image) * (size_t)(desc->n_images + 1));
    uint8_t *tx = (uint8_t *)malloc(desc->n_texel_bytes + 1);
    memcpy(o, desc->objects, sizeof(ptx_object) * (size_t)desc->n_objects);
    memcpy(nd, desc->nodes, sizeof(ptx_node) * (size_t)desc->n_nodes);
    memcpy(l, desc->lights, 4 * (size_t)desc->n_lights);
    memcpy(im, desc->images, sizeof(ptx_image) * (size_t)desc->n_images);
    memcpy(tx, desc->texels, desc->n_texel_bytes);
    s->d.objects = o; s->d.nodes = nd; s->d.lights = l; s->d.images = im; s->d.texels = tx;
    s->img = (double **)calloc((size_t)desc->n_images + 1, sizeof(double *));
    for (int j = 0; j < desc->n_images; j++) {
        size_t cnt = (size_t)im[j].sx * im[j].sy * im[j].channels;
        s->img[j] = (double *)malloc(sizeof(double) * (cnt + 1));
        const uint8_t *src = tx + im[j].data_offset;
        /* utils_path.c:1849 divides by the int maxval, :1921 by 255.0 */
        for (size_t k = 0; k < cnt; k++) s->img[j][k] = ((double)src[k]) / im[j].divisor;
    }
    return s;
}
orc_scene *orc_scene_load(const char *path) {
    ptx_scene_desc d;
    if (ptx_scene_read(path, &d) != 0) return NULL;
    orc_scene *s = orc_scene_create(&d);
    ptx_scene_free(&d);
    return s;
}
void orc_scene_destroy(orc_scene *s) {
    if (!s) return;
    for (int j = 0; j < s->d.n_images; j++) free(s->img[j]);
    free(s->img);
    ptx_scene_free(&s->d);
    free(s);
}
const ptx_scene_desc *orc_scene_desc(const orc_scene *s) { return &s->d; }

void orc_first_hit(const orc_scene *s, const ptx_ray *rays, int n, ptx_hit *hits) {
    /* rays are independent; threads only matter for scenes of a million objects */
#pragma omp parallel for schedule(dynamic, 16) if (n >= 64 && s->d.n_objects > 100000)
    for (int i = 0; i < n; i++) {
        pt p0 = {rays[i].p0[0], rays[i].p0[1], rays[i].p0[2], 1};
        pt d = {rays[i].d[0], rays[i].d[1], rays[i].d[2], 1};
        double lambda = -1, a = 0, b = 0;
        int obj = -1;
        pt p = {0, 0, 0, 1}, nn = {0, 0, 0, 1};
        first_hit(&s->d, 0, &p0, &d, &lambda, rays[i].os, &obj, &p, &nn, &a, &b);
        ptx_hit *h = &hits[i];
        memset(h, 0, sizeof(*h));
        h->lambda = lambda;
        h->obj = -1;
        if (lambda > 0 && obj >= 0) {
            h->obj = obj;
            h->p[0] = p.x; h->p[1] = p.y; h->p[2] = p.z;
            h->n[0] = nn.x; h->n[1] = nn.y; h->n[2] = nn.z;
            h->a = a; h->b = b;
        }
    }
}

static void stats_add(orc_stats *a, const orc_stats *b) {
    a->paths += b->paths; a->closest_hits += b->closest_hits; a->shadow_rays += b->shadow_rays;
    a->rindex_oob += b->rindex_oob; a->end_depth += b->end_depth; a->end_miss += b->end_miss;
    a->end_light += b->end_light; a->end_rr += b->end_rr;
}

/* Sample loop, PathTracer.c:617-654 */
void orc_render(const orc_scene *s, const ptx_render_params *prm, int k0, int nk, double *rgb, double *wght, int threads,
                uint64_t *lcg_state, orc_stats *stats) {
    const ptx_camera *cam = &s->d.cam;
    int sx = prm->sx;
    pt fp = {0, 0, cam->focus_distance, 1};
    matvec(cam->C2W, &fp);
    int par = prm->rng_mode == PTX_RNG_COUNTER && threads != 1;
    if (threads <= 0) threads = omp_get_max_threads();
    int n_local = ptx_rows_owned(sx, prm->rank, prm->n_ranks, prm->band_rows);
    orc_stats total;
    memset(&total, 0, sizeof(total));
    for (int k = k0; k < k0 + nk; k++) {
#pragma omp parallel if (par) num_threads(threads)
        {
            orc_stats st;
            memset(&st, 0, sizeof(st));
#pragma omp for schedule(dynamic, 1)
            for (int jl = 0; jl < n_local; jl++) {
                int j = ptx_global_row(jl, prm->rank, prm->n_ranks, prm->band_rows);
                for (int i = 0; i < sx; i++) {
                    rng_t rng;
                    rng.mode = prm->rng_mode == PTX_RNG_COUNTER ? 0 : 1;
                    rng.key = ptx_path_key(prm->seed, (uint64_t)j * sx + i, (uint64_t)k);
                    rng.draw = 0;
                    rng.lcg = lcg_state;
                    rng.script = NULL;
                    path_t ps;
                    double col[3];
                    primary_ray_j(cam, sx, i, j, prm->aperture, &fp, &rng, &ps, prm->flags & PTX_CLEAN);
                    double wt = wght[i + j * sx];
                    path_trace(s, &ps, prm->max_depth, prm->flags, &rng, col, &st);
                    st.paths++;
                    if (prm->flags & PTX_CLEAN) { /* linear accumulator */
                        rgb[(i + j * sx) * 3 + 0] += col[0];
                        rgb[(i + j * sx) * 3 + 1] += col[1];
                        rgb[(i + j * sx) * 3 + 2] += col[2];
                        wght[i + j * sx] = wt + 1;
                        continue;
                    }
                    rgb[(i + j * sx) * 3 + 0] += col[0] * pow(2, -log(wt));
                    rgb[(i + j * sx) * 3 + 1] += col[1] * pow(2, -log(wt));
                    rgb[(i + j * sx) * 3 + 2] += col[2] * pow(2, -log(wt));
                    wt += col[0];
                    wt += col[1];
                    wt += col[2];
                    wght[i + j * sx] = wt;
                }
            }
#pragma omp critical
            stats_add(&total, &st);
        }
    }
    if (stats) stats_add(stats, &total);
}

void orc_trace_pixels(const orc_scene *s, const ptx_render_params *prm, const int32_t *pix_i, const int32_t *pix_j,
                      const int32_t *samp, int n, double *out) {
    const ptx_camera *cam = &s->d.cam;
    pt fp = {0, 0, cam->focus_distance, 1};
    matvec(cam->C2W, &fp);
    orc_stats st;
    memset(&st, 0, sizeof(st));
    for (int q = 0; q < n; q++) {
        rng_t rng = {0, ptx_path_key(prm->seed, (uint64_t)pix_j[q] * prm->sx + pix_i[q], (uint64_t)samp[q]), 0, NULL, NULL};
        path_t ps;
        primary_ray_j(cam, prm->sx, pix_i[q], pix_j[q], prm->aperture, &fp, &rng, &ps, prm->flags & PTX_CLEAN);
        path_trace(s, &ps, prm->max_depth, prm->flags, &rng, &out[3 * q], &st);
    }
}

/* ------------------------------------------------------------------ KATs */
static rng_t script_rng(const double *script) {
    rng_t r = {2, 0, 0, NULL, script};
    return r;
}
void orc_kat_cos_weighted(const double n[3], const double *script, double out[3]) {
    rng_t r = script_rng(script);
    pt nn = {n[0], n[1], n[2], 1}, d;
    cos_weighted_sample(&nn, &d, &r);
    out[0] = d.x; out[1] = d.y; out[2] = d.z;
}
void orc_kat_hemisphere_reorient(const double n[3], const double v[3], double out[3]) {
    pt nn = {n[0], n[1], n[2], 1}, vv = {v[0], v[1], v[2], 1};
    pt r = hemisphere_reorient(nn, vv);
    out[0] = r.x; out[1] = r.y; out[2] = r.z;
}
double orc_kat_random_normal(const double *script) {
    rng_t r = script_rng(script);
    return random_normal(&r);
}
void orc_kat_refract(const orc_scene *s, int obj_index, const double d_in[3], const double p[3], const double n_in[3],
                     double r_index[10], int *r_index_index, const double *script, double d_out[3], double n_out[3]) {
    rng_t r = script_rng(script);
    orc_stats st;
    memset(&st, 0, sizeof(st));
    path_t ps;
    memset(&ps, 0, sizeof(ps));
    ps.d.x = d_in[0]; ps.d.y = d_in[1]; ps.d.z = d_in[2]; ps.d.w = 1;
    ps.p0.w = 1;
    ps.mem[9] = 1;
    memcpy(&ps.mem[RBASE], r_index, sizeof(double) * 10);
    ps.idx = *r_index_index;
    pt pp = {p[0], p[1], p[2], 1}, nn = {n_in[0], n_in[1], n_in[2], 1};
    refract_ray(&ps, &pp, &nn, &s->d.objects[obj_index], &r, &st);
    memcpy(r_index, &ps.mem[RBASE], sizeof(double) * 10);
    *r_index_index = ps.idx;
    d_out[0] = ps.d.x; d_out[1] = ps.d.y; d_out[2] = ps.d.z;
    n_out[0] = nn.x; n_out[1] = nn.y; n_out[2] = nn.z;
}
void orc_kat_reflect(const double d_in[3], const double p[3], const double n_in[3], double d_out[3]) {
    path_t ps;
    memset(&ps, 0, sizeof(ps));
    ps.d.x = d_in[0]; ps.d.y = d_in[1]; ps.d.z = d_in[2]; ps.d.w = 1;
    pt pp = {p[0], p[1], p[2], 1}, nn = {n_in[0], n_in[1], n_in[2], 1};
    reflect_ray(&ps, &pp, &nn);
    d_out[0] = ps.d.x; d_out[1] = ps.d.y; d_out[2] = ps.d.z;
}
void orc_kat_texmap(const orc_scene *s, int obj_index, int which, double a, double b, double out[3]) {
    const ptx_object *o = &s->d.objects[obj_index];
    out[0] = out[1] = out[2] = 0;
    if (which == 0) tex_map(&s->d.images[o->tex_img], s->img[o->tex_img], a, b, &out[0], &out[1], &out[2]);
    else if (which == 1) tex_map(&s->d.images[o->normal_img], s->img[o->normal_img], a, b, &out[0], &out[1], &out[2]);
    else alpha_map(&s->d.images[o->alpha_img], s->img[o->alpha_img], a, b, &out[0]);
}
void orc_kat_normal_deform(const orc_scene *s, int obj_index, double a, double b, const double n_in[3], double n_out[3]) {
    pt nn = {n_in[0], n_in[1], n_in[2], 1};
    normal_deform(s, &s->d.objects[obj_index], a, b, &nn);
    n_out[0] = nn.x; n_out[1] = nn.y; n_out[2] = nn.z;
}
void orc_kat_primary_ray(const orc_scene *s, int sx, int i, int j, double aperture, const double *script, double p0[3],
                         double d_out[3]) {
    rng_t r = script_rng(script);
    const ptx_camera *cam = &s->d.cam;
    pt fp = {0, 0, cam->focus_distance, 1};
    matvec(cam->C2W, &fp);
    path_t ps;
    primary_ray(cam, sx, i, j, aperture, &fp, &r, &ps);
    p0[0] = ps.p0.x; p0[1] = ps.p0.y; p0[2] = ps.p0.z;
    d_out[0] = ps.d.x; d_out[1] = ps.d.y; d_out[2] = ps.d.z;
}
void orc_kat_light_sample(const orc_scene *s, int obj_index, const double *script, double out[3]) {
    rng_t r = script_rng(script);
    light_sample(&s->d.objects[obj_index], &r, &out[0], &out[1], &out[2]);
}
