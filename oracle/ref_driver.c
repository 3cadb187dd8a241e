/*
 * ref_driver.c -- TEST INFRASTRUCTURE.  Thin harness around the unmodified
 * reference sources, compiled where they lie under /root/reference by
 * oracle/Makefile into oracle/_ref/libptref_<flags>.so.  It exists only to
 * pin the CPU oracle (oracle/ptx_oracle.c) and to generate the golden
 * vectors under tests/golden/; nothing in the product links or loads it.
 *
 * What is reference code and what is harness:
 *   reference (compiled from /root/reference, not copied): findFirstHit,
 *     PathTrace, every intersect/sampling/texture routine, the scene kit,
 *     build_boxtree, the three shipped scenes;
 *   harness (this file): a replacement for drand48() (the build renames the
 *     reference's calls with -Ddrand48=ref_drand48), a walk over the
 *     reference's BoxTree that writes the flat scene file of
 *     include/ptx_scene.h, and ref_render(), which repeats the pixel loop of
 *     reference PathTracer.c:617-654 around the reference's own PathTrace so
 *     that the random stream can be chosen per pixel.
 *
 * Scene selection: PathTracer.c textually includes buildScene.c (BubbleTea);
 * the build renames that function to ref_scene_bubbletea.  The other two
 * shipped scenes are included below from /root/reference/Scenes.
 */
#include <omp.h>
#include "utils_path.h"
#include "magic.h"
#include "ptx_scene.h"
#include "ptx_scene_io.h"
#include "ptx_rng.h"

extern struct object3D *object_list;
extern struct object3D *light_list;
extern struct BoxTree *boxtree;
extern struct textureNode *texture_list;
extern struct view *cam;
extern int MAX_DEPTH;
void loadObj(const char *fname, double diff, double refl, double tran, double r, double g, double b,
             double refl_sig, double r_index, struct object3D **object_list, double T[4][4]);
double random_normal(void);
void ref_scene_bubbletea(void);

#define buildScene ref_scene_cornell
#include "Scenes/CornellBox.c"
#undef buildScene
#define buildScene ref_scene_scifi
#include "Scenes/SciFiCube.c"
#undef buildScene

/* ------------------------------------------------------------------ RNG */
enum { RNG_LCG = 0, RNG_COUNTER = 1, RNG_SCRIPT = 2 };
static int g_rng_mode = RNG_LCG;
static uint64_t g_lcg_state = 0x330E;
static uint64_t g_seed = 0;
static const double *g_script = NULL;
static int g_script_n = 0, g_script_pos = 0;
static __thread uint64_t t_key;
static __thread uint32_t t_draw;

double ref_drand48(void) {
    if (g_rng_mode == RNG_COUNTER) return ptx_counter_draw(t_key, t_draw++);
    if (g_rng_mode == RNG_SCRIPT) {
        double v = g_script_pos < g_script_n ? g_script[g_script_pos] : 0.5;
        g_script_pos++;
        return v;
    }
    return ptx_lcg48_draw(&g_lcg_state);
}

void ref_rng_lcg(uint64_t seed) { g_rng_mode = RNG_LCG; g_lcg_state = ptx_lcg48_seed(seed); }
void ref_rng_counter(uint64_t seed) { g_rng_mode = RNG_COUNTER; g_seed = seed; }
void ref_rng_script(const double *vals, int n) {
    g_rng_mode = RNG_SCRIPT; g_script = vals; g_script_n = n; g_script_pos = 0;
}
int ref_rng_script_used(void) { return g_script_pos; }

/* ---------------------------------------------------------- scene set-up */
static struct object3D **g_flat = NULL; /* flat index -> object */
static int g_nflat = 0;
static int g_ninserted = 0;

/* A scene made from a table, so that tests can exercise every primitive
 * and transform through the reference's own constructors. */
typedef struct ref_xform { int32_t op; int32_t pad; double a, b, c; } ref_xform;
/* op: 0 RotateX(a) 1 RotateY(a) 2 RotateZ(a) 3 Translate(a,b,c) 4 Scale(a,b,c) */
typedef struct ref_obj_spec {
    int32_t type, is_light, n_ops, pad;
    double diff, refl, tran, r, g, b, refl_sig, r_index;
    double verts[9];
    ref_xform ops[8];
    char tex[128], nrm[128], alpha[128];
} ref_obj_spec;
typedef struct ref_cam_spec { double e[3], g[3], up[3], f, wl, wt, wsize, focus; } ref_cam_spec;

static void reset_scene(void) {
    /* the reference never frees its BoxTree; test processes are short-lived */
    object_list = NULL;
    light_list = NULL;
    texture_list = NULL;
    boxtree = NULL;
    cam = NULL;
    free(g_flat);
    g_flat = NULL;
    g_nflat = 0;
}

/* Everything main() does between buildScene() and the sample loop
 * (reference PathTracer.c:566, :600-611), plus id bookkeeping in two
 * object3D fields the reference never reads (frontAndBack, isCSG). */
static void finish_scene(void) {
    int n = 0;
    struct object3D *o;
    for (o = object_list; o; o = o->next) n++;
    g_ninserted = n;
    /* insertObject (utils_path.c:125-138) keeps the first object at the head
     * and puts each later one right behind it: list = [0, n-1, n-2, ..., 1] */
    int pos = 0;
    for (o = object_list; o; o = o->next, pos++) o->frontAndBack = pos == 0 ? 0 : n - pos;
    boxtree = build_boxtree(object_list, 1, 0);
    double pct = 0;
    for (o = light_list; o; o = o->LSnext) { o->surfaceArea = o->LSweight; pct += o->LSweight; }
    for (o = light_list; o; o = o->LSnext) o->LSweight /= pct;
}

/* The flat scene comes from integration/ref_flatten.c, the file a maintainer of the reference would compile into
 * it: every comparison this harness makes with the oracle therefore also tests that file. */
#include "../integration/ref_flatten.c"

static ptx_scene_desc g_scene; /* of the scene built last */

static void flatten_tree(void) {
    ptx_ref_flatten_free(&g_scene);
    free(g_flat);
    g_flat = NULL;
    g_nflat = 0;
    if (ptx_ref_flatten(boxtree, light_list, cam, &g_scene, &g_flat) == 0) g_nflat = g_scene.n_objects;
}

int ref_select_scene(const char *name, int max_depth) {
    reset_scene();
    MAX_DEPTH = max_depth;
    if (!strcmp(name, "cornell")) ref_scene_cornell();
    else if (!strcmp(name, "scifi")) ref_scene_scifi();
    else if (!strcmp(name, "bubbletea")) ref_scene_bubbletea();
    else return -1;
    finish_scene();
    flatten_tree();
    return g_nflat;
}

int ref_build_custom(const ref_obj_spec *specs, int n, const ref_cam_spec *c, int max_depth) {
    reset_scene();
    MAX_DEPTH = max_depth;
    struct point3D e = {c->e[0], c->e[1], c->e[2], 1}, g = {c->g[0], c->g[1], c->g[2], 1},
                   up = {c->up[0], c->up[1], c->up[2], 1};
    cam = setupView(&e, &g, &up, c->f, c->wl, c->wt, c->wsize, c->focus);
    for (int i = 0; i < n; i++) {
        const ref_obj_spec *s = &specs[i];
        struct object3D *o = NULL;
        struct point3D v1 = {s->verts[0], s->verts[1], s->verts[2], 1}, v2 = {s->verts[3], s->verts[4], s->verts[5], 1},
                       v3 = {s->verts[6], s->verts[7], s->verts[8], 1};
        switch (s->type) {
        case PTX_PLANE: o = newPlane(s->diff, s->refl, s->tran, s->r, s->g, s->b, s->refl_sig, s->r_index); break;
        case PTX_SPHERE: o = newSphere(s->diff, s->refl, s->tran, s->r, s->g, s->b, s->refl_sig, s->r_index); break;
        case PTX_CYL: o = newCyl(s->diff, s->refl, s->tran, s->r, s->g, s->b, s->refl_sig, s->r_index); break;
        case PTX_TRIANGLE: o = newTriangle(v1, v2, v3, s->diff, s->refl, s->tran, s->r, s->g, s->b, s->refl_sig, s->r_index); break;
        case PTX_CUBE: o = newCube(s->diff, s->refl, s->tran, s->r, s->g, s->b, s->refl_sig, s->r_index); break;
        case PTX_ROOM: o = newRoom(s->diff, s->refl, s->tran, s->r, s->g, s->b, s->refl_sig, s->r_index); break;
        default: return -1;
        }
        for (int k = 0; k < s->n_ops; k++) {
            const ref_xform *x = &s->ops[k];
            switch (x->op) {
            case 0: RotateX(o, x->a); break;
            case 1: RotateY(o, x->a); break;
            case 2: RotateZ(o, x->a); break;
            case 3: Translate(o, x->a, x->b, x->c); break;
            case 4: Scale(o, x->a, x->b, x->c); break;
            default: return -2;
            }
        }
        if (s->tex[0]) loadTexture(o, s->tex, 1, &texture_list);
        if (s->nrm[0]) loadTexture(o, s->nrm, 2, &texture_list);
        if (s->alpha[0]) loadTexture(o, s->alpha, 3, &texture_list);
        invert(&o->T[0][0], &o->Tinv[0][0]);
        o->isLightSource = s->is_light;
        insertObject(o, &object_list);
    }
    finish_scene();
    flatten_tree();
    return g_nflat;
}

int ref_num_inserted(void) { return g_ninserted; }

/* ------------------------------------------------------------ scene dump */
int ref_dump_scene(const char *path) {
    if (g_nflat == 0) return -1;
    return ptx_scene_write(path, &g_scene);
}

/* ------------------------------------------------------------- first hit */
void ref_first_hit(const ptx_ray *rays, int n, ptx_hit *hits) {
    for (int i = 0; i < n; i++) {
        struct ray3D ray;
        struct point3D p0 = {rays[i].p0[0], rays[i].p0[1], rays[i].p0[2], 1};
        struct point3D d = {rays[i].d[0], rays[i].d[1], rays[i].d[2], 1};
        initRay(&ray, &p0, &d);
        double lambda = -1, a = 0, b = 0;
        struct object3D *obj = NULL;
        struct point3D p = {0, 0, 0, 1}, nn = {0, 0, 0, 1};
        struct object3D *Os = rays[i].os >= 0 && rays[i].os < g_nflat ? g_flat[rays[i].os] : NULL;
        findFirstHit(boxtree, &ray, &lambda, Os, &obj, &p, &nn, &a, &b);
        ptx_hit *h = &hits[i];
        memset(h, 0, sizeof(*h));
        h->lambda = lambda;
        h->obj = -1;
        if (lambda > 0 && obj) {
            h->obj = obj->isCSG;
            h->p[0] = p.px; h->p[1] = p.py; h->p[2] = p.pz;
            h->n[0] = nn.px; h->n[1] = nn.py; h->n[2] = nn.pz;
            h->a = a; h->b = b;
        }
    }
}

/* ---------------------------------------------------------------- render */
/* The pixel loop of reference PathTracer.c:617-654 around the reference's
 * own sampleLensPoint / focusRay / PathTrace.  rgb (sx*sx*3) and wght
 * (sx*sx) are read and updated, so passes can be continued.  In counter
 * mode rows run in parallel (each path's stream depends only on its pixel
 * and sample); in LCG mode the loop is serial like OMP_NUM_THREADS=1.
 * ray_count, if not NULL, receives nothing: the reference has no exact ray
 * counter. */
void ref_render(int sx, int k0, int nk, double aperture, double *rgbIm, double *wght, int threads) {
    double du = cam->wsize / (sx - 1);
    double dv = -cam->wsize / (sx - 1);
    struct point3D fp = {0, 0, cam->focus_distance, 1};
    matVecMult(cam->C2W, &fp);
    int par = g_rng_mode == RNG_COUNTER && threads != 1;
    if (threads <= 0) threads = omp_get_max_threads();
    for (int k = k0; k < k0 + nk; k++) {
#pragma omp parallel for schedule(dynamic, 1) if (par) num_threads(threads)
        for (int j = 0; j < sx; j++) {
            for (int i = 0; i < sx; i++) {
                struct point3D pc, d, p_lens;
                struct ray3D ray;
                struct colourRGB col;
                t_key = ptx_path_key(g_seed, (uint64_t)j * sx + i, (uint64_t)k);
                t_draw = 0;
                pc.px = cam->wl + (i * du);
                pc.py = cam->wt + (j * dv);
                pc.pz = cam->f;
                pc.pw = 1;
                sampleLensPoint(aperture, &pc, &p_lens);
                matVecMult(cam->C2W, &pc);
                memcpy(&d, &pc, sizeof(struct point3D));
                focusRay(&p_lens, &pc, &fp, &d);
                initRay(&ray, &p_lens, &d);
                double wt = *(wght + i + (j * sx));
                PathTrace(&ray, 1, &col, NULL, 0);
                (*(rgbIm + ((i + (j * sx)) * 3) + 0)) += col.R * pow(2, -log(wt));
                (*(rgbIm + ((i + (j * sx)) * 3) + 1)) += col.G * pow(2, -log(wt));
                (*(rgbIm + ((i + (j * sx)) * 3) + 2)) += col.B * pow(2, -log(wt));
                wt += col.R;
                wt += col.G;
                wt += col.B;
                *(wght + i + (j * sx)) = wt;
            }
        }
    }
}

/* Radiance of single paths, for path-level comparison: out[3*n]. */
void ref_trace_pixels(int sx, const int32_t *pix_i, const int32_t *pix_j, const int32_t *samp, int n, double aperture,
                      double *out) {
    double du = cam->wsize / (sx - 1);
    double dv = -cam->wsize / (sx - 1);
    struct point3D fp = {0, 0, cam->focus_distance, 1};
    matVecMult(cam->C2W, &fp);
    for (int q = 0; q < n; q++) {
        struct point3D pc, d, p_lens;
        struct ray3D ray;
        struct colourRGB col;
        int i = pix_i[q], j = pix_j[q];
        t_key = ptx_path_key(g_seed, (uint64_t)j * sx + i, (uint64_t)samp[q]);
        t_draw = 0;
        pc.px = cam->wl + (i * du);
        pc.py = cam->wt + (j * dv);
        pc.pz = cam->f;
        pc.pw = 1;
        sampleLensPoint(aperture, &pc, &p_lens);
        matVecMult(cam->C2W, &pc);
        memcpy(&d, &pc, sizeof(struct point3D));
        focusRay(&p_lens, &pc, &fp, &d);
        initRay(&ray, &p_lens, &d);
        PathTrace(&ray, 1, &col, NULL, 0);
        out[3 * q] = col.R; out[3 * q + 1] = col.G; out[3 * q + 2] = col.B;
    }
}

/* --------------------------------------------- known-answer entry points */
void ref_kat_cos_weighted(const double n[3], double out[3]) {
    struct point3D nn = {n[0], n[1], n[2], 1}, d;
    cosWeightedSample(&nn, &d);
    out[0] = d.px; out[1] = d.py; out[2] = d.pz;
}
void ref_kat_hemisphere_reorient(const double n[3], const double v[3], double out[3]) {
    struct point3D nn = {n[0], n[1], n[2], 1}, vv = {v[0], v[1], v[2], 1};
    struct point3D r = hemisphereReorient(nn, vv);
    out[0] = r.px; out[1] = r.py; out[2] = r.pz;
}
double ref_kat_random_normal(void) { return random_normal(); }
/* state: r_index[10], r_index_index in/out; d in/out; returns via out_d, out_p0 */
void ref_kat_refract(int obj_index, const double d_in[3], const double p[3], const double n_in[3], double r_index[10],
                     int *r_index_index, double d_out[3], double n_out[3]) {
    struct ray3D ray;
    struct point3D p0 = {0, 0, 0, 1}, d = {d_in[0], d_in[1], d_in[2], 1};
    initRay(&ray, &p0, &d);
    memcpy(ray.r_index, r_index, sizeof(double) * 10);
    ray.r_index_index = *r_index_index;
    struct point3D pp = {p[0], p[1], p[2], 1}, nn = {n_in[0], n_in[1], n_in[2], 1};
    refractRay(&ray, &pp, &nn, g_flat[obj_index]);
    memcpy(r_index, ray.r_index, sizeof(double) * 10);
    *r_index_index = ray.r_index_index;
    d_out[0] = ray.d.px; d_out[1] = ray.d.py; d_out[2] = ray.d.pz;
    n_out[0] = nn.px; n_out[1] = nn.py; n_out[2] = nn.pz;
}
void ref_kat_reflect(const double d_in[3], const double p[3], const double n_in[3], double d_out[3]) {
    struct ray3D ray;
    struct point3D p0 = {0, 0, 0, 1}, d = {d_in[0], d_in[1], d_in[2], 1};
    initRay(&ray, &p0, &d);
    struct point3D pp = {p[0], p[1], p[2], 1}, nn = {n_in[0], n_in[1], n_in[2], 1};
    reflectRay(&ray, &pp, &nn);
    d_out[0] = ray.d.px; d_out[1] = ray.d.py; d_out[2] = ray.d.pz;
}
void ref_kat_texmap(int obj_index, int which, double a, double b, double out[3]) {
    struct object3D *o = g_flat[obj_index];
    out[0] = out[1] = out[2] = 0;
    if (which == 0) texMap(o->texImg, a, b, &out[0], &out[1], &out[2]);
    else if (which == 1) texMap(o->normalImg, a, b, &out[0], &out[1], &out[2]);
    else alphaMap(o->alphaImg, a, b, &out[0]);
}
void ref_kat_normal_deform(int obj_index, double a, double b, const double n_in[3], double n_out[3]) {
    struct point3D nn = {n_in[0], n_in[1], n_in[2], 1};
    normalDeform(g_flat[obj_index], a, b, &nn);
    n_out[0] = nn.px; n_out[1] = nn.py; n_out[2] = nn.pz;
}
void ref_kat_primary_ray(int sx, int i, int j, double aperture, double p0[3], double d_out[3]) {
    double du = cam->wsize / (sx - 1);
    double dv = -cam->wsize / (sx - 1);
    struct point3D fp = {0, 0, cam->focus_distance, 1};
    matVecMult(cam->C2W, &fp);
    struct point3D pc = {cam->wl + (i * du), cam->wt + (j * dv), cam->f, 1}, d, p_lens;
    sampleLensPoint(aperture, &pc, &p_lens);
    matVecMult(cam->C2W, &pc);
    memcpy(&d, &pc, sizeof(struct point3D));
    focusRay(&p_lens, &pc, &fp, &d);
    p0[0] = p_lens.px; p0[1] = p_lens.py; p0[2] = p_lens.pz;
    d_out[0] = d.px; d_out[1] = d.py; d_out[2] = d.pz;
}
void ref_kat_light_sample(int obj_index, double out[3]) {
    struct object3D *o = g_flat[obj_index];
    o->randomPoint(o, &out[0], &out[1], &out[2]);
}
