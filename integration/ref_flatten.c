/*
 * ref_flatten.c -- the flat scene of include/ptx_scene.h filled from the reference's OWN structures.
 *
 * For a maintainer who keeps PathTracer's scene code and main() and only swaps the sample loop for the GPU library:
 * compile this file inside the reference tree (it includes the reference's utils_path.h and magic.h, nothing of
 * this repository but include/ptx_scene.h) and call ptx_ref_flatten() after buildScene(), build_boxtree() and the
 * light-weight normalisation (reference PathTracer.c:563-566, :600-611).
 *
 * What it does: walks the BoxTree in the order findFirstHit visits it (reference PathTracer.c:198-251: a node's own
 * objects, then child1, then child2), so that an object's position in the flat array is its tie-break rank and
 * objects build_boxtree dropped (magic.c:224-231) are absent; records the nodes; lists the lights in LSnext order;
 * turns the texture images back into the bytes they were read from (value = byte / 255, utils_path.c:1849, :1921).
 * The primitive type is recognised by the object's intersect function pointer (PathTracer.h:120-132).
 *
 * object3D.isCSG, a field the reference declares and never reads, receives the object's flat index (the light list
 * needs it); object3D.frontAndBack, equally unread, is copied to ptx_object.src_id for callers that number their
 * objects there.
 */
#include <stdlib.h>
#include <string.h>

#include "utils_path.h"
#include "magic.h"
#include "ptx_scene.h"

static int ptxr_count_nodes(struct BoxTree *bt) { return bt ? 1 + ptxr_count_nodes(bt->child1) + ptxr_count_nodes(bt->child2) : 0; }
static int ptxr_count_objs(struct BoxTree *bt) {
    if (!bt) return 0;
    int n = 0;
    for (struct object3D *o = bt->objs; o; o = o->next) n++;
    return n + ptxr_count_objs(bt->child1) + ptxr_count_objs(bt->child2);
}

static int ptxr_walk(struct BoxTree *bt, ptx_node *nodes, int *n_nodes, struct object3D **flat, int *n_flat) {
    if (!bt) return -1;
    int idx = (*n_nodes)++;
    memcpy(nodes[idx].box, &bt->box, sizeof(double) * 6);
    nodes[idx].first_obj = *n_flat;
    for (struct object3D *o = bt->objs; o; o = o->next) {
        o->isCSG = *n_flat;
        flat[(*n_flat)++] = o;
    }
    nodes[idx].n_objs = *n_flat - nodes[idx].first_obj;
    int c1 = ptxr_walk(bt->child1, nodes, n_nodes, flat, n_flat);
    nodes[idx].child1 = c1;
    int c2 = ptxr_walk(bt->child2, nodes, n_nodes, flat, n_flat);
    nodes[idx].child2 = c2;
    return idx;
}

static int ptxr_type(struct object3D *o) {
    if (o->intersect == &planeIntersect) return PTX_PLANE;
    if (o->intersect == &sphereIntersect) return PTX_SPHERE;
    if (o->intersect == &cylIntersect) return PTX_CYL;
    if (o->intersect == &triangleIntersect) return PTX_TRIANGLE;
    if (o->intersect == &cubeIntersect) return PTX_CUBE;
    return PTX_ROOM;
}

/* Releases what ptx_ref_flatten allocated. */
void ptx_ref_flatten_free(ptx_scene_desc *s) {
    free((void *)s->objects);
    free((void *)s->nodes);
    free((void *)s->lights);
    free((void *)s->images);
    free((void *)s->texels);
    memset(s, 0, sizeof(*s));
}

/* Fills *out from the reference's scene.  flat_out, if not NULL, receives a malloc'ed array mapping flat index to
 * object (n_objects entries).  Returns 0, or a negative number: -1 no objects, -2 out of memory, -5 more than 256
 * distinct images, -6 an image whose values are not multiples of 1/255. */
int ptx_ref_flatten(struct BoxTree *boxtree, struct object3D *light_list, struct view *cam, ptx_scene_desc *out,
                    struct object3D ***flat_out) {
    memset(out, 0, sizeof(*out));
    const int nn = ptxr_count_nodes(boxtree), no = ptxr_count_objs(boxtree);
    if (no == 0 || !cam) return -1;
    ptx_node *nodes = (ptx_node *)calloc((size_t)nn + 1, sizeof(ptx_node));
    struct object3D **flat = (struct object3D **)calloc((size_t)no + 1, sizeof(*flat));
    ptx_object *objs = (ptx_object *)calloc((size_t)no + 1, sizeof(ptx_object));
    if (!nodes || !flat || !objs) return -2;
    int n_nodes = 0, n_flat = 0;
    ptxr_walk(boxtree, nodes, &n_nodes, flat, &n_flat);

    struct image *imgs[256];
    int chans[256];
    int n_img = 0;
    for (int i = 0; i < n_flat; i++) {
        struct object3D *o = flat[i];
        ptx_object *d = &objs[i];
        d->type = ptxr_type(o);
        d->is_light = o->isLightSource;
        d->src_id = o->frontAndBack;
        struct image *three[3] = {o->texImg, o->normalImg, o->alphaImg};
        int32_t *slots[3] = {&d->tex_img, &d->normal_img, &d->alpha_img};
        for (int k = 0; k < 3; k++) {
            *slots[k] = -1;
            if (!three[k]) continue;
            int j;
            for (j = 0; j < n_img; j++)
                if (imgs[j] == three[k]) break;
            if (j == n_img) {
                if (n_img >= 256) return -5;
                imgs[n_img] = three[k];
                chans[n_img] = k == 2 ? 1 : 3;
                n_img++;
            }
            *slots[k] = j;
        }
        d->col[0] = o->col.R; d->col[1] = o->col.G; d->col[2] = o->col.B;
        memcpy(d->v[0], &o->v1, 32); memcpy(d->v[1], &o->v2, 32); memcpy(d->v[2], &o->v3, 32);
        memcpy(d->T, o->T, 128); memcpy(d->Tinv, o->Tinv, 128);
        memcpy(d->box, &o->box, 48);
        d->diffPct = o->diffPct; d->reflPct = o->reflPct; d->tranPct = o->tranPct;
        d->refl_sig = o->refl_sig; d->r_index = o->r_index;
        d->surfaceArea = o->surfaceArea; d->LSweight = o->LSweight;
    }
    int nl = 0;
    for (struct object3D *o = light_list; o; o = o->LSnext) nl++;
    int32_t *lights = (int32_t *)calloc((size_t)nl + 1, 4);
    nl = 0;
    for (struct object3D *o = light_list; o; o = o->LSnext) {
        /* a light that build_boxtree dropped is not in the flat array */
        int idx = (o->isCSG >= 0 && o->isCSG < n_flat && flat[o->isCSG] == o) ? o->isCSG : -1;
        lights[nl++] = idx;
    }
    ptx_image *images = (ptx_image *)calloc((size_t)n_img + 1, sizeof(ptx_image));
    uint64_t total = 0;
    for (int j = 0; j < n_img; j++) total += (uint64_t)imgs[j]->sx * imgs[j]->sy * chans[j];
    uint8_t *texels = (uint8_t *)malloc(total + 1);
    if (!lights || !images || !texels) return -2;
    uint64_t off = 0;
    for (int j = 0; j < n_img; j++) {
        uint64_t cnt = (uint64_t)imgs[j]->sx * imgs[j]->sy * chans[j];
        const double *src = (const double *)imgs[j]->rgbdata;
        for (uint64_t k = 0; k < cnt; k++) texels[off + k] = (uint8_t)(src[k] * 255.0 + 0.5);
        for (uint64_t k = 0; k < cnt; k++)
            if ((double)texels[off + k] / 255.0 != src[k]) return -6;
        images[j].sx = imgs[j]->sx; images[j].sy = imgs[j]->sy; images[j].channels = chans[j];
        images[j].divisor = 255.0; images[j].data_offset = off;
        off += cnt;
    }
    out->magic = PTX_SCENE_MAGIC;
    out->n_objects = n_flat; out->n_nodes = n_nodes; out->n_lights = nl; out->n_images = n_img; out->n_texel_bytes = total;
    out->objects = objs; out->nodes = nodes; out->lights = lights; out->images = images; out->texels = texels;
    memcpy(out->cam.e, &cam->e, 32); memcpy(out->cam.u, &cam->u, 32); memcpy(out->cam.v, &cam->v, 32); memcpy(out->cam.w, &cam->w, 32);
    out->cam.f = cam->f; out->cam.focus_distance = cam->focus_distance; out->cam.wl = cam->wl; out->cam.wt = cam->wt;
    out->cam.wsize = cam->wsize;
    memcpy(out->cam.C2W, cam->C2W, 128); memcpy(out->cam.W2C, cam->W2C, 128);
    if (flat_out) *flat_out = flat;
    else free(flat);
    return 0;
}
