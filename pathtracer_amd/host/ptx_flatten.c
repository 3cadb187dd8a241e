/*
 * ptx_flatten.c -- from the kit's linked scene to the flat description of
 * include/ptx_scene.h.
 *
 * What the reference's findFirstHit can reach, and in which order it visits
 * it, is decided by build_boxtree (magic.c:142-247): a binary tree of boxes
 * with objects kept in inner nodes, built by a greedy volume-cost pivot
 * search.  That builder also loses objects: when a node's partition is
 * collapsed back into one list (magic.c:224-231) only the first object of
 * each side list is kept (append(), magic.c:367-379).  The GPU must trace
 * against exactly the surviving set, in the reference's visiting order (ties
 * between equal lambdas go to the object visited first), so the builder is
 * restated here, on index arrays instead of re-linked list nodes, and the
 * tree is written out in visiting order.
 */
#include "ptx_kit.h"

typedef struct {
    int *v;
    int n, cap;
} ilist;

static void il_push(ilist *l, int x) {
    if (l->n == l->cap) {
        l->cap = l->cap ? 2 * l->cap : 8;
        l->v = (int *)realloc(l->v, sizeof(int) * (size_t)l->cap);
    }
    l->v[l->n++] = x;
}
/* append(list, elem) of the reference: an empty list becomes [elem]; otherwise elem goes right behind the head */
static void il_append(ilist *l, int x) {
    if (l->n == 0) {
        il_push(l, x);
        return;
    }
    il_push(l, x);
    memmove(&l->v[2], &l->v[1], sizeof(int) * (size_t)(l->n - 2));
    l->v[1] = x;
}

static struct object3D **g_obj; /* objects in object_list order */

static const struct Box *obox(int i) { return &g_obj[i]->box; }
static double lo_of(const struct Box *b, int dim) { return dim == 0 ? b->xl : (dim == 1 ? b->yl : b->zl); }
static double hi_of(const struct Box *b, int dim) { return dim == 0 ? b->xu : (dim == 1 ? b->yu : b->zu); }
#define KMIN(A, B) ((A) > (B) ? (B) : (A))
#define KMAX(A, B) ((A) < (B) ? (B) : (A))

static struct Box box_union(struct Box a, struct Box b) {
    struct Box r;
    r.xl = KMIN(a.xl, b.xl); r.xu = KMAX(a.xu, b.xu);
    r.yl = KMIN(a.yl, b.yl); r.yu = KMAX(a.yu, b.yu);
    r.zl = KMIN(a.zl, b.zl); r.zu = KMAX(a.zu, b.zu);
    return r;
}
static struct Box list_box(const ilist *l) { /* boxproduct, magic.c:98-116 */
    struct Box b = {INFINITY, -INFINITY, INFINITY, -INFINITY, INFINITY, -INFINITY};
    for (int k = 0; k < l->n; k++) {
        const struct Box *o = obox(l->v[k]);
        b.xl = KMIN(b.xl, o->xl); b.xu = KMAX(b.xu, o->xu);
        b.yl = KMIN(b.yl, o->yl); b.yu = KMAX(b.yu, o->yu);
        b.zl = KMIN(b.zl, o->zl); b.zu = KMAX(b.zu, o->zu);
    }
    return b;
}
static double box_volume(struct Box b) { return (b.xu - b.xl) * (b.yu - b.yl) * (b.zu - b.zl); }

/* Cost of splitting the list at `pivot` along `dim` (pivotquality, magic.c:250-304): objects entirely at or above
 * the pivot, entirely at or below it, and the rest; each group is charged its count times the volume of its box,
 * the rest being charged the volume of everything. */
static double pivot_cost(double pivot, int dim, const ilist *l) {
    if (l->n == 0) return INFINITY;
    struct Box lower = {0}, upper = {0}, rest = {0};
    int lc = 0, uc = 0, rc = 0;
    for (int k = 0; k < l->n; k++) {
        const struct Box *b = obox(l->v[k]);
        if (pivot <= lo_of(b, dim)) {
            if (uc == 0) upper = *b;
            upper = box_union(*b, upper);
            uc++;
        } else if (pivot >= hi_of(b, dim)) {
            if (lc == 0) lower = *b;
            lower = box_union(*b, lower);
            lc++;
        } else {
            if (rc == 0) rest = *b;
            rest = box_union(*b, rest);
            rc++;
        }
    }
    double cost = 0;
    if (lc != 0) {
        if (rc == 0) rest = lower;
        rest = box_union(lower, rest);
        cost += lc * box_volume(lower);
    }
    if (uc != 0) {
        if (rc == 0 && lc == 0) rest = upper;
        rest = box_union(upper, rest);
        cost += uc * box_volume(upper);
    }
    cost += rc * box_volume(rest);
    return cost;
}
/* every object bound is a candidate, upper bound before lower bound, first best wins (bestpivot, magic.c:307-332) */
static double exhaustive_pivot(const ilist *l, int dim) {
    double best = 0, best_cost = INFINITY;
    for (int k = 0; k < l->n; k++) {
        const struct Box *b = obox(l->v[k]);
        double c = pivot_cost(hi_of(b, dim), dim, l);
        if (c < best_cost) { best = hi_of(b, dim); best_cost = c; }
        c = pivot_cost(lo_of(b, dim), dim, l);
        if (c < best_cost) { best = lo_of(b, dim); best_cost = c; }
    }
    return best;
}
/* 20 evenly spaced candidates (okpivot, magic.c:335-364).  For the y axis the reference takes the upper end of
 * the range from the z extent (magic.c:344); that is kept, since it changes the tree. */
static double sampled_pivot(const ilist *l, int dim) {
    if (l->n == 0) return 0;
    struct Box b = list_box(l);
    double lo = b.xl, hi = b.xu;
    if (dim == 1) { lo = b.yl; hi = b.zu; }
    else if (dim == 2) { lo = b.zl; hi = b.zu; }
    const int NUM = 20;
    double best = 0, best_cost = INFINITY, step = (hi - lo) / (NUM + 2);
    for (int i = 0; i < NUM; i++) {
        double pivot = lo + (step * (i + 1));
        double c = pivot_cost(pivot, dim, l);
        if (c < best_cost) { best = pivot; best_cost = c; }
    }
    return best;
}

/* growing output arrays */
static ptx_node *g_nodes;
static int g_nnodes, g_nodecap;
static int *g_order; /* visiting order -> index into g_obj */
static int g_norder;

static int build(ilist *l) { /* build_boxtree, magic.c:142-247; consumes l */
    if (l->n == 0) {
        free(l->v);
        return -1;
    }
    struct Box b = list_box(l);
    double piv[3] = {(b.xu + b.xl) / 2, (b.yu + b.yl) / 2, (b.zu + b.zl) / 2};
    if (l->n < 1000) for (int d = 0; d < 3; d++) piv[d] = exhaustive_pivot(l, d);
    else if (l->n < 100000) for (int d = 0; d < 3; d++) piv[d] = sampled_pivot(l, d);
    double c0 = pivot_cost(piv[0], 0, l), c1 = pivot_cost(piv[1], 1, l), c2 = pivot_cost(piv[2], 2, l);
    int dim = 0;
    double pivot = piv[0], c = c0;
    if (c1 < c0) { pivot = piv[1]; c = c1; dim = 1; }
    if (c2 < c) { pivot = piv[2]; dim = 2; }
    ilist lower = {0}, upper = {0}, middle = {0};
    for (int k = 0; k < l->n; k++) {
        const struct Box *ob = obox(l->v[k]);
        if (pivot <= lo_of(ob, dim)) il_append(&upper, l->v[k]);
        else if (pivot >= hi_of(ob, dim)) il_append(&lower, l->v[k]);
        else il_append(&middle, l->v[k]);
    }
    if (middle.n + lower.n == 0 || middle.n + upper.n == 0 || middle.n + lower.n + upper.n <= 2) {
        /* no useful split: everything is meant to stay in this node, but only the first object of each side
         * list survives the reference's append() */
        if (lower.n) il_append(&middle, lower.v[0]);
        if (upper.n) il_append(&middle, upper.v[0]);
        lower.n = 0;
        upper.n = 0;
    }
    free(l->v);
    if (g_nnodes == g_nodecap) {
        g_nodecap = g_nodecap ? 2 * g_nodecap : 64;
        g_nodes = (ptx_node *)realloc(g_nodes, sizeof(ptx_node) * (size_t)g_nodecap);
    }
    int me = g_nnodes++;
    memcpy(g_nodes[me].box, &b, sizeof(double) * 6);
    g_nodes[me].first_obj = g_norder;
    g_nodes[me].n_objs = middle.n;
    for (int k = 0; k < middle.n; k++) g_order[g_norder++] = middle.v[k];
    free(middle.v);
    int ch1 = build(&lower);
    int ch2 = build(&upper);
    g_nodes[me].child1 = ch1;
    g_nodes[me].child2 = ch2;
    return me;
}

/* storage handed out through the scene description */
static ptx_object *g_flat_objs;
static int32_t *g_flat_lights;
static ptx_image *g_flat_images;
static uint8_t *g_flat_texels;

static void free_flat(void) {
    free(g_nodes); g_nodes = NULL; g_nnodes = g_nodecap = 0;
    free(g_order); g_order = NULL; g_norder = 0;
    free(g_obj); g_obj = NULL;
    free(g_flat_objs); g_flat_objs = NULL;
    free(g_flat_lights); g_flat_lights = NULL;
    free(g_flat_images); g_flat_images = NULL;
    free(g_flat_texels); g_flat_texels = NULL;
}

void ptxh_reset(void) {
    free_flat();
    cleanup(object_list, texture_list);
    free(cam);
    object_list = NULL;
    light_list = NULL;
    texture_list = NULL;
    cam = NULL;
    extern void ptxh_reset_counter(void);
    ptxh_reset_counter();
}

static int image_index(struct textureNode **tex, int ntex, struct image *im) {
    if (!im) return -1;
    for (int k = 0; k < ntex; k++)
        if (tex[k]->im == im) return k;
    return -1;
}

int ptxh_flatten(ptx_scene_desc *out) {
    if (!cam || !object_list || !out) return -1;
    free_flat();
    int n = 0;
    for (struct object3D *o = object_list; o; o = o->next) n++;
    g_obj = (struct object3D **)calloc((size_t)n, sizeof(*g_obj));
    g_order = (int *)calloc((size_t)n, sizeof(int));
    ilist all = {0};
    n = 0;
    for (struct object3D *o = object_list; o; o = o->next) {
        g_obj[n] = o;
        il_push(&all, n);
        n++;
    }
    const char *flat = getenv("PTX_FLAT_ORDER");
    if (flat && *flat && strcmp(flat, "0") != 0) {
        /* No BoxTree: every object, in object_list order, under one node.  For scenes far beyond what the
         * reference's builder was made for (it samples pivots above 1000 objects and above 100000 splits at the
         * middle, magic.c:154-171); nothing is lost and equal-lambda ties go to the earlier list entry. */
        struct Box b = list_box(&all);
        g_nodes = (ptx_node *)calloc(1, sizeof(ptx_node));
        g_nnodes = g_nodecap = 1;
        memcpy(g_nodes[0].box, &b, sizeof(double) * 6);
        g_nodes[0].child1 = g_nodes[0].child2 = -1;
        g_nodes[0].first_obj = 0;
        g_nodes[0].n_objs = n;
        for (int k = 0; k < n; k++) g_order[k] = k;
        g_norder = n;
        free(all.v);
    } else {
        build(&all);
    }

    /* light weights (PathTracer.c:600-611): surfaceArea is the raw weight, LSweight its share of the total.  The
     * kit's objects are left as they are, so flattening twice gives the same result. */
    double total = 0;
    for (struct object3D *o = light_list; o; o = o->LSnext) total += o->LSweight;

    int ntex = 0;
    for (struct textureNode *t = texture_list; t; t = t->next) ntex++;
    struct textureNode **tex = (struct textureNode **)calloc((size_t)ntex + 1, sizeof(*tex));
    ntex = 0;
    uint64_t nbytes = 0;
    for (struct textureNode *t = texture_list; t; t = t->next) {
        tex[ntex++] = t;
        nbytes += (uint64_t)t->im->sx * t->im->sy * (t->type == 3 ? 1 : 3);
    }
    g_flat_images = (ptx_image *)calloc((size_t)ntex + 1, sizeof(ptx_image));
    g_flat_texels = (uint8_t *)malloc(nbytes + 1);
    uint64_t off = 0;
    for (int k = 0; k < ntex; k++) {
        int ch = tex[k]->type == 3 ? 1 : 3;
        uint64_t cnt = (uint64_t)tex[k]->im->sx * tex[k]->im->sy * ch;
        memcpy(g_flat_texels + off, tex[k]->bytes, cnt);
        g_flat_images[k].sx = tex[k]->im->sx;
        g_flat_images[k].sy = tex[k]->im->sy;
        g_flat_images[k].channels = ch;
        g_flat_images[k].divisor = tex[k]->divisor;
        g_flat_images[k].data_offset = off;
        off += cnt;
    }

    int *flat_of = (int *)malloc(sizeof(int) * (size_t)n);
    for (int k = 0; k < n; k++) flat_of[k] = -1;
    g_flat_objs = (ptx_object *)calloc((size_t)g_norder + 1, sizeof(ptx_object));
    for (int k = 0; k < g_norder; k++) {
        const struct object3D *o = g_obj[g_order[k]];
        flat_of[g_order[k]] = k;
        ptx_object *d = &g_flat_objs[k];
        d->type = o->kind;
        d->is_light = o->isLightSource;
        d->tex_img = image_index(tex, ntex, o->texImg);
        d->normal_img = image_index(tex, ntex, o->normalImg);
        d->alpha_img = image_index(tex, ntex, o->alphaImg);
        d->src_id = o->insert_id;
        d->col[0] = o->col.R; d->col[1] = o->col.G; d->col[2] = o->col.B;
        memcpy(d->v[0], &o->v1, 32); memcpy(d->v[1], &o->v2, 32); memcpy(d->v[2], &o->v3, 32);
        memcpy(d->T, o->T, 128);
        memcpy(d->Tinv, o->Tinv, 128);
        memcpy(d->box, &o->box, 48);
        d->diffPct = o->diffPct; d->reflPct = o->reflPct; d->tranPct = o->tranPct;
        d->refl_sig = o->refl_sig; d->r_index = o->r_index;
        if (o->isLightSource) {
            d->surfaceArea = o->LSweight;
            d->LSweight = o->LSweight / total;
        } else {
            d->surfaceArea = 0;
            d->LSweight = o->LSweight;
        }
    }
    int nl = 0;
    for (struct object3D *o = light_list; o; o = o->LSnext) nl++;
    g_flat_lights = (int32_t *)calloc((size_t)nl + 1, sizeof(int32_t));
    nl = 0;
    for (struct object3D *o = light_list; o; o = o->LSnext) {
        int idx = -1;
        for (int k = 0; k < n; k++)
            if (g_obj[k] == o) { idx = flat_of[k]; break; }
        g_flat_lights[nl++] = idx;
    }
    free(flat_of);
    free(tex);

    memset(out, 0, sizeof(*out));
    out->magic = PTX_SCENE_MAGIC;
    out->n_objects = g_norder;
    out->n_nodes = g_nnodes;
    out->n_lights = nl;
    out->n_images = ntex;
    out->n_texel_bytes = nbytes;
    out->objects = g_flat_objs;
    out->nodes = g_nodes;
    out->lights = g_flat_lights;
    out->images = g_flat_images;
    out->texels = g_flat_texels;
    memcpy(out->cam.e, &cam->e, 32); memcpy(out->cam.u, &cam->u, 32);
    memcpy(out->cam.v, &cam->v, 32); memcpy(out->cam.w, &cam->w, 32);
    out->cam.f = cam->f; out->cam.focus_distance = cam->focus_distance;
    out->cam.wl = cam->wl; out->cam.wt = cam->wt; out->cam.wsize = cam->wsize;
    memcpy(out->cam.C2W, cam->C2W, 128);
    memcpy(out->cam.W2C, cam->W2C, 128);
    return 0;
}
