/* scene_util.h -- small helpers shared by the shipped scene files: a table row per object and a loop that turns
 * rows into kit calls.  A scene file may equally call the kit functions one by one, as the reference's scene
 * files do; this is only a compact notation. */
#ifndef PTX_SCENE_UTIL_H
#define PTX_SCENE_UTIL_H

enum { OP_END = 0, OP_RX, OP_RY, OP_RZ, OP_MOVE, OP_SCALE };
typedef struct { int op; double a, b, c; } scene_op;
typedef struct {
    int kind;                       /* KIT_PLANE ... KIT_ROOM */
    double diff, refl, tran;        /* material split, sums to 1 */
    double r, g, b;                 /* colour */
    double refl_sig, r_index;
    int light;                      /* 1: area light */
    const char *tex, *normal, *alpha; /* image files or NULL */
    scene_op ops[6];                /* applied in order, ends at OP_END */
} scene_row;

static void scene_apply(struct object3D *o, const scene_op *ops) {
    for (; ops->op != OP_END; ops++) switch (ops->op) {
        case OP_RX: RotateX(o, ops->a); break;
        case OP_RY: RotateY(o, ops->a); break;
        case OP_RZ: RotateZ(o, ops->a); break;
        case OP_MOVE: Translate(o, ops->a, ops->b, ops->c); break;
        case OP_SCALE: Scale(o, ops->a, ops->b, ops->c); break;
        }
}

static void scene_add_rows(const scene_row *rows, int n) {
    for (int k = 0; k < n; k++) {
        const scene_row *s = &rows[k];
        struct object3D *o = NULL;
        switch (s->kind) {
        case KIT_PLANE: o = newPlane(s->diff, s->refl, s->tran, s->r, s->g, s->b, s->refl_sig, s->r_index); break;
        case KIT_SPHERE: o = newSphere(s->diff, s->refl, s->tran, s->r, s->g, s->b, s->refl_sig, s->r_index); break;
        case KIT_CYL: o = newCyl(s->diff, s->refl, s->tran, s->r, s->g, s->b, s->refl_sig, s->r_index); break;
        case KIT_CUBE: o = newCube(s->diff, s->refl, s->tran, s->r, s->g, s->b, s->refl_sig, s->r_index); break;
        default: o = newRoom(s->diff, s->refl, s->tran, s->r, s->g, s->b, s->refl_sig, s->r_index); break;
        }
        scene_apply(o, s->ops);
        if (s->tex) loadTexture(o, s->tex, 1, &texture_list);
        if (s->normal) loadTexture(o, s->normal, 2, &texture_list);
        if (s->alpha) loadTexture(o, s->alpha, 3, &texture_list);
        invert(&o->T[0][0], &o->Tinv[0][0]);
        o->isLightSource = s->light;
        insertObject(o, &object_list);
    }
}

static void scene_camera(double ex, double ey, double ez, double lookx, double looky, double lookz, double f, double wl,
                         double wt, double wsize, double focus) {
    struct point3D e = {ex, ey, ez, 1}, g = {lookx - ex, looky - ey, lookz - ez, 1}, up = {0, 1, 0, 1};
    cam = setupView(&e, &g, &up, f, wl, wt, wsize, focus);
}
#endif
