/* Mesh stress scene: the Cornell room (wall spheres, plane light) around two procedurally built triangle meshes, a
 * bumpy glossy "boulder" and a knobbly diffuse torus, about 2*PTX_MESH_N*PTX_MESH_N triangles each (environment
 * variable PTX_MESH_N, default 512: 1.05 million triangles in all).  It stands in for the reference's benchmark
 * scene with dragon.obj and bunny.obj, whose meshes are not part of the reference checkout and which no scene file
 * of the reference builds.  Every triangle is an object of its own, made with newTriangle() and insertObject()
 * like the triangles loadObj() produces (reference meshes.c:81-99).
 *
 * With this many objects the host must not run the reference's BoxTree builder: flatten with PTX_FLAT_ORDER=1
 * (ptxh_flatten keeps insertion-list order and every object). */
#include <stdlib.h>

#include "scene_util.h"

static struct point3D mesh_point(int which, double u, double v) {
    /* u, v in [0,1) */
    const double tu = 2 * PI * u, tv = which == 0 ? PI * v : 2 * PI * v;
    struct point3D p;
    if (which == 0) { /* boulder: sphere of radius 3.2 with two frequencies of bumps */
        double r = 3.2 * (1.0 + 0.06 * sin(9 * tu) * sin(7 * tv) + 0.02 * sin(41 * tu + 1.3) * sin(37 * tv));
        p.px = -4.2 + r * cos(tu) * sin(tv);
        p.py = -6.0 + r * cos(tv);
        p.pz = 5.5 + r * sin(tu) * sin(tv);
    } else { /* torus: major radius 2.6, minor 1.1 with knobs */
        double rr = 1.1 * (1.0 + 0.08 * sin(13 * tu) * sin(11 * tv) + 0.015 * sin(53 * tu) * sin(47 * tv + 0.7));
        double ring = 2.6 + rr * cos(tv);
        p.px = 4.3 + ring * cos(tu);
        p.py = -2.5 + rr * sin(tv) + 0.9 * sin(tu);
        p.pz = 7.0 + ring * sin(tu) * 0.8;
    }
    p.pw = 1;
    return p;
}

static void add_mesh(int which, int n, double diff, double refl, double r, double g, double b, double sig) {
    for (int iu = 0; iu < n; iu++)
        for (int iv = 0; iv < n; iv++) {
            double u0 = (double)iu / n, u1 = (double)(iu + 1) / n, v0 = (double)iv / n, v1 = (double)(iv + 1) / n;
            if (which == 0) { /* keep away from the poles' degenerate triangles */
                v0 = 0.002 + 0.996 * v0;
                v1 = 0.002 + 0.996 * v1;
            }
            struct point3D a = mesh_point(which, u0, v0), bq = mesh_point(which, u1, v0), c = mesh_point(which, u1, v1),
                           d = mesh_point(which, u0, v1);
            struct object3D *t1 = newTriangle(a, bq, c, diff, refl, 0, r, g, b, sig, 1.0);
            insertObject(t1, &object_list);
            struct object3D *t2 = newTriangle(a, c, d, diff, refl, 0, r, g, b, sig, 1.0);
            insertObject(t2, &object_list);
        }
}

void buildScene(void) {
    scene_camera(0, 0, -15, 0, 0, 0, -3, -2, 2, 4, -20);
    static const scene_row rows[] = {
        {KIT_SPHERE, 1, 0, 0, .75, .25, .25, .05, 1.4, 0, 0, 0, 0, {{OP_SCALE, 500, 500, 500}, {OP_MOVE, -510, 0, 5}}},
        {KIT_SPHERE, 1, 0, 0, .25, .25, .75, .05, 1.4, 0, 0, 0, 0, {{OP_SCALE, 500, 500, 500}, {OP_MOVE, 510, 0, 5}}},
        {KIT_SPHERE, 1, 0, 0, .75, .75, .75, .05, 1.4, 0, 0, 0, 0, {{OP_SCALE, 500, 500, 500}, {OP_MOVE, 0, 0, 515}}},
        {KIT_SPHERE, 1, 0, 0, .75, .75, .75, .02, 1.4, 0, 0, 0, 0, {{OP_SCALE, 500, 500, 500}, {OP_MOVE, 0, -510, 5}}},
        {KIT_SPHERE, 1, 0, 0, .75, .75, .75, .05, 1.4, 0, 0, 0, 0, {{OP_SCALE, 500, 500, 500}, {OP_MOVE, 0, 510, 5}}},
        {KIT_PLANE, 1.00, 0.00, 0.0, 1.0, 1.0, 1.0, 0.0, 1.54, 1, 0, 0, 0, {{OP_SCALE, .5, 2.5, 1}, {OP_RX, PI / 2}, {OP_MOVE, 0, 9.995, 5}}},
    };
    scene_add_rows(rows, (int)(sizeof(rows) / sizeof(rows[0])));
    int n = 512;
    if (getenv("PTX_MESH_N")) n = atoi(getenv("PTX_MESH_N"));
    if (n < 2) n = 2;
    add_mesh(0, n, 0.7, 0.3, .85, .65, .35, 0.08);
    add_mesh(1, n, 1.0, 0.0, .35, .75, .45, 0.0);
}
