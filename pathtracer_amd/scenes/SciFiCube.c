/* SciFiCube: a level-3 Menger sponge of 8000 half-mirror half-glass cubes turned 45 degrees about z, a sphere
 * light inside it and a plane light above, in a room of textured, normal-mapped planes with two pillars.  Same
 * objects, order, transforms and camera as the reference's Scenes/SciFiCube.c:8-181. */
#include "scene_util.h"

/* which of the 27 sub-cells of a cell are kept, in the reference's order: the front layer (8), the four corners
 * of the middle layer, the back layer (8) */
static const int sponge_cells[20][3] = {
    {0, 0, 0}, {1, 0, 0}, {2, 0, 0}, {0, 1, 0}, {2, 1, 0}, {0, 2, 0}, {1, 2, 0}, {2, 2, 0},
    {0, 0, 1}, {0, 2, 1}, {2, 0, 1}, {2, 2, 1},
    {0, 0, 2}, {1, 0, 2}, {2, 0, 2}, {0, 1, 2}, {2, 1, 2}, {0, 2, 2}, {1, 2, 2}, {2, 2, 2},
};

static void sponge(int level, double T[4][4], double size, double x, double y, double z) {
    if (level == 0) {
        struct object3D *o = newCube(0, 0.5, 0.5, .7, .3, .9, .05, 1);
        Scale(o, size, size, size);
        Scale(o, .5, .5, .5);
        Translate(o, x, y, z);
        matMult(T, o->T);
        invert(&o->T[0][0], &o->Tinv[0][0]);
        insertObject(o, &object_list);
        return;
    }
    double s = size / 3.0;
    for (int k = 0; k < 20; k++) {
        const int *c = sponge_cells[k];
        /* offsets are written as x, x+s, x+s*2 in the reference; keep the same roundings */
        double ox = c[0] == 0 ? x : (c[0] == 1 ? x + s : x + s * 2);
        double oy = c[1] == 0 ? y : (c[1] == 1 ? y + s : y + s * 2);
        double oz = c[2] == 0 ? z : (c[2] == 1 ? z + s : z + s * 2);
        sponge(level - 1, T, size / 3.0, ox, oy, oz);
    }
}

void buildScene(void) {
    scene_camera(0, 8, -19.5, 0, 3.5, 0, -5, -2, 2, 4, -22);
    double T[4][4] = {{1, 0, 0, 0}, {0, 1, 0, 0}, {0, 0, 1, 0}, {0, 0, 0, 1}};
    RotateZMat(T, PI / 4);
    sponge(3, T, 4, 0, 0, 0);
    static const char *alu = "./Textures/aluminium.ppm", *alu_n = "./Textures/aluminium_normal.ppm";
    static const char *wall = "./Textures/scifi_wall.ppm", *wall_n = "./Textures/scifi_wall_normal.ppm";
    const scene_row rows[] = {
        {KIT_SPHERE, 1, 0, 0, .7, .3, .9, 0, 1, 1, 0, 0, 0, {{OP_SCALE, .7, .7, .7}, {OP_MOVE, 0, 2.7, 2}}},
        {KIT_PLANE, 1, 0, 0, 1, 1, 1, 0, 1, 1, 0, 0, 0, {{OP_SCALE, 3, 1, 1}, {OP_RX, PI / 2}, {OP_MOVE, 0, 14.9999, 3}}},
        {KIT_PLANE, 0.4, 0.6, 0, 1, 1, 1, .05, 1, 0, alu, alu_n, 0, {{OP_RX, -PI / 2}, {OP_SCALE, 25, 25, 25}, {OP_MOVE, 0, -15, 5}}},  /* floor */
        {KIT_PLANE, 0.4, 0.6, 0, 1, 1, 1, .05, 1, 0, alu, alu_n, 0, {{OP_RX, PI / 2}, {OP_SCALE, 25, 30, 25}, {OP_MOVE, 0, 15, 5}}},    /* roof */
        {KIT_PLANE, .4, .6, 0, .5, .5, .5, 0, 1, 0, 0, 0, 0, {{OP_SCALE, 25, 25, 25}, {OP_MOVE, 0, 0, 25}}},                            /* mirrored wall */
        {KIT_PLANE, 1, 0, 0, 1, 1, 1, 0, 1, 0, wall, wall_n, 0, {{OP_SCALE, 25, 15, 25}, {OP_RX, PI}, {OP_MOVE, 0, 0, -20}}},           /* behind camera */
        {KIT_PLANE, 1, 0, 0, .25, .25, .75, 0, 1, 0, wall, wall_n, 0,
         {{OP_RX, PI}, {OP_RY, PI / 2}, {OP_RY, -PI / 6}, {OP_SCALE, 25, 15, 25}, {OP_MOVE, 20, 0, 5}}},                                 /* right wall */
        {KIT_PLANE, 1, 0, 0, .75, .25, .25, 0, 1, 0, wall, wall_n, 0,
         {{OP_RX, PI}, {OP_RY, PI / 2}, {OP_RY, PI / 6}, {OP_SCALE, 25, 15, 25}, {OP_MOVE, -20, 0, 5}}},                                 /* left wall */
        {KIT_CYL, .8, .2, 0, 1, 1, 1, .1, 1, 0, 0, 0, 0, {{OP_RX, PI / 2}, {OP_SCALE, 1, 25, 1}, {OP_MOVE, -7.5, -5, 25}}},
        {KIT_CYL, .8, .2, 0, 1, 1, 1, .1, 1, 0, 0, 0, 0, {{OP_RX, PI / 2}, {OP_SCALE, 1, 25, 1}, {OP_MOVE, 7.5, -5, 25}}},
    };
    scene_add_rows(rows, (int)(sizeof(rows) / sizeof(rows[0])));
}
