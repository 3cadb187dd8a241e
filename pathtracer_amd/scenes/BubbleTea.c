/* BubbleTea: a mirror teapot (objs/teapot.obj) on a granite pedestal in a panelled room with two facing mirrors,
 * playing cards with an alpha-mapped outline scattered on the floor, soap bubbles (an outer water sphere around
 * an inner air sphere) and a plane light.  Same objects, order, transforms and camera as the reference's
 * buildScene.c:5-367. */
#include "scene_util.h"

#define TEX(name) "./Textures/" name ".ppm"

static scene_row card(const char *face, double turn_y, double x, double y, double z) {
    scene_row r = {KIT_PLANE, 1, 0, 0, 1, 1, 1, 0, 1, 0, face, 0, "./Textures/card_alpha.pgm",
                   {{OP_RX, -PI / 2}, {OP_SCALE, 6.4, 1, 8.9}, {OP_SCALE, .1, .1, .1}, {OP_RY, turn_y}, {OP_MOVE, x, y, z}}};
    return r;
}

static void bubble(double radius, double inner, double x, double y, double z) {
    scene_row shell[2] = {
        {KIT_SPHERE, 0, 0, 1, 1, 1, 1, 0, 1.333, 0, 0, 0, 0, {{OP_SCALE, radius, radius, radius}, {OP_MOVE, x, y, z}}},
        {KIT_SPHERE, 0, 0, 1, 1, 1, 1, 0, 1, 0, 0, 0, 0,
         {{OP_SCALE, radius, radius, radius}, {OP_SCALE, inner, inner, inner}, {OP_MOVE, x, y, z}}},
    };
    scene_add_rows(shell, 2);
}

void buildScene(void) {
    scene_camera(0, 1, -15, 0, 0, 0, -3, -2, 2, 4, -22);

    const scene_row pedestal = {KIT_CUBE, .9, .1, 0, 1, 1, 1, .05, 1, 0, TEX("granite"), TEX("granite_normal"), 0,
                                {{OP_SCALE, 3, 3, 3}, {OP_MOVE, 0, -7, 7}}};
    scene_add_rows(&pedestal, 1);

    double T[4][4] = {{1, 0, 0, 0}, {0, 1, 0, 0}, {0, 0, 1, 0}, {0, 0, 0, 1}};
    ScaleMat(T, 1.2, 1.2, 1.2);
    TranslateMat(T, 0, -3.7, 7);
    loadObj("objs/teapot.obj", 0, 1, 0, 1, 1, 1, 0, 1, &object_list, T);

    const scene_row room[] = {
        {KIT_PLANE, 1, 0, 0, 1, 1, 1, 0, 1, 0, TEX("woodfloor"), TEX("woodfloor_normal"), 0,
         {{OP_RX, PI / 2}, {OP_SCALE, 25, 25, 25}, {OP_MOVE, 0, -10, 5}}},                                                   /* floor */
        {KIT_PLANE, 1, 0, 0, 1, 1, 1, 0, 1, 0, 0, TEX("ceiling_normal"), 0,
         {{OP_RX, -PI / 2}, {OP_SCALE, 25, 25, 25}, {OP_MOVE, 0, 10, 5}}},                                                   /* roof */
        {KIT_PLANE, 0, 1, 0, 1, 1, 1, 0, 1, 0, 0, 0, 0, {{OP_SCALE, 25, 25, 25}, {OP_MOVE, 0, 0, 25}}},                     /* mirror, back */
        {KIT_PLANE, 0, 1, 0, 1, 1, 1, 0, 1, 0, 0, 0, 0, {{OP_SCALE, 25, 25, 25}, {OP_MOVE, 0, 0, -20}}},                    /* mirror, behind camera */
        {KIT_PLANE, 1, 0, 0, .64, .03, .03, 0, 1, 0, 0, 0, 0, {{OP_RY, PI / 2}, {OP_SCALE, 25, 25, 25}, {OP_MOVE, 10, 0, 5}}},   /* right wall */
        {KIT_PLANE, 1, 0, 0, .2, .19, .38, 0, 1, 0, 0, 0, 0, {{OP_RY, PI / 2}, {OP_SCALE, 25, 25, 25}, {OP_MOVE, -10, 0, 5}}},   /* left wall */
        {KIT_CUBE, 1, 0, 0, 1, 1, 1, 0, 1, 0, 0, 0, 0, {{OP_SCALE, .2, .5, 25}, {OP_MOVE, 10, -9.75, 5}}},                  /* baseboards */
        {KIT_CUBE, 1, 0, 0, 1, 1, 1, 0, 1, 0, 0, 0, 0, {{OP_SCALE, .2, .5, 25}, {OP_MOVE, -10, -9.75, 5}}},
        {KIT_CUBE, 1, 0, 0, 1, 1, 1, 0, 1, 0, 0, 0, 0, {{OP_SCALE, .2, .6, 25}, {OP_MOVE, 10, 9.75, 5}}},                   /* crown trim */
        {KIT_CUBE, 1, 0, 0, 1, 1, 1, 0, 1, 0, 0, 0, 0, {{OP_SCALE, .2, .6, 25}, {OP_MOVE, -10, 9.75, 5}}},
        {KIT_PLANE, 1, 0, 0, 1, 1, 1, 0, 1, 1, 0, 0, 0, {{OP_SCALE, .5, 2.5, 1}, {OP_RX, PI / 2}, {OP_MOVE, 0, 9.9999, 5}}}, /* light */
    };
    scene_add_rows(room, (int)(sizeof(room) / sizeof(room[0])));

    const scene_row cards[] = {
        card(TEX("card_back_red"), PI / 4.5, -4, -9.99, 3),
        card(TEX("card_ace_spade"), PI / 3.14, -6, -9.99, 5),
        card(TEX("card_9_club"), PI / 3.14, -2.5, -3.99, 5.2),
        card(TEX("card_queen_heart"), -PI / 26, 4, -9.985, 3.2),
        card(TEX("card_back_black"), -PI / 3.2, 8, -9.99, 7),
        card(TEX("card_back_red"), PI / 4.7, 7.6, -9.99, 8),
        card(TEX("card_back_black"), -PI / 3.14, -8.2, -9.99, 5.8),
        card(TEX("card_4_diamond"), -PI / 3.27, 0.7, -9.99, 1.6),
        card(TEX("card_back_red"), PI / 2.78, 6.6, -9.99, 11),
        card(TEX("card_back_black"), PI / 1.1, -6.2, -9.99, 10.4),
        card(TEX("card_back_black"), PI / 4.2, 6.5, -9.99, 1),
        card(TEX("card_king_club"), -PI / 3.6, -8, -9.99, 2),
        card(TEX("card_back_black"), -PI / 3, -2.5, -9.985, 3.1),
        card(TEX("card_8_diamond"), PI / 6, 4.6, -9.99, 4.3),
    };
    scene_add_rows(cards, (int)(sizeof(cards) / sizeof(cards[0])));

    bubble(.4, .98, 3.9, 0, 7);
    bubble(.6, .98, 4.5, 1.5, 7);
    bubble(.7, .98, 7, 3.7, 6);
    bubble(2, .99, 5, 8, 11);
    bubble(1.75, .99, 3.2, 3.75, 9);
    bubble(2.5, .99, 6, 6, 0);
}
