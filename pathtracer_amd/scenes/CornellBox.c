/* Cornell box: five radius-500 wall spheres, a refracting and a glossy mirror sphere, a 1 x 5 plane light just
 * under the ceiling.  Same objects, order, transforms and camera as the reference's Scenes/CornellBox.c:18-112. */
#include "scene_util.h"

void buildScene(void) {
    scene_camera(0, 0, -15, 0, 0, 0, -3, -2, 2, 4, -20);
    static const scene_row rows[] = {
        {KIT_SPHERE, 1, 0, 0, .75, .25, .25, .05, 1.4, 0, 0, 0, 0, {{OP_SCALE, 500, 500, 500}, {OP_MOVE, -510, 0, 5}}},  /* left */
        {KIT_SPHERE, 1, 0, 0, .25, .25, .75, .05, 1.4, 0, 0, 0, 0, {{OP_SCALE, 500, 500, 500}, {OP_MOVE, 510, 0, 5}}},   /* right */
        {KIT_SPHERE, 1, 0, 0, .75, .75, .75, .05, 1.4, 0, 0, 0, 0, {{OP_SCALE, 500, 500, 500}, {OP_MOVE, 0, 0, 515}}},   /* back */
        {KIT_SPHERE, 1, 0, 0, .75, .75, .75, .02, 1.4, 0, 0, 0, 0, {{OP_SCALE, 500, 500, 500}, {OP_MOVE, 0, -510, 5}}},  /* bottom */
        {KIT_SPHERE, 1, 0, 0, .75, .75, .75, .05, 1.4, 0, 0, 0, 0, {{OP_SCALE, 500, 500, 500}, {OP_MOVE, 0, 510, 5}}},   /* top */
        {KIT_SPHERE, 0, 0, 1, .99, .99, .99, .01, 1.54, 0, 0, 0, 0, {{OP_SCALE, 3.75, 3.75, 3.75}, {OP_MOVE, -5, -4.0, 4.5}}},  /* glass */
        {KIT_SPHERE, 0, 1, 0, .99, .99, .99, .05, 2.47, 0, 0, 0, 0, {{OP_SCALE, 3.75, 3.75, 3.75}, {OP_MOVE, 4, -3.75, 6.5}}},  /* mirror */
        {KIT_PLANE, 1.00, 0.00, 0.0, 1.0, 1.0, 1.0, 0.0, 1.54, 1, 0, 0, 0, {{OP_SCALE, .5, 2.5, 1}, {OP_RX, PI / 2}, {OP_MOVE, 0, 9.995, 5}}},
    };
    scene_add_rows(rows, (int)(sizeof(rows) / sizeof(rows[0])));
}
