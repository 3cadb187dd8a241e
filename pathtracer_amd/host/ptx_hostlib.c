/*
 * ptx_hostlib.c -- the shipped scenes behind one call, for programs that load the host kit as a shared library
 * (the Python tests and tools do) instead of compiling a scene in.
 */
#include "ptx_kit.h"
#include "ptx_scene_io.h"

void buildScene_cornell(void);
void buildScene_scifi(void);
void buildScene_bubbletea(void);
void buildScene_meshstress(void);

/* Builds the named scene ("cornell", "scifi", "bubbletea", "meshstress") with the kit and flattens it.  Assets (./Textures,
 * objs/) are looked up relative to the current directory, as in the reference.  Returns 0 or -1. */
int ptxh_build_scene(const char *name, ptx_scene_desc *out) {
    ptxh_reset();
    if (!strcmp(name, "cornell")) buildScene_cornell();
    else if (!strcmp(name, "scifi")) buildScene_scifi();
    else if (!strcmp(name, "bubbletea")) buildScene_bubbletea();
    else if (!strcmp(name, "meshstress")) buildScene_meshstress();
    else return -1;
    return ptxh_flatten(out);
}

/* The same, written to a scene file. */
int ptxh_build_scene_file(const char *name, const char *path) {
    ptx_scene_desc d;
    if (ptxh_build_scene(name, &d) != 0) return -1;
    return ptx_scene_write(path, &d);
}

/* A triangle mesh file in a plain setting, written to a scene file: the Cornell camera, one plane light above,
 * the mesh with the identity transform and a grey diffuse material, every object kept in list order.  For
 * looking at a mesh and for testing loadObj(). */
int ptxh_obj_scene_file(const char *obj_path, const char *path) {
    ptxh_reset();
    struct point3D e = {0, 0, -15, 1}, g = {0, 0, 15, 1}, up = {0, 1, 0, 1};
    cam = setupView(&e, &g, &up, -3, -2, 2, 4, -20);
    if (!cam) return -1;
    struct object3D *l = newPlane(1, 0, 0, 1, 1, 1, 0, 1.54);
    Scale(l, .5, 2.5, 1);
    RotateX(l, PI / 2);
    Translate(l, 0, 9.995, 5);
    invert(&l->T[0][0], &l->Tinv[0][0]);
    l->isLightSource = 1;
    insertObject(l, &object_list);
    double T[4][4] = {{1, 0, 0, 0}, {0, 1, 0, 0}, {0, 0, 1, 0}, {0, 0, 0, 1}};
    loadObj(obj_path, 1, 0, 0, .7, .7, .7, 0, 1, &object_list, T);
    setenv("PTX_FLAT_ORDER", "1", 1);
    ptx_scene_desc d;
    if (ptxh_flatten(&d) != 0) return -1;
    return ptx_scene_write(path, &d);
}
