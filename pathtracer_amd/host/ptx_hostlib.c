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
