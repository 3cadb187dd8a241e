/*
 * ref_gpu_cli.c -- TEST INFRASTRUCTURE.  The reference program with its sample loop swapped for the GPU library,
 * the way a maintainer of the reference would do it: the reference's own scene construction, BoxTree, light weights and
 * dataOutput (compiled from /root/reference by oracle/Makefile), integration/ref_flatten.c and
 * integration/ref_gpu_loop.c in between.  This main() repeats the set-up steps of the reference's main()
 * (PathTracer.c:521-611) by calling the reference's functions; it exists because that main() cannot be edited
 * where it lies.
 *   ref_gpu_cli scene size depth spp out [aperture]     scene: cornell | scifi | bubbletea
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "utils_path.h"
#include "magic.h"

extern struct object3D *object_list;
extern struct object3D *light_list;
extern struct BoxTree *boxtree;
extern struct view *cam;
extern int MAX_DEPTH;
int ref_select_scene(const char *name, int max_depth); /* ref_driver.c: buildScene + build_boxtree + light weights */
long ptx_ref_gpu_loop(int sx, int max_depth, int num_samples, double aperture, double *rgbIm, double *wght, char *output_name);

int main(int argc, char *argv[]) {
    if (argc < 6) {
        fprintf(stderr, "usage: ref_gpu_cli scene size depth spp out [aperture]\n");
        return 2;
    }
    int sx = atoi(argv[2]), depth = atoi(argv[3]), spp = atoi(argv[4]);
    double aperture = argc > 6 ? atof(argv[6]) : 0;
    struct image *im = newImage(sx, sx);
    double *wght = (double *)calloc((size_t)sx * sx, sizeof(double));
    if (!im || !wght) return 3;
    for (long i = 0; i < (long)sx * sx; i++) wght[i] = 1.0;
    if (ref_select_scene(argv[1], depth) < 0) return 4;
    long paths = ptx_ref_gpu_loop(sx, depth, spp, aperture, (double *)im->rgbdata, wght, argv[5]);
    fprintf(stderr, "Total number of rays created: %ld\n", paths);
    return 0;
}
