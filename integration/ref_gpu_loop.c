/*
 * ref_gpu_loop.c -- the reference's sample loop (PathTracer.c:617-661) carried out by libptx_hip.so, for a main()
 * that keeps everything else of the reference: argument parsing, newImage, buildScene(), build_boxtree(), the
 * light weights, dataOutput().  Compile inside the reference tree next to ref_flatten.c; link with -lptx_hip.
 *
 * In the reference's main(), replace the lines from `for (k=0; k<num_samples; k++)` (:617) to the final
 * dataOutput (:661) by
 *     NUM_RAYS = ptx_ref_gpu_loop(sx, MAX_DEPTH, num_samples, aperture, rgbIm, wght, output_name);
 * The build flags IS= ES= RR= (-D__USE_IS/-D__USE_ES/-D__USE_RR, reference Makefile:6-16) select the sampling
 * features as they do there, provided this file is compiled with them.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "utils_path.h"
#include "magic.h"
#include "ptx_hip.h"

extern struct object3D *light_list;
extern struct BoxTree *boxtree;
extern struct view *cam;
int ptx_ref_flatten(struct BoxTree *boxtree, struct object3D *light_list, struct view *cam, ptx_scene_desc *out,
                    struct object3D ***flat_out);
void ptx_ref_flatten_free(ptx_scene_desc *s);

static void ptxr_die(const char *what) {
    fprintf(stderr, "%s failed: %s\n", what, ptx_last_error());
    exit(0); /* the reference leaves with exit(0) on every error (PathTracer.c:511-520, :552-555) */
}

/* Returns the number of paths traced (what the reference counts in NUM_RAYS, PathTracer.c:645). */
long ptx_ref_gpu_loop(int sx, int max_depth, int num_samples, double aperture, double *rgbIm, double *wght, char *output_name) {
    ptx_scene_desc scene;
    if (ptx_ref_flatten(boxtree, light_list, cam, &scene, NULL) != 0) {
        fprintf(stderr, "The scene could not be flattened for the GPU\n");
        exit(0);
    }
    ptx_context *gpu = NULL;
    if (ptx_init(0, &gpu) != 0) ptxr_die("ptx_init");
    if (ptx_scene_upload(gpu, &scene) != 0) ptxr_die("ptx_scene_upload");
    ptx_ref_flatten_free(&scene); /* the library holds its own copy */
    ptx_render_params prm;
    memset(&prm, 0, sizeof(prm));
    prm.sx = sx;
    prm.max_depth = max_depth;
    prm.rng_mode = PTX_RNG_COUNTER;
    prm.aperture = aperture;
    prm.seed = (uint64_t)time(NULL); /* srand48(time(NULL)), PathTracer.c:560 */
    if (getenv("PTX_SEED")) prm.seed = strtoull(getenv("PTX_SEED"), NULL, 10);
    prm.n_ranks = 1;
#ifdef __USE_IS
    prm.flags |= PTX_USE_IS;
#endif
#ifdef __USE_ES
    prm.flags |= PTX_USE_ES;
#endif
#ifdef __USE_RR
    prm.flags |= PTX_USE_RR;
#endif
    if (ptx_render_begin(gpu, &prm) != 0) ptxr_die("ptx_render_begin");
    fprintf(stderr, "Rendering pass... ");
    int k = 0;
    while (k < num_samples) {
        int next_due = k % 25 == 0 ? k : k + (25 - k % 25); /* images are written after pass 0, 25, 50, ... (:655) */
        int upto = next_due + 1 < num_samples ? next_due + 1 : num_samples;
        for (int q = k; q < upto; q++) fprintf(stderr, "%d/%d, ", q, num_samples);
        if (ptx_render_passes(gpu, k, upto - k) != 0) ptxr_die("ptx_render_passes");
        k = upto;
        if ((k - 1) % 25 == 0) {
            if (ptx_download(gpu, rgbIm, wght) != 0) ptxr_die("ptx_download");
            dataOutput(rgbIm, sx, output_name);
        }
    }
    fprintf(stderr, "\nDone!\n");
    if (ptx_download(gpu, rgbIm, wght) != 0) ptxr_die("ptx_download");
    dataOutput(rgbIm, sx, output_name);
    ptx_stats st;
    if (ptx_get_stats(gpu, &st) != 0) ptxr_die("ptx_get_stats");
    ptx_shutdown(gpu);
    return (long)st.paths;
}
