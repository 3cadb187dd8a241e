/*
 * ptx_main.c -- the PathTracer command line on top of the GPU library.
 *
 *   ./PathTracer size rec_depth num_samples output_name [aperture]
 *
 * Same arguments, messages, progressive output (every 25 passes) and exit behaviour as the reference's main()
 * (PathTracer.c:486-672).  The scene is compiled in, as there: the build includes the file named by
 * PTX_SCENE_FILE (a buildScene() definition; default scenes/CornellBox.c), and the sampling features are chosen
 * at build time with IS=1 ES=1 RR=1 (-D__USE_IS/-D__USE_ES/-D__USE_RR, reference Makefile:6-16).  What differs is
 * the body of the sample loop: instead of tracing on the CPU it asks libptx_hip.so for the passes and downloads
 * the accumulator whenever an image is due.
 */
#include <time.h>

#include "ptx_hip.h"
#include "ptx_kit.h"

#ifndef PTX_SCENE_FILE
#define PTX_SCENE_FILE "../scenes/CornellBox.c"
#endif
#include PTX_SCENE_FILE

static int die_gpu(const char *what) {
    fprintf(stderr, "%s failed: %s\n", what, ptx_last_error());
    exit(0);
}

#define PTX_MAX_GPUS 64

/* dataOutput() (reference PathTracer.c:655, :661) with the tone mapping done on the GPU: same two files, same
 * bytes, same messages.  With several contexts every one downloads its rows, and the tone mapping runs on the
 * first context's GPU over the gathered image.  PTX_HOST_TONEMAP=1 in the environment runs the host routine
 * instead. */
static void output_image(ptx_context **gpu, int n_gpus, double *rgbIm, double *wght, unsigned char *bits, int sx, char *name) {
    for (int r = 0; r < n_gpus; r++)
        if (ptx_download(gpu[r], rgbIm, wght) != 0) die_gpu("ptx_download");
    if (getenv("PTX_HOST_TONEMAP")) {
        dataOutput(rgbIm, sx, name);
        return;
    }
    ptx_tonemap_info info;
    if (n_gpus == 1) {
        if (ptx_tonemap(gpu[0], bits, &info) != 0) die_gpu("ptx_tonemap");
    } else {
        double *d_rgb = NULL;
        if (ptx_gather(gpu, n_gpus, &d_rgb, NULL) != 0) die_gpu("ptx_gather");
        if (ptx_tonemap_device(gpu[0], d_rgb, (uint64_t)sx * sx * 3, bits, &info) != 0) die_gpu("ptx_tonemap_device");
    }
    ptxh_write_output(rgbIm, bits, sx, name, info.min, info.max, info.low_cut, info.high_cut);
}

int main(int argc, char *argv[]) {
    if (argc < 5) {
        fprintf(stderr, "PathTracer: Can not parse input parameters\n");
        fprintf(stderr, "USAGE: PathTracer size rec_depth num_samples output_name\n");
        fprintf(stderr, "   size = Image size (both along x and y)\n");
        fprintf(stderr, "   rec_depth = Recursion depth\n");
        fprintf(stderr, "   num_samples = Number of samples per pixel\n");
        fprintf(stderr, "   output_name = Name of the output file, e.g. MyRender.ppm\n");
        fprintf(stderr, "   aperture = (OPTIONAL) Aperture for thin-lens model, eg. 0 = pinhole \n");
        exit(0);
    }
    int sx = atoi(argv[1]);
    int max_depth = atoi(argv[2]);
    int num_samples = atoi(argv[3]);
    char output_name[1024];
    strncpy(output_name, argv[4], sizeof(output_name) - 1);
    output_name[sizeof(output_name) - 1] = 0;
    double aperture = argc == 6 ? atof(argv[5]) : 0;

    int flags = 0;
    fprintf(stderr, "Rendering image at %d x %d\n", sx, sx);
    fprintf(stderr, "Recursion depth = %d\n", max_depth);
    fprintf(stderr, "Number of samples = %d\n", num_samples);
    fprintf(stderr, "Output file name: %s\n\n", output_name);
#ifdef __USE_ES
    fprintf(stderr, "Using Explicit Light Sampling\n");
    flags |= PTX_USE_ES;
#endif
#ifdef __USE_IS
    fprintf(stderr, "Using Importance Sampling\n");
    flags |= PTX_USE_IS;
#endif
#ifdef __USE_RR
    fprintf(stderr, "Using Russian Roulette Termination\n");
    flags |= PTX_USE_RR;
#endif
    if (aperture > 0) fprintf(stderr, "Simulating Depth of Field\n");

    struct image *im = newImage(sx, sx);
    double *wght = (double *)calloc((size_t)sx * sx, sizeof(double));
    unsigned char *bits = (unsigned char *)malloc((size_t)sx * sx * 3);
    if (!im || !wght || !bits) {
        fprintf(stderr, "Unable to allocate memory for image\n");
        exit(0);
    }
    double *rgbIm = (double *)im->rgbdata;
    for (long i = 0; i < (long)sx * sx; i++) wght[i] = 1.0;

    buildScene();
    if (cam == NULL) {
        fprintf(stderr, "Unable to set up the view and camera parameters. Out of memory!\n");
        exit(0);
    }
    ptx_scene_desc scene;
    if (ptxh_flatten(&scene) != 0) {
        fprintf(stderr, "The scene has no objects\n");
        exit(0);
    }
    fprintf(stderr, "\nView parameters:\n");
    fprintf(stderr, "Left=%f, Top=%f, Width=%f, f=%f\n", cam->wl, cam->wt, cam->wsize, cam->f);
    if (cam->focus_distance > cam->f) {
        fprintf(stderr, "focus_distance is too close! It must be farther than the focal length, %.1f.\n", cam->f);
        ptxh_reset();
        deleteImage(im);
        exit(0);
    }
    struct point3D fp = {0, 0, cam->focus_distance, 1};
    matVecMult(cam->C2W, &fp);
    printf("Camera is at: (%.2f, %.2f, %.2f)\n", cam->e.px, cam->e.py, cam->e.pz);
    printf("Focusing on plane at: (%.2f, %.2f, %.2f)\n", fp.px, fp.py, fp.pz);
    fprintf(stderr, "\n");

    /* PTX_GPUS=N: the image rows are dealt out in bands of eight to N contexts, context r on device r modulo the
     * number of devices (so N may exceed it).  The reference is one process on one machine's cores
     * (PathTracer.c:620); here every GPU runs all passes for its rows and the rows are put together for output. */
    int n_gpus = getenv("PTX_GPUS") ? atoi(getenv("PTX_GPUS")) : 1;
    int n_devices = 0;
    if (ptx_device_count(&n_devices) != 0 || n_devices < 1) die_gpu("ptx_device_count");
    if (n_gpus < 1) n_gpus = 1;
    if (n_gpus > PTX_MAX_GPUS) n_gpus = PTX_MAX_GPUS;
    int band_rows = 8;
    while (band_rows > 1 && (sx + band_rows - 1) / band_rows < n_gpus) band_rows /= 2;
    if (n_gpus > sx) n_gpus = sx;
    ptx_context *gpu[PTX_MAX_GPUS];
    ptx_render_params prm;
    memset(&prm, 0, sizeof(prm));
    prm.sx = sx;
    prm.max_depth = max_depth;
    prm.flags = flags;
    if (getenv("PTX_CLEAN")) prm.flags |= PTX_CLEAN; /* not the reference's estimator: see ptx_scene.h */
    prm.rng_mode = PTX_RNG_COUNTER;
    prm.aperture = aperture;
    prm.seed = (uint64_t)time(NULL); /* the reference seeds drand48 from the clock too (PathTracer.c:560) */
    if (getenv("PTX_SEED")) prm.seed = strtoull(getenv("PTX_SEED"), NULL, 10);
    prm.n_ranks = n_gpus;
    prm.band_rows = band_rows;
    for (int r = 0; r < n_gpus; r++) {
        gpu[r] = NULL;
        if (ptx_init(r % n_devices, &gpu[r]) != 0) die_gpu("ptx_init");
        if (ptx_scene_upload(gpu[r], &scene) != 0) die_gpu("ptx_scene_upload");
        prm.rank = r;
        if (ptx_render_begin(gpu[r], &prm) != 0) die_gpu("ptx_render_begin");
    }
    if (n_gpus > 1) fprintf(stderr, "Rows in bands of %d over %d GPU contexts on %d device(s)\n", band_rows, n_gpus, n_devices);

    struct timespec t1, t2;
    clock_gettime(CLOCK_MONOTONIC, &t1);
    fprintf(stderr, "Rendering pass... ");
    /* images are due after pass 0, 25, 50, ... (k % 25 == 0, PathTracer.c:655): run the passes up to the next
     * such k in one call */
    int k = 0;
    while (k < num_samples) {
        int next_due = k % 25 == 0 ? k : k + (25 - k % 25); /* first pass index >= k whose image is written */
        int upto = next_due + 1 < num_samples ? next_due + 1 : num_samples;
        for (int q = k; q < upto; q++) fprintf(stderr, "%d/%d, ", q, num_samples);
        for (int r = 0; r < n_gpus; r++) /* queued: the GPUs work side by side */
            if (ptx_render_passes(gpu[r], k, upto - k) != 0) die_gpu("ptx_render_passes");
        k = upto;
        if ((k - 1) % 25 == 0) output_image(gpu, n_gpus, rgbIm, wght, bits, sx, output_name);
    }
    for (int r = 0; r < n_gpus; r++)
        if (ptx_sync(gpu[r]) != 0) die_gpu("ptx_sync");
    clock_gettime(CLOCK_MONOTONIC, &t2);
    fprintf(stderr, "\nDone!\n");
    output_image(gpu, n_gpus, rgbIm, wght, bits, sx, output_name);

    ptx_stats st;
    memset(&st, 0, sizeof(st));
    for (int r = 0; r < n_gpus; r++) {
        ptx_stats one;
        if (ptx_get_stats(gpu[r], &one) != 0) die_gpu("ptx_get_stats");
        st.paths += one.paths;
        st.closest_hits += one.closest_hits;
        st.shadow_rays += one.shadow_rays;
        st.rindex_oob += one.rindex_oob;
    }
    double secs = (double)(t2.tv_sec - t1.tv_sec) + 1e-9 * (double)(t2.tv_nsec - t1.tv_nsec);
    /* the reference's "rays" are paths, one per pixel sample (PathTracer.c:645) */
    fprintf(stderr, "Total number of rays created: %ld\n", (long)st.paths);
    fprintf(stderr, "Rays per second: %f\n", (double)st.paths / secs);
    fprintf(stderr, "Ray queries (closest hit + shadow): %ld, %f per second\n", (long)(st.closest_hits + st.shadow_rays),
            (double)(st.closest_hits + st.shadow_rays) / secs);

    for (int r = 0; r < n_gpus; r++) ptx_shutdown(gpu[r]);
    ptxh_reset();
    deleteImage(im);
    free(wght);
    free(bits);
    exit(0);
}
