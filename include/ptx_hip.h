/*
 * ptx_hip.h -- C ABI of libptx_hip.so, the MI355X (gfx950) implementation of
 * PathTracer's per-pixel Monte-Carlo loop.
 *
 * The reference has no plugin or FFI layer.  Its hot path is reached through
 * C globals (object_list, light_list, boxtree, texture_list, cam, MAX_DEPTH:
 * PathTracer.c:34-40) and the body of the sample loop in main()
 * (PathTracer.c:617-654).  The functions below are what a host program calls
 * in place of that loop body; each comment names the reference code it
 * stands in for.  Everything is plain C: pointers, sizes, ints.  All
 * functions return 0 on success and a negative PTX_E* code on failure;
 * ptx_last_error() then describes the failure.  (The reference itself only
 * prints to stderr and calls exit(0): PathTracer.c:511-520, :552-555,
 * :570-581.)
 *
 * Threading: a context is used from one host thread at a time, as the
 * reference's main() is.  One context drives one GPU; a multi-GPU job is
 * one process (and one context) per GPU, each given its share of the image
 * rows through ptx_render_params.rank / n_ranks / band_rows.
 */
#ifndef PTX_HIP_H
#define PTX_HIP_H

#include <stddef.h>

#include "ptx_scene.h"

#ifdef __cplusplus
extern "C" {
#endif

enum {
    PTX_OK = 0,
    PTX_EINVAL = -1,   /* bad argument or scene */
    PTX_EDEVICE = -2,  /* HIP runtime error */
    PTX_ESTATE = -3,   /* call made in the wrong order */
    PTX_ENOMEM = -4
};

typedef struct ptx_context ptx_context;

/* Exact work counters since ptx_render_begin.  Stands in for NUM_RAYS
 * (PathTracer.c:39, :645), which counts paths only and loses updates under
 * OpenMP. */
typedef struct ptx_stats {
    uint64_t paths;         /* pixel-samples traced */
    uint64_t closest_hits;  /* closest-hit queries (PathTracer.c:282) */
    uint64_t shadow_rays;   /* explicit-light queries (PathTracer.c:428) */
    uint64_t rindex_oob;    /* refraction-stack accesses outside [0,9] */
} ptx_stats;

/* Creates a context on HIP device `device`.  No counterpart in the
 * reference (it has no device). */
int ptx_init(int device, ptx_context **ctx);

/* Releases every device buffer and the context. */
void ptx_shutdown(ptx_context *ctx);

/* Copies the scene to the GPU and builds the acceleration structure.  Call
 * after buildScene(), build_boxtree() and the light-weight normalisation
 * (PathTracer.c:563-566, :600-611) have produced the flat description; it
 * replaces the hot path's reads of object_list / light_list / boxtree /
 * texture_list / cam.  The description may be freed afterwards. */
int ptx_scene_upload(ptx_context *ctx, const ptx_scene_desc *scene);

/* Starts a render: allocates the radiance accumulator (all zero) and the
 * per-pixel weights (all one) for this rank's rows, as newImage() and the
 * wght loop do (PathTracer.c:550-557), and zeroes the counters.  Only
 * PTX_RNG_COUNTER is implemented on the GPU. */
int ptx_render_begin(ptx_context *ctx, const ptx_render_params *params);

/* As ptx_render_begin, but the accumulators live in device memory the
 * caller owns (for example torch tensors that a collective will read):
 * d_rgb holds rows_owned*sx*3 doubles, d_wght rows_owned*sx doubles, local
 * row order.  The caller initialises them (0 and 1 for a new render). */
int ptx_render_begin_external(ptx_context *ctx, const ptx_render_params *params, double *d_rgb, double *d_wght);

/* Runs passes k0 .. k0+nk-1 of the sample loop for every pixel this rank
 * owns: primary ray, PathTrace, and the ordered fold into the accumulator
 * (PathTracer.c:617-654 without the dataOutput call at :655).  Returns
 * once the work is queued on the context's stream.  The paths are traced
 * first, their colours going to a buffer in device memory (32 bytes per
 * pixel and pass, allocated by the first call that needs it, kept until
 * ptx_shutdown, at most 32 GiB and half of the free memory), and folded
 * into the accumulators in pass order afterwards; a call too large for the
 * buffer is carried out in several rounds, with the same result. */
int ptx_render_passes(ptx_context *ctx, int k0, int nk);

/* ptx_render_passes for callers that cannot allocate device memory (code under HIP graph capture, serving loops
 * and harnesses that forbid hipMalloc).  No counterpart in the reference.  It allocates nothing: the traced
 * samples go through a fixed 256 MiB array in the library's device image (one per GPU, held until the process
 * ends) instead of the buffer ptx_render_passes allocates, in as many rounds of trace + fold as that takes (eight
 * passes a round at 1024 x 1024); the accumulators are the ones ptx_render_begin or ptx_render_begin_external
 * set up, and the result is the same bit for bit.  Limits: this rank's share of the image may have at most
 * ptx_static_max_pixels() = 2 097 152 pixels (PTX_EINVAL beyond, e.g. a whole 2048 x 2048 image on one rank);
 * ptx_last_trace_ms reports no launches for it (per-launch events would have to be created).  All contexts on
 * one GPU share the array: calls that overlap in time on one GPU race; calls in sequence need no clearing (every
 * record is written before it is read). */
int ptx_render_passes_static(ptx_context *ctx, int k0, int nk);
size_t ptx_static_max_pixels(void);

/* Waits for queued work (the reference's loop is synchronous). */
int ptx_sync(ptx_context *ctx);

/* Copies this rank's rows of the accumulators into whole-image host
 * buffers laid out like the reference's rgbIm (sx*sx*3 doubles) and wght
 * (sx*sx doubles) (PathTracer.c:550-557); rows of other ranks are left
 * untouched.  Either pointer may be NULL.  What the host calls before
 * dataOutput() (PathTracer.c:655, :661). */
int ptx_download(ptx_context *ctx, double *rgbIm, double *wght);

/* Device self-check of the arithmetic that stands in for IEEE divisions and libm routines on the hot path; no
 * counterpart in the reference.  results[0]: number of (byte, divisor) pairs, 0..255 / 1..255, for which the
 * texel conversion differs from the reference's byte/divisor (utils_path.c:1849) -- must be 0.  results[1..4]:
 * largest error, in units in the last place, of the logarithm, square root, reciprocal and reciprocal square
 * root used in place of log, sqrt, 1/x and 1/sqrt(x). */
int ptx_selftest(ptx_context *ctx, double results[5]);

/* What dataOutput() prints and chooses while tone mapping (reference utils_path.c:2011-2047): the minimum and
 * maximum of the accumulator values, and the low and high cut picked from the 1000-bin histogram. */
typedef struct ptx_tonemap_info {
    double min, max;
    double low_cut, high_cut;
    uint64_t recomputed_on_host; /* values whose byte depended on the last bit of pow(): redone with libm */
} ptx_tonemap_info;

/* The tone mapping of dataOutput() (reference utils_path.c:2011-2060: normalise by the global minimum and
 * maximum, 1000-bin histogram, cut selection, clip, power 0.75, bytes) of this context's radiance accumulator,
 * which must hold the whole image (one rank).  bits receives sx*sx*3 bytes, the payload of the reference's P6
 * file, identical to what dataOutput() computes from the same accumulator.  info may be NULL. */
int ptx_tonemap(ptx_context *ctx, uint8_t *bits, ptx_tonemap_info *info);

/* The same for any device buffer of n_values doubles (an image gathered from several ranks, for instance). */
int ptx_tonemap_device(ptx_context *ctx, const double *d_im, uint64_t n_values, uint8_t *bits, ptx_tonemap_info *info);

/* The reverse of ptx_download: sets this rank's rows of the accumulators from whole-image host buffers (the
 * doubles of a <name>.pfm written earlier and the matching weights), so that a render can be taken up again at
 * the pass where it stopped.  The reference has no such path (its accumulators live and die with the process,
 * PathTracer.c:550-557).  Either pointer may be NULL. */
int ptx_upload(ptx_context *ctx, const double *rgbIm, const double *wght);

/* Device addresses of this rank's accumulators and the number of rows they
 * hold (for a device-side gather).  */
int ptx_accumulators(ptx_context *ctx, double **d_rgb, double **d_wght, int32_t *rows_owned);

/* Number of HIP devices.  The reference has none. */
int ptx_device_count(int *n);

/* Several GPUs driven from one process (the reference is one process: PathTracer.c:486-672): ranks[r] is a
 * context that renders rank r of n_ranks of one image (ptx_render_params.rank / n_ranks / band_rows), on whatever
 * device it was created for.  Waits for all of them and copies every context's rows, device to device, into
 * whole-image buffers on ranks[0]'s device laid out like the reference's rgbIm and wght (PathTracer.c:550-557);
 * the buffers belong to ranks[0] and live until its ptx_shutdown.  What a host needs before ptx_tonemap_device
 * when it has split the image; for the host copy of the image call ptx_download on every context. */
int ptx_gather(ptx_context *const *ranks, int n_ranks, double **d_rgb, double **d_wght);

/* findFirstHit (PathTracer.c:198-251) for n rays in host memory; the ray
 * direction is used as given, as in the reference.  hits[i].obj is the
 * index into the uploaded objects[] or -1.  For triangles and rooms a and b
 * are returned as 0 (the reference leaves them unset). */
int ptx_first_hit(ptx_context *ctx, const ptx_ray *rays, int64_t n, ptx_hit *hits);

/* The same for rays and hits already in device memory. */
int ptx_first_hit_device(ptx_context *ctx, const ptx_ray *d_rays, int64_t n, ptx_hit *d_hits);

/* Radiance of n individual paths (pixel i,j of an sx-wide image, sample
 * index k): the value PathTrace returns in *col (PathTracer.c:644) before
 * it is folded into the image.  out holds 3*n doubles, host memory.  Uses
 * the flags, depth, aperture and seed of params. */
int ptx_trace_paths(ptx_context *ctx, const ptx_render_params *params, const int32_t *pix_i, const int32_t *pix_j,
                    const int32_t *sample, int64_t n, double *out);

/* What the closest-hit queries (PathTracer.c:282, :428) of the most recent ptx_trace_paths call tested: that entry
 * point runs the instrumented build of the path kernel.  The reference counts nothing of the kind; these figures
 * turn a scene into FLOP and bytes per ray (24 FLOP and 32 bytes per node visit, the per-type cost of each
 * primitive test).  Synchronises. */
typedef struct ptx_walk_stats {
    uint64_t queries;        /* closest-hit + shadow queries */
    uint64_t hits;           /* of which found a primitive */
    uint64_t list_tests;     /* primitive tests on the always-tested list */
    uint64_t node_visits;    /* hierarchy nodes whose box was tested (inner nodes and leaves) */
    uint64_t leaf_tests[6];  /* primitive tests at leaves, by PTX_PLANE .. PTX_ROOM */
    uint64_t texel_lookups;  /* texMap / normalDeform / alphaMap calls (four texels each) */
} ptx_walk_stats;
int ptx_last_walk_stats(ptx_context *ctx, ptx_walk_stats *stats);

/* Counters accumulated since ptx_render_begin (synchronises). */
int ptx_get_stats(ptx_context *ctx, ptx_stats *stats);

/* Device time in milliseconds of the most recent ptx_render_passes call,
 * from HIP events recorded on the context's stream around its kernels
 * (synchronises).  The reference times with time(NULL) at 1 s resolution
 * (PathTracer.c:614, :658). */
int ptx_last_pass_ms(ptx_context *ctx, float *ms);

/* Of that time, the part spent in the path-tracing kernel alone (the rest is the in-order accumulation of the
 * traced samples), summed over the kernel's launches in that call; *launches, if not NULL, receives their number.
 * No counterpart in the reference; used to report the kernel against its roofline. */
int ptx_last_trace_ms(ptx_context *ctx, float *ms, int *launches);

/* Makes the context launch on an existing hipStream_t (passed as void*),
 * e.g. the stream a caller's framework is using.  NULL restores the
 * context's own stream. */
int ptx_set_stream(ptx_context *ctx, void *hip_stream);

/* Text of the last error on the calling thread. */
const char *ptx_last_error(void);

/* Library version, "major.minor". */
const char *ptx_version(void);

#ifdef __cplusplus
}
#endif
#endif
