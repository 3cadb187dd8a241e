/*
 * ptx_oracle.h -- TEST INFRASTRUCTURE: interface of the CPU oracle.
 *
 * The oracle is a plain-C, double-precision restatement of the reference's
 * hot path (PathTracer.c:46-484 and :617-654, utils_path.c:29-114,
 * :439-1428, magic.c:457-474) working on the flat scene of
 * include/ptx_scene.h.  It is the checker for the HIP library and is never
 * part of the product: only tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py may load it.
 */
#ifndef PTX_ORACLE_H
#define PTX_ORACLE_H

#include "ptx_scene.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct orc_scene orc_scene;

/* Exact work counters of a render (the reference's NUM_RAYS counts paths
 * only, and racily: PathTracer.c:645). */
typedef struct orc_stats {
    uint64_t paths;          /* pixel-samples traced */
    uint64_t closest_hits;   /* findFirstHit calls from PathTrace (:282) */
    uint64_t shadow_rays;    /* findFirstHit calls for explicit sampling (:428) */
    uint64_t rindex_oob;     /* r_index accesses outside [0,9] (reference UB) */
    uint64_t end_depth, end_miss, end_light, end_rr;
} orc_stats;

orc_scene *orc_scene_create(const ptx_scene_desc *desc);
orc_scene *orc_scene_load(const char *path);
void orc_scene_destroy(orc_scene *s);
const ptx_scene_desc *orc_scene_desc(const orc_scene *s);

/* findFirstHit for n rays (directions used as given, not normalised). */
void orc_first_hit(const orc_scene *s, const ptx_ray *rays, int n, ptx_hit *hits);

/* Passes k0 .. k0+nk-1 of the sample loop over the rows of the sx*sx image
 * that prm assigns to its rank (all rows when n_ranks <= 1).  rgb (sx*sx*3) and wght (sx*sx) are whole-image buffers
 * that are read and updated.  threads<=0: all cores (counter stream only;
 * the LCG48 stream always runs serially).  lcg_state is the running
 * drand48 state for PTX_RNG_LCG48 (in/out), ignored otherwise. */
void orc_render(const orc_scene *s, const ptx_render_params *prm, int k0, int nk, double *rgb, double *wght,
                int threads, uint64_t *lcg_state, orc_stats *stats);

/* Radiance of individual paths (counter stream): out[3*n]. */
void orc_trace_pixels(const orc_scene *s, const ptx_render_params *prm, const int32_t *pix_i, const int32_t *pix_j,
                      const int32_t *samp, int n, double *out);

/* Debugging aid: per-bounce state on stderr. */
void orc_set_debug(int on);

/* Known-answer entry points; random numbers come from script[] in order. */
void orc_kat_cos_weighted(const double n[3], const double *script, double out[3]);
void orc_kat_hemisphere_reorient(const double n[3], const double v[3], double out[3]);
double orc_kat_random_normal(const double *script);
void orc_kat_refract(const orc_scene *s, int obj_index, const double d_in[3], const double p[3], const double n_in[3],
                     double r_index[10], int *r_index_index, const double *script, double d_out[3], double n_out[3]);
void orc_kat_reflect(const double d_in[3], const double p[3], const double n_in[3], double d_out[3]);
void orc_kat_texmap(const orc_scene *s, int obj_index, int which, double a, double b, double out[3]);
void orc_kat_normal_deform(const orc_scene *s, int obj_index, double a, double b, const double n_in[3], double n_out[3]);
void orc_kat_primary_ray(const orc_scene *s, int sx, int i, int j, double aperture, const double *script, double p0[3],
                         double d_out[3]);
void orc_kat_light_sample(const orc_scene *s, int obj_index, const double *script, double out[3]);

#ifdef __cplusplus
}
#endif
#endif
