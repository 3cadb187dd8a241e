/*
 * ptx_scene.h -- flat, pointer-free description of a PathTracer scene.
 *
 * The reference keeps its scene as pointer-linked C structs reachable from
 * the globals object_list, light_list, boxtree, texture_list and cam
 * (reference PathTracer.c:34-40, PathTracer.h:42-182, magic.h:13-30).  This
 * header is the same information laid out as plain arrays so that it can
 * cross a C ABI, be uploaded to a GPU, or be written to a file.
 *
 * Objects are stored in the order findFirstHit (reference
 * PathTracer.c:198-251) visits them: a pre-order walk of the BoxTree, each
 * node's own object list first, then child1, then child2.  Because the
 * reference keeps the first of two equal lambdas (strict '<' at
 * PathTracer.c:233), the position of an object in this array is also its
 * tie-break rank.  Objects that build_boxtree loses (reference
 * magic.c:224-231 with append() at :367-379) are not in the array.
 *
 * All doubles are IEEE binary64 exactly as the reference computed them.
 */
#ifndef PTX_SCENE_H
#define PTX_SCENE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PTX_SCENE_MAGIC 0x31435358545450ULL /* "PTTXSC1" */

/* Primitive kinds: one per intersect function of the reference
 * (utils_path.c:488 plane, :546 sphere, :625 cylinder, :781 triangle,
 * :832 cube, :997 room). */
enum {
    PTX_PLANE = 0,
    PTX_SPHERE = 1,
    PTX_CYL = 2,
    PTX_TRIANGLE = 3,
    PTX_CUBE = 4,
    PTX_ROOM = 5
};

/* Which light-sampling routine object3D.randomPoint pointed at
 * (utils_path.c:1189 planeSample, :1199 sphereSample, :1211 cylSample;
 * triangle/cube/room have none). */

typedef struct ptx_object {
    int32_t type;        /* PTX_PLANE .. PTX_ROOM */
    int32_t is_light;    /* object3D.isLightSource */
    int32_t tex_img;     /* index into images[] or -1 (object3D.texImg) */
    int32_t normal_img;  /* index into images[] or -1 (object3D.normalImg) */
    int32_t alpha_img;   /* index into images[] or -1 (object3D.alphaImg) */
    int32_t src_id;      /* insertObject() call order, 0-based */
    double col[3];       /* object3D.col */
    double v[3][4];      /* triangle vertices v1..v3 (px,py,pz,pw) */
    double T[16];        /* object3D.T, row major */
    double Tinv[16];     /* object3D.Tinv, row major */
    double box[6];       /* world AABB: xl xu yl yu zl zu */
    double diffPct, reflPct, tranPct;
    double refl_sig, r_index;
    double surfaceArea;  /* LSweight before normalisation (PathTracer.c:603) */
    double LSweight;     /* normalised weight (PathTracer.c:609) */
} ptx_object;

/* One BoxTree node (magic.h:25-30) with its direct objects as a range of
 * the object array. */
typedef struct ptx_node {
    double box[6];
    int32_t child1, child2;  /* node index or -1 */
    int32_t first_obj, n_objs;
} ptx_node;

/* A texture, normal map (3 channels) or alpha map (1 channel).  The
 * reference stores double(u8)/divisor (utils_path.c:1849, :1921); the
 * bytes and the divisor are kept so every consumer can rebuild those
 * doubles bit for bit. */
typedef struct ptx_image {
    int32_t sx, sy, channels, pad;
    double divisor;       /* PPM: the file's maxval; PGM: 255.0 */
    uint64_t data_offset; /* into the texel byte pool */
} ptx_image;

/* struct view (PathTracer.h:170-182). */
typedef struct ptx_camera {
    double e[4], u[4], v[4], w[4];
    double f, focus_distance, wl, wt, wsize;
    double C2W[16], W2C[16];
} ptx_camera;

typedef struct ptx_scene_desc {
    uint64_t magic;
    int32_t n_objects, n_nodes, n_lights, n_images;
    uint64_t n_texel_bytes;
    const ptx_object *objects;
    const ptx_node *nodes;      /* nodes[0] is the root; n_nodes may be 0 */
    const int32_t *lights;      /* object indices in light_list/LSnext order */
    const ptx_image *images;
    const uint8_t *texels;
    ptx_camera cam;
} ptx_scene_desc;

/* Build flags of the reference Makefile (:6-16). */
enum { PTX_USE_IS = 1, PTX_USE_ES = 2, PTX_USE_RR = 4 };
/* Not in the reference, never the default: an estimator without the reference's known defects, for users who want
 * the picture rather than parity.  With this flag
 *  - the weight of the explicit light sample is clamped to [0, 1] (the reference clamps at 1 only, PathTracer.c:437,
 *    so a light seen from behind subtracts light);
 *  - a light met by a path counts unless the previous bounce already sampled the lights explicitly: the flag that
 *    says so (CEL) is set after every diffuse bounce with explicit sampling, whether or not the shadow ray reached
 *    the light, and cleared by mirror and refraction bounces (the reference sets it only when the light was
 *    reached and lets it leak through the bounces that follow, PathTracer.c:431-445);
 *  - the primary ray goes through a uniformly random point of its pixel, not the centre (two more random numbers
 *    per path, drawn first);
 *  - the accumulator is linear: rgb += colour, wght += 1 (the reference's is rgb += colour * 2^(-ln wght),
 *    wght += R + G + B, PathTracer.c:646-652); the mean image is rgb / (wght - 1). */
enum { PTX_CLEAN = 8 };

/* Which stream of uniform numbers stands in for the reference's drand48().
 * PTX_RNG_COUNTER: stateless, u = mix(seed, pixel, sample, draw index); the
 *   only mode the GPU implements.
 * PTX_RNG_LCG48: glibc drand48 restated (one global 48-bit LCG consumed in
 *   pixel order); CPU oracle only, for bit-exact comparison with a
 *   single-threaded run of the reference. */
enum { PTX_RNG_COUNTER = 0, PTX_RNG_LCG48 = 1 };

typedef struct ptx_render_params {
    int32_t sx;          /* image is sx by sx (PathTracer.c:521) */
    int32_t max_depth;   /* MAX_DEPTH (PathTracer.c:522) */
    int32_t flags;       /* PTX_USE_IS | PTX_USE_ES | PTX_USE_RR [| PTX_CLEAN] */
    int32_t rng_mode;    /* PTX_RNG_* */
    double aperture;     /* PathTracer.c:525-529 */
    uint64_t seed;
    /* Which image rows this call works on.  Rows are grouped in bands of
     * band_rows consecutive rows; band b belongs to rank (b % n_ranks).
     * n_ranks = 1 (or 0) means the whole image.  The reference has one
     * process and no such split (PathTracer.c:620 shares rows among OpenMP
     * threads instead). */
    int32_t rank, n_ranks, band_rows, pad;
} ptx_render_params;

#if defined(__HIPCC__)
#define PTX_SCENE_FN __host__ __device__ static inline
#else
#define PTX_SCENE_FN static inline
#endif

/* Number of rows owned by (rank, n_ranks, band_rows) in an sx-row image, and
 * the local index of an owned row j (or -1 if j is not owned). */
PTX_SCENE_FN int32_t ptx_rows_owned(int32_t sx, int32_t rank, int32_t n_ranks, int32_t band_rows) {
    if (n_ranks <= 1) return sx;
    int32_t n = 0;
    for (int32_t j0 = 0; j0 < sx; j0 += band_rows)
        if ((j0 / band_rows) % n_ranks == rank) n += (sx - j0 < band_rows ? sx - j0 : band_rows);
    return n;
}
PTX_SCENE_FN int32_t ptx_local_row(int32_t j, int32_t rank, int32_t n_ranks, int32_t band_rows) {
    if (n_ranks <= 1) return j;
    int32_t b = j / band_rows;
    if (b % n_ranks != rank) return -1;
    return (b / n_ranks) * band_rows + j % band_rows;
}
PTX_SCENE_FN int32_t ptx_global_row(int32_t local, int32_t rank, int32_t n_ranks, int32_t band_rows) {
    if (n_ranks <= 1) return local;
    return ((local / band_rows) * n_ranks + rank) * band_rows + local % band_rows;
}

/* One ray for the first-hit entry points.  os is the index of the object
 * to skip (the reference's Os pointer) or -1. */
typedef struct ptx_ray {
    double p0[3];
    double d[3];
    int32_t os;
    int32_t pad;
} ptx_ray;

/* Result of findFirstHit: obj = index in objects[] or -1 on a miss (then
 * the other fields are unspecified). */
typedef struct ptx_hit {
    double lambda;
    double p[3];
    double n[3];
    double a, b;
    int32_t obj;
    int32_t pad;
} ptx_hit;

#ifdef __cplusplus
}
#endif
#endif
