// ptx_hip.hip -- kernels and C ABI of libptx_hip.so (gfx950 / MI355X).
//
// Execution model.  A render_passes call runs two kernels per round.  k_trace_samples: a wave owns a run of
// consecutive samples of an 8x8 pixel tile, one lane per pixel; when a lane's path ends it writes the path's colour
// to the sample buffer and starts its pixel's next sample in the same loop iteration, so the wave stays busy until
// its slowest pixel has finished the run.  k_fold_samples: one lane per pixel folds the pixel's samples into the
// accumulator in sample order, which keeps the reference's order-dependent accumulation (PathTracer.c:643-652)
// exact.  The scene of a small configuration is read with wave-uniform addresses (scalar loads of a few KB that
// stay in the scalar cache); larger scenes add a per-lane walk of the bounding-volume hierarchy.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "ptx_device.h"
#include "ptx_hip.h"
#include "ptx_internal.h"
#include "ptx_trace_bvh.h"

// ------------------------------------------------------------------ kernels
// Rendering is two kernels.  The reference folds each finished path into its pixel's accumulator with a
// non-linear, order-dependent step (rgb += col * pow(2, -log(wt)); wt += R + G + B, PathTracer.c:646-652), so the
// samples of a pixel must be folded in sample order -- but they can be traced in any order.  k_trace_samples traces
// paths and writes each path's colour to a sample buffer in global memory, [pixel][sample] records of 32 bytes
// (r, g, b, 0), so that a lane fills one cache line after another with its pixel's consecutive samples;
// k_fold_samples then walks every pixel's samples in order, a line at a time.  That leaves the hot kernel free of the fold (a log and
// a power per sample, which only the lanes that have just finished a path would execute) and free to cut the work
// into as many waves as the GPU needs: a wave traces samples ks..ke of an 8x8 pixel tile, whatever the image size
// or the share of it this GPU holds.  The buffer costs 32 bytes written and read per path, against ~6000 vector
// instructions traced.
#define PTX_SAMPLE_DOUBLES 4

// One sample folded into a pixel's accumulator (reference PathTracer.c:646-652)
__device__ __forceinline__ void fold_sample(double &ar, double &ag, double &ab, double &wt, double cr, double cg, double cb) {
    const double f = exp2(-lean_log(wt));
    ar += cr * f;
    ag += cg * f;
    ab += cb * f;
    wt += cr;
    wt += cg;
    wt += cb;
}

// The samples 0 .. n_samples-1 of one pixel folded into its accumulator, in sample order: four samples (one 128-byte line
// of the pixel's stream) per trip.  linear: PTX_CLEAN's accumulator instead of the reference's, rgb += colour, wght += 1.
__device__ __forceinline__ void fold_pixel(const double *samples, size_t pix, size_t stride, int n_samples, double *rgb, double *wght,
                                           int linear) {
    double ar = rgb[pix * 3 + 0], ag = rgb[pix * 3 + 1], ab = rgb[pix * 3 + 2], wt = wght[pix];
    const double4 *in = (const double4 *)(samples + pix * stride * PTX_SAMPLE_DOUBLES);
    int q = 0;
    if (linear) {
        for (; q < n_samples; q++) {
            const double4 s0 = in[q];
            ar += s0.x;
            ag += s0.y;
            ab += s0.z;
        }
        wt += (double)n_samples;
    }
    for (; q + 4 <= n_samples; q += 4) {
        const double4 s0 = in[q], s1 = in[q + 1], s2 = in[q + 2], s3 = in[q + 3];
        fold_sample(ar, ag, ab, wt, s0.x, s0.y, s0.z);
        fold_sample(ar, ag, ab, wt, s1.x, s1.y, s1.z);
        fold_sample(ar, ag, ab, wt, s2.x, s2.y, s2.z);
        fold_sample(ar, ag, ab, wt, s3.x, s3.y, s3.z);
    }
    for (; q < n_samples; q++) {
        const double4 s0 = in[q];
        fold_sample(ar, ag, ab, wt, s0.x, s0.y, s0.z);
    }
    rgb[pix * 3 + 0] = ar;
    rgb[pix * 3 + 1] = ag;
    rgb[pix * 3 + 2] = ab;
    wght[pix] = wt;
}

// Waves per SIMD the trace kernel is compiled for.  The kernel for small scenes with a zero aperture (the benchmark
// case) is held to 128 registers, a little below what it wants, and runs four waves per SIMD; the variants with a
// hierarchy or a lens need more registers and lose 10 % that way: three waves.
template <bool USE_BVH, bool PINHOLE>
struct TraceShape {
    static constexpr int waves_per_simd = (USE_BVH || !PINHOLE) ? 3 : 4;
};

// One lane per pixel of the tile; a lane starts its pixel's next sample as soon as its path ends (path
// regeneration), so the wave stays full until its slowest pixel is done with the range.
// PINHOLE: the aperture is zero, so every sample of a pixel starts along the same direction.
// samples: colours of samples k0..k1-1 of this rank's pixels, `stride` records per pixel; sub_len: samples per
// wave, a multiple of four.
template <int FLAGS, bool USE_BVH, bool PINHOLE>
__global__ void __launch_bounds__(PTX_BLOCK, (TraceShape<USE_BVH, PINHOLE>::waves_per_simd))
    k_trace_samples(DevScene sc, DevRender rp, int k0, int k1, int sub_len, int n_full, int tail_len, int n_tiles, size_t stride,
                    double *__restrict__ samples, unsigned long long *__restrict__ stats) {
    __shared__ double lds_d_[LS_PARKED + 9][PTX_BLOCK];
    __shared__ int lds_i_[LS_ISLOTS][PTX_BLOCK];
    const LaneStore ls{&lds_d_[0][threadIdx.x], &lds_i_[0][threadIdx.x], LS_PEND};
    const int lane = threadIdx.x & 63;
    const int tile = blockIdx.x % n_tiles, sub = blockIdx.x / n_tiles;
    const int tiles_x = (rp.sx + 7) >> 3;
    const int tx = tile % tiles_x, ty = tile / tiles_x;
    const int i = tx * 8 + (lane & 7);
    const int jl = ty * 8 + (lane >> 3);
    const bool inside = i < rp.sx && jl < rp.rows_owned;
    const int j = inside ? ptx_global_row(jl, rp.rank, rp.n_ranks, rp.band_rows) : 0;
    const size_t pix = inside ? (size_t)jl * rp.sx + i : 0;
    // the first n_full ranges have sub_len samples, those after them tail_len: the waves the GPU runs last are short
    // ones, so that its compute units run dry at about the same time
    const int ks = sub < n_full ? k0 + sub * sub_len : k0 + n_full * sub_len + (sub - n_full) * tail_len;
    const int my_len = sub < n_full ? sub_len : tail_len;
    const int ke = inside ? (ks + my_len < k1 ? ks + my_len : k1) : ks;

    Counters cnt = {0, 0, 0};
    uint32_t paths = 0;
    Path ps;
    bool alive = false;
    int k = ks;
#ifdef PTX_PHASE_TIMERS
    WalkCount wc_dbg = {};
    __shared__ unsigned long long ph_lds_[3 * PTX_N_PHASES];
    const long long t_begin_dbg = clock64();
    if (lane < 3 * PTX_N_PHASES) ph_lds_[lane] = lane == PH_LAST ? (unsigned long long)t_begin_dbg : 0ULL;
    __syncthreads();
    wc_dbg.ph = USE_BVH ? nullptr : ph_lds_;
    long long iters_dbg = 0;
#endif
    if (PINHOLE && inside) {
        primary_ray(sc, rp, ls, i, j, ks, ps);
        ls.put(LS_PRIMARY_D + 0, ps.d.x); ls.put(LS_PRIMARY_D + 1, ps.d.y); ls.put(LS_PRIMARY_D + 2, ps.d.z);
    }
#ifdef PTX_PHASE_TIMERS
    constexpr bool COUNT = true;  // for the PH_* macros
    WalkCount *const wcp = &wc_dbg;
#else
    constexpr bool COUNT = false;
    WalkCount *const wcp = nullptr;
#endif
    for (;;) {
        bool ended = false;
        long long ph_t = PH_T();
        if (!alive && k < ke) {
            if (PINHOLE) primary_ray_cached(sc, rp, ls, i, j, k, ps);
            else primary_ray(sc, rp, ls, i, j, k, ps);
            alive = true;
            ended = rp.max_depth < 1;  // reference PathTracer.c:273 with depth = 1
            PH_ADD(wcp, PH_NEW_PATH, ph_t);
        }
        if (__ballot(alive) == 0ULL) break;
        double glossy = 0;
#ifdef PTX_PHASE_TIMERS
        if (alive && !ended) ended = bounce<FLAGS, USE_BVH, true, true, !PINHOLE>(sc, rp, ls, ps, cnt, glossy, &wc_dbg);
        iters_dbg++;
#else
        if (alive && !ended) ended = bounce<FLAGS, USE_BVH, true, false, !PINHOLE>(sc, rp, ls, ps, cnt, glossy);
#endif
        ph_t = PH_T();
        glossy_service(ps, glossy, lane);
        if (__ballot(glossy > 0) != 0ULL) PH_ADD(wcp, PH_GLOSSY, ph_t);
        ph_t = PH_T();
        if (alive && ended) {
            // Four samples make a 128-byte line of the lane's stream.  The first three wait in LDS; the fourth goes
            // out together with them, so that every line is written once and whole (sub_len and the stride are
            // multiples of four, so lines start at k - ks = 0, 4, ...).  The last samples of a range go out singly.
            const int slot = (k - ks) & 3;
            double4 *out = (double4 *)(samples + (pix * stride + (size_t)(k - k0)) * PTX_SAMPLE_DOUBLES);
            if (slot == 3) {
#pragma unroll
                for (int q = 0; q < 3; q++)
                    out[q - 3] = make_double4(ls.get(LS_PARKED + 3 * q), ls.get(LS_PARKED + 3 * q + 1), ls.get(LS_PARKED + 3 * q + 2), 0.0);
                out[0] = make_double4(ps.I.x, ps.I.y, ps.I.z, 0.0);
            } else if (k + (3 - slot) >= ke) {
                out[0] = make_double4(ps.I.x, ps.I.y, ps.I.z, 0.0);
            } else {
                ls.put(LS_PARKED + 3 * slot, ps.I.x);
                ls.put(LS_PARKED + 3 * slot + 1, ps.I.y);
                ls.put(LS_PARKED + 3 * slot + 2, ps.I.z);
            }
            alive = false;
            k++;
            paths++;
            PH_ADD(wcp, PH_SAMPLE_END, ph_t);
        }
    }
#ifdef PTX_PHASE_TIMERS
    if (!USE_BVH) {
        // the parts of the loop: the wave's cycles, runs and lane visits
        __syncthreads();
        if (lane == 0) {
            ph_lds_[3 * PH_LOOP] = (unsigned long long)(clock64() - t_begin_dbg);
            ph_lds_[3 * PH_LOOP + 1] = (unsigned long long)iters_dbg;
        }
        __syncthreads();
        if (lane < 3 * PTX_N_PHASES && lane != PH_LAST && ph_lds_[lane]) atomicAdd(&stats[PTX_STATS_DBG + lane], ph_lds_[lane]);
    }
    {
        // per-lane maxima are what the wave paid: the slowest lane's cycle counts cover the whole loops
        double v[10] = {(double)(clock64() - t_begin_dbg), (double)wc_dbg.cyc_list, (double)wc_dbg.cyc_walk, (double)iters_dbg,
                        wc_dbg.steps_wave, wc_dbg.leaf_passes_wave, (double)wc_dbg.node_visits, (double)wc_dbg.queries, (double)wc_dbg.cyc_leaf,
                        (double)(wc_dbg.leaf_tests[0] + wc_dbg.leaf_tests[1] + wc_dbg.leaf_tests[2] + wc_dbg.leaf_tests[3] + wc_dbg.leaf_tests[4])};
        for (int c = 0; c < 10; c++) {
            double m = v[c], sum = v[c];
            for (int off = 32; off > 0; off >>= 1) {
                m = fmax(m, __shfl_down(m, off));
                sum += __shfl_down(sum, off);
            }
            const bool as_max = c < 4 || c == 8;
            if (lane == 0) atomicAdd(&stats[PTX_STATS_WALK + c], (unsigned long long)((as_max ? m : sum) + 0.5));
        }
    }
#endif
    // one set of atomics per wave
    unsigned long long c0 = paths, c1 = cnt.closest, c2 = cnt.shadow, c3 = cnt.oob;
    // (as a loop: unrolled, these 48 lane permutes at the end change the register assignment of the loop above, which
    // then runs 0.9 % slower)
#pragma nounroll
    for (int off = 32; off > 0; off >>= 1) {
        c0 += __shfl_down(c0, off);
        c1 += __shfl_down(c1, off);
        c2 += __shfl_down(c2, off);
        c3 += __shfl_down(c3, off);
    }
    if (lane == 0) {
        atomicAdd(&stats[0], c0);
        atomicAdd(&stats[1], c1);
        atomicAdd(&stats[2], c2);
        if (c3) atomicAdd(&stats[3], c3);
    }
}

// The accumulation for n_samples samples of every pixel, in sample order; one lane per pixel, four samples (one
// 128-byte line of the lane's own stream) per trip.
// linear: PTX_CLEAN's accumulator instead of the reference's, rgb += colour and wght += 1.
__global__ void __launch_bounds__(256) k_fold_samples(const double *__restrict__ samples, size_t n_pix, size_t stride,
                                                       int n_samples, double *__restrict__ rgb, double *__restrict__ wght, int linear) {
    const size_t pix = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (pix >= n_pix) return;
    fold_pixel(samples, pix, stride, n_samples, rgb, wght, linear);
}

template <int FLAGS, bool USE_BVH>
__global__ void __launch_bounds__(PTX_BLOCK) k_trace_paths(DevScene sc, DevRender rp, const int32_t *__restrict__ pi,
                                                            const int32_t *__restrict__ pj, const int32_t *__restrict__ pk,
                                                            long n, double *__restrict__ out, unsigned long long *__restrict__ walk) {
    PTX_LANE_STORE(ls);
    long q = (long)blockIdx.x * blockDim.x + threadIdx.x;
    Counters cnt = {0, 0, 0};
    WalkCount wc = {};
    Path ps;
    if (q < n) {
        primary_ray(sc, rp, ls, pi[q], pj[q], pk[q], ps);
        if (rp.max_depth >= 1) {
            double glossy;
            while (!bounce<FLAGS, USE_BVH, false, true>(sc, rp, ls, ps, cnt, glossy, &wc)) {
            }
        }
        out[3 * q + 0] = ps.I.x;
        out[3 * q + 1] = ps.I.y;
        out[3 * q + 2] = ps.I.z;
    }
    // this kernel is the instrumented build: what its queries tested, summed over the wave
    unsigned long long v[PTX_N_WALK] = {wc.queries, wc.hits, wc.list_tests, wc.node_visits, wc.leaf_tests[0], wc.leaf_tests[1],
                                        wc.leaf_tests[2], wc.leaf_tests[3], wc.leaf_tests[4], wc.leaf_tests[5], wc.texel_lookups};
#pragma unroll
    for (int c = 0; c < PTX_N_WALK; c++) {
        for (int off = 32; off > 0; off >>= 1) v[c] += __shfl_down(v[c], off);
        if ((threadIdx.x & 63) == 0 && v[c]) atomicAdd(&walk[c], v[c]);
    }
}

template <bool USE_BVH>
__global__ void __launch_bounds__(256) k_first_hit(DevScene sc, const ptx_ray *__restrict__ rays, long n,
                                                    ptx_hit *__restrict__ hits) {
    long q = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= n) return;
    ptx_ray r = rays[q];
    V3 p0 = mk(r.p0[0], r.p0[1], r.p0[2]), d = mk(r.d[0], r.d[1], r.d[2]);
    Hit h = closest_hit_all<USE_BVH>(sc, p0, d, r.os);
    ptx_hit o;
    o.lambda = -1;
    o.obj = -1;
    o.pad = 0;
    o.p[0] = o.p[1] = o.p[2] = 0;
    o.n[0] = o.n[1] = o.n[2] = 0;
    o.a = o.b = 0;
    if (h.obj >= 0) {
        V3 p, nn;
        double a, b;
        finish_hit<true>(sc, h, p0, d, p, nn, a, b);
        o.lambda = h.lambda;
        o.obj = h.obj;
        o.p[0] = p.x; o.p[1] = p.y; o.p[2] = p.z;
        o.n[0] = nn.x; o.n[1] = nn.y; o.n[2] = nn.z;
        o.a = a; o.b = b;
    }
    hits[q] = o;
}

__global__ void k_fill(double *p, size_t n, double v) {
    size_t q = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (q < n) p[q] = v;
}

// Device self-check of the arithmetic helpers that stand in for IEEE operations or library routines.
// out[0] byte/divisor pairs (0..255 / 1..255) where texel_value differs from the division
// out[1..4] largest error in ulp of lean_log, fast_sqrt, fast_rcp, fast_rsqrt against log, sqrt, 1/x, 1/sqrt(x)
__global__ void k_selftest(double *out) {
    __shared__ double s_err[5][256];
    const int t = threadIdx.x;
    double bad = 0, e_log = 0, e_sqrt = 0, e_rcp = 0, e_rsq = 0;
    for (int b = 1; b <= 255; b++) {
        const double inv = fast_rcp((double)b);
        if (texel_value((double)t, (double)b, inv) != (double)t / (double)b) bad += 1;
    }
    uint64_t z = 0x1234567ULL + (uint64_t)t * PTX_GOLDEN + (uint64_t)blockIdx.x * 0x9E3779B1ULL;
    for (int q = 0; q < 4096; q++) {
        z += PTX_GOLDEN;
        const double u = ptx_bits_to_unit(ptx_mix64(z));
        double x = (q & 3) == 0 ? u : ((q & 3) == 1 ? 1.0 + 4000.0 * u : exp((u - 0.5) * 1200.0));
        if ((q & 3) == 3) x = 1.0 + (u - 0.5) * 1e-3;
        if (!(x > 0)) continue;
        auto ulps = [](double got, double want) {
            if (got == want) return 0.0;
            const double ulp = ldexp(1.0, ilogb(want) - 52);
            return fabs(got - want) / ulp;
        };
        e_log = fmax(e_log, x == 1.0 ? fabs(lean_log(x)) * 0x1p52 : ulps(lean_log(x), log(x)));
        e_sqrt = fmax(e_sqrt, ulps(fast_sqrt(x), sqrt(x)));
        e_rcp = fmax(e_rcp, ulps(fast_rcp(x), 1.0 / x));
        e_rsq = fmax(e_rsq, ulps(fast_rsqrt(x), 1.0 / sqrt(x)));
    }
    s_err[0][t] = bad; s_err[1][t] = e_log; s_err[2][t] = e_sqrt; s_err[3][t] = e_rcp; s_err[4][t] = e_rsq;
    __syncthreads();
    if (t < 5) {
        double r = 0;
        for (int q = 0; q < 256; q++) r = t == 0 ? r + s_err[0][q] : fmax(r, s_err[t][q]);
        out[blockIdx.x * 5 + t] = r;
    }
}

// Sample buffer of ptx_render_passes_static, the entry point for callers that cannot allocate device memory: a
// fixed array in the library's own device image, one copy per GPU, held until the process ends.  Host code never
// uses this name as an address: ptx_init asks the runtime for the device address (hipGetSymbolAddress) once per
// context, with the context's device current, and everything else goes through that pointer.
#define PTX_STATIC_SAMPLE_BYTES ((size_t)256 << 20)
static __device__ __align__(256) unsigned char scratch[PTX_STATIC_SAMPLE_BYTES];

// ---------------------------------------------------------------- host side
static thread_local std::string g_err;
static int fail(int code, const std::string &msg) {
    g_err = msg;
    return code;
}
#define HIPCHK(expr)                                                                         \
    do {                                                                                     \
        hipError_t e_ = (expr);                                                              \
        if (e_ != hipSuccess)                                                                \
            return fail(PTX_EDEVICE, std::string(#expr) + ": " + hipGetErrorString(e_));     \
    } while (0)

struct ptx_context {
    int device = 0;
    hipStream_t own_stream = nullptr;
    hipStream_t stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;  // around the last ptx_render_passes
    std::vector<hipEvent_t> trace_ev;         // pairs around each k_trace_samples launch of that call
    int n_trace_launches = 0;
    bool have_timing = false;
    int n_cus = 256;                 // compute units of the device
    double *d_samples = nullptr;     // sample buffer of the render kernels, grown on demand
    size_t samples_bytes = 0;
    double *d_static_samples = nullptr;  // this device's copy of `scratch`
    // scene
    bool have_scene = false;
    DevScene sc{};
    void *d_geom = nullptr, *d_mat = nullptr, *d_lights = nullptr, *d_images = nullptr, *d_texels = nullptr;
    void *d_list_geom = nullptr, *d_list_ids = nullptr, *d_bvh = nullptr, *d_class_geom = nullptr, *d_class_ids = nullptr;
    void *d_wide = nullptr;   // the hierarchy four children to a node (large scenes), or null
    int *d_spill = nullptr;   // stack entries beyond the LDS part of the wide walk's stack, per lane of the largest grid
    bool use_bvh = false;
    int n_objects = 0;
    // render
    bool rendering = false;
    ptx_render_params prm{};
    DevRender rp{};
    double *d_rgb = nullptr, *d_wght = nullptr;
    bool own_accum = false;
    unsigned long long *d_stats = nullptr;
    // whole-image buffers of ptx_gather (on the context that receives the rows), grown on demand
    double *d_whole_rgb = nullptr, *d_whole_wght = nullptr;
    size_t whole_pixels = 0;
    ptx_tonemap_scratch tonemap;  // scratch of ptx_tonemap / ptx_tonemap_device, allocated at the first image
};

static void free_scene(ptx_context *c) {
    for (void *q : {c->d_geom, c->d_mat, c->d_lights, c->d_images, c->d_texels, c->d_list_geom, c->d_list_ids, c->d_bvh, c->d_class_geom,
                    c->d_class_ids, c->d_wide, (void *)c->d_spill})
        (void)hipFree(q);
    c->d_geom = c->d_mat = c->d_lights = c->d_images = c->d_texels = nullptr;
    c->d_list_geom = c->d_list_ids = c->d_bvh = c->d_class_geom = c->d_class_ids = c->d_wide = nullptr;
    c->d_spill = nullptr;
    c->have_scene = false;
}
static void free_render(ptx_context *c) {
    if (c->own_accum) { (void)hipFree(c->d_rgb); (void)hipFree(c->d_wght); }
    c->d_rgb = c->d_wght = nullptr;
    c->own_accum = false;
    c->rendering = false;
}

extern "C" const char *ptx_last_error(void) { return g_err.c_str(); }
extern "C" const char *ptx_version(void) { return "0.1"; }

extern "C" void ptx_shutdown(ptx_context *c);

static int init_context(ptx_context *c) {
    HIPCHK(hipDeviceGetAttribute(&c->n_cus, hipDeviceAttributeMultiprocessorCount, c->device));
    HIPCHK(hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking));
    c->stream = c->own_stream;
    HIPCHK(hipEventCreate(&c->ev0));
    HIPCHK(hipEventCreate(&c->ev1));
    HIPCHK(hipMalloc(&c->d_stats, PTX_N_STATS * sizeof(unsigned long long)));
    HIPCHK(hipMemset(c->d_stats, 0, PTX_N_STATS * sizeof(unsigned long long)));
    HIPCHK(hipGetSymbolAddress((void **)&c->d_static_samples, HIP_SYMBOL(scratch)));  // the current device's copy
    return PTX_OK;
}

extern "C" int ptx_init(int device, ptx_context **out) {
    if (!out) return fail(PTX_EINVAL, "ptx_init: null output pointer");
    int ndev = 0;
    HIPCHK(hipGetDeviceCount(&ndev));
    if (device < 0 || device >= ndev) return fail(PTX_EINVAL, "ptx_init: no such device");
    HIPCHK(hipSetDevice(device));
    ptx_context *c = new ptx_context();
    c->device = device;
    const int rc = init_context(c);
    if (rc) {
        ptx_shutdown(c);  // releases whatever had been created
        return rc;
    }
    *out = c;
    return PTX_OK;
}

extern "C" void ptx_shutdown(ptx_context *c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    free_render(c);
    free_scene(c);
    (void)hipFree(c->d_stats);
    (void)hipFree(c->d_samples);
    (void)hipFree(c->d_whole_rgb);
    (void)hipFree(c->d_whole_wght);
    ptx_tonemap_release(c->tonemap);
    for (hipEvent_t e : c->trace_ev) (void)hipEventDestroy(e);
    if (c->ev0) (void)hipEventDestroy(c->ev0);
    if (c->ev1) (void)hipEventDestroy(c->ev1);
    if (c->own_stream) (void)hipStreamDestroy(c->own_stream);
    delete c;
}

extern "C" int ptx_set_stream(ptx_context *c, void *s) {
    if (!c) return fail(PTX_EINVAL, "null context");
    c->stream = s ? (hipStream_t)s : c->own_stream;
    return PTX_OK;
}

// 3x4 affine part of a row-major 4x4 matrix
static void affine34(const double *m4, double *m) {
    for (int r = 0; r < 3; r++)
        for (int q = 0; q < 4; q++) m[4 * r + q] = m4[4 * r + q];
}
// normalize(cross(v2-v1, v3-v1)) in the reference's operation order
// (triangleIntersect, utils_path.c:788-795), on the host so that the stored
// normal has the bits the reference recomputes at every test
static void tri_normal(const ptx_object &o, double *n) {
    double e12[3], e13[3];
    for (int q = 0; q < 3; q++) {
        e12[q] = o.v[1][q] + o.v[0][q] * -1;
        e13[q] = o.v[2][q] + o.v[0][q] * -1;
    }
    double x = (e12[1] * e13[2]) - (e13[1] * e12[2]);
    double y = (e13[0] * e12[2]) - (e12[0] * e13[2]);
    double z = (e12[0] * e13[1]) - (e13[0] * e12[1]);
    double l = x * x;
    l += (y * y);
    l += (z * z);
    l = 1.0 / sqrt(l);
    n[0] = x * l;
    n[1] = y * l;
    n[2] = z * l;
}

extern "C" int ptx_scene_upload(ptx_context *c, const ptx_scene_desc *s) {
    if (!c || !s) return fail(PTX_EINVAL, "ptx_scene_upload: null argument");
    if (s->magic != PTX_SCENE_MAGIC) return fail(PTX_EINVAL, "ptx_scene_upload: bad magic");
    if (s->n_objects <= 0 || !s->objects) return fail(PTX_EINVAL, "ptx_scene_upload: scene has no objects");
    if (s->n_lights < 0 || s->n_images < 0) return fail(PTX_EINVAL, "ptx_scene_upload: negative count");
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(hipStreamSynchronize(c->stream));
    free_scene(c);
    // a failure part of the way leaves no half-built scene behind
    struct Guard {
        ptx_context *c;
        bool done;
        ~Guard() { if (!done) free_scene(c); }
    } guard{c, false};
    const int n = s->n_objects;
    std::vector<DevGeom> geom(n);
    std::vector<DevMat> mat(n);
    int has_refr = 0;
    for (int k = 0; k < n; k++) {
        const ptx_object &o = s->objects[k];
        if (o.type < PTX_PLANE || o.type > PTX_ROOM) return fail(PTX_EINVAL, "ptx_scene_upload: unknown object type");
        // texMap and normalDeform read three channels per texel, alphaMap one (reference utils_path.c:1322-1406)
        for (int32_t img : {o.tex_img, o.normal_img, o.alpha_img})
            if (img < -1 || img >= s->n_images) return fail(PTX_EINVAL, "ptx_scene_upload: image index out of range");
        if ((o.tex_img >= 0 && s->images[o.tex_img].channels != 3) || (o.normal_img >= 0 && s->images[o.normal_img].channels != 3))
            return fail(PTX_EINVAL, "ptx_scene_upload: a texture or normal map must have three channels");
        if (o.alpha_img >= 0 && s->images[o.alpha_img].channels != 1)
            return fail(PTX_EINVAL, "ptx_scene_upload: an alpha map must have one channel");
        DevGeom &g = geom[k];
        memset(&g, 0, sizeof(g));
        g.type = o.type;
        if (o.type == PTX_TRIANGLE) {
            for (int v = 0; v < 3; v++)
                for (int q = 0; q < 3; q++) g.m[3 * v + q] = o.v[v][q];
            tri_normal(o, &g.m[9]);
        } else {
            affine34(o.Tinv, g.m);
            // a sphere under uniform scale + translation (T exactly diag(r,r,r | c)) is intersected in world space
            const double *T = o.T;
            bool diag = T[1] == 0 && T[2] == 0 && T[4] == 0 && T[6] == 0 && T[8] == 0 && T[9] == 0;
            if (o.type == PTX_SPHERE && diag && T[0] > 0 && T[0] == T[5] && T[0] == T[10]) {
                g.fast = 1;
                g.c[0] = T[3]; g.c[1] = T[7]; g.c[2] = T[11];
                g.e[0] = T[0] * T[0];
                g.e[1] = 1.0 / T[0];
#ifdef PTX_CULL32_PROBE  // (measurement build: single-precision copies in the transform words, which such a sphere does not use)
                float c32[4] = {(float)g.c[0], (float)g.c[1], (float)g.c[2], (float)g.e[0]};
                memcpy(&g.m[0], c32, sizeof(c32));
#endif
            }
            if (o.type == PTX_PLANE) {
                // normalTransform (reference utils_path.c:58-76) of the two canonical normals, term by term
                for (int side = 0; side < 2; side++) {
                    const double nz = side ? -1.0 : 1.0;
                    double v[3];
                    for (int q = 0; q < 3; q++) v[q] = g.m[q] * 0.0 + g.m[4 + q] * 0.0 + g.m[8 + q] * nz;
                    double l = 1.0 / sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
                    for (int q = 0; q < 3; q++) (side ? g.e : g.c)[q] = v[q] * l;
                }
            }
        }
        DevMat &m = mat[k];
        memset(&m, 0, sizeof(m));
        for (int q = 0; q < 3; q++) m.col[q] = o.col[q];
        m.diffPct = o.diffPct; m.reflPct = o.reflPct; m.tranPct = o.tranPct;
        m.refl_sig = o.refl_sig; m.r_index = o.r_index; m.surfaceArea = o.surfaceArea;
        m.is_light = o.is_light; m.tex_img = o.tex_img; m.normal_img = o.normal_img; m.alpha_img = o.alpha_img;
        if (o.tranPct > 0 || o.alpha_img >= 0) has_refr = 1;
    }
    std::vector<DevLight> lights(s->n_lights > 0 ? s->n_lights : 1);
    memset(lights.data(), 0, lights.size() * sizeof(DevLight));
    for (int l = 0; l < s->n_lights; l++) {
        int oi = s->lights[l];
        if (oi < 0 || oi >= n) return fail(PTX_EINVAL, "ptx_scene_upload: light index out of range");
        const ptx_object &o = s->objects[oi];
        lights[l].obj = oi;
        lights[l].type = o.type;
        lights[l].weight = o.LSweight;
        affine34(o.T, lights[l].T);
    }
    std::vector<DevImage> images(s->n_images > 0 ? s->n_images : 1);
    memset(images.data(), 0, images.size() * sizeof(DevImage));
    for (int q = 0; q < s->n_images; q++) {
        const ptx_image &im = s->images[q];
        uint64_t bytes = (uint64_t)im.sx * im.sy * im.channels;
        if (im.sx <= 0 || im.sy <= 0 || (im.channels != 1 && im.channels != 3) || im.data_offset + bytes > s->n_texel_bytes)
            return fail(PTX_EINVAL, "ptx_scene_upload: bad image descriptor");
        images[q].sx = im.sx; images[q].sy = im.sy; images[q].channels = im.channels;
        images[q].divisor = im.divisor; images[q].offset = im.data_offset;
    }
    // Which primitives every ray tests directly, and which go into the hierarchy.  A small scene is tested
    // exhaustively.  In a large one, primitives whose box is far larger than the typical box (walls, floors,
    // lights around a mesh) stay on the direct list: inside a hierarchy their boxes would cover everything.
    std::vector<int32_t> list_ids;
    std::vector<int> bvh_ids;
    const int SMALL_SCENE = 48, MAX_DIRECT = 96;
    if (n <= SMALL_SCENE) {
        for (int k = 0; k < n; k++) list_ids.push_back(k);
    } else {
        std::vector<double> extent(n);
        for (int k = 0; k < n; k++) {
            const double *b = s->objects[k].box;
            extent[k] = (b[1] - b[0]) + (b[3] - b[2]) + (b[5] - b[4]);
        }
        std::vector<double> sorted(extent);
        std::nth_element(sorted.begin(), sorted.begin() + n / 2, sorted.end());
        const double median = sorted[n / 2];
        std::vector<int> order(n);
        for (int k = 0; k < n; k++) order[k] = k;
        std::sort(order.begin(), order.end(), [&](int a, int b) { return extent[a] > extent[b]; });
        // Direct: boxes more than 16 times the median (measured on SciFiCube and BubbleTea: smaller factors put
        // BubbleTea's cards and bubbles on the list, which costs more than walking to them; larger ones send
        // SciFiCube's pillars into the hierarchy, whose boxes most rays cross), and light sources of at least
        // half that size, which every shadow ray heads for.
        std::vector<char> direct(n, 0);
        int n_direct = 0;
        double direct_factor = 16.0;
        if (const char *e = getenv("PTX_DIRECT_FACTOR")) direct_factor = std::max(8.0, atof(e));  // for measurements
        for (int q = 0; q < n && n_direct < MAX_DIRECT; q++) {
            int k = order[q];
            if (!std::isfinite(extent[k])) continue;
            const double factor = s->objects[k].is_light ? 8.0 : direct_factor;
            if (!(extent[k] > 8.0 * median)) break;
            if (!(extent[k] > factor * median)) continue;
            direct[k] = 1;
            n_direct++;
        }
        for (int k = 0; k < n; k++) {
            if (!std::isfinite(extent[k])) direct[k] = 1;  // a box that is not finite cannot be sorted into a hierarchy
            if (direct[k]) list_ids.push_back(k); else bvh_ids.push_back(k);
        }
    }
    // Device copy of the texels: in the file's order, or three-channel images in tiles of 8 x 4 texels, one tile to a
    // 128-byte line (DevImage; scenes with a hierarchy only: the kernels for small scenes are compiled without the tiled
    // look-up).
    // Off unless PTX_TEXEL_LAYOUT=tiled: measured on SciFiCube (textured, normal-mapped walls; tools/texel_layout_ab.sh) the
    // tiles save 6 % of the L2 requests and no time.
    bool tile_texels = false;
    if (const char *e = getenv("PTX_TEXEL_LAYOUT")) tile_texels = !bvh_ids.empty() && strcmp(e, "tiled") == 0;
    std::vector<uint8_t> texels;
    for (int q = 0; q < s->n_images; q++) {
        const ptx_image &im = s->images[q];
        while (texels.size() % 128) texels.push_back(0);
        images[q].offset = texels.size();
        if (im.channels == 3 && tile_texels) {
            const int tiles_x = (im.sx + 7) / 8, tiles_y = (im.sy + 3) / 4;
            images[q].tiles_x = tiles_x;
            texels.resize(texels.size() + (size_t)tiles_x * tiles_y * 128, 0);
            uint8_t *dst = texels.data() + images[q].offset;
            const uint8_t *src = s->texels + im.data_offset;
            for (int y = 0; y < im.sy; y++)
                for (int x = 0; x < im.sx; x++)
                    memcpy(dst + ((size_t)(y >> 2) * tiles_x + (x >> 3)) * 128 + ((y & 3) * 8 + (x & 7)) * 3, src + 3 * ((size_t)y * im.sx + x), 3);
        } else {
            images[q].tiles_x = 0;
            const size_t bytes = (size_t)im.sx * im.sy * im.channels;
            texels.insert(texels.end(), s->texels + im.data_offset, s->texels + im.data_offset + bytes);
        }
    }
    texels.resize(texels.size() + 16, 0);  // spare bytes: a row's two texels are read as one 8-byte word
    std::vector<DevGeom> list_geom(list_ids.size() ? list_ids.size() : 1);
    for (size_t q = 0; q < list_ids.size(); q++) list_geom[q] = geom[list_ids[q]];
    // the same list grouped by intersection routine, for the kernel that sets its queries up group by group
    std::vector<DevGeom> class_geom(list_geom.size());
    std::vector<int32_t> class_ids(list_ids.size() ? list_ids.size() : 1);
    int32_t class_end[LC_N];
    {
        size_t w = 0;
        for (int cls = 0; cls < LC_N; cls++) {
            for (size_t q = 0; q < list_ids.size(); q++) {
                const DevGeom &g = list_geom[q];
                const int of = g.fast ? LC_FAST_SPHERE
                               : g.type == PTX_SPHERE ? LC_SPHERE
                               : g.type == PTX_PLANE ? LC_PLANE
                               : g.type == PTX_TRIANGLE ? LC_TRIANGLE
                               : g.type == PTX_CYL ? LC_CYL : LC_BOX;
                if (of != cls) continue;
                class_geom[w] = g;
                class_ids[w] = list_ids[q];
                w++;
            }
            class_end[cls] = (int32_t)w;
        }
    }
    PtxBvhBuilt built;
    int n_bvh_nodes = 0;
    float bvh_bound = 0;
    // One node array per octant of ray directions (512 bytes per primitive in all; +56 % on the 8000-cube sponge and
    // +23 % on a million triangles over a single array), a single array beyond 16 M primitives.
    // PTX_BVH_ORDERS=1 or 8 overrides the choice (for measurements).
    int bvh_orders = bvh_ids.size() <= (16u << 20) ? 8 : 1;
    if (const char *e = getenv("PTX_BVH_ORDERS")) bvh_orders = atoi(e) == 8 ? 8 : 1;
    if (!bvh_ids.empty()) {
        std::vector<double> boxes(6 * bvh_ids.size());
        for (size_t q = 0; q < bvh_ids.size(); q++) memcpy(&boxes[6 * q], s->objects[bvh_ids[q]].box, 48);
        std::string err;
        if (n > PTX_LEAF_INDEX) return fail(PTX_EINVAL, "ptx_scene_upload: too many objects for the hierarchy's leaf encoding");
        std::vector<int> leaves(bvh_ids.size());
        for (size_t q = 0; q < bvh_ids.size(); q++) {
            const DevGeom &g = geom[bvh_ids[q]];
            leaves[q] = bvh_ids[q] | (g.fast ? PTX_LEAF_FAST : 0) | (g.type << PTX_LEAF_TYPE_SHIFT);
        }
        // Topology: surface-area heuristic on the host up to 4 M primitives (PTX_BVH_BUILD=lbvh or sah overrides)
        bool use_sah = leaves.size() <= (4u << 20);
        if (const char *e = getenv("PTX_BVH_BUILD")) use_sah = strcmp(e, "lbvh") != 0;
        // Node arrays far larger than the L2 caches (eight of 4 MB): the four-wide form of the tree as well, which the
        // render kernel then walks (PTX_BVH_WALK=wide or binary overrides the choice; the images are the same).
        const size_t node_bytes = (size_t)(2 * leaves.size() - 1) * sizeof(PtxBvhNode) * bvh_orders;
        bool want_wide = node_bytes > ((size_t)32 << 20);
        if (const char *e = getenv("PTX_BVH_WALK")) want_wide = strcmp(e, "wide") == 0;
        if (ptx_bvh_build(boxes.data(), leaves.data(), (int)leaves.size(), bvh_orders, use_sah, want_wide, c->stream, &built, err) != 0)
            return fail(PTX_EDEVICE, "ptx_scene_upload: hierarchy build failed: " + err);
        n_bvh_nodes = built.n_nodes;
        bvh_bound = built.coord_bound;
    }
    c->d_bvh = built.d_nodes;
    c->d_wide = built.d_wide;
    c->use_bvh = n_bvh_nodes > 0;
    // The wide walk's stack never holds more than three pending children per wide node on the way down plus the four
    // of the last: a tree too deep for the stack (LDS part plus PTX_WIDE_SPILL entries of device memory per lane) is
    // walked in its binary form instead.
    int wide_stack_bound = 0;
    if (built.d_wide) {
        wide_stack_bound = 3 * built.wide_depth + 1;
        if (wide_stack_bound > WIDE_STACK_LDS + PTX_WIDE_SPILL) {
            (void)hipFree(c->d_wide);
            c->d_wide = nullptr;
        } else {
            // one region per lane of the largest grid the render kernel is launched with (persistent waves: what the GPU holds)
            const size_t lanes = (size_t)c->n_cus * 4 * PTX_BVH_WAVES * PTX_BLOCK;
            HIPCHK(hipMalloc((void **)&c->d_spill, lanes * PTX_WIDE_SPILL * sizeof(int)));
        }
    }
    HIPCHK(hipMalloc(&c->d_list_geom, list_geom.size() * sizeof(DevGeom)));
    HIPCHK(hipMalloc(&c->d_list_ids, (list_ids.size() ? list_ids.size() : 1) * sizeof(int32_t)));
    HIPCHK(hipMemcpy(c->d_list_geom, list_geom.data(), list_geom.size() * sizeof(DevGeom), hipMemcpyHostToDevice));
    if (!list_ids.empty())
        HIPCHK(hipMemcpy(c->d_list_ids, list_ids.data(), list_ids.size() * sizeof(int32_t), hipMemcpyHostToDevice));
    HIPCHK(hipMalloc(&c->d_class_geom, class_geom.size() * sizeof(DevGeom)));
    HIPCHK(hipMalloc(&c->d_class_ids, class_ids.size() * sizeof(int32_t)));
    HIPCHK(hipMemcpy(c->d_class_geom, class_geom.data(), class_geom.size() * sizeof(DevGeom), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(c->d_class_ids, class_ids.data(), class_ids.size() * sizeof(int32_t), hipMemcpyHostToDevice));
    HIPCHK(hipMalloc(&c->d_geom, geom.size() * sizeof(DevGeom)));
    HIPCHK(hipMalloc(&c->d_mat, mat.size() * sizeof(DevMat)));
    HIPCHK(hipMalloc(&c->d_lights, lights.size() * sizeof(DevLight)));
    HIPCHK(hipMalloc(&c->d_images, images.size() * sizeof(DevImage)));
    HIPCHK(hipMalloc(&c->d_texels, texels.size()));
    HIPCHK(hipMemcpy(c->d_geom, geom.data(), geom.size() * sizeof(DevGeom), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(c->d_mat, mat.data(), mat.size() * sizeof(DevMat), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(c->d_lights, lights.data(), lights.size() * sizeof(DevLight), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(c->d_images, images.data(), images.size() * sizeof(DevImage), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(c->d_texels, texels.data(), texels.size(), hipMemcpyHostToDevice));
    DevScene &d = c->sc;
    memset(&d, 0, sizeof(d));
    d.geom = (const DevGeom *)c->d_geom;
    d.mat = (const DevMat *)c->d_mat;
    d.lights = (const DevLight *)c->d_lights;
    d.images = (const DevImage *)c->d_images;
    d.texels = (const uint8_t *)c->d_texels;
    d.list_geom = (const DevGeom *)c->d_list_geom;
    d.list_ids = (const int32_t *)c->d_list_ids;
    d.class_geom = (const DevGeom *)c->d_class_geom;
    d.class_ids = (const int32_t *)c->d_class_ids;
    for (int cls = 0; cls < LC_N; cls++) d.class_end[cls] = class_end[cls];
    d.bvh = (const PtxBvhNode *)c->d_bvh;
    d.n_list = (int32_t)list_ids.size();
    d.n_bvh_nodes = n_bvh_nodes;
    d.bvh_per_octant = bvh_orders == 8;
    d.bvh_bound = bvh_bound;
    d.wide = (const PtxWideNode *)c->d_wide;
    d.n_wide = c->d_wide ? built.n_wide : 0;
    d.wide_stack_bound = c->d_wide ? wide_stack_bound : 0;
    d.n_objects = n; d.n_lights = s->n_lights; d.n_images = s->n_images; d.has_refraction = has_refr;
    const ptx_camera &cam = s->cam;
    for (int q = 0; q < 3; q++) { d.cam.e[q] = cam.e[q]; d.cam.w[q] = cam.w[q]; }
    affine34(cam.C2W, d.cam.C2W);
    d.cam.f = cam.f; d.cam.wl = cam.wl; d.cam.wt = cam.wt; d.cam.wsize = cam.wsize;
    // fp = C2W (0,0,focus_distance,1), reference PathTracer.c:589-594
    for (int r = 0; r < 3; r++)
        d.cam.fp[r] = (cam.C2W[4 * r + 0] * 0.0) + (cam.C2W[4 * r + 1] * 0.0) + (cam.C2W[4 * r + 2] * cam.focus_distance) +
                      (cam.C2W[4 * r + 3] * 1.0);
    c->n_objects = n;
    c->have_scene = true;
    guard.done = true;
    return PTX_OK;
}

static int fill_render(ptx_context *c, const ptx_render_params *p) {
    if (!c || !p) return fail(PTX_EINVAL, "null argument");
    if (!c->have_scene) return fail(PTX_ESTATE, "no scene uploaded");
    if (p->sx < 2) return fail(PTX_EINVAL, "image size must be at least 2");
    if (p->rng_mode != PTX_RNG_COUNTER) return fail(PTX_EINVAL, "only PTX_RNG_COUNTER runs on the GPU");
    if (p->flags & ~(7 | PTX_CLEAN)) return fail(PTX_EINVAL, "unknown flag bits");
    int n_ranks = p->n_ranks <= 1 ? 1 : p->n_ranks;
    if (n_ranks > 1 && (p->rank < 0 || p->rank >= n_ranks || p->band_rows <= 0))
        return fail(PTX_EINVAL, "bad rank / n_ranks / band_rows");
    if ((p->flags & PTX_USE_ES) && c->sc.n_lights == 0) return fail(PTX_EINVAL, "explicit sampling needs a light");
    c->prm = *p;
    DevRender &r = c->rp;
    r.sx = p->sx; r.max_depth = p->max_depth;
    r.rank = n_ranks > 1 ? p->rank : 0; r.n_ranks = n_ranks; r.band_rows = n_ranks > 1 ? p->band_rows : p->sx;
    r.rows_owned = ptx_rows_owned(p->sx, r.rank, n_ranks, r.band_rows);
    r.aperture = p->aperture; r.seed = p->seed;
    r.clean = (p->flags & PTX_CLEAN) ? 1 : 0;
    r.pad = 0;
    r.du = c->sc.cam.wsize / (p->sx - 1);   // reference PathTracer.c:583-584
    r.dv = -c->sc.cam.wsize / (p->sx - 1);
    return PTX_OK;
}

extern "C" int ptx_render_begin(ptx_context *c, const ptx_render_params *p) {
    int rc = fill_render(c, p);
    if (rc) return rc;
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(hipStreamSynchronize(c->stream));
    free_render(c);
    size_t npix = (size_t)c->rp.rows_owned * c->rp.sx;
    if (npix == 0) npix = 1;
    HIPCHK(hipMalloc((void **)&c->d_rgb, npix * 3 * sizeof(double)));
    HIPCHK(hipMalloc((void **)&c->d_wght, npix * sizeof(double)));
    c->own_accum = true;
    HIPCHK(hipMemsetAsync(c->d_rgb, 0, npix * 3 * sizeof(double), c->stream));
    hipLaunchKernelGGL(k_fill, dim3((unsigned)((npix + 255) / 256)), dim3(256), 0, c->stream, c->d_wght, npix, 1.0);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemsetAsync(c->d_stats, 0, PTX_STATS_WALK * sizeof(unsigned long long), c->stream));
    c->rendering = true;
    c->have_timing = false;
    return PTX_OK;
}

extern "C" int ptx_render_begin_external(ptx_context *c, const ptx_render_params *p, double *d_rgb, double *d_wght) {
    if (!d_rgb || !d_wght) return fail(PTX_EINVAL, "null accumulator");
    int rc = fill_render(c, p);
    if (rc) return rc;
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(hipStreamSynchronize(c->stream));
    free_render(c);
    c->d_rgb = d_rgb;
    c->d_wght = d_wght;
    c->own_accum = false;
    HIPCHK(hipMemsetAsync(c->d_stats, 0, PTX_STATS_WALK * sizeof(unsigned long long), c->stream));
    c->rendering = true;
    c->have_timing = false;
    return PTX_OK;
}

template <typename F>
static void dispatch_flags_1(int flags, F &&f) {
    switch (flags & 7) {
#ifndef PTX_ONLY_FLAGS7  // (experimental builds, tools/build_variants.sh: only the IS+ES+RR kernels, a quarter of the compile time)
    case 0: f(std::integral_constant<int, 0>()); break;
    case 1: f(std::integral_constant<int, 1>()); break;
    case 2: f(std::integral_constant<int, 2>()); break;
    case 3: f(std::integral_constant<int, 3>()); break;
    case 4: f(std::integral_constant<int, 4>()); break;
    case 5: f(std::integral_constant<int, 5>()); break;
    case 6: f(std::integral_constant<int, 6>()); break;
#endif
    default: f(std::integral_constant<int, 7>()); break;
    }
}

// calls f(flags constant, hierarchy constant)
template <typename F>
static void dispatch_flags(int flags, bool use_bvh, F &&f) {
    dispatch_flags_1(flags, [&](auto fl) {
        if (use_bvh) f(fl, std::true_type());
        else f(fl, std::false_type());
    });
}

// Largest sample buffer ptx_render_passes allocates: 32 GiB (PTX_SAMPLE_BUFFER_MB overrides), and never more than
// half of what the device has free.  A call with more samples than fit is carried out in several rounds.
static size_t sample_buffer_cap(size_t already_ours) {
    size_t cap = (size_t)32 << 30;
    if (const char *e = getenv("PTX_SAMPLE_BUFFER_MB")) {
        const long long mb = atoll(e);
        if (mb > 0) cap = (size_t)mb << 20;
    }
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) == hipSuccess) cap = std::min(cap, (free_b + already_ours) / 2);
    return cap;
}

// Passes k0 .. k0+nk-1 in rounds of trace + fold through the sample buffer `buf` of `buf_bytes` bytes.
// per_launch_timing: bracket every launch of the tracing kernel with events (created on demand).
static int render_rounds(ptx_context *c, int k0, int nk, double *buf, size_t buf_bytes, bool per_launch_timing) {
    const int tiles_x = (c->rp.sx + 7) / 8, tiles_y = (c->rp.rows_owned + 7) / 8;
    const long n_tiles = (long)tiles_x * tiles_y;
    const size_t n_pix = (size_t)c->rp.rows_owned * c->rp.sx;
    const size_t bytes_per_pass = n_pix * PTX_SAMPLE_DOUBLES * sizeof(double);
    // samples per round: what the buffer holds; a pixel's records start on a 128-byte line, so the stride is a
    // multiple of four records
    const int round_max = (int)std::min<size_t>(((size_t)nk + 3) & ~(size_t)3, (buf_bytes / bytes_per_pass) & ~(size_t)3);
    if (round_max < 4) return fail(PTX_EINVAL, "sample buffer too small for four passes of this image");
    bool pinhole = c->rp.aperture == 0 && !c->rp.clean;  // every sample of a pixel starts along the same direction
    if (getenv("PTX_NO_PINHOLE")) pinhole = false;          // (the general variant on a zero aperture: for comparisons)
    const long slots = (long)c->n_cus * 4 * (c->use_bvh ? PTX_BVH_WAVES : (!pinhole ? 3 : 4));  // waves the GPU holds at once
    // Scenes with a hierarchy: the kernel that carries queries in slices (PTX_BVH_KERNEL=whole selects the kernel
    // that runs whole queries, for comparisons; the images are the same).  Its settings were swept at 2048 x 2048 on
    // SciFiCube and BubbleTea (8000 and 6300 primitives in the hierarchy: node arrays of 4 MB, served by the L2 caches,
    // the walk a quarter of the wave's time) and on the mesh scene of a million triangles (512 MB of nodes: a step
    // waits for device memory and the walk is two thirds of the time).  What the second kind wants is many lanes
    // walking at once, so it leaves the walk early to refill it (exit_lanes 24), shades smaller groups and takes
    // short bursts: 1.43 times faster on that scene than with the first kind's settings, which in turn are 2 %
    // better on the small scenes.
    bool sliced = true;
    if (const char *e = getenv("PTX_BVH_KERNEL")) sliced = strcmp(e, "whole") != 0;
    // Waves wanted per slot: enough of them that the GPU stays full while waves finish at different times, and short
    // enough runs that a slow tile does not hold the end of the launch.  k_trace_samples: 6, with the last sample range
    // cut into four (below) -- a wave's start and its wait for its slowest pixel cost the same whatever its length, so
    // long waves with a fine tail are best; with all ranges alike it was 16 to 32.  That matters most where a GPU holds a
    // share of an image: rank 0's rows of an 8-rank split of the benchmark take 25.2 ms instead of 26.2 ms, the whole
    // image 195.8 ms either way.  The kernel for hierarchy scenes hands its work tiles out itself: 16.
    long fill = (c->use_bvh && sliced) ? 16 : 6;
    if (const char *e = getenv("PTX_WAVES_PER_SLOT")) fill = std::max(1L, atol(e));
    const size_t node_bytes = (size_t)c->sc.n_bvh_nodes * sizeof(PtxBvhNode) * (c->sc.bvh_per_octant ? 8 : 1);
    const bool nodes_in_l2 = node_bytes <= ((size_t)32 << 20);  // eight L2 caches of 4 MB
    bool wide = c->sc.wide != nullptr && sliced;                 // the four-wide walk (built for large node arrays only)
    // entries of its stack kept in LDS (PTX_WIDE_LDS_ENTRIES lowers it, for tests of the part in device memory)
    int stack_lds = WIDE_STACK_LDS;
    if (const char *e = getenv("PTX_WIDE_LDS_ENTRIES")) stack_lds = std::max(4, std::min(WIDE_STACK_LDS, atoi(e)));
    if (wide && c->sc.wide_stack_bound > stack_lds + PTX_WIDE_SPILL) wide = false;
    int exit_lanes = nodes_in_l2 ? 12 : 24, leaf_batch = nodes_in_l2 ? 4 : 8, shade_min = nodes_in_l2 ? 40 : 24,
        burst = nodes_in_l2 ? 6 : 4, small_run = 8;
    // the four-wide walk: a step is four box tests and a stack operation, worth two or three steps of the binary walk, and a
    // lane needs four or five of them for a query; short bursts, and the walk is left early (swept on the mesh of a
    // million triangles, tools/sweep_walk.py: bursts of 2, 40 lanes, leaves tested eight at a time; flat within 1 %
    // between 32 and 48 lanes and between 8 and 16 leaves)
    if (c->sc.wide != nullptr) {
        exit_lanes = 40;
        leaf_batch = 8;
        shade_min = 24;
        burst = 2;
    }
    if (const char *e = getenv("PTX_WALK_EXIT_LANES")) exit_lanes = std::max(0, std::min(63, atoi(e)));
    if (const char *e = getenv("PTX_WALK_LEAF_BATCH")) leaf_batch = std::max(1, std::min(64, atoi(e)));
    if (const char *e = getenv("PTX_WALK_SHADE_MIN")) shade_min = std::max(1, std::min(64, atoi(e)));
    if (const char *e = getenv("PTX_WALK_BURST")) burst = std::max(1, atoi(e));
    if (const char *e = getenv("PTX_WALK_SMALL_RUN")) small_run = std::max(1, std::min(64, atoi(e)));
    int lds_nodes = 0;
    if (const char *e = getenv("PTX_WALK_LDS_NODES")) lds_nodes = std::max(0, std::min(PTX_LDS_TOP_MAX, atoi(e)));
    c->n_trace_launches = 0;
    // (no events on the path that must stay capturable into a HIP graph: the timing getters then report PTX_ESTATE)
    c->have_timing = false;
    if (per_launch_timing) HIPCHK(hipEventRecord(c->ev0, c->stream));
    for (int ka = k0; ka < k0 + nk; ka += round_max) {
        const int nka = std::min(round_max, k0 + nk - ka);
        // a wave traces sub_len samples of a tile: as long a run as leaves enough waves, and at least 8
        long n_sub = std::max(1L, std::min((fill * slots + n_tiles - 1) / n_tiles, (long)nka / 8));
        const int sub_len = (int)(((nka + n_sub - 1) / n_sub + 3) & ~3L);
        n_sub = (nka + sub_len - 1) / sub_len;
        // k_trace_samples cuts the samples differently.  The lanes of a wave wait for the one whose paths were longest, and
        // what they wait is the spread of a sum of path lengths, which grows with the root of the number of samples: 8 % of
        // a wave of 125 samples, 4 % of one of 500 (measured: 60 of 64 lanes alive with [500, 125 x 4]).  So the waves
        // should be as long as possible -- but blocks start in the order of their index, and the GPU empties as slowly as
        // its last waves are long.  Hence a head of long ranges, as many as fill the GPU head_rounds times, over
        // head_permille of the samples, and the rest in tail_split short ranges that level the end of the launch.  Swept on
        // the benchmark (1024^2, 1000 samples) and on one GPU's share of a 2-, 4- and 8-GPU split: 4 rounds, 875, 4 pieces
        // (800 to 920 within 0.5 %; beyond 930 the tail no longer covers the spread of the last long waves).
        // PTX_HEAD_PERMILLE=0: ranges of equal length with the last one in tail_split pieces; PTX_TAIL_SPLIT=1: all alike.
        int n_full = (int)n_sub, tail_len = sub_len;
        long tail_split = 4;
        if (const char *e = getenv("PTX_TAIL_SPLIT")) tail_split = std::max(1L, atol(e));
        int sub_len_k = sub_len;
        long head_permille = 875, head_rounds = 4;
        if (const char *e = getenv("PTX_HEAD_PERMILLE")) head_permille = std::max(0L, std::min(990L, atol(e)));
        if (const char *e = getenv("PTX_HEAD_ROUNDS")) head_rounds = std::max(1L, atol(e));
        const long n_head = std::max(1L, (head_rounds * slots + n_tiles - 1) / n_tiles);
        const long head_len = ((nka * head_permille / 1000 + n_head - 1) / n_head + 3) & ~3L;
        // (not where the head's ranges would be shorter than 32 samples: few samples on a small image)
        if (!(c->use_bvh && sliced) && head_permille > 0 && tail_split > 1 && head_len >= 32) {
            sub_len_k = (int)head_len;
            n_full = (int)std::min(n_head, (long)nka / sub_len_k);
            const long rest = nka - (long)n_full * sub_len_k;
            if (rest > 0) {  // (even a tail of four samples is better than none: 50-pass launches 10.2 against 10.8 ms)
                const long pieces = std::max(1L, std::min(tail_split, rest / 8));  // none shorter than 8 samples
                tail_len = (int)(((rest + pieces - 1) / pieces + 3) & ~3L);
                n_sub = n_full + (rest + tail_len - 1) / tail_len;
            } else {
                tail_len = sub_len_k;
                n_sub = n_full;
            }
        } else if (!(c->use_bvh && sliced) && tail_split > 1 && n_sub >= 2) {
            n_full = (int)n_sub - 1;
            const long rest = nka - (long)n_full * sub_len;
            const long pieces = std::max(1L, std::min(tail_split, rest / 8));  // none shorter than 8 samples: a wave's start
            tail_len = (int)(((rest + pieces - 1) / pieces + 3) & ~3L);       // costs as much as a few samples
            n_sub = n_full + (rest + tail_len - 1) / tail_len;
        }
        if (n_tiles * n_sub > 0x7fffffffL) return fail(PTX_EINVAL, "ptx_render_passes: image too large for one launch");
        if (per_launch_timing) {
            while ((int)c->trace_ev.size() < 2 * (c->n_trace_launches + 1)) {
                hipEvent_t e;
                HIPCHK(hipEventCreate(&e));
                c->trace_ev.push_back(e);
            }
            HIPCHK(hipEventRecord(c->trace_ev[2 * c->n_trace_launches], c->stream));
        }
        unsigned int *work_counter = (unsigned int *)(c->d_stats + PTX_STATS_WORK);
        if (c->use_bvh && sliced) HIPCHK(hipMemsetAsync(work_counter, 0, sizeof(unsigned long long), c->stream));
        dispatch_flags(c->prm.flags, c->use_bvh, [&](auto fl, auto bv) {
            constexpr int F = decltype(fl)::value;
            constexpr bool B = decltype(bv)::value;
            const dim3 grid((unsigned)(n_tiles * n_sub));
            // persistent waves: as many as the GPU holds at once, or as there are work tiles
            const dim3 pgrid((unsigned)std::min<long>(n_tiles * n_sub, slots));
            if (B && sliced) {
                const TraceBvhArgs a{c->sc, c->rp, ka, ka + nka, sub_len, (int)n_tiles, (int)(n_tiles * n_sub), (size_t)round_max, buf, c->d_stats,
                                     work_counter, exit_lanes, leaf_batch, shade_min, burst, small_run, lds_nodes, c->d_spill, stack_lds};
                if (wide) ptx_launch_trace_bvh_wide(F, pinhole, pgrid, c->stream, a);
                else if (pinhole) launch_trace_bvh<F, true, false>(pgrid, c->stream, a);
                else launch_trace_bvh<F, false, false>(pgrid, c->stream, a);
            } else if (pinhole)
                hipLaunchKernelGGL((k_trace_samples<F, B, true>), grid, dim3(PTX_BLOCK), 0, c->stream, c->sc, c->rp, ka, ka + nka,
                                   sub_len_k, n_full, tail_len, (int)n_tiles, (size_t)round_max, buf, c->d_stats);
            else
                hipLaunchKernelGGL((k_trace_samples<F, B, false>), grid, dim3(PTX_BLOCK), 0, c->stream, c->sc, c->rp, ka, ka + nka,
                                   sub_len_k, n_full, tail_len, (int)n_tiles, (size_t)round_max, buf, c->d_stats);
        });
        HIPCHK(hipGetLastError());
        if (per_launch_timing) {
            HIPCHK(hipEventRecord(c->trace_ev[2 * c->n_trace_launches + 1], c->stream));
            c->n_trace_launches++;
        }
        hipLaunchKernelGGL(k_fold_samples, dim3((unsigned)((n_pix + 255) / 256)), dim3(256), 0, c->stream, buf, n_pix, (size_t)round_max, nka,
                           c->d_rgb, c->d_wght, c->rp.clean);
        HIPCHK(hipGetLastError());
    }
    if (per_launch_timing) {
        HIPCHK(hipEventRecord(c->ev1, c->stream));
        c->have_timing = true;
    }
    return PTX_OK;
}

extern "C" int ptx_render_passes(ptx_context *c, int k0, int nk) {
    if (!c) return fail(PTX_EINVAL, "null context");
    if (!c->rendering) return fail(PTX_ESTATE, "ptx_render_begin has not been called");
    if (k0 < 0 || nk < 0) return fail(PTX_EINVAL, "negative pass range");
    if (nk == 0 || c->rp.rows_owned == 0) return PTX_OK;
    HIPCHK(hipSetDevice(c->device));
    const size_t n_pix = (size_t)c->rp.rows_owned * c->rp.sx;
    const size_t bytes_per_pass = n_pix * PTX_SAMPLE_DOUBLES * sizeof(double);
    const size_t cap = sample_buffer_cap(c->samples_bytes);
    const size_t round_max = std::min<size_t>(((size_t)nk + 3) & ~(size_t)3, (cap / bytes_per_pass) & ~(size_t)3);
    if (round_max < 4)
        return fail(PTX_ENOMEM, "ptx_render_passes: the sample buffer cap (PTX_SAMPLE_BUFFER_MB, or half the free device memory) cannot "
                                "hold four passes of this rank's " + std::to_string(n_pix) + " pixels (" +
                                std::to_string(4 * bytes_per_pass >> 20) + " MiB needed)");
    if (round_max * bytes_per_pass > c->samples_bytes) {
        HIPCHK(hipStreamSynchronize(c->stream));
        (void)hipFree(c->d_samples);
        c->d_samples = nullptr;
        c->samples_bytes = 0;
        HIPCHK(hipMalloc((void **)&c->d_samples, round_max * bytes_per_pass));
        c->samples_bytes = round_max * bytes_per_pass;
    }
    return render_rounds(c, k0, nk, c->d_samples, round_max * bytes_per_pass, true);  // the cap, not what an earlier call left
}

extern "C" int ptx_render_passes_static(ptx_context *c, int k0, int nk) {
    if (!c) return fail(PTX_EINVAL, "null context");
    if (!c->rendering) return fail(PTX_ESTATE, "ptx_render_begin has not been called");
    if (k0 < 0 || nk < 0) return fail(PTX_EINVAL, "negative pass range");
    if (nk == 0 || c->rp.rows_owned == 0) return PTX_OK;
    const size_t n_pix = (size_t)c->rp.rows_owned * c->rp.sx;
    if (n_pix > PTX_STATIC_SAMPLE_BYTES / (4 * PTX_SAMPLE_DOUBLES * sizeof(double)))
        return fail(PTX_EINVAL, "ptx_render_passes_static: more than 2097152 pixels on this rank (the fixed 256 MiB sample buffer "
                                "holds four passes of at most that many); use ptx_render_passes or more ranks");
    HIPCHK(hipSetDevice(c->device));
    return render_rounds(c, k0, nk, c->d_static_samples, PTX_STATIC_SAMPLE_BYTES, false);
}

extern "C" size_t ptx_static_max_pixels(void) { return PTX_STATIC_SAMPLE_BYTES / (4 * PTX_SAMPLE_DOUBLES * sizeof(double)); }

extern "C" int ptx_sync(ptx_context *c) {
    if (!c) return fail(PTX_EINVAL, "null context");
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(hipStreamSynchronize(c->stream));
    return PTX_OK;
}

extern "C" int ptx_last_pass_ms(ptx_context *c, float *ms) {
    if (!c || !ms) return fail(PTX_EINVAL, "null argument");
    if (!c->have_timing) return fail(PTX_ESTATE, "no pass has been launched");
    HIPCHK(hipEventSynchronize(c->ev1));
    HIPCHK(hipEventElapsedTime(ms, c->ev0, c->ev1));
    return PTX_OK;
}

extern "C" int ptx_last_trace_ms(ptx_context *c, float *ms, int *launches) {
    if (!c || !ms) return fail(PTX_EINVAL, "null argument");
    if (!c->have_timing) return fail(PTX_ESTATE, "no pass has been launched");
    HIPCHK(hipEventSynchronize(c->ev1));
    float total = 0;
    for (int q = 0; q < c->n_trace_launches; q++) {
        float t = 0;
        HIPCHK(hipEventElapsedTime(&t, c->trace_ev[2 * q], c->trace_ev[2 * q + 1]));
        total += t;
    }
    *ms = total;
    if (launches) *launches = c->n_trace_launches;
    return PTX_OK;
}

extern "C" int ptx_tonemap_device(ptx_context *c, const double *d_im, uint64_t n_values, uint8_t *bits, ptx_tonemap_info *info) {
    if (!c || !d_im || !bits) return fail(PTX_EINVAL, "ptx_tonemap_device: null argument");
    if (n_values == 0) return fail(PTX_EINVAL, "ptx_tonemap_device: empty image");
    HIPCHK(hipSetDevice(c->device));
    std::string err;
    int rc = ptx_tonemap_run(c->stream, c->tonemap, d_im, (size_t)n_values, bits, info, err);
    return rc ? fail(rc, err) : PTX_OK;
}

extern "C" int ptx_tonemap(ptx_context *c, uint8_t *bits, ptx_tonemap_info *info) {
    if (!c) return fail(PTX_EINVAL, "null context");
    if (!c->rendering) return fail(PTX_ESTATE, "ptx_render_begin has not been called");
    if (c->rp.rows_owned != c->rp.sx) return fail(PTX_ESTATE, "ptx_tonemap: this rank holds only part of the image");
    return ptx_tonemap_device(c, c->d_rgb, (uint64_t)c->rp.sx * c->rp.sx * 3, bits, info);
}

extern "C" int ptx_selftest(ptx_context *c, double *results) {
    if (!c || !results) return fail(PTX_EINVAL, "null argument");
    HIPCHK(hipSetDevice(c->device));
    const int blocks = 64;
    double *d_out = nullptr;
    HIPCHK(hipMalloc((void **)&d_out, blocks * 5 * sizeof(double)));
    hipLaunchKernelGGL(k_selftest, dim3(blocks), dim3(256), 0, c->stream, d_out);
    std::vector<double> h(blocks * 5);
    int rc = PTX_OK;
    if (hipGetLastError() != hipSuccess || hipMemcpyAsync(h.data(), d_out, h.size() * sizeof(double), hipMemcpyDeviceToHost, c->stream) != hipSuccess ||
        hipStreamSynchronize(c->stream) != hipSuccess)
        rc = fail(PTX_EDEVICE, "ptx_selftest: kernel failed");
    (void)hipFree(d_out);
    if (rc) return rc;
    for (int q = 0; q < 5; q++) results[q] = 0;
    for (int b = 0; b < blocks; b++) {
        results[0] = h[b * 5];  // the same exhaustive count in every block
        for (int q = 1; q < 5; q++) results[q] = std::max(results[q], h[b * 5 + q]);
    }
    return PTX_OK;
}

extern "C" int ptx_accumulators(ptx_context *c, double **d_rgb, double **d_wght, int32_t *rows) {
    if (!c) return fail(PTX_EINVAL, "null context");
    if (!c->rendering) return fail(PTX_ESTATE, "ptx_render_begin has not been called");
    if (d_rgb) *d_rgb = c->d_rgb;
    if (d_wght) *d_wght = c->d_wght;
    if (rows) *rows = c->rp.rows_owned;
    return PTX_OK;
}

extern "C" int ptx_download(ptx_context *c, double *rgbIm, double *wght) {
    if (!c) return fail(PTX_EINVAL, "null context");
    if (!c->rendering) return fail(PTX_ESTATE, "ptx_render_begin has not been called");
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(hipStreamSynchronize(c->stream));
    const DevRender &r = c->rp;
    const size_t sx = r.sx;
    // bands of consecutive local rows map to consecutive global rows
    int local = 0;
    while (local < r.rows_owned) {
        int g0 = ptx_global_row(local, r.rank, r.n_ranks, r.band_rows);
        int run = r.n_ranks <= 1 ? r.rows_owned : r.band_rows - local % r.band_rows;
        if (run > r.rows_owned - local) run = r.rows_owned - local;
        if (rgbIm)
            HIPCHK(hipMemcpy(rgbIm + (size_t)g0 * sx * 3, c->d_rgb + (size_t)local * sx * 3, (size_t)run * sx * 3 * sizeof(double),
                             hipMemcpyDeviceToHost));
        if (wght)
            HIPCHK(hipMemcpy(wght + (size_t)g0 * sx, c->d_wght + (size_t)local * sx, (size_t)run * sx * sizeof(double),
                             hipMemcpyDeviceToHost));
        local += run;
    }
    return PTX_OK;
}

extern "C" int ptx_device_count(int *n) {
    if (!n) return fail(PTX_EINVAL, "null argument");
    HIPCHK(hipGetDeviceCount(n));
    return PTX_OK;
}

extern "C" int ptx_gather(ptx_context *const *ranks, int n_ranks, double **d_rgb, double **d_wght) {
    if (!ranks || n_ranks < 1 || !ranks[0]) return fail(PTX_EINVAL, "ptx_gather: no contexts");
    ptx_context *dst = ranks[0];
    for (int r = 0; r < n_ranks; r++) {
        const ptx_context *c = ranks[r];
        if (!c || !c->rendering) return fail(PTX_ESTATE, "ptx_gather: a context has no render in progress");
        if (c->rp.sx != dst->rp.sx || c->rp.n_ranks != n_ranks || c->rp.rank != r || c->rp.band_rows != dst->rp.band_rows)
            return fail(PTX_EINVAL, "ptx_gather: context r must render rank r of n_ranks of the same image");
    }
    const size_t sx = dst->rp.sx, n_pix = sx * sx;
    HIPCHK(hipSetDevice(dst->device));
    if (dst->whole_pixels < n_pix) {
        (void)hipFree(dst->d_whole_rgb);
        (void)hipFree(dst->d_whole_wght);
        dst->d_whole_rgb = dst->d_whole_wght = nullptr;
        dst->whole_pixels = 0;
        HIPCHK(hipMalloc((void **)&dst->d_whole_rgb, n_pix * 3 * sizeof(double)));
        HIPCHK(hipMalloc((void **)&dst->d_whole_wght, n_pix * sizeof(double)));
        dst->whole_pixels = n_pix;
    }
    for (int r = 0; r < n_ranks; r++) {
        ptx_context *c = ranks[r];
        HIPCHK(hipSetDevice(c->device));
        HIPCHK(hipStreamSynchronize(c->stream));  // the rows must be final before they are copied
        HIPCHK(hipSetDevice(dst->device));
        const DevRender &rp = c->rp;
        int local = 0;
        while (local < rp.rows_owned) {  // bands of consecutive local rows are consecutive rows of the image
            const int g0 = ptx_global_row(local, rp.rank, rp.n_ranks, rp.band_rows);
            int run = rp.n_ranks <= 1 ? rp.rows_owned : rp.band_rows - local % rp.band_rows;
            if (run > rp.rows_owned - local) run = rp.rows_owned - local;
            HIPCHK(hipMemcpyPeerAsync(dst->d_whole_rgb + (size_t)g0 * sx * 3, dst->device, c->d_rgb + (size_t)local * sx * 3, c->device,
                                      (size_t)run * sx * 3 * sizeof(double), dst->stream));
            HIPCHK(hipMemcpyPeerAsync(dst->d_whole_wght + (size_t)g0 * sx, dst->device, c->d_wght + (size_t)local * sx, c->device,
                                      (size_t)run * sx * sizeof(double), dst->stream));
            local += run;
        }
    }
    HIPCHK(hipStreamSynchronize(dst->stream));
    if (d_rgb) *d_rgb = dst->d_whole_rgb;
    if (d_wght) *d_wght = dst->d_whole_wght;
    return PTX_OK;
}

extern "C" int ptx_upload(ptx_context *c, const double *rgbIm, const double *wght) {
    if (!c) return fail(PTX_EINVAL, "null context");
    if (!c->rendering) return fail(PTX_ESTATE, "ptx_render_begin has not been called");
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(hipStreamSynchronize(c->stream));
    const DevRender &r = c->rp;
    const size_t sx = r.sx;
    int local = 0;
    while (local < r.rows_owned) {  // the row bands of ptx_download, in the other direction
        int g0 = ptx_global_row(local, r.rank, r.n_ranks, r.band_rows);
        int run = r.n_ranks <= 1 ? r.rows_owned : r.band_rows - local % r.band_rows;
        if (run > r.rows_owned - local) run = r.rows_owned - local;
        if (rgbIm)
            HIPCHK(hipMemcpy(c->d_rgb + (size_t)local * sx * 3, rgbIm + (size_t)g0 * sx * 3, (size_t)run * sx * 3 * sizeof(double),
                             hipMemcpyHostToDevice));
        if (wght)
            HIPCHK(hipMemcpy(c->d_wght + (size_t)local * sx, wght + (size_t)g0 * sx, (size_t)run * sx * sizeof(double),
                             hipMemcpyHostToDevice));
        local += run;
    }
    return PTX_OK;
}

extern "C" int ptx_get_stats(ptx_context *c, ptx_stats *st) {
    if (!c || !st) return fail(PTX_EINVAL, "null argument");
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(hipStreamSynchronize(c->stream));
    unsigned long long h[4];
    HIPCHK(hipMemcpy(h, c->d_stats, sizeof(h), hipMemcpyDeviceToHost));
    st->paths = h[0]; st->closest_hits = h[1]; st->shadow_rays = h[2]; st->rindex_oob = h[3];
    return PTX_OK;
}

extern "C" int ptx_debug_counters(ptx_context *c, unsigned long long *out, int clear) {
    if (!c || !out) return fail(PTX_EINVAL, "null argument");
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(hipStreamSynchronize(c->stream));
    HIPCHK(hipMemcpy(out, c->d_stats + PTX_STATS_DBG, PTX_N_DBG * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    if (clear) HIPCHK(hipMemset(c->d_stats + PTX_STATS_DBG, 0, PTX_N_DBG * sizeof(unsigned long long)));
    return PTX_OK;
}

extern "C" int ptx_last_walk_stats(ptx_context *c, ptx_walk_stats *ws) {
    if (!c || !ws) return fail(PTX_EINVAL, "null argument");
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(hipStreamSynchronize(c->stream));
    unsigned long long h[PTX_N_WALK];
    HIPCHK(hipMemcpy(h, c->d_stats + PTX_STATS_WALK, sizeof(h), hipMemcpyDeviceToHost));
    ws->queries = h[0]; ws->hits = h[1]; ws->list_tests = h[2]; ws->node_visits = h[3];
    for (int q = 0; q < 6; q++) ws->leaf_tests[q] = h[4 + q];
    ws->texel_lookups = h[10];
    return PTX_OK;
}

extern "C" int ptx_first_hit_device(ptx_context *c, const ptx_ray *d_rays, int64_t n, ptx_hit *d_hits) {
    if (!c || (n > 0 && (!d_rays || !d_hits))) return fail(PTX_EINVAL, "null argument");
    if (!c->have_scene) return fail(PTX_ESTATE, "no scene uploaded");
    if (n < 0) return fail(PTX_EINVAL, "negative ray count");
    if (n == 0) return PTX_OK;
    HIPCHK(hipSetDevice(c->device));
    if (c->use_bvh)
        hipLaunchKernelGGL(k_first_hit<true>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream, c->sc, d_rays,
                           (long)n, d_hits);
    else
        hipLaunchKernelGGL(k_first_hit<false>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream, c->sc, d_rays,
                           (long)n, d_hits);
    HIPCHK(hipGetLastError());
    return PTX_OK;
}

extern "C" int ptx_first_hit(ptx_context *c, const ptx_ray *rays, int64_t n, ptx_hit *hits) {
    if (!c || (n > 0 && (!rays || !hits))) return fail(PTX_EINVAL, "null argument");
    if (!c->have_scene) return fail(PTX_ESTATE, "no scene uploaded");
    if (n < 0) return fail(PTX_EINVAL, "negative ray count");
    if (n == 0) return PTX_OK;
    HIPCHK(hipSetDevice(c->device));
    ptx_ray *d_rays = nullptr;
    ptx_hit *d_hits = nullptr;
    HIPCHK(hipMalloc((void **)&d_rays, (size_t)n * sizeof(ptx_ray)));
    if (hipMalloc((void **)&d_hits, (size_t)n * sizeof(ptx_hit)) != hipSuccess) {
        (void)hipFree(d_rays);
        return fail(PTX_ENOMEM, "ptx_first_hit: out of device memory");
    }
    int rc = PTX_OK;
    if (hipMemcpyAsync(d_rays, rays, (size_t)n * sizeof(ptx_ray), hipMemcpyHostToDevice, c->stream) != hipSuccess)
        rc = fail(PTX_EDEVICE, "ptx_first_hit: copy to device failed");
    if (!rc) rc = ptx_first_hit_device(c, d_rays, n, d_hits);
    if (!rc && hipMemcpyAsync(hits, d_hits, (size_t)n * sizeof(ptx_hit), hipMemcpyDeviceToHost, c->stream) != hipSuccess)
        rc = fail(PTX_EDEVICE, "ptx_first_hit: copy from device failed");
    if (hipStreamSynchronize(c->stream) != hipSuccess && !rc) rc = fail(PTX_EDEVICE, "ptx_first_hit: kernel failed");
    (void)hipFree(d_rays);
    (void)hipFree(d_hits);
    return rc;
}

extern "C" int ptx_trace_paths(ptx_context *c, const ptx_render_params *p, const int32_t *pi, const int32_t *pj,
                               const int32_t *pk, int64_t n, double *out) {
    if (!c || !p || (n > 0 && (!pi || !pj || !pk || !out))) return fail(PTX_EINVAL, "null argument");
    if (n < 0) return fail(PTX_EINVAL, "negative path count");
    ptx_render_params saved_prm = c->prm;
    DevRender saved_rp = c->rp;
    ptx_render_params whole = *p;
    whole.n_ranks = 1;
    whole.rank = 0;
    int rc = fill_render(c, &whole);
    DevRender rp = c->rp;
    c->prm = saved_prm;
    c->rp = saved_rp;
    if (rc) return rc;
    if (n == 0) return PTX_OK;
    for (int64_t q = 0; q < n; q++)
        if (pi[q] < 0 || pi[q] >= p->sx || pj[q] < 0 || pj[q] >= p->sx || pk[q] < 0)
            return fail(PTX_EINVAL, "ptx_trace_paths: pixel or sample out of range");
    HIPCHK(hipSetDevice(c->device));
    int32_t *d_idx = nullptr;
    double *d_out = nullptr;
    HIPCHK(hipMalloc((void **)&d_idx, (size_t)n * 3 * sizeof(int32_t)));
    if (hipMalloc((void **)&d_out, (size_t)n * 3 * sizeof(double)) != hipSuccess) {
        (void)hipFree(d_idx);
        return fail(PTX_ENOMEM, "ptx_trace_paths: out of device memory");
    }
    rc = PTX_OK;
    if (hipMemsetAsync(c->d_stats + PTX_STATS_WALK, 0, PTX_N_WALK * sizeof(unsigned long long), c->stream) != hipSuccess)
        rc = fail(PTX_EDEVICE, "ptx_trace_paths: clearing the counters failed");
    if (!rc && (hipMemcpyAsync(d_idx, pi, (size_t)n * 4, hipMemcpyHostToDevice, c->stream) != hipSuccess ||
        hipMemcpyAsync(d_idx + n, pj, (size_t)n * 4, hipMemcpyHostToDevice, c->stream) != hipSuccess ||
        hipMemcpyAsync(d_idx + 2 * n, pk, (size_t)n * 4, hipMemcpyHostToDevice, c->stream) != hipSuccess))
        rc = fail(PTX_EDEVICE, "ptx_trace_paths: copy to device failed");
    if (!rc) {
        dispatch_flags(p->flags, c->use_bvh, [&](auto fl, auto bv) {
            constexpr int F = decltype(fl)::value;
            constexpr bool B = decltype(bv)::value;
            hipLaunchKernelGGL((k_trace_paths<F, B>), dim3((unsigned)((n + PTX_BLOCK - 1) / PTX_BLOCK)), dim3(PTX_BLOCK), 0, c->stream, c->sc, rp, d_idx,
                               d_idx + n, d_idx + 2 * n, (long)n, d_out, c->d_stats + PTX_STATS_WALK);
        });
        if (hipGetLastError() != hipSuccess) rc = fail(PTX_EDEVICE, "ptx_trace_paths: launch failed");
    }
    if (!rc && hipMemcpyAsync(out, d_out, (size_t)n * 3 * sizeof(double), hipMemcpyDeviceToHost, c->stream) != hipSuccess)
        rc = fail(PTX_EDEVICE, "ptx_trace_paths: copy from device failed");
    if (hipStreamSynchronize(c->stream) != hipSuccess && !rc) rc = fail(PTX_EDEVICE, "ptx_trace_paths: kernel failed");
    (void)hipFree(d_idx);
    (void)hipFree(d_out);
    return rc;
}
