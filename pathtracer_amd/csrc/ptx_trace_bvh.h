// ptx_trace_bvh.h -- the tracing kernel for scenes with a hierarchy (included by ptx_hip.hip, which instantiates it for
// the binary hierarchy, and by ptx_wide.hip, which instantiates it for the four-wide one).
#pragma once

#include "ptx_device.h"
#include "ptx_internal.h"

#ifndef PTX_SAMPLE_DOUBLES
#define PTX_SAMPLE_DOUBLES 4  // doubles per record of the sample buffer: r, g, b, 0
#endif

#ifndef PTX_BVH_WAVES
#define PTX_BVH_WAVES 3
#endif
// The tracing kernel for scenes with a hierarchy.  Same tiles, sample ranges and sample buffer as k_trace_samples,
// but a lane does not run whole closest-hit queries: it carries one query (of a path segment or of a shadow ray)
// at a time through the walk (query_begin / query_step / query_leaf), and the loop alternates between
//   walking  -- every lane with an unfinished query visits hierarchy nodes, until at most exit_lanes lanes are
//               still walking and there is something to shade; a ray that needs thirty steps no longer keeps
//               sixty-three rays that needed five waiting, it simply goes on in the next walking phase;
//   shading  -- the lanes whose query has finished take its result.  Resolving a shadow ray (apply_shadow) is
//               cheap and done whenever one is ready; the surface interaction (shade_hit) is a few thousand
//               instructions that the wave pays for whatever the number of lanes in it, so it waits until
//               shade_min lanes have a closest hit to shade (or nothing else can make progress).  Most lanes thus
//               still move in step -- query, shade, shadow query, resolve -- and only the stragglers drift.
// Leaves are not tested where they are met: a lane pauses at a leaf whose box it enters, and the wave tests the
// leaves of leaf_batch lanes together (or of those there are, when no lane can step).
// Work is handed out while the kernel runs.  A work tile is what a wave of k_trace_samples traces, sub_len samples
// of the 64 pixels of a tile; here the waves are persistent (the grid is what the GPU holds at once), a wave takes
// the next work tile from a counter in device memory (*work_counter, zero at launch) whenever its lanes have used
// up the pixels of the last one, and a lane that has finished its pixel's samples takes any pixel of the wave's
// current work tile that has not been handed out.  Paths differ much in length (a pixel's 25 samples take between
// half and twice the average number of queries), and with fixed pixels a wave ends up waiting for its slowest
// pixel with most lanes idle; this way every lane has work until the image runs out.
// Every path performs the operations of bounce() in the same order, so the image has the same bits.
// WIDE: the walk goes over the four-wide hierarchy (DevScene::wide, query_step_wide) with a stack per lane.  The stack
// takes the LDS that otherwise parks three finished samples per lane until their 128-byte line is whole, so this
// variant writes every sample as its path ends (the scenes it is for move a hundred times the sample buffer in nodes).
// spill: device memory for the stack entries beyond the first stack_lds (<= WIDE_STACK_LDS), a region per lane of the grid.
template <int FLAGS, bool PINHOLE, bool WIDE>
__global__ void __launch_bounds__(PTX_BLOCK, PTX_BVH_WAVES)
    k_trace_samples_bvh(DevScene sc, DevRender rp, int k0, int k1, int sub_len, int n_tiles, int n_work, size_t stride,
                        double *__restrict__ samples, unsigned long long *__restrict__ stats, unsigned int *__restrict__ work_counter,
                        int exit_lanes, int leaf_batch, int shade_min, int burst, int small_run, int lds_nodes, int *__restrict__ spill,
                        int stack_lds) {
    constexpr int PEND = WIDE ? LS_PARKED : LS_PEND;  // where the pending light sample waits (no parked samples when WIDE)
    __shared__ double lds_d_[PEND + LS_PEND_N][PTX_BLOCK];
    __shared__ int lds_i_[LS_ISLOTS][PTX_BLOCK];
    __shared__ int lds_stack_[WIDE ? WIDE_STACK_LDS : 1][PTX_BLOCK];
    const LaneStore ls{&lds_d_[0][threadIdx.x], &lds_i_[0][threadIdx.x], PEND};
    const int lane = threadIdx.x & 63;
    const int tiles_x = (rp.sx + 7) >> 3;
    const int n_nodes = sc.n_bvh_nodes;
    const WideStack wstack{(LdsIntPtr)&lds_stack_[0][threadIdx.x], spill + ((size_t)blockIdx.x * PTX_BLOCK + lane), (size_t)gridDim.x * PTX_BLOCK, stack_lds};
    // whether the lane's walk has nodes left to visit
    auto walking = [&](const Query &qq) { return WIDE ? qq.node > 0 : qq.node < n_nodes; };
#ifdef PTX_LDS_TOP
    // Experiment (PTX_WALK_LDS_NODES=n, n <= 12, in a build with -DPTX_LDS_TOP): the first n entries of every octant's node
    // array in LDS, a copy per wave.  The arrays are in depth-first order, so these are the root and the chain of nearer
    // children below it: the first steps of every walk.
    __shared__ float4 lds_top_[8 * PTX_LDS_TOP_MAX * 2];
    if (lds_nodes > n_nodes) lds_nodes = n_nodes;
    for (int t = lane; t < 8 * lds_nodes * 2; t += 64) {
        const int oct = t / (2 * lds_nodes), rest = t - oct * 2 * lds_nodes;
        const int src_oct = sc.bvh_per_octant ? oct : 0;
        lds_top_[2 * oct * PTX_LDS_TOP_MAX + rest] = ((const float4 *)(sc.bvh + (size_t)src_oct * n_nodes))[rest];
    }
    __syncthreads();
    const float4 *top = lds_top_;
#else
    const float4 *top = nullptr;
    lds_nodes = 0;
#endif
    // the lane's pixel and sample range (none yet: k == ke)
    int i = 0, j = 0, ks = 0, ke = 0, k = 0;
    size_t pix = 0;
    // the wave's current work tile and which of its pixels have not been handed out; both the same in every lane
    int cur_tile = 0, cur_sub = 0;
    unsigned long long cur_free = 0ULL;
    bool exhausted = false;  // the counter has run past the last work tile

    enum { M_NONE = 0, M_CLOSEST = 1, M_SHADOW = 2 };
    // Paths and queries are counted per wave, from ballots taken where every lane of the wave is present: the counts
    // are then the same in every lane and live in scalar registers (four counters per lane would cost four vector
    // registers of a kernel that has none to spare).  Only the out-of-range refraction-stack accesses, counted deep
    // inside refract_ray, are kept per lane.
    Counters cnt = {0, 0, 0};
    uint32_t w_paths = 0, w_closest = 0, w_shadow = 0;
    Path ps;
    V3 shadow_d = mk(0, 0, 0);  // direction of the lane's shadow query
    double shadow_cb = 0;       // the one field of its ShadowPending besides the direction that stays in registers
    Query q;
#ifdef PTX_PHASE_TIMERS
    // per wave: cycles spent in each part of the loop, how often it ran and with how many lanes
    enum { D_TOTAL, D_SHADE, D_SHADOW, D_SETUP, D_STEPS, D_LEAF, D_REGEN, D_N_ITER, D_N_SHADE, D_L_SHADE, D_N_SHADOW, D_L_SHADOW,
           D_N_SETUP, D_L_SETUP, D_N_BURST, D_L_BURST, D_N_LEAF, D_L_LEAF, D_N_REGEN, D_L_REGEN, D_L_STEPS, D_L_ACTIVE };
    unsigned long long dbg[PTX_N_DBG] = {};
    const long long dbg_t0 = clock64();
    long long dbg_t;
#define DBG_BEGIN() (dbg_t = clock64())
#define DBG_END(slot_cyc, slot_n, slot_l, pred)                 \
    do {                                                        \
        const int lanes_ = __popcll(__ballot(pred));            \
        dbg[slot_cyc] += (unsigned long long)(clock64() - dbg_t); \
        dbg[slot_n] += lanes_ > 0;                              \
        dbg[slot_l] += lanes_;                                  \
    } while (0)
#else
#define DBG_BEGIN() ((void)0)
#define DBG_END(a, b, c, d) ((void)0)
#endif
    q.node = WIDE ? 0 : n_nodes;
    q.leaf = -1;
    int mode = M_NONE;   // what the lane's query is for
    bool fresh = false;  // the query has yet to be set up
    bool over = false;   // M_SHADOW: the path ends when the shadow ray is resolved
    for (;;) {
        // ---- shading: lanes whose query is over
        bool ended = false;
        bool shaded = false;  // shade_hit ran in this iteration
        {
            const bool ready = mode != M_NONE && !fresh && !walking(q) && q.leaf < 0;
            const bool ready_closest = ready && mode == M_CLOSEST;
            const int n_closest = __popcll(__ballot(ready_closest));
            const int n_busy = __popcll(__ballot(mode != M_NONE && !ready));
            const int n_active = __popcll(__ballot(mode != M_NONE));
            const bool ready_shadow = ready && mode == M_SHADOW;
            const bool shade_now = n_closest > 0 && (64 * n_closest >= shade_min * n_active || n_busy == 0);
            const int n_shadow = __popcll(__ballot(ready_shadow));
            if (n_shadow > 0 && (n_shadow >= small_run || shade_now || n_busy == 0)) {
                DBG_BEGIN();
                if (ready_shadow) {
                    ShadowPending pend;
                    pend.ed = shadow_d;
                    pend.cb = shadow_cb;
                    shadow_pending_load(ls, pend);
                    apply_shadow(sc, rp, ps, q.h, pend);
                    ended = over;
                    mode = M_CLOSEST;
                    fresh = true;
                }
                DBG_END(D_SHADOW, D_N_SHADOW, D_L_SHADOW, ready_shadow);
            }
            shaded = shade_now;
            if (shade_now) {
                DBG_BEGIN();
                double glossy = 0;
                if (ready_closest) {
                    ShadowPending pend;
                    pend.ed = mk(0, 0, 0);
                    pend.cb = 0;
                    const int r = shade_hit<FLAGS>(sc, rp, ls, ps, cnt, q.h, glossy, pend, over);
                    ended = r == SH_END;
                    mode = r == SH_SHADOW ? M_SHADOW : M_CLOSEST;
                    fresh = true;
                    // (unconditional: a lane without a shadow query does not read them, and the old values do not
                    // have to be kept through shade_hit)
                    shadow_d = pend.ed;
                    shadow_cb = pend.cb;
                }
                glossy_service(ps, glossy, lane);
                DBG_END(D_SHADE, D_N_SHADE, D_L_SHADE, ready_closest);
            }
        }
#ifdef PTX_PHASE_TIMERS
        dbg[D_N_ITER]++;
        dbg[D_L_ACTIVE] += __popcll(__ballot(mode != M_NONE || k < ke || !exhausted));
        const bool dbg_regen = ended || (mode == M_NONE && k < ke);
#endif
        DBG_BEGIN();
        w_paths += (uint32_t)__popcll(__ballot(ended));
        if (ended) {
            // four samples make a 128-byte line of the lane's stream, written once and whole (see k_trace_samples)
            const int slot = (k - ks) & 3;
            double4 *out = (double4 *)(samples + (pix * stride + (size_t)(k - k0)) * PTX_SAMPLE_DOUBLES);
            if (WIDE) {
                out[0] = make_double4(ps.I.x, ps.I.y, ps.I.z, 0.0);
            } else if (slot == 3) {
#pragma unroll
                for (int c = 0; c < 3; c++)
                    out[c - 3] = make_double4(ls.get(LS_PARKED + 3 * c), ls.get(LS_PARKED + 3 * c + 1), ls.get(LS_PARKED + 3 * c + 2), 0.0);
                out[0] = make_double4(ps.I.x, ps.I.y, ps.I.z, 0.0);
            } else if (k + (3 - slot) >= ke) {
                out[0] = make_double4(ps.I.x, ps.I.y, ps.I.z, 0.0);
            } else {
                ls.put(LS_PARKED + 3 * slot, ps.I.x);
                ls.put(LS_PARKED + 3 * slot + 1, ps.I.y);
                ls.put(LS_PARKED + 3 * slot + 2, ps.I.z);
            }
            mode = M_NONE;
            fresh = false;
            k++;
        }
        // ---- a new pixel for the lanes that have done theirs
        for (;;) {
            const bool need = mode == M_NONE && k >= ke;
            const unsigned long long m_need = __ballot(need);
            if (m_need == 0ULL) break;
            if (cur_free == 0ULL) {
                if (exhausted) break;
                unsigned int wt = 0;
                if (lane == 0) wt = atomicAdd(work_counter, 1u);
                wt = __shfl(wt, 0);
                if (wt >= (unsigned int)n_work) {
                    exhausted = true;
                    break;
                }
                cur_tile = (int)(wt % (unsigned int)n_tiles);
                cur_sub = (int)(wt / (unsigned int)n_tiles);
                const int tx = cur_tile % tiles_x, ty = cur_tile / tiles_x;
                cur_free = __ballot(tx * 8 + (lane & 7) < rp.sx && ty * 8 + (lane >> 3) < rp.rows_owned);
                continue;
            }
            // the r-th lane in need takes the r-th free pixel
            const int r = __popcll(m_need & ((1ULL << lane) - 1ULL));
            const int n_free = __popcll(cur_free), n_need = __popcll(m_need);
            if (need && r < n_free) {
                int pos = 0, left = r;  // position of the r-th set bit of cur_free
#pragma unroll
                for (int w = 32; w >= 1; w >>= 1) {
                    const int c = __popcll((cur_free >> pos) & ((1ULL << w) - 1ULL));
                    if (left >= c) {
                        left -= c;
                        pos += w;
                    }
                }
                const int tx = cur_tile % tiles_x, ty = cur_tile / tiles_x;
                i = tx * 8 + (pos & 7);
                const int jl = ty * 8 + (pos >> 3);
                j = ptx_global_row(jl, rp.rank, rp.n_ranks, rp.band_rows);
                pix = (size_t)jl * rp.sx + i;
                ks = k0 + cur_sub * sub_len;
                ke = ks + sub_len < k1 ? ks + sub_len : k1;
                k = ks;
                if (PINHOLE) {
                    // (into a path state of its own: the lane's `ps` is left alone until its next sample starts)
                    Path first;
                    primary_ray(sc, rp, ls, i, j, ks, first);
                    ls.put(LS_PRIMARY_D + 0, first.d.x); ls.put(LS_PRIMARY_D + 1, first.d.y); ls.put(LS_PRIMARY_D + 2, first.d.z);
                }
            }
            // drop the pixels just handed out: the lowest min(n_need, n_free) set bits
            const int taken = n_need < n_free ? n_need : n_free;
            unsigned long long rest = cur_free;
            for (int t = 0; t < taken; t++) rest &= rest - 1ULL;
            cur_free = rest;
        }
        // ---- a pixel's next sample (not for a few lanes at a time: they wait for the iteration in which the others
        // have been shaded, unless the wave has nothing else to do)
        const int n_start = __popcll(__ballot(mode == M_NONE && k < ke));
        const bool start_now = n_start >= small_run || shaded || __ballot(mode != M_NONE && (fresh || walking(q) || q.leaf >= 0)) == 0ULL;
        if (start_now && mode == M_NONE && k < ke) {
            if (PINHOLE) primary_ray_cached(sc, rp, ls, i, j, k, ps);
            else primary_ray(sc, rp, ls, i, j, k, ps);
            if (rp.max_depth < 1) {  // reference PathTracer.c:273 with depth = 1: the path is over before its first query
                mode = M_SHADOW;
                over = true;
                q.h.obj = -1;
            } else {
                mode = M_CLOSEST;
                fresh = true;
            }
        }
        DBG_END(D_REGEN, D_N_REGEN, D_L_REGEN, dbg_regen);
        if (__ballot(mode != M_NONE || k < ke) == 0ULL) break;
        // ---- the next query of the lanes that have one: the primitives every ray tests
        DBG_BEGIN();
#ifdef PTX_PHASE_TIMERS
        const bool dbg_fresh = fresh;
#endif
        w_closest += (uint32_t)__popcll(__ballot(fresh && mode == M_CLOSEST));
        w_shadow += (uint32_t)__popcll(__ballot(fresh && mode != M_CLOSEST));
        if (fresh) {
            V3 qd;
            if (mode == M_CLOSEST) {
                ps.d = norm3(ps.d);
                qd = ps.d;
            } else {
                qd = shadow_d;
            }
            if (WIDE) query_begin_wide(sc, ps.p0, qd, ps.os, q, wstack);
            else query_begin(sc, ps.p0, qd, ps.os, q);
            fresh = false;
        }
        DBG_END(D_SETUP, D_N_SETUP, D_L_SETUP, dbg_fresh);
        // ---- walking: bursts of up to `burst` steps per lane, with the wave-level decisions in between
        const int n_active = __popcll(__ballot(mode != M_NONE));
        for (;;) {
            DBG_BEGIN();
#ifdef PTX_PHASE_TIMERS
            const bool dbg_walking = q.leaf < 0 && walking(q);
            for (int st = 0; st < burst && q.leaf < 0 && walking(q); st++) {
                if (WIDE) query_step_wide(sc, ps.os, q, wstack);
                else query_step(sc, ps.os, q, top, lds_nodes);
                dbg[D_L_STEPS]++;
            }
#else
            for (int st = 0; st < burst && q.leaf < 0 && walking(q); st++) {
                if (WIDE) query_step_wide(sc, ps.os, q, wstack);
                else query_step(sc, ps.os, q, top, lds_nodes);
            }
#endif
            DBG_END(D_STEPS, D_N_BURST, D_L_BURST, dbg_walking);
            const bool paused = q.leaf >= 0;
            const bool stepping = !paused && walking(q);
            const int n_step = __popcll(__ballot(stepping)), n_paused = __popcll(__ballot(paused));
            if (n_paused > 0 && (n_paused >= leaf_batch || n_step == 0)) {
                DBG_BEGIN();
                if (paused) query_leaf(sc, ps.p0, mode == M_SHADOW ? shadow_d : ps.d, q);
                DBG_END(D_LEAF, D_N_LEAF, D_L_LEAF, paused);
                continue;
            }
            if (n_step == 0) break;
            if (n_step + n_paused <= exit_lanes) {
                const bool done = mode != M_NONE && !stepping && !paused;
                const int n_closest = __popcll(__ballot(done && mode == M_CLOSEST));
                if (64 * n_closest >= shade_min * n_active || __popcll(__ballot(done && mode == M_SHADOW)) >= small_run) break;
            }
        }
    }
    #ifdef PTX_PHASE_TIMERS
    dbg[D_TOTAL] = (unsigned long long)(clock64() - dbg_t0);
    for (int c = 0; c < PTX_N_DBG; c++) {
        // cycle counts and invocation counts are the same in every lane; D_L_STEPS is per lane
        unsigned long long v = dbg[c];
        if (c == D_L_STEPS)
            for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off);
        if (lane == 0 && v) atomicAdd(&stats[PTX_STATS_DBG + c], v);
    }
#endif
    // one set of atomics per wave
    unsigned long long c3 = cnt.oob;
    for (int off = 32; off > 0; off >>= 1) c3 += __shfl_down(c3, off);
    if (lane == 0) {
        atomicAdd(&stats[0], (unsigned long long)w_paths);
        atomicAdd(&stats[1], (unsigned long long)w_closest);
        atomicAdd(&stats[2], (unsigned long long)w_shadow);
        if (c3) atomicAdd(&stats[3], c3);
    }
}


// What a launch of the kernel takes besides the grid and the stream
struct TraceBvhArgs {
    DevScene sc;
    DevRender rp;
    int k0, k1, sub_len, n_tiles, n_work;
    size_t stride;
    double *samples;
    unsigned long long *stats;
    unsigned int *work_counter;
    int exit_lanes, leaf_batch, shade_min, burst, small_run, lds_nodes;
    int *spill;
    int stack_lds;
};
template <int FLAGS, bool PINHOLE, bool WIDE>
inline void launch_trace_bvh(dim3 grid, hipStream_t stream, const TraceBvhArgs &a) {
    hipLaunchKernelGGL((k_trace_samples_bvh<FLAGS, PINHOLE, WIDE>), grid, dim3(PTX_BLOCK), 0, stream, a.sc, a.rp, a.k0, a.k1, a.sub_len, a.n_tiles,
                       a.n_work, a.stride, a.samples, a.stats, a.work_counter, a.exit_lanes, a.leaf_batch, a.shade_min, a.burst, a.small_run,
                       a.lds_nodes, a.spill, a.stack_lds);
}
// The instantiations for the four-wide hierarchy live in a translation unit of their own (ptx_wide.hip), so that the
// two sets compile side by side.  flags: the IS/ES/RR bits.
void ptx_launch_trace_bvh_wide(int flags, bool pinhole, dim3 grid, hipStream_t stream, const TraceBvhArgs &a);
