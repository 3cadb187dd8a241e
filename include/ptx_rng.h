/*
 * ptx_rng.h -- the uniform-number streams that stand in for drand48().
 *
 * The reference draws every random number from libc drand48(), one global
 * 48-bit LCG seeded from the wall clock and shared, racily, by all OpenMP
 * threads (reference PathTracer.c:560, :620).  Nothing about a render is
 * therefore reproducible.  Two replacements are defined here:
 *
 *  - counter stream (PTX_RNG_COUNTER): u(seed, pixel, sample, i) is a pure
 *    function of its four arguments, i being the position of the draw inside
 *    one path (primary-ray draws first, then each bounce's draws in the order
 *    the reference makes them).  Any device may evaluate any pixel in any
 *    order and get the same numbers; this is what the GPU uses.
 *  - LCG48 stream (PTX_RNG_LCG48): the drand48 recurrence
 *    X <- (0x5DEECE66D * X + 0xB) mod 2^48, u = X / 2^48, consumed in the
 *    order a single-threaded run of the reference consumes it.  CPU only.
 *
 * Both are written so that the same source compiles as C, C++ and HIP
 * device code and returns bit-identical doubles everywhere.
 */
#ifndef PTX_RNG_H
#define PTX_RNG_H

#include <stdint.h>

#if defined(__HIPCC__)
#define PTX_HD __host__ __device__ static inline
#else
#define PTX_HD static inline
#endif

#define PTX_GOLDEN 0x9E3779B97F4A7C15ULL

/* splitmix64 output function (Steele, Lea, Flood 2014). */
PTX_HD uint64_t ptx_mix64(uint64_t z) {
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}

/* Key of one path: pixel is j*sx+i in whole-image coordinates. */
PTX_HD uint64_t ptx_path_key(uint64_t seed, uint64_t pixel, uint64_t sample) {
    return ptx_mix64(ptx_mix64(seed + PTX_GOLDEN * (pixel + 1)) + sample);
}

/* 52 random mantissa bits -> [1,2) -> [0,1).  Exact in binary64. */
PTX_HD double ptx_bits_to_unit(uint64_t z) {
    union { uint64_t u; double d; } c;
    c.u = (z >> 12) | 0x3FF0000000000000ULL;
    return c.d - 1.0;
}

/* The i-th draw (i = 0,1,2,...) of the path with the given key. */
PTX_HD double ptx_counter_draw(uint64_t key, uint32_t i) {
    return ptx_bits_to_unit(ptx_mix64(key + PTX_GOLDEN * ((uint64_t)i + 1)));
}

/* drand48 recurrence; *state holds X (48 bits). */
PTX_HD double ptx_lcg48_draw(uint64_t *state) {
    *state = (0x5DEECE66DULL * *state + 0xBULL) & 0xFFFFFFFFFFFFULL;
    return (double)*state * (1.0 / 281474976710656.0);
}

/* srand48(seed): X = (seed << 16) + 0x330E. */
PTX_HD uint64_t ptx_lcg48_seed(uint64_t seed) {
    return ((seed << 16) + 0x330EULL) & 0xFFFFFFFFFFFFULL;
}

#endif
