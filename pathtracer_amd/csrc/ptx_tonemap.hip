// ptx_tonemap.hip -- the tone mapping of dataOutput() (reference utils_path.c:2011-2060) on the device.
//
// dataOutput turns the 3*sx*sx accumulator values into bytes in four steps: global minimum and maximum and a
// normalisation to [0,1]; a 1000-bin histogram; a low and a high cut found by scanning the histogram; clip,
// power 0.75, truncation to 0..255.  The first, second and fourth step touch every value and run here as three
// kernels; the end of the reduction and the scan of the 1000 bins are one-wave kernels in between, so that the five
// launches queue up behind each other without the host (one synchronisation per image, at the end).  The scratch
// buffers belong to the context and are allocated once.  Every operation is the reference's IEEE
// operation (subtract, divide, compare, multiply), so minimum, maximum, histogram and cuts are the reference's
// numbers exactly.  The one exception is pow(): the device's and glibc's may differ in the last bit, which can
// change a byte only when 255*pow(.) lies within rounding of an integer.  Such values (none in practice) are
// reported back and recomputed with the host's pow(), which makes the bytes those of the reference in every case.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <string>
#include <vector>

#include "ptx_internal.h"

namespace {

constexpr int TM_BLOCK = 256;
constexpr int TM_BINS = 1000;
constexpr unsigned TM_MAX_UNSURE = 65536;

// "if (t < mi) mi = t; if (t > mx) mx = t;" over a slice, then over the block (any order gives the same result:
// values that compare false, NaN included, never replace anything)
__global__ void __launch_bounds__(TM_BLOCK) k_tm_minmax(const double *__restrict__ t, size_t n, double *__restrict__ part) {
    double mi = 10e6, mx = -10e6;  // the reference's starting values
    for (size_t q = (size_t)blockIdx.x * TM_BLOCK + threadIdx.x; q < n; q += (size_t)gridDim.x * TM_BLOCK) {
        const double v = t[q];
        if (v < mi) mi = v;
        if (v > mx) mx = v;
    }
    for (int off = 32; off > 0; off >>= 1) {
        const double a = __shfl_down(mi, off), b = __shfl_down(mx, off);
        if (a < mi) mi = a;
        if (b > mx) mx = b;
    }
    __shared__ double s_mi[TM_BLOCK / 64], s_mx[TM_BLOCK / 64];
    if ((threadIdx.x & 63) == 0) {
        s_mi[threadIdx.x >> 6] = mi;
        s_mx[threadIdx.x >> 6] = mx;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < TM_BLOCK / 64; w++) {
            if (s_mi[w] < mi) mi = s_mi[w];
            if (s_mx[w] > mx) mx = s_mx[w];
        }
        part[2 * blockIdx.x] = mi;
        part[2 * blockIdx.x + 1] = mx;
    }
}

// The block minima and maxima folded into one pair: prm[0] = min, prm[1] = max, prm[2] = max - min (one wave)
__global__ void __launch_bounds__(64) k_tm_minmax_final(const double *__restrict__ part, int blocks, double *__restrict__ prm) {
    double mi = 10e6, mx = -10e6;
    for (int b = threadIdx.x; b < blocks; b += 64) {
        if (part[2 * b] < mi) mi = part[2 * b];
        if (part[2 * b + 1] > mx) mx = part[2 * b + 1];
    }
    for (int off = 32; off > 0; off >>= 1) {
        const double a = __shfl_down(mi, off), b = __shfl_down(mx, off);
        if (a < mi) mi = a;
        if (b > mx) mx = b;
    }
    if (threadIdx.x == 0) {
        prm[0] = mi;
        prm[1] = mx;
        prm[2] = mx - mi;
    }
}

// value -> [0,1] as the reference does it in place: t = t - mi; t = t / (mx - mi)
__device__ __forceinline__ double tm_normalise(double v, double mi, double range) {
    v = v - mi;
    return v / range;
}

// The reference finds the bin of v by scanning j = 0..999 for biw*j <= v < biw*(j+1); the same pair of
// comparisons decides here, starting from the arithmetic guess.  prm: minimum, maximum, range (k_tm_minmax_final)
__global__ void __launch_bounds__(TM_BLOCK) k_tm_hist(const double *__restrict__ t, size_t n, const double *__restrict__ prm, double biw,
                                                       unsigned long long *__restrict__ hist) {
    __shared__ unsigned s_hist[TM_BINS];
    for (int j = threadIdx.x; j < TM_BINS; j += TM_BLOCK) s_hist[j] = 0;
    __syncthreads();
    const double mi = prm[0], range = prm[2];
    for (size_t q = (size_t)blockIdx.x * TM_BLOCK + threadIdx.x; q < n; q += (size_t)gridDim.x * TM_BLOCK) {
        const double v = tm_normalise(t[q], mi, range);
        if (!(v >= 0)) continue;  // NaN or negative: no bin
        double g = v / biw;
        int j = g > 999.0 ? 999 : (int)g;
        while (j > 0 && !(v >= biw * j)) j--;
        while (j < 999 && !(v < biw * (j + 1))) j++;
        if (v >= biw * j && v < biw * (j + 1)) atomicAdd(&s_hist[j], 1u);
    }
    __syncthreads();
    for (int j = threadIdx.x; j < TM_BINS; j += TM_BLOCK)
        if (s_hist[j]) atomicAdd(&hist[j], (unsigned long long)s_hist[j]);
}

// The two scans of dataOutput over the 1000 bins (reference utils_path.c:2034-2047), by one lane; counts are exact
// in double.  prm[3] = low cut, prm[4] = high cut, prm[5] = their difference.  No multiply is fused with an add here:
// the cuts must be the products and sums the reference computes.
__global__ void k_tm_cuts(const unsigned long long *__restrict__ hist, double n_values, double biw, double *__restrict__ prm) {
#pragma clang fp contract(off)
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    const double pct = .005 * n_values;  // 0.5 % of the 3*sx*sx values
    double acc = 0;
    int j;
    for (j = 5; j < 990; j++) {
        acc += (double)hist[j];
        if ((double)hist[j + 5] - (double)hist[j - 5] > pct) break;
        if (acc > pct) break;
    }
    const double lo = (biw * (.90 * j));
    for (j = 990; j > 5; j--)
        if ((double)hist[j - 5] - (double)hist[j + 5] > pct) break;
    const double hi = (biw * (j + (.25 * (999 - j))));
    prm[3] = lo;
    prm[4] = hi;
    prm[5] = hi - lo;
}

// clip, gamma, byte (reference utils_path.c:2050-2059).  unsure[0] counts, unsure[1..] lists the values whose
// byte could depend on the last bit of pow().
__global__ void __launch_bounds__(TM_BLOCK) k_tm_bytes(const double *__restrict__ t, size_t n, const double *__restrict__ prm,
                                                        uint8_t *__restrict__ bits, unsigned long long *__restrict__ unsure) {
    const double mi = prm[0], range = prm[2], lo = prm[3], cut_range = prm[5];
    for (size_t q = (size_t)blockIdx.x * TM_BLOCK + threadIdx.x; q < n; q += (size_t)gridDim.x * TM_BLOCK) {
        double v = tm_normalise(t[q], mi, range);
        v = v - lo;
        v = v / cut_range;
        if (v < 0.0) v = 0.0;
        if (v > 1.0) v = 1.0;
        const double x = 255.0 * pow(v, .75);
        bits[q] = (uint8_t)x;  // NaN gives 0, as the x86 conversion of the reference does
        if (v > 0.0 && v < 1.0 && fabs(x - rint(x)) < 1e-9) {
            const unsigned long long at = atomicAdd(&unsure[0], 1ULL);
            if (at < TM_MAX_UNSURE) unsure[1 + at] = (unsigned long long)q;
        }
    }
}

#define TMCHK(expr)                                                      \
    do {                                                                 \
        hipError_t e_ = (expr);                                          \
        if (e_ != hipSuccess) {                                          \
            err = std::string(#expr) + ": " + hipGetErrorString(e_);     \
            return -2; /* PTX_EDEVICE */                                 \
        }                                                                \
    } while (0)

constexpr int TM_MAX_BLOCKS = 1024;

}  // namespace

void ptx_tonemap_release(ptx_tonemap_scratch &ws) {
    (void)hipFree(ws.d_part);
    (void)hipFree(ws.d_prm);
    (void)hipFree(ws.d_hist);
    (void)hipFree(ws.d_unsure);
    (void)hipFree(ws.d_bits);
    ws = ptx_tonemap_scratch();
}

int ptx_tonemap_run(hipStream_t stream, ptx_tonemap_scratch &ws, const double *d_im, size_t n, uint8_t *h_bits, ptx_tonemap_info *info,
                    std::string &err) {
    const int blocks = (int)std::min<size_t>((n + TM_BLOCK - 1) / TM_BLOCK, TM_MAX_BLOCKS);
    const double biw = 1.000001 / 1000.0;
    // scratch of the context: the small buffers once, the byte image whenever a larger one is asked for
    if (!ws.d_part) TMCHK(hipMalloc((void **)&ws.d_part, 2 * (size_t)TM_MAX_BLOCKS * sizeof(double)));
    if (!ws.d_prm) TMCHK(hipMalloc((void **)&ws.d_prm, 6 * sizeof(double)));
    if (!ws.d_hist) TMCHK(hipMalloc((void **)&ws.d_hist, TM_BINS * sizeof(unsigned long long)));
    if (!ws.d_unsure) TMCHK(hipMalloc((void **)&ws.d_unsure, (1 + (size_t)TM_MAX_UNSURE) * sizeof(unsigned long long)));
    if (ws.bits_bytes < n) {
        TMCHK(hipStreamSynchronize(stream));
        (void)hipFree(ws.d_bits);
        ws.d_bits = nullptr;
        ws.bits_bytes = 0;
        TMCHK(hipMalloc((void **)&ws.d_bits, n));
        ws.bits_bytes = n;
    }
    TMCHK(hipMemsetAsync(ws.d_hist, 0, TM_BINS * sizeof(unsigned long long), stream));
    TMCHK(hipMemsetAsync(ws.d_unsure, 0, sizeof(unsigned long long), stream));
    hipLaunchKernelGGL(k_tm_minmax, dim3(blocks), dim3(TM_BLOCK), 0, stream, d_im, n, ws.d_part);
    hipLaunchKernelGGL(k_tm_minmax_final, dim3(1), dim3(64), 0, stream, ws.d_part, blocks, ws.d_prm);
    hipLaunchKernelGGL(k_tm_hist, dim3(blocks), dim3(TM_BLOCK), 0, stream, d_im, n, ws.d_prm, biw, ws.d_hist);
    hipLaunchKernelGGL(k_tm_cuts, dim3(1), dim3(64), 0, stream, ws.d_hist, (double)n, biw, ws.d_prm);
    hipLaunchKernelGGL(k_tm_bytes, dim3(blocks), dim3(TM_BLOCK), 0, stream, d_im, n, ws.d_prm, ws.d_bits, ws.d_unsure);
    TMCHK(hipGetLastError());
    double prm[6];
    unsigned long long n_unsure = 0;
    TMCHK(hipMemcpyAsync(h_bits, ws.d_bits, n, hipMemcpyDeviceToHost, stream));
    TMCHK(hipMemcpyAsync(prm, ws.d_prm, sizeof(prm), hipMemcpyDeviceToHost, stream));
    TMCHK(hipMemcpyAsync(&n_unsure, ws.d_unsure, sizeof(n_unsure), hipMemcpyDeviceToHost, stream));
    TMCHK(hipStreamSynchronize(stream));  // the one wait of an image
    if (n_unsure > TM_MAX_UNSURE) {
        err = "ptx_tonemap: more values on a byte boundary than the list holds";
        return -3;  // PTX_ESTATE
    }
    if (n_unsure) {  // (never seen) bytes that could depend on the last bit of pow(): once more with the host's
        std::vector<unsigned long long> unsure(n_unsure);
        TMCHK(hipMemcpy(unsure.data(), ws.d_unsure + 1, n_unsure * sizeof(unsigned long long), hipMemcpyDeviceToHost));
        for (unsigned long long q : unsure) {
            double v;
            TMCHK(hipMemcpy(&v, d_im + q, sizeof(double), hipMemcpyDeviceToHost));
            v = v - prm[0];
            v = v / prm[2];
            v = v - prm[3];
            v = v / prm[5];
            if (v < 0.0) v = 0.0;
            if (v > 1.0) v = 1.0;
            v = pow(v, .75);
            h_bits[q] = (unsigned char)(255.0 * v);
        }
    }
    if (info) {
        info->min = prm[0];
        info->max = prm[1];
        info->low_cut = prm[3];
        info->high_cut = prm[4];
        info->recomputed_on_host = n_unsure;
    }
    return 0;
}
