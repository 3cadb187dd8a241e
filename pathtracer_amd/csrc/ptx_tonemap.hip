// ptx_tonemap.hip -- the tone mapping of dataOutput() (reference utils_path.c:2011-2060) on the device.
//
// dataOutput turns the 3*sx*sx accumulator values into bytes in four steps: global minimum and maximum and a
// normalisation to [0,1]; a 1000-bin histogram; a low and a high cut found by scanning the histogram; clip,
// power 0.75, truncation to 0..255.  The first, second and fourth step touch every value and run here as three
// kernels; the scan of the 1000 bins stays on the host.  Every per-value operation is the reference's IEEE
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

// value -> [0,1] as the reference does it in place: t = t - mi; t = t / (mx - mi)
__device__ __forceinline__ double tm_normalise(double v, double mi, double range) {
    v = v - mi;
    return v / range;
}

// The reference finds the bin of v by scanning j = 0..999 for biw*j <= v < biw*(j+1); the same pair of
// comparisons decides here, starting from the arithmetic guess.
__global__ void __launch_bounds__(TM_BLOCK) k_tm_hist(const double *__restrict__ t, size_t n, double mi, double range, double biw,
                                                       unsigned long long *__restrict__ hist) {
    __shared__ unsigned s_hist[TM_BINS];
    for (int j = threadIdx.x; j < TM_BINS; j += TM_BLOCK) s_hist[j] = 0;
    __syncthreads();
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

// clip, gamma, byte (reference utils_path.c:2050-2059).  unsure[0] counts, unsure[1..] lists the values whose
// byte could depend on the last bit of pow().
__global__ void __launch_bounds__(TM_BLOCK) k_tm_bytes(const double *__restrict__ t, size_t n, double mi, double range, double lo,
                                                        double cut_range, uint8_t *__restrict__ bits,
                                                        unsigned long long *__restrict__ unsure) {
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
            goto done;                                                   \
        }                                                                \
    } while (0)

}  // namespace

int ptx_tonemap_run(hipStream_t stream, const double *d_im, size_t n, uint8_t *h_bits, ptx_tonemap_info *info, std::string &err) {
    int rc = -2;  // PTX_EDEVICE
    const int blocks = (int)std::min<size_t>((n + TM_BLOCK - 1) / TM_BLOCK, 1024);
    double *d_part = nullptr;
    unsigned long long *d_hist = nullptr, *d_unsure = nullptr;
    uint8_t *d_bits = nullptr;
    std::vector<double> part(2 * (size_t)blocks);
    std::vector<unsigned long long> hist(TM_BINS), unsure;
    double mi = 10e6, mx = -10e6, range, lo, hi, acc;
    const double biw = 1.000001 / 1000.0;
    const double pct = .005 * (double)n;  // 0.5 % of the 3*sx*sx values
    int j;
    unsigned long long n_unsure = 0;

    TMCHK(hipMalloc((void **)&d_part, part.size() * sizeof(double)));
    TMCHK(hipMalloc((void **)&d_hist, TM_BINS * sizeof(unsigned long long)));
    TMCHK(hipMalloc((void **)&d_unsure, (1 + (size_t)TM_MAX_UNSURE) * sizeof(unsigned long long)));
    TMCHK(hipMalloc((void **)&d_bits, n));
    TMCHK(hipMemsetAsync(d_hist, 0, TM_BINS * sizeof(unsigned long long), stream));
    TMCHK(hipMemsetAsync(d_unsure, 0, sizeof(unsigned long long), stream));

    hipLaunchKernelGGL(k_tm_minmax, dim3(blocks), dim3(TM_BLOCK), 0, stream, d_im, n, d_part);
    TMCHK(hipGetLastError());
    TMCHK(hipMemcpyAsync(part.data(), d_part, part.size() * sizeof(double), hipMemcpyDeviceToHost, stream));
    TMCHK(hipStreamSynchronize(stream));
    for (int b = 0; b < blocks; b++) {
        if (part[2 * b] < mi) mi = part[2 * b];
        if (part[2 * b + 1] > mx) mx = part[2 * b + 1];
    }
    range = mx - mi;

    hipLaunchKernelGGL(k_tm_hist, dim3(blocks), dim3(TM_BLOCK), 0, stream, d_im, n, mi, range, biw, d_hist);
    TMCHK(hipGetLastError());
    TMCHK(hipMemcpyAsync(hist.data(), d_hist, TM_BINS * sizeof(unsigned long long), hipMemcpyDeviceToHost, stream));
    TMCHK(hipStreamSynchronize(stream));

    // the two scans of dataOutput (reference utils_path.c:2034-2047); counts are exact in double
    acc = 0;
    for (j = 5; j < 990; j++) {
        acc += (double)hist[j];
        if ((double)hist[j + 5] - (double)hist[j - 5] > pct) break;
        if (acc > pct) break;
    }
    lo = (biw * (.90 * j));
    for (j = 990; j > 5; j--)
        if ((double)hist[j - 5] - (double)hist[j + 5] > pct) break;
    hi = (biw * (j + (.25 * (999 - j))));

    hipLaunchKernelGGL(k_tm_bytes, dim3(blocks), dim3(TM_BLOCK), 0, stream, d_im, n, mi, range, lo, hi - lo, d_bits, d_unsure);
    TMCHK(hipGetLastError());
    TMCHK(hipMemcpyAsync(h_bits, d_bits, n, hipMemcpyDeviceToHost, stream));
    TMCHK(hipMemcpyAsync(&n_unsure, d_unsure, sizeof(n_unsure), hipMemcpyDeviceToHost, stream));
    TMCHK(hipStreamSynchronize(stream));
    if (n_unsure > TM_MAX_UNSURE) {
        err = "ptx_tonemap: more values on a byte boundary than the list holds";
        rc = -3;  // PTX_ESTATE
        goto done;
    }
    if (n_unsure) {
        unsure.resize(n_unsure);
        TMCHK(hipMemcpy(unsure.data(), d_unsure + 1, n_unsure * sizeof(unsigned long long), hipMemcpyDeviceToHost));
        for (unsigned long long q : unsure) {
            double v;
            TMCHK(hipMemcpy(&v, d_im + q, sizeof(double), hipMemcpyDeviceToHost));
            v = v - mi;
            v = v / range;
            v = v - lo;
            v = v / (hi - lo);
            if (v < 0.0) v = 0.0;
            if (v > 1.0) v = 1.0;
            v = pow(v, .75);
            h_bits[q] = (unsigned char)(255.0 * v);
        }
    }
    if (info) {
        info->min = mi;
        info->max = mx;
        info->low_cut = lo;
        info->high_cut = hi;
        info->recomputed_on_host = n_unsure;
    }
    rc = 0;
done:
    (void)hipFree(d_part);
    (void)hipFree(d_hist);
    (void)hipFree(d_unsure);
    (void)hipFree(d_bits);
    return rc;
}
