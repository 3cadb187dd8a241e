// rcp_accuracy.hip -- how exact the hardware's double-precision reciprocal and reciprocal square root are, raw and
// after Newton / Goldschmidt steps: the largest relative error over 2^24 random arguments, in units of 2^-53.
//   hipcc --offload-arch=gfx950 -O2 tools/rcp_accuracy.hip -o tools/bin/rcp_accuracy && tools/bin/rcp_accuracy
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>

__device__ double rel(double got, double want) { return fabs(got - want) / fabs(want) * 0x1p53; }

__global__ void k(double *out) {
    unsigned long long z = 0x9E3779B97F4A7C15ULL * (blockIdx.x * 256 + threadIdx.x + 1);
    double e[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    for (int q = 0; q < 4096; q++) {
        z ^= z >> 12; z ^= z << 25; z ^= z >> 27;
        const double u = (double)((z * 0x2545F4914F6CDD1DULL) >> 11) * 0x1p-53;
        const double x = (q & 1) ? 0.25 + 15.75 * u : exp((u - 0.5) * 80.0);
        // reciprocal
        double r = __builtin_amdgcn_rcp(x);
        e[0] = fmax(e[0], rel(r, 1.0 / x));
        double t = fma(-x, r, 1.0);
        const double t0 = t;
        double r1 = fma(r, t, r);
        e[1] = fmax(e[1], rel(r1, 1.0 / x));
        t = fma(-x, r1, 1.0);
        e[2] = fmax(e[2], rel(fma(r1, t, r1), 1.0 / x));
        e[8] = fmax(e[8], rel(fma(r, fma(t0, t0, t0), r), 1.0 / x));  // one second-order step
        // reciprocal square root and square root
        double y = __builtin_amdgcn_rsq(x);
        e[3] = fmax(e[3], rel(y, 1.0 / sqrt(x)));
        double g = x * y, h = 0.5 * y;
        double rr = fma(-h, g, 0.5);
        double g1 = fma(g, rr, g);
        double d = fma(-g1, g1, x);
        e[4] = fmax(e[4], rel(fma(d, h, g1), sqrt(x)));                 // h not refined
        e[5] = fmax(e[5], rel(fma(d, fma(h, rr, h), g1), sqrt(x)));     // h refined
        double d0 = fma(-g, g, x);
        e[6] = fmax(e[6], rel(fma(d0, h, g), sqrt(x)));                 // no Goldschmidt step at all
        double ee = fma(-x * y, y, 1.0);
        e[7] = fmax(e[7], rel(fma(y * ee, 0.5, y), 1.0 / sqrt(x)));      // rsqrt, one second-order step
    }
    for (int c = 0; c < 9; c++) {
        double m = e[c];
        for (int off = 32; off > 0; off >>= 1) m = fmax(m, __shfl_down(m, off));
        if ((threadIdx.x & 63) == 0) atomicMax((unsigned long long *)&out[c], (unsigned long long)__double_as_longlong(m));
    }
}

int main() {
    double *d;
    hipMalloc(&d, 9 * sizeof(double));
    hipMemset(d, 0, 9 * sizeof(double));
    hipLaunchKernelGGL(k, dim3(16), dim3(256), 0, 0, d);
    double h[9];
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    const char *names[9] = {"v_rcp_f64 raw", "rcp + 1 Newton step", "rcp + 2 Newton steps", "v_rsq_f64 raw", "sqrt: 1 Goldschmidt step + residual, h as it is",
                            "sqrt: 1 Goldschmidt step + residual, h refined", "sqrt: residual correction only", "rsqrt: one first-order step", "rcp + 1 second-order step (3 instructions)"};
    for (int c = 0; c < 9; c++) printf("%-52s largest relative error %.3g x 2^-53\n", names[c], h[c]);
    return 0;
}
