// valu_rates.hip -- issue cost of the vector instructions the path-tracing kernels are made of, on the GPU at hand.
//
//   hipcc --offload-arch=gfx950 -O2 tools/valu_rates.hip -o tools/valu_rates && tools/valu_rates
//
// For each instruction: a loop of 64 independent copies of it (eight chains, no result feeds the next instruction
// of its own chain sooner than eight instructions later), run by 1, 2 and 4 waves per SIMD on every compute unit;
// printed is the time per wave-instruction and SIMD in cycles of the shader clock as hipDeviceProp reports it.
// The render kernels run four waves per SIMD and are bound by vector issue, so the four-wave column is what an
// instruction costs them.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHK(e)                                                                    \
    do {                                                                          \
        hipError_t r_ = (e);                                                      \
        if (r_ != hipSuccess) {                                                   \
            fprintf(stderr, "%s: %s\n", #e, hipGetErrorString(r_));               \
            exit(1);                                                              \
        }                                                                         \
    } while (0)

// eight chains x eight rounds of one instruction per loop trip
#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
#define ROUNDS8(X) REP8(X) REP8(X) REP8(X) REP8(X) REP8(X) REP8(X) REP8(X) REP8(X)

#define KERNEL(name, decl, body, sink)                                                        \
    __global__ void __launch_bounds__(64) name(int trips, double *out) {                     \
        decl;                                                                                 \
        for (int t = 0; t < trips; t++) {                                                     \
            ROUNDS8(body)                                                                     \
        }                                                                                     \
        if (trips < 0) out[threadIdx.x] = sink;                                               \
    }

#define D_DECL                                                                                          \
    double a[8], x = 1.0 + threadIdx.x * 1e-9, y = 0.999999 + trips * 1e-12;                            \
    for (int q = 0; q < 8; q++) a[q] = q + threadIdx.x
#define F_DECL                                                                                          \
    float a[8], x = 1.0f + threadIdx.x * 1e-6f, y = 0.999f + trips * 1e-9f;                             \
    for (int q = 0; q < 8; q++) a[q] = q + threadIdx.x
#define U_DECL                                                                                          \
    unsigned a[8], x = 0x9E3779B1u + threadIdx.x, y = 12345u + trips;                                   \
    for (int q = 0; q < 8; q++) a[q] = q + threadIdx.x
#define L_DECL                                                                                          \
    unsigned long long a[8];                                                                            \
    unsigned x = 0x9E3779B1u + threadIdx.x, y = 12345u + trips;                                         \
    for (int q = 0; q < 8; q++) a[q] = q + threadIdx.x
#define F2_DECL                                                                                         \
    typedef float f2 __attribute__((ext_vector_type(2)));                                               \
    f2 a[8], x = {1.0f + threadIdx.x * 1e-6f, 1.0f}, y = {0.999f + trips * 1e-9f, 0.5f};                \
    for (int q = 0; q < 8; q++) a[q] = f2{(float)q, (float)threadIdx.x}

#define B_FMA64(i) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a[i]) : "v"(x), "v"(y));
#define B_ADD64(i) asm volatile("v_add_f64 %0, %0, %1" : "+v"(a[i]) : "v"(y));
#define B_MUL64(i) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(a[i]) : "v"(x));
#define B_RSQ64(i) asm volatile("v_rsq_f64 %0, %0" : "+v"(a[i]));
#define B_RCP64(i) asm volatile("v_rcp_f64 %0, %0" : "+v"(a[i]));
#define B_CMP64(i) asm volatile("v_cmp_lt_f64 vcc, %0, %1" : : "v"(a[i]), "v"(x) : "vcc");
#define B_CVT3264(i) asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(((float *)&a[i])[0]) : "v"(a[i]));
#define B_FMA32(i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(x), "v"(y));
#define B_ADD32(i) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(y));
#define B_SQRT32(i) asm volatile("v_sqrt_f32 %0, %0" : "+v"(a[i]));
#define B_RSQ32(i) asm volatile("v_rsq_f32 %0, %0" : "+v"(a[i]));
#define B_CMP32(i) asm volatile("v_cmp_lt_f32 vcc, %0, %1" : : "v"(a[i]), "v"(x) : "vcc");
#define B_PKFMA32(i) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(x), "v"(y));
#define B_MOV(i) asm volatile("v_mov_b32 %0, %1" : "=v"(a[i]) : "v"(x));
#define B_CNDMASK(i) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(x) : "vcc");
#define B_XOR(i) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(a[i]) : "v"(x));
#define B_ADDU(i) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[i]) : "v"(x));
#define B_ADDCO(i) asm volatile("v_add_co_u32 %0, vcc, %0, %1" : "+v"(a[i]) : "v"(x) : "vcc");
#define B_MULLO(i) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a[i]) : "v"(x));
#define B_MULHI(i) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(a[i]) : "v"(x));
#define B_MAD64(i) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(a[i]) : "v"(x), "v"(y) : "vcc");
#define B_SHL64(i) asm volatile("v_lshlrev_b64 %0, 3, %0" : "+v"(a[i]));
#define B_SHR64(i) asm volatile("v_lshrrev_b64 %0, 27, %0" : "+v"(a[i]));
#define B_MAX64(i) asm volatile("v_max_f64 %0, %0, %1" : "+v"(a[i]) : "v"(x));
#define B_CVT6432(i) asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(a[i]) : "v"(((float *)&a[i])[0]));
#define B_CVTU(i) asm volatile("v_cvt_f64_u32 %0, %1" : "=v"(a[i]) : "v"(((unsigned *)&a[i])[0]));
#define B_LDEXP(i) asm volatile("v_ldexp_f64 %0, %0, %1" : "+v"(a[i]) : "v"(((int *)&x)[0]));
#define B_FREXP(i) asm volatile("v_frexp_mant_f64 %0, %0" : "+v"(a[i]));
#define B_FRACT(i) asm volatile("v_fract_f64 %0, %0" : "+v"(a[i]));

KERNEL(k_fma64, D_DECL, B_FMA64, a[0] + a[1] + a[2] + a[3] + a[4] + a[5] + a[6] + a[7])
KERNEL(k_add64, D_DECL, B_ADD64, a[0] + a[1] + a[2] + a[3] + a[4] + a[5] + a[6] + a[7])
KERNEL(k_mul64, D_DECL, B_MUL64, a[0] + a[1] + a[2] + a[3] + a[4] + a[5] + a[6] + a[7])
KERNEL(k_max64, D_DECL, B_MAX64, a[0] + a[1] + a[2] + a[3] + a[4] + a[5] + a[6] + a[7])
KERNEL(k_rsq64, D_DECL, B_RSQ64, a[0] + a[1] + a[2] + a[3] + a[4] + a[5] + a[6] + a[7])
KERNEL(k_rcp64, D_DECL, B_RCP64, a[0] + a[1] + a[2] + a[3] + a[4] + a[5] + a[6] + a[7])
KERNEL(k_cmp64, D_DECL, B_CMP64, a[0] + a[1] + a[2] + a[3] + a[4] + a[5] + a[6] + a[7])
KERNEL(k_cvt3264, D_DECL, B_CVT3264, a[0] + a[1] + a[2] + a[3] + a[4] + a[5] + a[6] + a[7])
KERNEL(k_cvt6432, D_DECL, B_CVT6432, a[0] + a[1] + a[2] + a[3] + a[4] + a[5] + a[6] + a[7])
KERNEL(k_cvtu, D_DECL, B_CVTU, a[0] + a[1] + a[2] + a[3] + a[4] + a[5] + a[6] + a[7])
KERNEL(k_ldexp, D_DECL, B_LDEXP, a[0] + a[1] + a[2] + a[3] + a[4] + a[5] + a[6] + a[7])
KERNEL(k_frexp, D_DECL, B_FREXP, a[0] + a[1] + a[2] + a[3] + a[4] + a[5] + a[6] + a[7])
KERNEL(k_fract, D_DECL, B_FRACT, a[0] + a[1] + a[2] + a[3] + a[4] + a[5] + a[6] + a[7])
KERNEL(k_fma32, F_DECL, B_FMA32, a[0] + a[1] + a[2] + a[3] + a[4] + a[5] + a[6] + a[7])
KERNEL(k_add32, F_DECL, B_ADD32, a[0] + a[1] + a[2] + a[3] + a[4] + a[5] + a[6] + a[7])
KERNEL(k_sqrt32, F_DECL, B_SQRT32, a[0] + a[1] + a[2] + a[3] + a[4] + a[5] + a[6] + a[7])
KERNEL(k_rsq32, F_DECL, B_RSQ32, a[0] + a[1] + a[2] + a[3] + a[4] + a[5] + a[6] + a[7])
KERNEL(k_cmp32, F_DECL, B_CMP32, a[0] + a[1] + a[2] + a[3] + a[4] + a[5] + a[6] + a[7])
KERNEL(k_pkfma32, F2_DECL, B_PKFMA32, (a[0] + a[1] + a[2] + a[3] + a[4] + a[5] + a[6] + a[7]).x)
KERNEL(k_mov, U_DECL, B_MOV, a[0] + a[1] + a[2] + a[3] + a[4] + a[5] + a[6] + a[7])
KERNEL(k_cndmask, U_DECL, B_CNDMASK, a[0] + a[1] + a[2] + a[3] + a[4] + a[5] + a[6] + a[7])
KERNEL(k_xor, U_DECL, B_XOR, a[0] + a[1] + a[2] + a[3] + a[4] + a[5] + a[6] + a[7])
KERNEL(k_addu, U_DECL, B_ADDU, a[0] + a[1] + a[2] + a[3] + a[4] + a[5] + a[6] + a[7])
KERNEL(k_addco, U_DECL, B_ADDCO, a[0] + a[1] + a[2] + a[3] + a[4] + a[5] + a[6] + a[7])
KERNEL(k_mullo, U_DECL, B_MULLO, a[0] + a[1] + a[2] + a[3] + a[4] + a[5] + a[6] + a[7])
KERNEL(k_mulhi, U_DECL, B_MULHI, a[0] + a[1] + a[2] + a[3] + a[4] + a[5] + a[6] + a[7])
KERNEL(k_mad64, L_DECL, B_MAD64, (double)(a[0] + a[1] + a[2] + a[3] + a[4] + a[5] + a[6] + a[7]))
KERNEL(k_shl64, L_DECL, B_SHL64, (double)(a[0] + a[1] + a[2] + a[3] + a[4] + a[5] + a[6] + a[7]))
KERNEL(k_shr64, L_DECL, B_SHR64, (double)(a[0] + a[1] + a[2] + a[3] + a[4] + a[5] + a[6] + a[7]))

typedef void (*kern_t)(int, double *);
struct Entry {
    const char *name;
    kern_t k;
};

int main() {
    hipDeviceProp_t prop;
    CHK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    const double ghz = prop.clockRate * 1e-6;
    printf("%s: %d compute units, %.2f GHz\n", prop.name, cus, ghz);
    double *out;
    CHK(hipMalloc(&out, 64 * sizeof(double)));
    const Entry es[] = {{"v_fma_f64", k_fma64},         {"v_add_f64", k_add64},       {"v_mul_f64", k_mul64},
                        {"v_max_f64", k_max64},         {"v_rsq_f64", k_rsq64},       {"v_rcp_f64", k_rcp64},
                        {"v_cmp_lt_f64", k_cmp64},      {"v_cvt_f32_f64", k_cvt3264}, {"v_cvt_f64_f32", k_cvt6432},
                        {"v_cvt_f64_u32", k_cvtu},      {"v_ldexp_f64", k_ldexp},     {"v_frexp_mant_f64", k_frexp},
                        {"v_fract_f64", k_fract},       {"v_fma_f32", k_fma32},       {"v_add_f32", k_add32},
                        {"v_sqrt_f32", k_sqrt32},       {"v_rsq_f32", k_rsq32},       {"v_cmp_lt_f32", k_cmp32},
                        {"v_pk_fma_f32", k_pkfma32},    {"v_mov_b32", k_mov},         {"v_cndmask_b32", k_cndmask},
                        {"v_xor_b32", k_xor},           {"v_add_u32", k_addu},        {"v_add_co_u32", k_addco},
                        {"v_mul_lo_u32", k_mullo},      {"v_mul_hi_u32", k_mulhi},    {"v_mad_u64_u32", k_mad64},
                        {"v_lshlrev_b64", k_shl64},     {"v_lshrrev_b64", k_shr64}};
    hipEvent_t e0, e1;
    CHK(hipEventCreate(&e0));
    CHK(hipEventCreate(&e1));
    const int trips = 2000;
    printf("%-18s %8s %8s %8s   cycles per wave-instruction and SIMD at 1, 2, 4 waves per SIMD\n", "instruction", "1", "2", "4");
    for (const Entry &e : es) {
        printf("%-18s", e.name);
        for (int waves : {1, 2, 4}) {
            const int grid = cus * 4 * waves;
            hipLaunchKernelGGL(e.k, dim3(grid), dim3(64), 0, 0, 10, out);  // warm-up
            CHK(hipEventRecord(e0, 0));
            hipLaunchKernelGGL(e.k, dim3(grid), dim3(64), 0, 0, trips, out);
            CHK(hipEventRecord(e1, 0));
            CHK(hipEventSynchronize(e1));
            float ms = 0;
            CHK(hipEventElapsedTime(&ms, e0, e1));
            const double cyc = ms * 1e-3 * ghz * 1e9 / ((double)trips * 64 * waves);
            printf(" %8.2f", cyc);
        }
        printf("\n");
    }
    return 0;
}
