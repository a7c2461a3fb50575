#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v2f __attribute__((ext_vector_type(2)));
#define REP16(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15)
template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters) {
    v2f a[16]; v2f b; b.x = 1.0001f; b.y = 0.9999f;
    for (int i = 0; i < 16; i++) { a[i].x = threadIdx.x * 0.001f + i; a[i].y = i * 0.5f; }
    for (int it = 0; it < iters; it++) {
#define PK_PLAIN(i) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
#define PK_SWZ(i) asm volatile("v_pk_add_f32 %0, %0, %1 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]" : "+v"(a[i]) : "v"(b));
#define PK_FMA(i) asm volatile("v_pk_fma_f32 %0, %0, %1, %1 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[1,0,0]" : "+v"(a[i]) : "v"(b));
#define PK_FMA_PLAIN(i) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(a[i]) : "v"(b));
#define SC_ADD(i) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i].x) : "v"(b.x)); asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i].y) : "v"(b.y));
#define SC_FMA(i) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(a[i].x) : "v"(b.x)); asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(a[i].y) : "v"(b.y));
#define PK_MUL(i) asm volatile("v_pk_mul_f32 %0, %0, %1 op_sel:[0,0] op_sel_hi:[0,1]" : "+v"(a[i]) : "v"(b));
        if (MODE == 0) { REP16(PK_PLAIN) }
        if (MODE == 1) { REP16(PK_SWZ) }
        if (MODE == 2) { REP16(PK_FMA) }
        if (MODE == 3) { REP16(SC_ADD) }
        if (MODE == 4) { REP16(SC_FMA) }
        if (MODE == 5) { REP16(PK_MUL) }
        if (MODE == 6) { REP16(PK_FMA_PLAIN) }
    }
    float s = 0; for (int i = 0; i < 16; i++) s += a[i].x + a[i].y;
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int MODE> void run(const char* name, float* d, int ops_per_iter) {
    const int iters = 4000, blocks = 256 * 8;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, 10);
    hipEventRecord(e0); hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, iters); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double winst = (double)blocks * 4 * iters * ops_per_iter;     // wave-instructions
    printf("%-14s %8.3f ms  %.2f cycles per wave-instruction per SIMD (at 2.4 GHz, 1024 SIMDs)\n", name, ms, ms * 1e-3 * 2.4e9 * 1024 / winst);
}
int main() {
    float* d; hipMalloc(&d, 256 * 8 * 256 * 4);
    run<0>("pk_add plain", d, 16); run<1>("pk_add swz", d, 16); run<2>("pk_fma swz", d, 16); run<6>("pk_fma plain", d, 16);
    run<5>("pk_mul swz", d, 16); run<3>("2x v_add", d, 32); run<4>("2x v_fma", d, 32);
    return 0;
}
