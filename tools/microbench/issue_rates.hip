// issue_rates.hip — per-SIMD issue cost of the instruction classes the pipeline is made of (MI355X).
// build: hipcc -O3 --offload-arch=gfx950 issue_rates.hip -o bin/issue_rates ; run on the GPU box.
// Every mode runs 16 independent chains per wave so that dependent-issue latency is hidden; `waves` per SIMD varies.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v2f __attribute__((ext_vector_type(2)));
#define REP16(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15)
template <int MODE>
__global__ __launch_bounds__(64) void k(float* out, int iters) {
    __shared__ float lds[64 * 17];
    v2f a[16]; v2f b; b.x = 1.0001f; b.y = 0.9999f;
    double d[16]; double db = 1.000001;
    int s[16];
    for (int i = 0; i < 16; i++) { a[i].x = threadIdx.x * 0.001f + i; a[i].y = i * 0.5f; d[i] = i + threadIdx.x; s[i] = i + (int)blockIdx.x; lds[threadIdx.x * 17 + i] = i; }
    int sidx = __builtin_amdgcn_readfirstlane(iters & 63);
    for (int it = 0; it < iters; it++) {
#define V_FMA(i) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(a[i].x) : "v"(b.x));
#define V_ADD(i) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i].x) : "v"(b.x));
#define PK_FMA(i) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(a[i]) : "v"(b));
#define PK_ADD(i) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
#define D_FMA(i) asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(d[i]) : "v"(db));
#define D_ADD(i) asm volatile("v_add_f64 %0, %0, %1" : "+v"(d[i]) : "v"(db));
#define D_MUL(i) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(d[i]) : "v"(db));
#define S_ADD(i) asm volatile("s_add_u32 %0, %0, %1" : "+s"(s[i]) : "s"(sidx));
#define V_RDL(i) asm volatile("v_readlane_b32 %0, %1, %2" : "=s"(s[i]) : "v"(a[i].x), "s"(sidx));
#define V_DPP(i) asm volatile("v_add_f32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a[i].x));
#define V_BPERM(i) asm volatile("ds_bpermute_b32 %0, %1, %0\n\ts_waitcnt lgkmcnt(0)" : "+v"(a[i].x) : "v"(s[0]));
#define V_IADD(i) asm volatile("v_add_u32 %0, %0, %1" : "+v"(s[i]) : "v"(s[(i + 1) & 15]));
#define V_CMP(i) asm volatile("v_cmp_gt_u32 vcc, %0, %1" : : "v"(a[i].x), "v"(b.x) : "vcc");
#define LDS_RD(i) asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(a[i].x) : "v"(threadIdx.x * 68), "i"(i * 4));
#define V_LOG(i) asm volatile("v_log_f32 %0, %0" : "+v"(a[i].x));
#define V_CVT(i) asm volatile("v_cvt_f64_u32 %0, %1" : "=v"(d[i]) : "v"(s[i]));
#define V_SWAP32(i) asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(a[i].x), "+v"(a[(i + 1) & 15].y));
#define V_SWAP16(i) asm volatile("v_permlane16_swap_b32 %0, %1" : "+v"(a[i].x), "+v"(a[(i + 1) & 15].y));
#define V_ROR8(i) asm volatile("v_mov_b32_dpp %0, %1 row_ror:8 row_mask:0xf bank_mask:0xc" : "+v"(a[i].x) : "v"(a[(i + 1) & 15].y));
#define V_CNDM(i) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i].x) : "v"(b.x) : "vcc");
#define LDS_WR64(i) asm volatile("ds_write_b64 %0, %1 offset:%2" : : "v"(threadIdx.x * 8), "v"(a[i]), "i"(i * 512) : "memory");
#define LDS_RD64(i) asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(a[i]) : "v"(threadIdx.x * 8), "i"(i * 512));
        if (MODE == 16) { REP16(V_SWAP32) }
        if (MODE == 17) { REP16(V_SWAP16) }
        if (MODE == 18) { REP16(V_ROR8) }
        if (MODE == 19) { REP16(V_CNDM) }
        if (MODE == 0) { REP16(V_FMA) }
        if (MODE == 1) { REP16(V_ADD) }
        if (MODE == 2) { REP16(PK_FMA) }
        if (MODE == 3) { REP16(PK_ADD) }
        if (MODE == 4) { REP16(D_FMA) }
        if (MODE == 5) { REP16(D_ADD) }
        if (MODE == 6) { REP16(S_ADD) }
        if (MODE == 7) { REP16(V_RDL) }
        if (MODE == 8) { REP16(V_DPP) }
        if (MODE == 9) { REP16(V_BPERM) }
        if (MODE == 10) { REP16(V_IADD) }
        if (MODE == 11) { REP16(V_FMA) REP16(S_ADD) }          // VALU + SALU interleaved in one wave
        if (MODE == 12) { REP16(LDS_RD) asm volatile("s_waitcnt lgkmcnt(0)"); }
        if (MODE == 13) { REP16(D_MUL) }
        if (MODE == 14) { REP16(V_LOG) }
        if (MODE == 15) { REP16(V_CVT) }
    }
    float r = 0; for (int i = 0; i < 16; i++) r += a[i].x + a[i].y + (float)d[i] + (float)s[i];
    out[blockIdx.x * 64 + threadIdx.x] = r + lds[threadIdx.x];
}
template <int MODE> void run(const char* name, float* d, int ops_per_iter, int waves_per_simd) {
    const int iters = 2000, blocks = 256 * 4 * waves_per_simd;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(64), 0, 0, d, 10);
    hipEventRecord(e0); hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(64), 0, 0, d, iters); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double winst = (double)blocks * iters * ops_per_iter;     // wave-instructions
    printf("%-22s w/SIMD %d  %8.3f ms  %6.2f ns/1k-inst/SIMD  %.2f cycles per wave-instruction per SIMD (2.4 GHz)\n", name, waves_per_simd, ms, ms * 1e6 * 1024 / winst * 1e3 / 1e3, ms * 1e-3 * 2.4e9 * 1024 / winst);
}
int main() {
    float* d; hipMalloc(&d, 256 * 4 * 8 * 64 * 4);
    for (int w : {1, 2, 4, 8}) {
        run<0>("v_fma_f32", d, 16, w); run<1>("v_add_f32", d, 16, w); run<2>("v_pk_fma_f32", d, 16, w); run<3>("v_pk_add_f32", d, 16, w);
        run<4>("v_fma_f64", d, 16, w); run<5>("v_add_f64", d, 16, w); run<13>("v_mul_f64", d, 16, w); run<6>("s_add_u32", d, 16, w); run<7>("v_readlane_b32", d, 16, w);
        run<8>("v_add_f32_dpp", d, 16, w); run<9>("ds_bpermute+wait", d, 16, w); run<10>("v_add_u32", d, 16, w); run<11>("v_fma+s_add (32)", d, 32, w);
        run<12>("ds_read_b32 x16+wait", d, 16, w); run<14>("v_log_f32", d, 16, w); run<15>("v_cvt_f64_u32", d, 16, w);
        run<16>("v_permlane32_swap", d, 16, w); run<17>("v_permlane16_swap", d, 16, w); run<18>("v_mov_dpp row_ror:8 bank", d, 16, w); run<19>("v_cndmask", d, 16, w);
    }
    return 0;
}
