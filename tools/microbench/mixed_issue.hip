// mixed_issue.hip — does an SALU / LDS / VMEM instruction of ANOTHER wave cost a SIMD a VALU issue slot on gfx950?
//   build: hipcc -O3 --offload-arch=gfx950 mixed_issue.hip -o bin/mixed_issue ; run on the GPU box.
//
// One workgroup of 256·W threads per CU (W = waves per SIMD, 2 … 4): a workgroup's waves go to the SIMDs in cyclic order, so
// the waves w, w + 4, w + 8, … share a SIMD and "slot" s = wave / 4 gives every SIMD one wave of each slot.  Each slot runs a
// ROLE (an instruction class, 16 independent chains or ONE dependent chain); slot 0 is the measured one: its own s_memtime
// around its loop / its instruction count = cycles per instruction of that wave, while the other slots run their roles for
// longer than slot 0 does (they spin until slot 0's waves have all finished).  The table printed per configuration is the
// measured wave's cycles per instruction and the partner slots' own rates over the same interval.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <vector>
#include <string>
typedef float v2f __attribute__((ext_vector_type(2)));
enum Role { IDLE = 0, VALU_I, VALU_D, SALU_I, SALU_D, LDS_RD, PK_I, F64_I, MIX_VS, MIX_VSL, VMEM_RD, LDS_WR, VALU_T, VALU_WAIT, VALU_NOP, VALU_BR, N_ROLES };
static const char* role_name[N_ROLES] = {"idle", "valu x16", "valu dep", "salu x16", "salu dep", "ds_read_b64 x16", "v_pk_fma x16", "v_fma_f64 x16",
                                         "V,S interleaved", "V,S,V,L interleaved", "global_load x8", "ds_write_b64 x16", "valu 3-src x16",
                                         "valu + s_waitcnt (per valu)", "valu + s_nop 0 (per valu)", "valu + s_cbranch nt (per valu)"};
struct Args { int role[4]; int iters0; unsigned long long* out; unsigned int* done; const float* gmem; };
#define REP16(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15)
#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)

__device__ __forceinline__ unsigned long long now() { unsigned long long t; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)); return t; }

// one pass of a role's loop body; returns the number of instructions of the role's class it issued
template <int ROLE> __device__ __forceinline__ int body(v2f (&a)[16], double (&d)[16], int (&s)[16], const v2f b, const int sb, float* lds, const float* g, int lane) {
#define V_ADD(i) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i].x) : "v"(b.x));
#define V_ADD_D(i) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[0].x) : "v"(b.x));
#define V_FMA3(i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i].x) : "v"(b.x), "v"(b.y));
#define S_ADD(i) asm volatile("s_add_u32 %0, %0, %1" : "+s"(s[i]) : "s"(sb) : "scc");
#define S_ADD_D(i) asm volatile("s_add_u32 %0, %0, %1" : "+s"(s[0]) : "s"(sb) : "scc");
#define PK_FMA(i) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(a[i]) : "v"(b));
#define D_FMA(i) asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(d[i]) : "v"(d[15]));
#define L_RD(i) asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(a[i]) : "v"(lane * 8), "i"(i * 512));
#define L_WR(i) asm volatile("ds_write_b64 %0, %1 offset:%2" : : "v"(lane * 8), "v"(a[i]), "i"(i * 512) : "memory");
#define G_RD(i) asm volatile("global_load_dwordx2 %0, %1, off offset:%2" : "=v"(a[i]) : "v"(g + lane * 2), "i"(i * 512));
#define V_WAIT(i) V_ADD(i) asm volatile("s_waitcnt lgkmcnt(0)");
#define V_NOP(i) V_ADD(i) asm volatile("s_nop 0");
#define V_BR(i) V_ADD(i) asm volatile("s_cbranch_execz 1f\n1:" ::: "memory");
#define VS(i) V_ADD(i) S_ADD(i)
#define VSVL(i) V_ADD(i) S_ADD(i) V_FMA3(i)
    if (ROLE == VALU_I) { REP16(V_ADD) return 16; }
    if (ROLE == VALU_D) { REP16(V_ADD_D) return 16; }
    if (ROLE == VALU_T) { REP16(V_FMA3) return 16; }
    if (ROLE == VALU_WAIT) { REP16(V_WAIT) return 16; }        // (cycles are quoted per VALU instruction: the gap to "valu x16" is what the s_waitcnt / s_nop / untaken branch costs)
    if (ROLE == VALU_NOP) { REP16(V_NOP) return 16; }
    if (ROLE == VALU_BR) { REP16(V_BR) return 16; }
    if (ROLE == SALU_I) { REP16(S_ADD) return 16; }
    if (ROLE == SALU_D) { REP16(S_ADD_D) return 16; }
    if (ROLE == PK_I) { REP16(PK_FMA) return 16; }
    if (ROLE == F64_I) { REP8(D_FMA) REP8(D_FMA) return 16; }
    if (ROLE == LDS_RD) { REP16(L_RD) asm volatile("s_waitcnt lgkmcnt(0)"); return 16; }
    if (ROLE == LDS_WR) { REP16(L_WR) asm volatile("s_waitcnt lgkmcnt(0)"); return 16; }
    if (ROLE == VMEM_RD) { REP8(G_RD) asm volatile("s_waitcnt vmcnt(0)"); return 8; }
    if (ROLE == MIX_VS) { REP16(VS) return 32; }
    if (ROLE == MIX_VSL) { REP8(VSVL) asm volatile("ds_read_b64 %0, %1" : "=v"(a[15]) : "v"(lane * 8)); REP8(VSVL) asm volatile("s_waitcnt lgkmcnt(0)"); return 49; }
    return 0;
}

template <int ROLE> __device__ void run_role(const Args& A, int slot, int wave, float* lds, float* lds_base) {
    const int lane = threadIdx.x & 63;
    v2f a[16]; double d[16]; int s[16];
    v2f b; b.x = 1.0001f; b.y = 0.9999f;
    for (int i = 0; i < 16; i++) { a[i].x = lane * 0.001f + i; a[i].y = i * 0.5f; d[i] = 1.0 + 1e-9 * (i + lane); s[i] = i + (int)blockIdx.x; }
    const int sb = __builtin_amdgcn_readfirstlane(A.iters0 & 63);
    unsigned long long n = 0;
    const unsigned long long t0 = now();
    if (slot == 0) {
        for (int it = 0; it < A.iters0; it++) n += body<ROLE>(a, d, s, b, sb, lds, A.gmem, lane);
    } else {
        // partner slots keep going until every measured wave of this workgroup is through (checked every 16 passes through LDS)
        volatile unsigned int* flag = reinterpret_cast<volatile unsigned int*>(lds_base + 64 * 16 * 2 * 16);
        for (int guard = 0; guard < 100000; guard++) {      // (bounded: a lost flag must not hang the box)
            for (int k = 0; k < 16; k++) n += body<ROLE>(a, d, s, b, sb, lds, A.gmem, lane);
            if (*flag >= 4u) break;
        }
    }
    const unsigned long long t1 = now();
    if (slot == 0 && lane == 0) atomicAdd(reinterpret_cast<unsigned int*>(lds_base + 64 * 16 * 2 * 16), 1u);
    float r = 0; for (int i = 0; i < 16; i++) r += a[i].x + a[i].y + (float)d[i] + (float)s[i];
    if (lane == 0) { A.out[((size_t)blockIdx.x * 16 + wave) * 2] = t1 - t0; A.out[((size_t)blockIdx.x * 16 + wave) * 2 + 1] = n; }
    if (r == 12345.678f) A.out[0] = 0;     // keeps the chains alive
}

__global__ __launch_bounds__(1024) void k(Args A) {
    extern __shared__ float lds[];
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), slot = wave >> 2;
    if (threadIdx.x == 0) reinterpret_cast<unsigned int*>(lds + 64 * 16 * 2 * 16)[0] = 0u;
    for (int i = threadIdx.x; i < 16 * 64 * 16 * 2; i += blockDim.x) lds[i] = i;
    __syncthreads();
    float* my = lds + (size_t)wave * 64 * 16 * 2;
    switch (A.role[slot]) {
#define CASE(R) case R: run_role<R>(A, slot, wave, my, lds); break;
        CASE(VALU_I) CASE(VALU_D) CASE(SALU_I) CASE(SALU_D) CASE(LDS_RD) CASE(PK_I) CASE(F64_I) CASE(MIX_VS) CASE(MIX_VSL) CASE(VMEM_RD) CASE(LDS_WR) CASE(VALU_T) CASE(VALU_WAIT) CASE(VALU_NOP) CASE(VALU_BR)
        default: break;
    }
}

int main(int argc, char** argv) {
    int dev_cus = 256;
    hipDeviceProp_t pr; hipGetDeviceProperties(&pr, 0); dev_cus = pr.multiProcessorCount;
    unsigned long long* out; hipMalloc(&out, (size_t)dev_cus * 16 * 2 * 8);
    float* g; hipMalloc(&g, 1 << 20); hipMemset(g, 0, 1 << 20);
    std::vector<unsigned long long> h((size_t)dev_cus * 16 * 2);
    auto run = [&](std::vector<int> roles) {
        Args A; memset(&A, 0, sizeof A);
        const int W = (int)roles.size();
        for (int i = 0; i < W; i++) A.role[i] = roles[i];
        A.iters0 = 4000; A.out = out; A.gmem = g;
        const size_t shm = (size_t)16 * 64 * 16 * 2 * 4 + 16;             // 128 KB + flag: ONE workgroup per CU
        hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm);
        hipMemset(out, 0, h.size() * 8);
        hipLaunchKernelGGL(k, dim3(dev_cus), dim3(256 * W), shm, 0, A);
        hipDeviceSynchronize();
        hipMemcpy(h.data(), out, h.size() * 8, hipMemcpyDeviceToHost);
        std::string line;
        for (int s = 0; s < W; s++) {
            double cyc = 0, n = 0;
            for (int b = 0; b < dev_cus; b++) for (int w = 4 * s; w < 4 * s + 4; w++) { cyc += (double)h[((size_t)b * 16 + w) * 2]; n += (double)h[((size_t)b * 16 + w) * 2 + 1]; }
            char buf[128];
            snprintf(buf, sizeof buf, "%s%-20s %6.2f", s ? " | " : "", role_name[roles[s]], n > 0 ? cyc / n : 0.0);
            line += buf;
        }
        printf("%d waves/SIMD: %s   (cycles per own instruction, slot 0 = measured wave)\n", W, line.c_str()); fflush(stdout);
    };
    if (argc > 1 && !strcmp(argv[1], "quick")) {
        for (int me : {VALU_I, VALU_WAIT, VALU_NOP, VALU_BR}) { run({me}); run({me, VALU_I}); run({me, VALU_I, VALU_I, VALU_I}); run({me, me, me, me}); printf("\n"); }
        return 0;
    }
    for (int me : {VALU_I, VALU_D, VALU_T, PK_I, F64_I, MIX_VS, MIX_VSL}) {
        run({me});
        for (int other : {IDLE, VALU_I, VALU_D, SALU_I, SALU_D, LDS_RD, LDS_WR, VMEM_RD, MIX_VS}) {
            if (other == IDLE) continue;
            run({me, other});
        }
        run({me, SALU_I, LDS_RD});
        run({me, VALU_I, SALU_I});
        run({me, VALU_I, SALU_I, LDS_RD});
        run({me, VALU_I, VALU_I, VALU_I});
        run({me, MIX_VS, MIX_VS, MIX_VS});
        run({me, SALU_I, SALU_I, SALU_I});
        printf("\n"); fflush(stdout);
    }
    for (int me : {SALU_I, SALU_D, LDS_RD}) {
        run({me});
        run({me, VALU_I}); run({me, SALU_I}); run({me, LDS_RD}); run({me, VALU_I, VALU_I, VALU_I});
        printf("\n"); fflush(stdout);
    }
    return 0;
}
