// fe_common.hpp — device helpers shared by the front-end kernels (frontend.hip: K1, fused.hip: K1 + peak scan in one launch):
// the FE-1 arithmetic as packed-fp32 operations and the radix-8 butterfly.
#pragma once
#include "wsa_internal.hpp"
#include "wave_ops.hpp"

namespace wsa {

struct __attribute__((packed, aligned(4))) pcm2 { float x, y; };

__device__ __forceinline__ float2 cmul(float2 x, float2 w) {        // FE-1 generic complex multiply
    float2 y;
    y.x = __builtin_fmaf(-x.y, w.y, x.x * w.x);
    y.y = __builtin_fmaf(x.y, w.x, x.x * w.y);
    return y;
}
__device__ __forceinline__ float2 cadd(float2 a, float2 b) { return make_float2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ float2 csub(float2 a, float2 b) { return make_float2(a.x - b.x, a.y - b.y); }
__device__ __forceinline__ float2 mul_mi(float2 t) { return make_float2(t.y, -t.x); }                 // * (-i)
__device__ __forceinline__ float2 mul_w8(float2 t) {                                                   // * W8
    const float s = 0.70710678118654752440f;
    return make_float2(s * (t.x + t.y), s * (t.y - t.x));
}
__device__ __forceinline__ float2 mul_w83(float2 t) {                                                  // * W8^3
    const float s = 0.70710678118654752440f;
    return make_float2(s * (t.y - t.x), -(s * (t.x + t.y)));
}

// radix-8 DIF butterfly = three radix-2 stages (FE-1); result returned in natural order.
// NZ = number of leading non-zero inputs (inputs >= NZ are exactly zero and pruned).
template <int NZ>
__device__ __forceinline__ void radix8(float2 (&v)[8]) {
    float2 a0, a1, a2, a3, b0, b1, b2, b3;
    if (NZ > 4) {
        a0 = cadd(v[0], v[4]); b0 = csub(v[0], v[4]);
        a1 = cadd(v[1], v[5]); b1 = mul_w8(csub(v[1], v[5]));
        a2 = cadd(v[2], v[6]); b2 = mul_mi(csub(v[2], v[6]));
        a3 = cadd(v[3], v[7]); b3 = mul_w83(csub(v[3], v[7]));
    } else {           // v[4..7] == 0: u + 0 = u, (u - 0) * W = u * W
        a0 = v[0]; b0 = v[0];
        a1 = v[1]; b1 = mul_w8(v[1]);
        a2 = v[2]; b2 = mul_mi(v[2]);
        a3 = v[3]; b3 = mul_w83(v[3]);
    }
    // stage h = 2 on (a0..a3) and (b0..b3)
    float2 c0 = cadd(a0, a2), c2 = csub(a0, a2);
    float2 c1 = cadd(a1, a3), c3 = mul_mi(csub(a1, a3));
    float2 d0 = cadd(b0, b2), d2 = csub(b0, b2);
    float2 d1 = cadd(b1, b3), d3 = mul_mi(csub(b1, b3));
    // stage h = 1; bit-reversed positions -> natural order: out[k] = pos[bitrev3(k)]
    v[0] = cadd(c0, c1); v[4] = csub(c0, c1);      // pos 0,1 -> k 0,4
    v[2] = cadd(c2, c3); v[6] = csub(c2, c3);      // pos 2,3 -> k 2,6
    v[1] = cadd(d0, d1); v[5] = csub(d0, d1);      // pos 4,5 -> k 1,5
    v[3] = cadd(d2, d3); v[7] = csub(d2, d3);      // pos 6,7 -> k 3,7
}

// lanes of one wave exchange through LDS: hardware runs a wave's LDS instructions in order, so only
// the compiler has to be kept from moving accesses across (a fence would also drain the PCM prefetch)
__device__ __forceinline__ void wave_lds_sync() { wsync(); }

typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
// a 64-bit value every lane holds the same of, moved into scalar registers
__device__ __forceinline__ uint64_t uniform_u64(uint64_t v) {
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)v), hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(v >> 32));
    return ((uint64_t)hi << 32) | lo;
}

__device__ __forceinline__ uint32_t to_u32(float x) {               // FE-1 F8: trunc, saturate, NaN -> 0
    // exactly what v_cvt_u32_f32 does (rounds toward zero, clamps out-of-range values and infinities to 0 / 0xffffffff, NaN -> 0); written as the
    // instruction because the C++ conversion of an out-of-range value is undefined and the explicit tests cost two exec-mask branches per band
    uint32_t r; asm("v_cvt_u32_f32 %0, %1" : "=v"(r) : "v"(x)); return r;
}
// d = lane's bit in mask ? t : f, the mask in scalar registers (per-lane loop invariants cost no vector register and no branch this way)
__device__ __forceinline__ float sel_mask(float f, float t, uint64_t mask) { float d; asm("v_cndmask_b32 %0, %1, %2, %3" : "=v"(d) : "v"(f), "v"(t), "s"(mask)); return d; }

// ---- packed-fp32 forms (VOP3P on gfx950: two fp32 lanes per instruction; op_sel / op_sel_hi pick the low or high
// half of each source for the low / high result, neg_lo / neg_hi negate a source for that half).  Written out
// because the compiler spends ~85 register moves per frame arranging operands for the packed adds it forms
// itself; the swizzles below are free.  Every lane operation is a single IEEE fp32 add / mul / fma, i.e. exactly
// the FE-1 operation sequence (a negated operand is exact, -i is a swap + sign folded into the next add).
typedef float v2f __attribute__((ext_vector_type(2)));
#ifndef WSA_FE_PK_ASM
#define WSA_FE_PK_ASM 1
#endif
#if WSA_FE_PK_ASM
__device__ __forceinline__ v2f pk_add(v2f a, v2f b) { v2f d; asm("v_pk_add_f32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b)); return d; }
__device__ __forceinline__ v2f pk_sub(v2f a, v2f b) { v2f d; asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(d) : "v"(a), "v"(b)); return d; }
__device__ __forceinline__ v2f pk_add_mi(v2f a, v2f b) { v2f d; asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]" : "=v"(d) : "v"(a), "v"(b)); return d; }
__device__ __forceinline__ v2f pk_sub_mi(v2f a, v2f b) { v2f d; asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]" : "=v"(d) : "v"(a), "v"(b)); return d; }
__device__ __forceinline__ v2f pk_mul_w8(v2f t, v2f ss) {
    v2f u, d;
    asm("v_pk_add_f32 %0, %1, %1 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]" : "=v"(u) : "v"(t));
    asm("v_pk_mul_f32 %0, %1, %2" : "=v"(d) : "v"(u), "v"(ss));
    return d;
}
__device__ __forceinline__ v2f pk_mul_w83(v2f t, v2f ss) {
    v2f u, d;
    asm("v_pk_add_f32 %0, %1, %1 op_sel:[1,0] op_sel_hi:[0,1] neg_lo:[0,1]" : "=v"(u) : "v"(t));
    asm("v_pk_mul_f32 %0, %1, %2 neg_hi:[0,1]" : "=v"(d) : "v"(u), "v"(ss));
    return d;
}
__device__ __forceinline__ v2f pk_cmul(v2f x, v2f w) {
    v2f t, y;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[0,1]" : "=v"(t) : "v"(x), "v"(w));
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[1,0,0]" : "=v"(y) : "v"(x), "v"(w), "v"(t));
    return y;
}
__device__ __forceinline__ v2f pk_mul(v2f a, v2f b) { v2f d; asm("v_pk_mul_f32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b)); return d; }
__device__ __forceinline__ v2f pk_fma(v2f a, v2f b, v2f c) { v2f d; asm("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c)); return d; }
__device__ __forceinline__ v2f pk_add_conj(v2f a, v2f b) { v2f d; asm("v_pk_add_f32 %0, %1, %2 neg_hi:[0,1]" : "=v"(d) : "v"(a), "v"(b)); return d; }
__device__ __forceinline__ v2f pk_sub_conj(v2f a, v2f b) { v2f d; asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1]" : "=v"(d) : "v"(a), "v"(b)); return d; }
#else
// the same operations in plain C++ (the compiler picks packed or scalar forms and schedules them itself)
__device__ __forceinline__ v2f mk2(float x, float y) { v2f d; d.x = x; d.y = y; return d; }
__device__ __forceinline__ v2f pk_add(v2f a, v2f b) { return mk2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ v2f pk_sub(v2f a, v2f b) { return mk2(a.x - b.x, a.y - b.y); }
__device__ __forceinline__ v2f pk_add_mi(v2f a, v2f b) { return mk2(a.x + b.y, a.y - b.x); }
__device__ __forceinline__ v2f pk_sub_mi(v2f a, v2f b) { return mk2(a.x - b.y, a.y + b.x); }
__device__ __forceinline__ v2f pk_mul_w8(v2f t, v2f ss) { return mk2(ss.x * (t.x + t.y), ss.y * (t.y - t.x)); }
__device__ __forceinline__ v2f pk_mul_w83(v2f t, v2f ss) { return mk2(ss.x * (t.y - t.x), -(ss.y * (t.x + t.y))); }
__device__ __forceinline__ v2f pk_cmul(v2f x, v2f w) { return mk2(__builtin_fmaf(-x.y, w.y, x.x * w.x), __builtin_fmaf(x.y, w.x, x.x * w.y)); }
__device__ __forceinline__ v2f pk_mul(v2f a, v2f b) { return mk2(a.x * b.x, a.y * b.y); }
__device__ __forceinline__ v2f pk_fma(v2f a, v2f b, v2f c) { return mk2(__builtin_fmaf(a.x, b.x, c.x), __builtin_fmaf(a.y, b.y, c.y)); }
__device__ __forceinline__ v2f pk_add_conj(v2f a, v2f b) { return mk2(a.x + b.x, a.y - b.y); }
__device__ __forceinline__ v2f pk_sub_conj(v2f a, v2f b) { return mk2(a.x - b.x, a.y + b.y); }
#endif
__device__ __forceinline__ v2f to_v2f(float2 a) { v2f d; d.x = a.x; d.y = a.y; return d; }

// radix-8 DIF butterfly on packed values, same stages and operation order as radix8<NZ> above; the -i twiddles
// of stages 1 and 2 are folded into the adds of the following stage.
// One asm block per butterfly (fe_blocks.inc, generated and checked by tools/gen/fe_blocks.py): between separate asm statements the compiler's hazard recognizer puts an `s_nop` in front of every
// statement that reads a register an asm statement wrote (an asm statement has unknown length, so the producer always counts as
// "zero wait states ago", and an asm result is assumed to need the wait state of a dst_sel write) — 45 of them per frame.  Inside a
// block the order is ours (consumers sit a few instructions behind their producers) and the registers rotate: a result takes the
// place of an operand that dies there, so a butterfly needs two registers on top of its eight values.
#if WSA_FE_PK_ASM
#include "fe_blocks.inc"
template <int NZ>
__device__ __forceinline__ void radix8_pk(v2f (&v)[8], const v2f ss) {
    if (NZ > 4) WSA_R8_FULL(v, ss); else WSA_R8_HALF(v, ss);
}
// v[k] *= w[k], k = 1 .. 7 (the inter-pass twiddles)
__device__ __forceinline__ void pk_cmul7(v2f (&v)[8], const v2f (&w)[8]) { WSA_CMUL7(v, w); }
// real-FFT split + power of five rows at once: pw[c] = 4 |X[k]|^2 from za = Z[k], zb = Z[512 - k], w = W_1024^k
__device__ __forceinline__ void pk_split5(const v2f (&za)[5], const v2f (&zb)[5], const v2f (&w)[5], float (&pw)[5]) { WSA_SPLIT5(za, zb, w, pw); }
#else
__device__ __forceinline__ void pk_split5(const v2f (&za)[5], const v2f (&zb)[5], const v2f (&w)[5], float (&pw)[5]) {
#pragma unroll
    for (int c = 0; c < 5; c++) {
        const v2f e = pk_add_conj(za[c], zb[c]), o = pk_sub_conj(za[c], zb[c]);
        const v2f t = pk_cmul(o, w[c]);
        const v2f xx = pk_add_mi(e, t);
        pw[c] = __builtin_fmaf(xx.x, xx.x, xx.y * xx.y);
    }
}
template <int NZ>
__device__ __forceinline__ void radix8_pk(v2f (&v)[8], const v2f ss) {
    v2f a0, a1, a2, a3, b0, b1, t2, b3;           // t2 = (v2 - v6) before its -i
    if (NZ > 4) {
        a0 = pk_add(v[0], v[4]); b0 = pk_sub(v[0], v[4]);
        a1 = pk_add(v[1], v[5]); b1 = pk_mul_w8(pk_sub(v[1], v[5]), ss);
        a2 = pk_add(v[2], v[6]); t2 = pk_sub(v[2], v[6]);
        a3 = pk_add(v[3], v[7]); b3 = pk_mul_w83(pk_sub(v[3], v[7]), ss);
    } else {
        a0 = v[0]; b0 = v[0];
        a1 = v[1]; b1 = pk_mul_w8(v[1], ss);
        a2 = v[2]; t2 = v[2];
        a3 = v[3]; b3 = pk_mul_w83(v[3], ss);
    }
    const v2f c0 = pk_add(a0, a2), c2 = pk_sub(a0, a2);
    const v2f c1 = pk_add(a1, a3), u3 = pk_sub(a1, a3);             // c3 = -i u3
    const v2f d0 = pk_add_mi(b0, t2), d2 = pk_sub_mi(b0, t2);       // b0 +- (-i) t2
    const v2f d1 = pk_add(b1, b3), w3 = pk_sub(b1, b3);             // d3 = -i w3
    v[0] = pk_add(c0, c1); v[4] = pk_sub(c0, c1);
    v[2] = pk_add_mi(c2, u3); v[6] = pk_sub_mi(c2, u3);
    v[1] = pk_add(d0, d1); v[5] = pk_sub(d0, d1);
    v[3] = pk_add_mi(d2, w3); v[7] = pk_sub_mi(d2, w3);
}
__device__ __forceinline__ void pk_cmul7(v2f (&v)[8], const v2f (&w)[8]) {
#pragma unroll
    for (int k = 1; k < 8; k++) v[k] = pk_cmul(v[k], w[k]);
}
#endif

constexpr int XROW = 72;                  // float2 row stride of the transpose buffer
constexpr int MELW = 12;                  // mel taps per band kept in registers (wider bands take the LDS loop)
constexpr int XBUF = 8 * XROW;            // float2 per wave

}  // namespace wsa
