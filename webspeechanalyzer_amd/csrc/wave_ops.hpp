// wave_ops.hpp — wave64 cross-lane helpers for gfx950 (DPP scans / reductions, lane exchange).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace wsa {

// Lanes of ONE wave exchange data through LDS / global memory.  The hardware executes a wave's LDS
// (and vector-memory) instructions in issue order, so no s_waitcnt is needed — only the compiler
// must not move memory accesses across the exchange point.  (A wavefront-scope fence would also do,
// but it drains vmcnt and so kills prefetches in flight.)
__device__ __forceinline__ void wsync() {
    asm volatile("" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    asm volatile("" ::: "memory");
}

__device__ __forceinline__ uint64_t lanemask_lt(int lane) { return lane == 0 ? 0ull : (~0ull >> (64 - lane)); }

// DPP controls (CDNA): row_shr:n = 0x110 + n, row_bcast:15 = 0x142, row_bcast:31 = 0x143
template <int CTRL, int ROW_MASK, int BANK_MASK>
__device__ __forceinline__ uint32_t dpp_or_zero(uint32_t v) {
    // lanes without a valid source (or masked off) receive 0.  A shift inside the rows with every row and bank enabled needs no `old` operand for that
    // (bound_ctrl: a source outside the row reads as 0, every lane is written) — and then the compiler folds the move into the instruction that uses it
    // (v_add_u32_dpp / v_max_u32_dpp) instead of a v_mov 0, an s_nop, the DPP move and the add
    if constexpr (ROW_MASK == 0xf && BANK_MASK == 0xf && CTRL >= 0x111 && CTRL <= 0x11f) return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, CTRL, 0xf, 0xf, true);
    else return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, ROW_MASK, BANK_MASK, false);
}

// inclusive add-scan over the 64 lanes, 6 DPP steps (Kogge-Stone inside rows of 16, then two row broadcasts)
__device__ __forceinline__ uint32_t wave_incl_scan_u32(uint32_t v) {
    v += dpp_or_zero<0x111, 0xf, 0xf>(v);
    v += dpp_or_zero<0x112, 0xf, 0xf>(v);
    v += dpp_or_zero<0x114, 0xf, 0xf>(v);
    v += dpp_or_zero<0x118, 0xf, 0xf>(v);
    v += dpp_or_zero<0x142, 0xa, 0xf>(v);
    v += dpp_or_zero<0x143, 0xc, 0xf>(v);
    return v;
}
__device__ __forceinline__ uint32_t wave_sum_u32(uint32_t v) {
    return (uint32_t)__builtin_amdgcn_readlane((int)wave_incl_scan_u32(v), 63);
}
__device__ __forceinline__ uint32_t wave_max_u32(uint32_t v) {
    uint32_t t;
    t = dpp_or_zero<0x111, 0xf, 0xf>(v); v = t > v ? t : v;
    t = dpp_or_zero<0x112, 0xf, 0xf>(v); v = t > v ? t : v;
    t = dpp_or_zero<0x114, 0xf, 0xf>(v); v = t > v ? t : v;
    t = dpp_or_zero<0x118, 0xf, 0xf>(v); v = t > v ? t : v;
    t = dpp_or_zero<0x142, 0xa, 0xf>(v); v = t > v ? t : v;
    t = dpp_or_zero<0x143, 0xc, 0xf>(v); v = t > v ? t : v;
    return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}
// inclusive max-scan over the 64 lanes (also of non-negative floats through their bit patterns: same order)
__device__ __forceinline__ uint32_t wave_incl_scan_max_u32(uint32_t v) {
    uint32_t t;
    t = dpp_or_zero<0x111, 0xf, 0xf>(v); v = t > v ? t : v;
    t = dpp_or_zero<0x112, 0xf, 0xf>(v); v = t > v ? t : v;
    t = dpp_or_zero<0x114, 0xf, 0xf>(v); v = t > v ? t : v;
    t = dpp_or_zero<0x118, 0xf, 0xf>(v); v = t > v ? t : v;
    t = dpp_or_zero<0x142, 0xa, 0xf>(v); v = t > v ? t : v;
    t = dpp_or_zero<0x143, 0xc, 0xf>(v); v = t > v ? t : v;
    return v;
}
// exact sum of <= 64 non-negative integers below 2^40 (band energies, amplitudes), as a double
__device__ __forceinline__ double wave_sum_int40(uint64_t x) {
    const uint32_t s0 = wave_sum_u32((uint32_t)(x & 0xfffffu));
    const uint32_t s1 = wave_sum_u32((uint32_t)((x >> 20) & 0xfffffu));
    return (double)s1 * 1048576.0 + (double)s0;
}
// f64 add-scan step: move both halves with the same DPP control, then one v_add_f64
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_f64_or_zero(double v) {
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, ROW_MASK, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, ROW_MASK, 0xf, false);
    return __hiloint2double(hi, lo);
}
// row_shr:n inside rows of 16 with bound_ctrl: a lane whose source lies outside its row receives 0 and every lane is written, so the
// DPP move needs no `old` operand (dpp_f64_or_zero's zeros cost two v_mov per step)
template <int CTRL>
__device__ __forceinline__ double dpp_f64_row(double v) {
    const int lo = __builtin_amdgcn_mov_dpp(__double2loint(v), CTRL, 0xf, 0xf, true);
    const int hi = __builtin_amdgcn_mov_dpp(__double2hiint(v), CTRL, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double read_lane_f64(double v, int src) {
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), src), __builtin_amdgcn_readlane(__double2loint(v), src));
}
// sums over the 64 lanes of N values at once (results valid in every lane).  The summation order is a fixed tree: inclusive scans inside the four
// rows of 16 lanes (Kogge-Stone: shifts by 1, 2, 4, 8), then (row3 + row2) + (row1 + row0) — the tree of the six-step DPP form
// (row_bcast:15 into rows 1 and 3, row_bcast:31 into rows 2 and 3) this replaces, so every sum keeps its bits.  The N chains advance step by step
// together: no instruction waits for the one before it (a v_add_f64 result read by the next DPP move costs two wait states).
template <int N>
__device__ __forceinline__ void wave_sums_f64(double (&v)[N]) {
#pragma unroll
    for (int k = 0; k < N; k++) v[k] += dpp_f64_row<0x111>(v[k]);
#pragma unroll
    for (int k = 0; k < N; k++) v[k] += dpp_f64_row<0x112>(v[k]);
#pragma unroll
    for (int k = 0; k < N; k++) v[k] += dpp_f64_row<0x114>(v[k]);
#pragma unroll
    for (int k = 0; k < N; k++) v[k] += dpp_f64_row<0x118>(v[k]);
#pragma unroll
    for (int k = 0; k < N; k++) {
        const double r0 = read_lane_f64(v[k], 15), r1 = read_lane_f64(v[k], 31), r2 = read_lane_f64(v[k], 47), r3 = read_lane_f64(v[k], 63);
        const double a = r1 + r0, b = r3 + r2;
        v[k] = b + a;
    }
}
__device__ __forceinline__ double wave_sum_f64(double v) { double a[1] = {v}; wave_sums_f64(a); return a[0]; }
// The same N <= 8 sums through an LDS transposition (red: 8 x 64 doubles of scratch, 16-byte aligned): every lane stores its N values as rows [k][lane]; lane
// (k = lane >> 3, h = lane & 7) adds the eight neighbours h*8 .. h*8+7 of row k as a balanced tree, three exchange steps inside the 8 lanes (lane ^ 1, ^ 2, then
// the mirror image inside the 8) complete the row, and v_readlane hands the totals out: ~3 N + 20 instructions instead of 23 N.  A fixed tree again — of another
// shape than wave_sums_f64's, so the two give sums that may differ in the last bits.  Ends with the scratch free for the next call.
template <int CTRL>
__device__ __forceinline__ double dpp_f64_perm(double v) {
    const int lo = __builtin_amdgcn_mov_dpp(__double2loint(v), CTRL, 0xf, 0xf, true);
    const int hi = __builtin_amdgcn_mov_dpp(__double2hiint(v), CTRL, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}
template <int N>
__device__ __forceinline__ void wave_sums_f64_lds(double (&v)[N], double* red, int lane) {
    static_assert(N >= 1 && N <= 8, "one row per value, eight lanes per row");
#pragma unroll
    for (int k = 0; k < N; k++) red[k * 64 + lane] = v[k];
    wsync();
    double s = 0;
    if ((lane >> 3) < N) {
        const double* r = red + (lane >> 3) * 64 + (lane & 7) * 8;
        s = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
    }
    s += dpp_f64_perm<0xB1>(s);          // quad_perm [1, 0, 3, 2]: lane ^ 1
    s += dpp_f64_perm<0x4E>(s);          // quad_perm [2, 3, 0, 1]: lane ^ 2
    s += dpp_f64_perm<0x141>(s);         // row_half_mirror: lane i of eight <- lane 7 - i (the other quad's sum)
#pragma unroll
    for (int k = 0; k < N; k++) v[k] = read_lane_f64(s, 8 * k);
    wsync();
}
__device__ __forceinline__ int read_lane_i32(int v, int src) { return __builtin_amdgcn_readlane(v, src); }

// ---- half-wave groups (lanes 0..31 / 32..63 work on two independent problems in lock step: the tracker's paired spans)
// the 32 ballot bits of the lane's own half
__device__ __forceinline__ uint32_t half_ballot(bool c, int lane) { const uint64_t m = __ballot(c); return lane < 32 ? (uint32_t)m : (uint32_t)(m >> 32); }
// inclusive add-scan inside each half: the wave scan without its last step
__device__ __forceinline__ uint32_t half_incl_scan_u32(uint32_t v) {
    v += dpp_or_zero<0x111, 0xf, 0xf>(v);
    v += dpp_or_zero<0x112, 0xf, 0xf>(v);
    v += dpp_or_zero<0x114, 0xf, 0xf>(v);
    v += dpp_or_zero<0x118, 0xf, 0xf>(v);
    v += dpp_or_zero<0x142, 0xa, 0xf>(v);
    return v;
}
// value of the half's last lane (31 / 63) in every lane of the half
__device__ __forceinline__ uint32_t half_last_u32(uint32_t v, int lane) {
    const uint32_t a = (uint32_t)__builtin_amdgcn_readlane((int)v, 31), b = (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
    return lane < 32 ? a : b;
}
// the larger of a value that is uniform inside each half (a scalar)
__device__ __forceinline__ int halves_max_i32(int v) { const int a = __builtin_amdgcn_readlane(v, 0), b = __builtin_amdgcn_readlane(v, 32); return a > b ? a : b; }

// ---- the same for groups of GW = 32 or 16 lanes (16: the four DPP rows of a wave work on four independent problems: the tracker's quads of spans)
template <int GW>
__device__ __forceinline__ uint32_t group_ballot(bool c, int lane) {
    if (GW == 32) return half_ballot(c, lane);
    const uint64_t m = __ballot(c);
    return (uint32_t)(m >> (lane & 48)) & 0xffffu;
}
template <int GW>
__device__ __forceinline__ uint32_t group_incl_scan_u32(uint32_t v) {
    if (GW == 32) return half_incl_scan_u32(v);
    v += dpp_or_zero<0x111, 0xf, 0xf>(v);          // a row of 16 lanes is a group: the scan stays inside it
    v += dpp_or_zero<0x112, 0xf, 0xf>(v);
    v += dpp_or_zero<0x114, 0xf, 0xf>(v);
    v += dpp_or_zero<0x118, 0xf, 0xf>(v);
    return v;
}
// value of the group's last lane in every lane of the group
template <int GW>
__device__ __forceinline__ uint32_t group_last_u32(uint32_t v, int lane) {
    if (GW == 32) return half_last_u32(v, lane);
    const uint32_t a = (uint32_t)__builtin_amdgcn_readlane((int)v, 15), b = (uint32_t)__builtin_amdgcn_readlane((int)v, 31);
    const uint32_t c = (uint32_t)__builtin_amdgcn_readlane((int)v, 47), d = (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
    const int g = lane >> 4;
    return g == 0 ? a : (g == 1 ? b : (g == 2 ? c : d));
}
// the largest of a value that is uniform inside each group (a scalar)
template <int GW>
__device__ __forceinline__ int groups_max_i32(int v) {
    if (GW == 32) return halves_max_i32(v);
    const int a = __builtin_amdgcn_readlane(v, 0), b = __builtin_amdgcn_readlane(v, 16), c = __builtin_amdgcn_readlane(v, 32), d = __builtin_amdgcn_readlane(v, 48);
    const int ab = a > b ? a : b, cd = c > d ? c : d;
    return ab > cd ? ab : cd;
}

}  // namespace wsa
