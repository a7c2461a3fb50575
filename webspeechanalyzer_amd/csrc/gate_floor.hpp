// gate_floor.hpp — the auto noise gate's floor law v(ctx_max) (ref dist/main.js:2 @B28615) for integer ctx_max.
//
//   t = Math.log10(y);  t>7: parseInt(10^(t-3)/20)   t>6: parseInt(10^(t-3)/2)   t>4: parseInt(10^(t-2)/2)
//                       t>2: parseInt(10^(t/3))      t>1: parseInt(y/10)         else 1
//
// floor_law_exact evaluates it with the V8 log10 / pow ports (jsmath_device.hpp).  floor_law is what the gate kernel calls: under
// the auto gate y is an integer (gate.hip), the real-valued result of every arm is y/20000, y/2000, y/200, cbrt(y) or y/10, and
// the f64 evaluation is off by a few 1e-15 relative — so parseInt of it can only differ from the integer quotient / cube root
// when the real value IS an integer (y a multiple of the divisor, a perfect cube) or y sits on an arm boundary (a power of ten),
// where the last bit of log10 / pow decides.  Those y go to floor_law_exact; everything else is integer arithmetic.
// tests/test_gpu_units.py compares the two over ALL 2^32 values of y on the device (wsa_debug_floor_law).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "jsmath_device.hpp"

namespace wsa {

__device__ inline uint32_t floor_law_exact(uint32_t ctx_max) {
    const double y = (double)ctx_max, t = jsm::log10(y);
    double v;
    if (t > 7) v = trunc(jsm::pow_pos(10, t - 3) / 20);
    else if (t > 6) v = trunc(jsm::pow_pos(10, t - 3) / 2);
    else if (t > 4) v = trunc(jsm::pow_pos(10, t - 2) / 2);
    else if (t > 2) v = trunc(jsm::pow_pos(10, t / 3));
    else if (t > 1) v = trunc(y / 10);
    else v = 1;
    return (uint32_t)v;
}

// true: the integer shortcut is not trusted for this y
__device__ __forceinline__ bool floor_law_needs_exact(uint32_t y, uint32_t& v) {
    if (y > 10000000u) { v = y / 20000u; return y % 20000u == 0u; }
    if (y > 1000000u) { v = y / 2000u; return y % 2000u == 0u; }         // y = 10^7 is a multiple of 2000
    if (y > 10000u) { v = y / 200u; return y % 200u == 0u; }             // y = 10^6
    if (y > 100u) {                                                       // cube root of 101 .. 10^4: 4 .. 21
        uint32_t k = (uint32_t)__builtin_amdgcn_exp2f(__builtin_amdgcn_logf((float)y) * (1.0f / 3.0f));
        if ((k + 1u) * (k + 1u) * (k + 1u) <= y) k++;
        if (k * k * k > y) k--;
        v = k;
        return k * k * k == y;                                            // includes y = 10^4? no: 10^4 is not a cube; it is an arm boundary
    }
    if (y > 10u) { v = y / 10u; return false; }                          // parseInt(y/10): the f64 quotient is exact enough for every integer y
    v = 1u;
    return false;
}

__device__ __forceinline__ uint32_t floor_law(uint32_t y) {
    uint32_t v;
    const bool edge = y == 10u || y == 100u || y == 10000u || y == 1000000u || y == 10000000u;
    if (floor_law_needs_exact(y, v) || edge) v = floor_law_exact(y);
    return v;
}

}  // namespace wsa
