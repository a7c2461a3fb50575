// debug.hip — test entries of libwsa that are NOT part of include/wsa.h: unit access to device-side pieces that the public
// entry points only exercise through their consequences (tests/test_gpu_units.py).
#include <vector>
#include "wsa_internal.hpp"
#include "jsmath_device.hpp"
#include "gate_floor.hpp"
#include "tracker_score.hpp"

namespace wsa {
// fn 0: jsm::log10(x[i]); fn 1: jsm::pow_pos(x[i], y[i]) — the V8 Math.log10 / Math.pow ports the noise gate's
// `parseInt(Math.pow(10, t - 3) / 20)` steps depend on (ref dist/main.js:2 @B28615)
__global__ void debug_jsmath_kernel(int fn, const double* x, const double* y, double* out, uint32_t n) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    out[i] = fn == 0 ? jsm::log10(x[i]) : jsm::pow_pos(x[i], y[i]);
}
// rows of 8 doubles {gap, dist, track length, track bin, peak bin, track amp, peak amp, velocity} -> match_score (ref dist/main.js:2 @B37340)
__global__ void debug_score_kernel(const double* a, double* out, uint32_t n) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double* r = a + 8 * (size_t)i;
    out[i] = match_score((int)r[0], r[1], r[2], r[3], r[4], r[5], r[6], r[7]);
}

// the gate's integer floor law against its f64 evaluation for every y in [lo, hi): out[0] = number of y where they differ, out[1] = the
// smallest such y, out[2] = number of y that took the f64 route inside floor_law
__global__ void debug_floor_law_kernel(uint64_t lo, uint64_t hi, unsigned long long* out) {
    unsigned long long bad = 0, first = ~0ull, exact = 0;
    for (uint64_t y = lo + (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; y < hi; y += (uint64_t)gridDim.x * blockDim.x) {
        uint32_t v;
        if (floor_law_needs_exact((uint32_t)y, v)) exact++;
        if (floor_law((uint32_t)y) != floor_law_exact((uint32_t)y)) { bad++; if (y < first) first = y; }
    }
    if (bad) { atomicAdd(&out[0], bad); atomicMin(&out[1], first); }
    if (exact) atomicAdd(&out[2], exact);
}
}  // namespace wsa

extern "C" int wsa_debug_score(int32_t device, const double* args8, double* out, uint32_t n) {
    if (!args8 || !out) return WSA_ERR_INVALID;
    if (hipSetDevice(device) != hipSuccess) return WSA_ERR_NO_DEVICE;
    double *da = nullptr, *dout = nullptr;
    bool ok = hipMalloc(&da, (size_t)(n ? n : 1) * 64) == hipSuccess && hipMalloc(&dout, (size_t)(n ? n : 1) * 8) == hipSuccess;
    ok = ok && hipMemcpy(da, args8, (size_t)n * 64, hipMemcpyHostToDevice) == hipSuccess;
    if (ok && n) {
        hipLaunchKernelGGL(wsa::debug_score_kernel, dim3((n + 255) / 256), dim3(256), 0, nullptr, da, dout, n);
        ok = hipGetLastError() == hipSuccess && hipMemcpy(out, dout, (size_t)n * 8, hipMemcpyDeviceToHost) == hipSuccess;
    }
    (void)hipFree(da); (void)hipFree(dout);
    return ok ? WSA_OK : WSA_ERR_HIP;
}

extern "C" int wsa_debug_floor_law(int32_t device, uint64_t lo, uint64_t hi, uint64_t* out3) {
    if (!out3 || hi > (1ull << 32) || lo > hi) return WSA_ERR_INVALID;
    if (hipSetDevice(device) != hipSuccess) return WSA_ERR_NO_DEVICE;
    unsigned long long* d = nullptr;
    unsigned long long init[3] = {0ull, ~0ull, 0ull};
    bool ok = hipMalloc(&d, sizeof(init)) == hipSuccess && hipMemcpy(d, init, sizeof(init), hipMemcpyHostToDevice) == hipSuccess;
    if (ok) {
        hipLaunchKernelGGL(wsa::debug_floor_law_kernel, dim3(256 * 32), dim3(256), 0, nullptr, lo, hi, d);
        ok = hipGetLastError() == hipSuccess && hipMemcpy(init, d, sizeof(init), hipMemcpyDeviceToHost) == hipSuccess;
    }
    (void)hipFree(d);
    out3[0] = init[0]; out3[1] = init[1]; out3[2] = init[2];
    return ok ? WSA_OK : WSA_ERR_HIP;
}

extern "C" int wsa_debug_jsmath(int32_t device, int32_t fn, const double* x, const double* y, double* out, uint32_t n) {
    if (!x || !out || (fn == 1 && !y) || fn < 0 || fn > 1) return WSA_ERR_INVALID;
    if (hipSetDevice(device) != hipSuccess) return WSA_ERR_NO_DEVICE;
    double *dx = nullptr, *dy = nullptr, *dout = nullptr;
    const size_t bytes = (size_t)(n ? n : 1) * sizeof(double);
    bool ok = hipMalloc(&dx, bytes) == hipSuccess && hipMalloc(&dy, bytes) == hipSuccess && hipMalloc(&dout, bytes) == hipSuccess;
    ok = ok && hipMemcpy(dx, x, (size_t)n * sizeof(double), hipMemcpyHostToDevice) == hipSuccess;
    if (ok && y) ok = hipMemcpy(dy, y, (size_t)n * sizeof(double), hipMemcpyHostToDevice) == hipSuccess;
    if (ok && n) {
        hipLaunchKernelGGL(wsa::debug_jsmath_kernel, dim3((n + 255) / 256), dim3(256), 0, nullptr, fn, dx, dy, dout, n);
        ok = hipGetLastError() == hipSuccess && hipMemcpy(out, dout, (size_t)n * sizeof(double), hipMemcpyDeviceToHost) == hipSuccess;
    }
    (void)hipFree(dx); (void)hipFree(dy); (void)hipFree(dout);
    return ok ? WSA_OK : WSA_ERR_HIP;
}
