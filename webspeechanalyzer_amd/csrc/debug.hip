// debug.hip — test entries of libwsa that are NOT part of include/wsa.h: unit access to device-side pieces that the public
// entry points only exercise through their consequences (tests/test_gpu_units.py).
#include <vector>
#include "wsa_internal.hpp"
#include "jsmath_device.hpp"

namespace wsa {
// fn 0: jsm::log10(x[i]); fn 1: jsm::pow_pos(x[i], y[i]) — the V8 Math.log10 / Math.pow ports the noise gate's
// `parseInt(Math.pow(10, t - 3) / 20)` steps depend on (ref dist/main.js:2 @B28615)
__global__ void debug_jsmath_kernel(int fn, const double* x, const double* y, double* out, uint32_t n) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    out[i] = fn == 0 ? jsm::log10(x[i]) : jsm::pow_pos(x[i], y[i]);
}
}  // namespace wsa

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
