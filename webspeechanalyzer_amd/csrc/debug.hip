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
    out[i] = fn == 0 ? jsm::log10(x[i]) : (fn == 2 ? jsm::log10_fin(x[i]) : jsm::pow_pos(x[i], y[i]));      // fn 2: the branch-free log10 of the feature reductions (positive normal arguments)
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
    if (!x || !out || (fn == 1 && !y) || fn < 0 || fn > 2) return WSA_ERR_INVALID;
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

// the peak-candidate scan on its own: `spec` = n_frames rows of `bands` u32 (host), mode 1 = lane-per-frame kernel, 2 = wave-per-frame
// kernel; out: hdr [n_frames][4], amp [n_frames * 64], ent [n_frames * 64][4] (u32, candidate tables zero-filled beforehand), flags
extern "C" int wsa_debug_peaks(int32_t device, const uint32_t* spec, uint32_t n_frames, int32_t bands, int32_t mode,
                               uint32_t* hdr, uint32_t* amp, uint32_t* ent, uint32_t* flags) {
    if (!spec || !hdr || !amp || !ent || !flags || bands < 1 || n_frames < 1) return WSA_ERR_INVALID;
    if (hipSetDevice(device) != hipSuccess) return WSA_ERR_NO_DEVICE;
    const size_t nsp = (size_t)n_frames * bands * 4, nh = (size_t)n_frames * 16, na = (size_t)n_frames * wsa::CAND_CAP * 4, ne = (size_t)n_frames * wsa::CAND_CAP * 16;
    char* d = nullptr;
    const size_t o_h = (nsp + 255) & ~(size_t)255, o_a = o_h + ((nh + 255) & ~(size_t)255), o_e = o_a + ((na + 255) & ~(size_t)255), o_f = o_e + ((ne + 255) & ~(size_t)255);
    bool ok = hipMalloc(&d, o_f + 256) == hipSuccess && hipMemset(d, 0, o_f + 256) == hipSuccess && hipMemcpy(d, spec, nsp, hipMemcpyHostToDevice) == hipSuccess;
    if (ok) {
        wsa::PkParams p{};
        p.spec = reinterpret_cast<const uint32_t*>(d);
        p.rec.hdr = reinterpret_cast<uint4*>(d + o_h); p.rec.amp = reinterpret_cast<uint32_t*>(d + o_a); p.rec.ent = reinterpret_cast<uint4*>(d + o_e);
        p.frame0 = 0; p.total_frames = n_frames; p.bands = bands; p.flags = reinterpret_cast<uint32_t*>(d + o_f);
        wsa::launch_peaks_mode(p, mode, nullptr);
        ok = hipGetLastError() == hipSuccess && hipDeviceSynchronize() == hipSuccess
             && hipMemcpy(hdr, d + o_h, nh, hipMemcpyDeviceToHost) == hipSuccess && hipMemcpy(amp, d + o_a, na, hipMemcpyDeviceToHost) == hipSuccess
             && hipMemcpy(ent, d + o_e, ne, hipMemcpyDeviceToHost) == hipSuccess && hipMemcpy(flags, d + o_f, 4, hipMemcpyDeviceToHost) == hipSuccess;
    }
    (void)hipFree(d);
    return ok ? WSA_OK : WSA_ERR_HIP;
}

// timing of the peak-candidate scan on its own (tuning): the same frames `reps` times, average kernel time in ms; dbg = PkParams::dbg
extern "C" int wsa_debug_peaks_time(int32_t device, const uint32_t* spec, uint32_t n_frames, int32_t bands, int32_t mode, int32_t dbg, int32_t reps, float* ms) {
    if (!spec || !ms || bands < 1 || n_frames < 1 || reps < 1) return WSA_ERR_INVALID;
    if (hipSetDevice(device) != hipSuccess) return WSA_ERR_NO_DEVICE;
    const size_t nsp = (size_t)n_frames * bands * 4, nh = (size_t)n_frames * 16, na = (size_t)n_frames * wsa::CAND_CAP * 4, ne = (size_t)n_frames * wsa::CAND_CAP * 16;
    char* d = nullptr;
    const size_t o_h = (nsp + 255) & ~(size_t)255, o_a = o_h + ((nh + 255) & ~(size_t)255), o_e = o_a + ((na + 255) & ~(size_t)255), o_f = o_e + ((ne + 255) & ~(size_t)255);
    bool ok = hipMalloc(&d, o_f + 256) == hipSuccess && hipMemset(d, 0, o_f + 256) == hipSuccess && hipMemcpy(d, spec, nsp, hipMemcpyHostToDevice) == hipSuccess;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    ok = ok && hipEventCreate(&e0) == hipSuccess && hipEventCreate(&e1) == hipSuccess;
    if (ok) {
        wsa::PkParams p{};
        p.spec = reinterpret_cast<const uint32_t*>(d);
        p.rec.hdr = reinterpret_cast<uint4*>(d + o_h); p.rec.amp = reinterpret_cast<uint32_t*>(d + o_a); p.rec.ent = reinterpret_cast<uint4*>(d + o_e);
        p.frame0 = 0; p.total_frames = n_frames; p.bands = bands; p.flags = reinterpret_cast<uint32_t*>(d + o_f); p.dbg = dbg;
        wsa::launch_peaks_mode(p, mode, nullptr);
        (void)hipEventRecord(e0, nullptr);
        for (int r = 0; r < reps; r++) wsa::launch_peaks_mode(p, mode, nullptr);
        (void)hipEventRecord(e1, nullptr);
        ok = hipGetLastError() == hipSuccess && hipDeviceSynchronize() == hipSuccess && hipEventElapsedTime(ms, e0, e1) == hipSuccess;
        *ms /= (float)reps;
    }
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
    (void)hipFree(d);
    return ok ? WSA_OK : WSA_ERR_HIP;
}
