// peaks.hip — K1b: per-frame peak-candidate scan, one LANE per frame (frames are independent).
//
// Stands in for the parallelisable half of the reference's frame loop D() (ref dist/main.js:2
// @B25717, scan @B25827): the strict 3-neighbour rising/falling classification, the direction /
// flat-run state machine and the /10 shoulder shrink.  The running noise floor `v` only decides
// WHETHER a candidate is accepted and whether it feeds n, d, h, p (SURVEY.md §8a note), never its
// geometry, so every candidate [i, s, l] is emitted here and the sequential tracker (tracker.hip)
// applies the gate.  Also emits g = sum e[1..B-1] (exact, 64-bit).
//
#include "wsa_internal.hpp"

namespace wsa {

__global__ __launch_bounds__(256) void peaks_kernel(PkParams p) {
    // one lane = one frame; the lane streams its own 4*bands-byte row with 16-byte loads (rows are
    // 512 B apart, so a wave touches 64 lines per load: TA-bound, which is cheap next to the
    // branchy scan) and writes its record straight to global memory.  No LDS: full occupancy.
    const uint32_t f = blockIdx.x * 256u + threadIdx.x;
    if (f >= p.total_frames) return;
    const int B = p.bands;
    constexpr int MAXC = 64;
    const uint32_t* e = p.spec + (uint64_t)f * (uint32_t)B;
    uint32_t* out = p.rec + (uint64_t)f * (uint32_t)p.rec_stride;
    uint32_t* out_amp = out + 4 + MAXC;
    double* out_plo = reinterpret_cast<double*>(out + 4 + 2 * MAXC);
    double* out_phi = reinterpret_cast<double*>(out + 4 + 4 * MAXC);
    int n = 0, i = 0, l = 0, s = 0, c = 0, u = 0;
    uint64_t g = 0;                             // sum e[1..a]; run = e[0] + g = sum e[0..a]
    uint64_t run0 = 0;
    // thr = e[l]/10 in the reference; e[x] < e[l]/10  <=>  10 e[x] < e[l] for u32 values
    // (e[l]/10 differs from an integer by 0 or >= 0.1, far more than a double ulp).
    // bit 24 marks the end-of-spectrum emission, which the reference adds to n and d but never
    // lets update h / p (ref @B26383: no `e[l]>h&&(h=e[l],p=l)` in that arm)
    // Each emission also records the exact prefix sums at its (shrunk) shoulders, so that the tracker
    // gets any band energy sum e[st..en] (ref @B36500 `for(t=a;t<=f;t++)d+=e[t]`) by one subtraction.
#define WSA_EMIT(last, a_now) do { const uint64_t el = e[l]; \
        while (i < l && 10ull * e[i] < el) i++; \
        while (s > l && 10ull * e[s] < el) s--; \
        uint64_t hi = run0 + g; for (int t_ = (a_now); t_ > s; t_--) hi -= e[t_]; \
        uint64_t lo = hi; for (int t_ = s; t_ >= i; t_--) lo -= e[t_]; \
        out[4 + n] = (uint32_t)i | ((uint32_t)s << 8) | ((uint32_t)l << 16) | ((uint32_t)(last) << 24); \
        out_amp[n] = (uint32_t)el; out_plo[n] = (double)lo; out_phi[n] = (double)hi; n++; } while (0)
#define WSA_STEP(a, ea) do { \
        g += (ea); \
        const bool rise = (ea) > e1 && ((a) < 2 || (ea) > e2) && ((a) < 3 || (ea) > e3); \
        const bool fall = (ea) < e1 && ((a) < 2 || (ea) < e2) && ((a) < 3 || (ea) < e3); \
        if (rise) { \
            if (u == -1 || u == 0) { if (u == -1 && i <= l && l < s) WSA_EMIT(0, a); i = (a) - 1; l = (a); } else l = (a); \
            u = 1; \
        } else if (fall) { if (u == 1 || u == -1) { s = (a); u = -1; } } \
        else if (u == -1) { c++; if (c > 2) { c = 0; if (i <= l && l < s) WSA_EMIT(0, a); u = 0; } } \
        else if (u == 1 && (ea) > e1) l = (a); \
        if ((a) == B - 1 && u == 1) { s = (a); l = (a); if (i < l && l <= s) WSA_EMIT(1, a); } \
        e3 = e2; e2 = e1; e1 = (ea); } while (0)
    uint32_t e1 = 0, e2 = 0, e3 = 0;            // e[a-1], e[a-2], e[a-3]
    if ((B & 3) == 0) {
        const uint4* row = reinterpret_cast<const uint4*>(e);
        for (int q = 0; q < B / 4; q++) {
            const uint4 v = row[q];
            const int a = 4 * q;
            if (q == 0) { e1 = v.x; run0 = v.x; } else WSA_STEP(a, v.x);
            WSA_STEP(a + 1, v.y); WSA_STEP(a + 2, v.z); WSA_STEP(a + 3, v.w);
        }
    } else {
        e1 = e[0]; run0 = e[0];
        for (int a = 1; a < B; a++) { const uint32_t ea = e[a]; WSA_STEP(a, ea); }
    }
#undef WSA_STEP
#undef WSA_EMIT
    *reinterpret_cast<double*>(out) = (double)g; out[2] = (uint32_t)n; out[3] = 0;
}

void launch_peaks(const PkParams& p, hipStream_t s) {
    if (p.total_frames == 0) return;
    hipLaunchKernelGGL(peaks_kernel, dim3((p.total_frames + 255) / 256), dim3(256), 0, s, p);
}

}  // namespace wsa
