// tracker_score.hpp — the formant tracker's match score (its own header so that the unit test entry of debug.hip evaluates the very
// function tracker.hip uses; fixture tests/golden/score_expected.json = the reference's `_` under Node).
#pragma once
#include <hip/hip_runtime.h>

namespace wsa {

// match score `_` (ref @B37340)
__device__ __forceinline__ double match_score(int gap, double dist, double n, double tbin, double pbin,
                                              double tamp, double pamp, double vel) {
    double s;
    if (tamp >= pamp) s = pamp / tamp;
    else { if (!(pamp > 0)) return 0; s = tamp / pamp; }
    if (gap == 0) {                      // 300 * s / dist: the window leaves dist in {0, 1, 2}: x / 1, x / 2 = x * 0.5, x / 0 = Infinity
        if (!(s > .1)) return 0;
        const double x = 300 * s;
        return dist == 1 ? x : (dist == 2 ? x * 0.5 : (dist == 0 ? __builtin_inf() : x / dist));
    }
    if (s < .001) return 0;
    if (s >= 1) s = 10; else if (s < .1) s = 1; else s *= 10;
    double t = 10 - fabs(pbin - tbin - vel);
    if (t < 0) return 0;
    if (t < 1) t = 1;
    double i = n;
    if (i > 10) i = 10;
    const double k = gap == 1 ? 10.0 : (gap == 2 ? 5.0 : (gap == 3 ? 10.0 / 3.0 : 10 / (double)gap));     // 10 / gap, gap in 1..3 inside the search window
    return k * (t * t + i * s);
}

}  // namespace wsa
