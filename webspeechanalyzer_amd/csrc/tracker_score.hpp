// tracker_score.hpp — the formant tracker's match score (its own header so that the unit test entry of debug.hip evaluates the very
// function tracker.hip uses; fixture tests/golden/score_expected.json = the reference's `_` under Node).
#pragma once
#include <hip/hip_runtime.h>

namespace wsa {

// match score `_` (ref @B37340).  The reference branches (`tamp >= pamp ? pamp / tamp : tamp / pamp`, early returns, two laws for gap == 0 and
// gap > 0); here every lane of a wave scores a different (track, peak) pair, so the branches would run one after the other under exec masks:
// ONE division serves both quotients (the operands are selected, the IEEE operation is the same) and both laws are evaluated and selected.
// Only what cannot occur inside the search window (gap == 0 with dist > 2, gap > 3) stays a branch.
__device__ __forceinline__ double match_score(int gap, double dist, double n, double tbin, double pbin,
                                              double tamp, double pamp, double vel) {
    const bool tge = tamp >= pamp;
    const bool none = !tge && !(pamp > 0);                         // ref: `else { if (!(pamp > 0)) return 0; ...`
    const double num = tge ? pamp : tamp, den = tge ? tamp : pamp;
    const double s = num / den;
    // gap == 0: 300 * s / dist — the window leaves dist in {0, 1, 2}: x / 1, x / 2 = x * 0.5, x / 0 = Infinity
    const double x = 300 * s;
    double r0 = dist == 1 ? x : (dist == 2 ? x * 0.5 : __builtin_inf());
    if (__builtin_expect(dist > 2, 0)) r0 = x / dist;
    r0 = s > .1 ? r0 : 0;
    // gap > 0
    const double s10 = s >= 1 ? 10 : (s < .1 ? 1 : s * 10);
    const double t0 = 10 - fabs(pbin - tbin - vel);
    const double t = t0 < 1 ? 1 : t0;
    const double i = n > 10 ? 10 : n;
    double k = gap == 1 ? 10.0 : (gap == 2 ? 5.0 : 10.0 / 3.0);                 // 10 / gap, gap in 1..3 inside the search window
    if (__builtin_expect(gap > 3, 0)) k = 10 / (double)gap;
    const double r1 = (s < .001 || t0 < 0) ? 0 : k * (t * t + i * s10);
    return none ? 0 : (gap == 0 ? r0 : r1);
}

}  // namespace wsa
