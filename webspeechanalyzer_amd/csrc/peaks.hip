// peaks.hip — K1b: per-frame peak-candidate scan, one LANE per frame (frames are independent).
//
// Stands in for the parallelisable half of the reference's frame loop D() (ref dist/main.js:2
// @B25717, scan @B25827): the strict 3-neighbour rising/falling classification, the direction /
// flat-run state machine and the /10 shoulder shrink.  The running noise floor `v` only decides
// WHETHER a candidate is accepted and whether it feeds n, d, h, p (SURVEY.md §8a note), never its
// geometry, so every candidate [i, s, l] is emitted here and the sequential tracker (tracker.hip)
// applies the gate.  Also emits g = sum e[1..B-1] (exact, 64-bit).
//
// A 64-frame x `bands` u32 tile is staged through LDS with coalesced loads (row stride bands+1 so
// the per-lane row walks are bank-conflict free); records leave through LDS the same way.
#include "wsa_internal.hpp"

namespace wsa {

__global__ __launch_bounds__(64) void peaks_kernel(PkParams p) {
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    const int lane = threadIdx.x;
    const int B = p.bands, RS = B + 1, RW = p.rec_words, RWS = RW | 1;
    uint32_t* tile = lds;                       // [64][RS]
    uint32_t* rec = lds + 64 * RS;              // [64][RWS]
    const uint32_t f0 = blockIdx.x * 64u;
    const uint32_t nf = min(64u, p.total_frames - f0);
    const uint32_t* src = p.spec + (uint64_t)f0 * (uint32_t)B;
    for (uint32_t idx = lane; idx < nf * (uint32_t)B; idx += 64) {
        const uint32_t r = idx / (uint32_t)B, c = idx - r * (uint32_t)B;
        tile[r * RS + c] = src[idx];
    }
    __syncthreads();
    if ((uint32_t)lane < nf) {
        const uint32_t* e = tile + lane * RS;
        uint32_t* out = rec + lane * RWS;
        int n = 0, i = 0, l = 0, s = 0, c = 0, u = 0;
        uint64_t g = 0;
        // thr = e[l]/10 in the reference; e[x] < e[l]/10  <=>  10 e[x] < e[l] for u32 values
        // (e[l]/10 differs from an integer by 0 or >= 0.1, far more than a double ulp)
        // bit 24 marks the end-of-spectrum emission, which the reference adds to n and d but never
        // lets update h / p (ref @B26383: no `e[l]>h&&(h=e[l],p=l)` in that arm)
#define WSA_EMIT(last) do { const uint64_t el = e[l]; \
            while (i < l && 10ull * e[i] < el) i++; \
            while (s > l && 10ull * e[s] < el) s--; \
            out[4 + n] = (uint32_t)i | ((uint32_t)s << 8) | ((uint32_t)l << 16) | ((uint32_t)(last) << 24); n++; } while (0)
        uint32_t e1 = e[0], e2 = 0, e3 = 0;     // e[a-1], e[a-2], e[a-3]
        for (int a = 1; a < B; a++) {
            const uint32_t ea = e[a];
            g += ea;
            const bool rise = ea > e1 && (a < 2 || ea > e2) && (a < 3 || ea > e3);
            const bool fall = ea < e1 && (a < 2 || ea < e2) && (a < 3 || ea < e3);
            if (rise) {
                if (u == -1 || u == 0) {
                    if (u == -1 && i <= l && l < s) WSA_EMIT(0);
                    i = a - 1; l = a;
                } else l = a;
                u = 1;
            } else if (fall) {
                if (u == 1 || u == -1) { s = a; u = -1; }
            } else if (u == -1) {
                c++;
                if (c > 2) { c = 0; if (i <= l && l < s) WSA_EMIT(0); u = 0; }
            } else if (u == 1 && ea > e1) l = a;
            if (a == B - 1 && u == 1) { s = a; l = a; if (i < l && l <= s) WSA_EMIT(1); }
            e3 = e2; e2 = e1; e1 = ea;
        }
#undef WSA_EMIT
        out[0] = (uint32_t)g; out[1] = (uint32_t)(g >> 32); out[2] = (uint32_t)n; out[3] = 0;
        for (int q = 4 + n; q < RW; q++) out[q] = 0;
    }
    __syncthreads();
    uint32_t* dst = p.cand + (uint64_t)f0 * (uint32_t)RW;
    for (uint32_t idx = lane; idx < nf * (uint32_t)RW; idx += 64) {
        const uint32_t r = idx / (uint32_t)RW, c = idx - r * (uint32_t)RW;
        dst[idx] = rec[r * RWS + c];
    }
}

void launch_peaks(const PkParams& p, hipStream_t s) {
    if (p.total_frames == 0) return;
    const int RS = p.bands + 1, RWS = p.rec_words | 1;
    const size_t lds = (size_t)64 * (RS + RWS) * 4;
    hipLaunchKernelGGL(peaks_kernel, dim3((p.total_frames + 63) / 64), dim3(64), lds, s, p);
}

}  // namespace wsa
