// utterance.hip — K4 (output_level 11 only): utterance features, ONE WAVEFRONT PER CLIP.
//
// Stands in for get_utterance_features(e, t) of the reference's inner module 7 (ref dist/main.js:2 @B107902) as the
// dispatcher uses it at process_level 11 (ref @B28869): after every new result the callback gets
// `(0, label, Y(), get_utterance_features(segments_ci, results))` computed over EVERYTHING so far — 15 histograms
// (264 bins) over the syllables / segments, each divided by its total.  Level 11 stores exactly what level 10
// stores (ref @B27713: `10==o.process_level||11==o.process_level`), so this kernel reads the compacted level-10
// rows (one per syllable), the segments_ci table and the straightened frames, and emits one 264-vector per result.
// Reproduced, not repaired: segments_ci is indexed with the RESULT index (after a segment whose straighten step
// threw, lengths / gaps come from the wrong entries); `a[idx]++` with an index that is NaN (0/0 when a syllable has
// no first / second formant) or negative creates a property outside the array part whose value is NaN
// (undefined + 1), the normalisation's `for (n in e) t += e[n]` then gives NaN, `t > 0` fails and that histogram is
// handed out as RAW COUNTS.  parseInt(x) of the Numbers that occur here is truncation (no value
// is below 1e-6 or above 1e21 in magnitude: every argument is a ratio of small integers, a mean of integer-valued
// floats or 3..4 x log10 of one).
#include "wsa_internal.hpp"
#include "wave_ops.hpp"
#include "jsmath_device.hpp"

namespace wsa {

// histogram layout in output order (ref: `t(w(i)),t(w(o)),t(w(l)),t(w(s)),t(w(c)),t(w(u)),t(w(f)),t(w(d)),t(w(h)),...`)
enum { H_I = 0, H_O = 10, H_L = 20, H_S = 30, H_C = 40, H_U = 60, H_F = 100, H_D = 140, H_H = 164, H_P = 188, H_M = 196,
       H_G = 204, H_Y = 214, H_V = 224, H_X = 244, H_END = 264, N_HIST = 15 };
__device__ __constant__ const int kHistOff[N_HIST + 1] = {H_I, H_O, H_L, H_S, H_C, H_U, H_F, H_D, H_H, H_P, H_M, H_G, H_Y, H_V, H_X, H_END};

// hist[off + parseInt(x)]++ with the reference's clamps; NaN / negative (unless clamped to 0) poisons the histogram
__device__ __forceinline__ void bump(uint32_t* hist, uint32_t* ghost, int h, double x, bool lo_clamp) {
    const int off = kHistOff[h], n = kHistOff[h + 1] - off;
    if (!(x == x) || fabs(x) == __builtin_inf()) { atomicAdd(&ghost[h], 1u); return; }       // parseInt(NaN / +-Infinity) = NaN
    const double t = trunc(x);
    int idx = t >= (double)n ? n - 1 : (t <= -1.0 ? -1 : (int)t);
    if (idx < 0) { if (lo_clamp) idx = 0; else { atomicAdd(&ghost[h], 1u); return; } }
    atomicAdd(&hist[off + idx], 1u);
}

__global__ __launch_bounds__(64) void utterance_kernel(UttParams p) {
    __shared__ uint32_t hist[H_END], ghost[N_HIST + 1];
    const uint32_t clip = blockIdx.x;
    const int lane = threadIdx.x;
    const uint32_t s0 = p.clip_seg_off[clip], s1 = p.clip_seg_off[clip + 1];
    const uint32_t r0 = p.clip_row_off[clip], r1 = p.clip_row_off[clip + 1];
    const int nseg = (int)(s1 - s0);
    const int32_t* segs = p.segments + (uint64_t)s0 * 4;                    // {clip, start, len, flag}
    // results before this clip = exclusive prefix over clips (filled by utterance_count_kernel)
    const uint32_t out0 = p.clip_utt_off[clip];
    // streams: what the launch has accumulated so far comes from the stream's state (cleared by a START)
    uint32_t* st = p.state ? p.state + (uint64_t)clip * UTT_STATE_WORDS : nullptr;
    const bool fresh = st && (p.ctl[clip] & 1u);
    for (int i = lane; i < H_END; i += 64) hist[i] = (st && !fresh) ? st[i] : 0;
    if (lane <= N_HIST) ghost[lane] = (st && !fresh && lane < N_HIST) ? st[H_END + lane] : 0;
    wsync();
    int k = 0, seg_before = 0, have_first = 0, first_start = 0;
    int tsum = 0, tsum_upto = 0;                                            // sum of segments_ci lengths up to (global) segment tsum_upto
    double prev_end = 0;
    const int32_t* cy = nullptr;
    if (st) {
        cy = p.carry + (uint64_t)clip * CARRY_WORDS;
        seg_before = cy[0] - nseg;                                          // K3 has counted this step's segments in already
        if (!fresh) {
            k = (int)st[280]; have_first = (int)st[286]; first_start = (int)st[285]; tsum = (int)st[284];
            prev_end = __hiloint2double((int)st[283], (int)st[282]);
        }
        tsum_upto = seg_before;                                             // every earlier segment is in tsum already
    }
    auto save_state = [&]() __attribute__((always_inline)) {
        if (!st) return;
        wsync();
        for (int i = lane; i < H_END; i += 64) st[i] = hist[i];
        if (lane < N_HIST) st[H_END + lane] = ghost[lane];
        if (lane == 0) { st[280] = (uint32_t)k; st[286] = (uint32_t)have_first; st[285] = (uint32_t)first_start; st[284] = (uint32_t)tsum;
                         st[282] = (uint32_t)__double2loint(prev_end); st[283] = (uint32_t)__double2hiint(prev_end); }
    };
    if (nseg == 0) { if (fresh) save_state(); return; }
    if (!have_first) { have_first = 1; first_start = segs[1]; prev_end = (double)segs[1]; }      // `let a = e[0][0]`
    // segments_ci entry g of the launch (a result index: the reference's own mix-up) — a batch holds them all, a stream the last CARRY_HIST
    auto seg_start_of = [&](int gidx) __attribute__((always_inline)) -> int { return st ? cy[2 + 2 * (gidx % CARRY_HIST)] : segs[4 * gidx + 1]; };
    auto seg_len_of = [&](int gidx) __attribute__((always_inline)) -> int { return st ? cy[3 + 2 * (gidx % CARRY_HIST)] : segs[4 * gidx + 2]; };
    const float* fm = p.formants + (uint64_t)p.frame_off[clip] * 9;
    uint32_t row = r0;
    int k_out = 0;
    for (int i = 0; i < nseg; i++) {
        if (segs[4 * i + 3] < 0) continue;                                  // no result entry for this segment
        // ---- syllables of result k: the rows whose callback index is k
        uint32_t row_end = row;
        while (row_end < r1 && p.row_meta[(uint64_t)row_end * 8 + 1] == k) row_end++;      // uniform scan (short)
        const int cnt = (int)(row_end - row);
        if (st && seg_before + nseg - k > CARRY_HIST && lane == 0) atomicOr(&p.totals[2], 1u);      // the entry has left the history (as in K3)
        const double seg_len = (double)seg_len_of(k), seg_start = (double)seg_start_of(k);   // u[r], r = RESULT index
        uint32_t osum = 0;
        for (int base = 0; base < cnt; base += 64) {
            const int j = base + lane;
            uint32_t my_len = 0;
            if (j < cnt) {
                const int32_t* m = p.row_meta + (uint64_t)(row + j) * 8;
                const int st_ = m[6], sl = m[7];
                my_len = (uint32_t)sl;
                double a = 0, e1 = 0, w1 = 0, dl1 = 0, c1 = 0, u = 0, e2 = 0, w2 = 0, dl2 = 0, c2 = 0;
                float pb1 = 0.f, pb2 = 0.f;
                // four frames per trip, their 24 values requested before the first is used (a load per use made every frame a memory round trip of its own)
                for (int o0 = 0; o0 < sl; o0 += 4) {
                    float v[4][6];
#pragma unroll
                    for (int q = 0; q < 4; q++) {
                        const float* F = fm + (uint64_t)(((uint32_t)st_ + (uint32_t)min(o0 + q, sl - 1)) & p.ring_mask) * 9;
#pragma unroll
                        for (int z = 0; z < 6; z++) v[q][z] = F[z];
                    }
#pragma unroll
                    for (int q = 0; q < 4; q++) {
                        const int o = o0 + q;
                        if (o >= sl) break;
                        const float b1 = v[q][0], b2 = v[q][3];
                        if (b1 > 0.f) { c1 += 1; a += (double)b1; e1 += (double)v[q][1]; w1 += (double)v[q][2]; if (o > 0) dl1 += (double)b1 - (double)pb1; }
                        if (b2 > 0.f) { c2 += 1; u += (double)b2; e2 += (double)v[q][4]; w2 += (double)v[q][5]; if (o > 0) dl2 += (double)b2 - (double)pb2; }
                        pb1 = b1; pb2 = b2;
                    }
                }
                a /= c1; e1 /= c1; w1 /= c1; u /= c2; e2 /= c2; w2 /= c2;               // 0/0 = NaN as in the reference
                const double e = (double)sl;
                bump(hist, ghost, 4, e / 2, false);                                      // c: syllable length
                bump(hist, ghost, 5, a / 2, false);                                      // u: mean bin of formant 1
                bump(hist, ghost, 6, u / 2, false);                                      // f: mean bin of formant 2
                bump(hist, ghost, 7, 3 * jsm::log10(e1), false);                         // d: energy of formant 1
                bump(hist, ghost, 8, 4 * jsm::log10(e2), false);                         // h: energy of formant 2
                bump(hist, ghost, 9, w1 / 2, false);                                     // p: width of formant 1
                bump(hist, ghost, 10, w2 / 2, false);                                    // m: width of formant 2
                bump(hist, ghost, 11, 10 * (e - c1) / e, false);                         // g: frames without formant 1
                bump(hist, ghost, 12, 10 * (e - c2) / e, false);                         // y: frames without formant 2
                bump(hist, ghost, 13, 20 * (dl1 + 50) / 100, true);                      // v: net bin movement of formant 1
                bump(hist, ghost, 14, 20 * (dl2 + 50) / 100, true);                      // x: ... of formant 2
            }
            osum += wave_sum_u32(my_len);
        }
        wsync();
        if (lane == 0) {                                                     // `_`(seg_len, n_syllables, gap, syllabic ratio)
            bump(hist, ghost, 0, 10 * seg_len / 150, false);
            bump(hist, ghost, 1, (double)cnt, false);
            bump(hist, ghost, 2, 10 * (seg_start - prev_end) / 150, false);
            bump(hist, ghost, 3, 2 * ((double)osum / seg_len - .3) * 10, true);
        }
        prev_end = seg_start + seg_len;
        wsync();
        // ---- Y() (ref: `[u[0][0]*step, (sum of u[n][1] + 1)*step]` over the entries pushed so far = up to this segment)
        while (tsum_upto <= seg_before + i) { tsum += segs[4 * (tsum_upto - seg_before) + 2]; tsum_upto++; }
        // ---- emit: every histogram divided by its total (array bins + ghost)
        double* out = p.utt_feat + (uint64_t)(out0 + k_out) * H_END;
        for (int h = 0; h < N_HIST; h++) {
            const int off = kHistOff[h], n = kHistOff[h + 1] - off;
            uint32_t c = (lane < n) ? hist[off + lane] : 0u;                 // every histogram has at most 40 bins
            const uint32_t tot = wave_sum_u32(c);
            if (lane < n) out[off + lane] = (tot > 0 && ghost[h] == 0) ? (double)c / (double)tot : (double)c;
        }
        if (lane == 0) {
            int32_t* mo = p.utt_meta + (uint64_t)(out0 + k_out) * 4;
            mo[0] = (int32_t)clip; mo[1] = k; mo[2] = first_start; mo[3] = tsum;
        }
        row = row_end;
        k++; k_out++;
        wsync();
    }
    // (segments behind the last result still count in tsum when a later step's result asks for it)
    while (st && tsum_upto < seg_before + nseg) { tsum += segs[4 * (tsum_upto - seg_before) + 2]; tsum_upto++; }
    save_state();
}

// results per clip -> exclusive offsets (single block; n_clips is small next to the frame counts)
__global__ void utterance_count_kernel(UttParams p) {
    __shared__ uint32_t part[256];
    const int tid = threadIdx.x;
    const uint32_t per = (p.n_clips + 255) / 256;
    const uint32_t c0 = min(p.n_clips, tid * per), c1 = min(p.n_clips, c0 + per);
    uint32_t sum = 0;
    for (uint32_t c = c0; c < c1; c++) {
        uint32_t n = 0;
        for (uint32_t s = p.clip_seg_off[c]; s < p.clip_seg_off[c + 1]; s++) n += p.segments[(uint64_t)s * 4 + 3] >= 0 ? 1u : 0u;
        p.clip_utt_off[c] = n;
        sum += n;
    }
    part[tid] = sum;
    __syncthreads();
    if (tid == 0) { uint32_t a = 0; for (int i = 0; i < 256; i++) { const uint32_t v = part[i]; part[i] = a; a += v; } p.clip_utt_off[p.n_clips] = a; p.totals[3] = a; }
    __syncthreads();
    uint32_t a = part[tid];
    for (uint32_t c = c0; c < c1; c++) { const uint32_t n = p.clip_utt_off[c]; p.clip_utt_off[c] = a; a += n; }
}

void launch_utterance(const UttParams& p, hipStream_t s) {
    if (p.n_clips == 0) return;
    hipLaunchKernelGGL(utterance_count_kernel, dim3(1), dim3(256), 0, s, p);
    hipLaunchKernelGGL(utterance_kernel, dim3(p.n_clips), dim3(64), 0, s, p);
}

}  // namespace wsa
