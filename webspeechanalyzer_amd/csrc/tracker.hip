// tracker.hip — K2b: formant tracking + segment finalize, ONE WAVEFRONT PER SEGMENT SPAN.
//
// Stands in for (ref = /root/reference/dist/main.js line 2, byte offsets):
//   accumulate_fm x(e,t,n,r,a) + match score _    @B35952, @B37340
//   the result part of finalize O(e)              @B27190-28506
//   get_ranked_formants y(), straighten m()       @B35670, @B35074
//   sep_syllables p(), formant_features u()       @B34757, @B32369 (+ stats helpers @B1978-2277)
//   clear_fm                                      @B35919
// including the reference's quirks: stale first-frame filing index, first-peak
// amplitude of a merged association, fp32 storage in straighten, and the segments_ci entry that
// survives a throwing straighten step.
//
// The tracker (`l`, `s`, `c` of ref module 4) is cleared by every reset_segment, so the frames
// between two resets form an independent span; gate.hip (K2a) has already decided which spans end in
// a finalize and with which arguments accumulate_fm is called on each frame.  Spans are dealt out to the
// waves statically.  All decision arithmetic is IEEE double exactly as JavaScript Numbers (-ffp-contract=off;
// Math.log10 from jsmath_device.hpp).  Lanes parallelise the inner loops: peak acceptance (lane =
// candidate), (track, peak) pair scoring (lane = pair), track update (lane = track), new tracks
// (lane = peak), ranking (lane = track), straighten (lane = frame), features (lane = formant).
#include <type_traits>
#include "wsa_internal.hpp"
#include "jsmath_device.hpp"
#include "wave_ops.hpp"
#include "tracker_score.hpp"

// Tuning switches of the tracker (WSA_DBG bits 1, 2, 4, 8, 16, 32, 64, 512: tools/README.md) exist only in a library built with
// `make TUNING=1`: a dozen tests of a kernel argument per frame are not free in a kernel that is bound by instruction issue.  The
// two switches the tests use (256: generic finalize, 1024: small track table) are always there.
#ifdef WSA_TUNING
#define WSA_TUNE(bits_) (p.dbg & (bits_))
#else
#define WSA_TUNE(bits_) false
#endif

namespace wsa {

constexpr int MAXC = 64;            // peak candidates per frame record (bands <= 128)
constexpr int AC_MAX = 320;         // worst case of the active-track table: tracks not yet 4 filing indices old (<= 5 x 63)
constexpr int PAIR_AC = 64;         // active-track table of one half-wave in the paired variant
constexpr int PAIR_GSZ = 4384;      // LDS bytes per half there: table (48 B per entry) + peak / pair scratch (40 B per peak) + the bin map
constexpr int QUAD_AC = 38;         // ... of one quarter-wave (16 lanes = a DPP row) in the variant that tracks four spans per wave
constexpr int QUAD_GSZ = 2496;      // 38 x 48 + 16 x 40 + 24, 16-byte aligned: four of them stay inside 8 LDS allocation units (10 240 B)
constexpr int AC_FAST = 140;        // what the default kernel variant holds in LDS (16 waves per CU); see the kernels at the end of tracker_body

struct Ws {                          // per-wave work space carved out of global memory
    int32_t *tr_len, *tr_slot, *tr_rank;           // per track id: summary (written when the track leaves the active table) + finalize scratch (tr_slot: rank << 2 | slot of the
                                                   // generic finalize; tr_rank: unused since round 6, kept so that the span regions' layout — tracker_pool_bpf — stays what it was)
    double *tr_sumE, *tr_sumEbin;
    int4* pt;                                      // per point: {track id, bin | width << 8 | min(filing index, 0x7fff) << 17, band energy (f64 in .z/.w)}
    int4* ptx;                                     // level 3 only: {start bin, amplitude, filing index, end bin} of the point
    int32_t* pt_key;
    int32_t *d_p0, *d_p1, *d_gen;
    float *fr, *sm1;
    double *dB, *Aev;
    int32_t *q_idx, *sorted; double* q_mb;
};

__host__ __device__ inline size_t align16(size_t x) { return (x + 15) & ~(size_t)15; }

// lays the per-wave arrays out back to back (16-byte aligned); returns the pointers by value so
// that they live in registers, and the total size through *bytes
__host__ __device__ __forceinline__ Ws carve_ws(char* base, int T, int P, int F, int PX, size_t* bytes) {
    Ws w;
    size_t o = 0;
#define WSA_CARVE(field, type, count) do { w.field = reinterpret_cast<type*>(base + o); \
        o = align16(o + sizeof(type) * (size_t)(count)); } while (0)
    WSA_CARVE(tr_len, int32_t, T); WSA_CARVE(tr_slot, int32_t, T); WSA_CARVE(tr_rank, int32_t, T);
    WSA_CARVE(tr_sumE, double, T); WSA_CARVE(tr_sumEbin, double, T);
    WSA_CARVE(pt, int4, P); WSA_CARVE(ptx, int4, PX); WSA_CARVE(pt_key, int32_t, P);
    WSA_CARVE(d_p0, int32_t, F + 2); WSA_CARVE(d_p1, int32_t, F + 2); WSA_CARVE(d_gen, int32_t, F + 2);
    WSA_CARVE(fr, float, (size_t)(F + 2) * 9); WSA_CARVE(sm1, float, F + 2);
    WSA_CARVE(dB, double, (size_t)3 * (F + 2)); WSA_CARVE(Aev, double, (size_t)3 * (F + 2));
    WSA_CARVE(q_idx, int32_t, T); WSA_CARVE(sorted, int32_t, T); WSA_CARVE(q_mb, double, T);
#undef WSA_CARVE
    if (bytes) *bytes = o;
    return w;
}

// bytes per frame of the span regions of the split finalize (TrParams::pool): a span of F frames needs carve_ws(64 F, 64 F, F) <= F * carve_ws(64, 64, 1)
size_t tracker_pool_bpf() { size_t b = 0; (void)carve_ws(nullptr, MAXC, MAXC, 1, 0, &b); return align16(b) + 256; }
size_t tracker_ws_bytes(int tcap, int pcap, int fcap, bool raw_tracks) { size_t b = 0; (void)carve_ws(nullptr, tcap, pcap, fcap, raw_tracks ? pcap : 0, &b); return align16(b) + 256; }

// formant_features (ref @B32369) for all three formant columns, executed by the whole wave.
// Per-frame quantities (validity, dB = 20 log10 E, the products, neighbour differences, run starts)
// are computed with lane = frame and reduced with wave sums (a fixed tree instead of the reference's
// left-to-right order: differences of a few ulp, far inside the 1e-4 feature tolerance); only the
// energy peak-then-halve state machine (L, S) is inherently sequential and runs on lanes 0..2
// (lane = formant).  Writes x[5 .. 52]; the caller writes x[0 .. 4].
__device__ __forceinline__ void formant_features_wave(const float* fr, int a, double ctx_max, double* x, double* Aev, int aev_stride, int lane) {
    double res[16];
#pragma unroll 1
    for (int n = 0; n < 3; n++) {
        double sc = 0, sw = 0, sM = 0, sT = 0, sK = 0, sKpos = 0, up = 0, dn = 0;
        uint32_t cnt = 0, runs = 0, nKpos = 0;
        int carry_valid = 0; float carry_r = 0.f;
        for (int base = 0; base < a; base += 64) {
            const int t = base + lane;
            float rf = 0.f, Ef = 0.f, wf = 0.f;
            if (t < a) { rf = fr[9 * t + 3 * n]; Ef = fr[9 * t + 3 * n + 1]; wf = fr[9 * t + 3 * n + 2]; }
            const bool valid = t < a && rf > 0.f && Ef > 0.f;
            int pv = __shfl_up((int)valid, 1, 64); float pr = __shfl_up(rf, 1, 64);
            if (lane == 0) { pv = carry_valid; pr = carry_r; }
            carry_valid = read_lane_i32((int)valid, 63); carry_r = __builtin_bit_cast(float, read_lane_i32(__builtin_bit_cast(int, rf), 63));
            if (valid) {
                const double r = rf, E = Ef, wd = wf, dB = 20 * jsm::log10(E);
                sc += r * dB; sw += r; sM += wd * dB; sT += E; sK += dB;
                if (dB > 0) { sKpos += dB; nKpos++; }
                cnt++;
                if (pv) { const double dl = r - (double)pr; if (dl > 1) up += dl; else if (dl < -1) dn += -1 * dl; }
                else runs++;
            }
        }
        { double r8[8] = {sc, sw, sM, sT, sK, sKpos, up, dn}; wave_sums_f64(r8); sc = r8[0]; sw = r8[1]; sM = r8[2]; sT = r8[3]; sK = r8[4]; sKpos = r8[5]; up = r8[6]; dn = r8[7]; }
        const double m = wave_sum_u32(cnt), nruns = wave_sum_u32(runs), nkp = wave_sum_u32(nKpos);
#pragma unroll
        for (int q = 0; q < 16; q++) res[q] = 0;
        if (nruns > 0) {
            const double mw = sw / m, mk = sKpos / nkp;
            double vw = 0, vk = 0;
            for (int base = 0; base < a; base += 64) {
                const int t = base + lane;
                if (t < a) {
                    const float rf = fr[9 * t + 3 * n], Ef = fr[9 * t + 3 * n + 1];
                    if (rf > 0.f && Ef > 0.f) {
                        const double d1 = (double)rf - mw, d2 = 20 * jsm::log10((double)Ef) - mk;
                        vw += d1 * d1; vk += d2 * d2;
                    }
                }
            }
            { double r2[2] = {vw, vk}; wave_sums_f64(r2); vw = r2[0]; vk = r2[1]; }
            res[4] = sT / a * 100 / ctx_max; res[5] = sT / m * 100 / ctx_max;
            res[0] = sc / sK; res[1] = sqrt(vw / m); res[6] = sM / sK; res[2] = mk; res[3] = sqrt(vk / m);
        }
        res[7] = m; res[8] = nruns; res[9] = up; res[10] = dn; res[15] = 100 * m / a;
        // keep sK / m for the event statistics of this column
        const double meanK = sK / m;
        if (lane == n) {
            // energy peak-then-halve events (sequential in the frame order)
            double* A = Aev + (size_t)n * aev_stride;
            bool prev = false; double S = 0, L = 0; int nA = 0;
            for (int t = 0; t < a; t++) {
                const float rf = fr[9 * t + 3 * n], Ef = fr[9 * t + 3 * n + 1];
                if (rf > 0.f && Ef > 0.f) {
                    const double E = Ef;
                    if (prev) {
                        if (E > L) { L = E; S = 1; }
                        else if (S == 1 && E < L / 2) { if (L > 10) A[nA++] = 20 * jsm::log10(E); L = 0; S = -1; }
                    }
                    prev = true;
                } else { prev = false; S = 0; L = 0; }
            }
            res[11] = nA;
            if (nA > 0 && nruns > 0) {
                double sa = 0, na = 0;
                for (int q = 0; q < nA; q++) if (A[q] > 0) { sa += A[q]; na += 1; }
                const double ma = sa / na;
                double va = 0;
                for (int q = 0; q < nA; q++) { const double d = A[q] - ma; va += d * d; }
                res[12] = ma; res[13] = sqrt(va / nA); res[14] = 100 * (ma / meanK - 1);
            }
            if (!(nruns > 0)) res[11] = 0;
#pragma unroll
            for (int q = 0; q < 16; q++) x[5 + 16 * n + q] = res[q];
        }
    }
}


// Energy peak-then-halve events (ref @B32369: `E > L ? (L = E, S = 1) : S == 1 && E < L / 2 && (L > 10 && events.push(...), L = 0, S = -1)`,
// state cleared by every invalid frame) of one 64-frame block, lane = frame.  The reference walks the frames one by one; here the walk
// advances per RUN of valid frames and per EVENT: inside a run the state is a running maximum L (S == 1 exactly when L > 0: valid frames
// have E > 0), so the next event is the first frame whose energy is below half the maximum of the frames before it — an exclusive
// prefix-max scan, a compare and a ballot.  Energies are fp32 values, so the scan and the compares run in fp32 (max, x 0.5 and the
// comparisons are exact there).  vm = valid frames, Ef = this lane's energy, Ep = the previous lane's, run_on = frame 0 continues a run of
// the block before, L = its running maximum (in: carried, out: state behind frame 63).  Returns the mask of event frames.
__device__ __forceinline__ uint64_t energy_events_block(uint64_t vm, float Ef, float Ep, bool run_on, float& L, int lane) {
    uint64_t ev = 0ull;
    int c = 0;
    bool cont = run_on;
    for (;;) {
        const uint64_t rest = c >= 64 ? 0ull : (vm >> c) << c;
        if (!rest) break;
        const int j0 = __ffsll((long long)rest) - 1;                       // first valid frame at or after c
        int s;                                                             // first frame that is compared
        if (j0 == c && cont) s = j0;                                       // the run comes over from the block before
        else { s = j0 + 1; L = 0.f; }                                      // a run starts: its first frame only clears the state
        const uint64_t inv = j0 >= 63 ? 0ull : (~vm >> (j0 + 1)) << (j0 + 1);
        const int e = inv ? __ffsll((long long)inv) - 1 : 64;              // the run is [j0, e)
        while (s < e) {
            // X = max of the run's energies in [s, lane): inclusive max-scan of the previous lane's energy over lanes (s, e]
            const uint32_t src = (lane > s && lane <= e) ? __builtin_bit_cast(uint32_t, Ep) : 0u;
            const float X = __builtin_bit_cast(float, wave_incl_scan_max_u32(src));
            const float before = X > L ? X : L;
            const uint64_t hm = __ballot(lane >= s && lane < e && Ef < before * 0.5f);
            if (!hm) {
                // no event: the run's maximum becomes the state
                const int last = e - 1;
                const float bl = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, before), last));
                const float el = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, Ef), last));
                L = el > bl ? el : bl;
                break;
            }
            const int jh = __ffsll((long long)hm) - 1;
            const float bh = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, before), jh));
            if (bh > 10.f) ev |= 1ull << jh;
            L = 0.f; s = jh + 1;
        }
        if (e >= 64) break;
        c = e; cont = false; L = 0.f;
    }
    return ev;
}

constexpr int FEAT_FX = 18;                          // per column: 15 sums / counts, a zero and a one for the lanes without a quotient
// ---- the same features for inputs of at most 15 frames (the syllables of level 13: 15 frames on average), ALL THREE formant columns at once: lane 16 n + t = frame t
// of column n.  Lanes 16 n + 15 and 48 .. 63 hold no frame, so a run of valid frames never crosses into the next column and energy_events_block walks the three
// columns' runs in one call.  Every sum keeps the tree it has in formant_features_lds, where a column's frames sit in lanes 0 .. 14: the f64 sums that go through
// the LDS transposition are two octet sums added (the other six octets only contribute exact zeros there), the event sum is the in-row Kogge-Stone scan (rows
// 1 .. 3 are zeros there), the counts are integers — so the rows are bit-identical to the one-column-at-a-time form (tests: WSA_DBG bit 65536 switches this off).
template <int CTRL>
__device__ __forceinline__ uint32_t dpp_row_u32(uint32_t v) { return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, CTRL, 0xf, 0xf, true); }
__device__ __forceinline__ uint32_t row_allsum_u32(uint32_t v) {          // every lane of a row of 16 receives the row's sum
    v += dpp_row_u32<0x128>(v); v += dpp_row_u32<0x124>(v); v += dpp_row_u32<0x122>(v); v += dpp_row_u32<0x121>(v);
    return v;
}
// (a function of its own — called, not inlined: inlined into the finalize loop it cost the kernel 21 spilled vector registers —, so the two LDS pointers arrive as
//  generic ones and are cast back to the LDS address space: ds_ instructions, not flat ones)
typedef __attribute__((address_space(3))) const float lds_cf;
typedef __attribute__((address_space(3))) double lds_d;
__device__ __attribute__((noinline)) void formant_columns_packed(const float* fr_g, int a, int lane, double* red_g) {
    lds_cf* const fr = (lds_cf*)fr_g;
    lds_d* const red = (lds_d*)red_g;
    lds_d* const fx = red + 8 * 64;
    lds_d* const col = red + 5 * 64;                     // [3][8]: the columns' f64 totals (rows 5 .. 7 of the reduction scratch are free)
    const int n = lane >> 4, t = lane & 15;
    const bool in = n < 3 && t < a;
    float rf = 0.f, Ef = 0.f, wf = 0.f;
    if (in) { rf = fr[9 * t + 3 * n]; Ef = fr[9 * t + 3 * n + 1]; wf = fr[9 * t + 3 * n + 2]; }
    const bool valid = in && rf > 0.f && Ef > 0.f;
    int pv = __shfl_up((int)valid, 1, 64); float pr = __shfl_up(rf, 1, 64);
    if (lane == 0) { pv = 0; pr = 0.f; }                 // (a column's first lane looks at the frameless lane in front of it: not valid)
    const uint64_t vm = __ballot(valid);
    const float Ep = __shfl_up(Ef, 1, 64);
    float evL = 0.f;
    const uint64_t ev = energy_events_block(vm, Ef, Ep, false, evL, lane);
    const bool my_event = ((ev >> lane) & 1ull) != 0ull;
    const int nA = __popcll(ev & (0xffffull << (lane & 48)));
    double sc = 0, sM = 0, sT = 0, sK = 0, sKpos = 0, sa = 0, dB = 0;
    uint32_t cnt = 0, runs = 0, nKpos = 0, na = 0, swi = 0, upi = 0, dni = 0;
    if (valid) {
        const double r = rf, E = Ef, wd = wf;
        dB = 20 * jsm::log10_fin(E);
        sc += r * dB; swi += (uint32_t)rf; sM += wd * dB; sT += E; sK += dB;
        if (dB > 0) { sKpos += dB; nKpos++; }
        cnt++;
        if (pv) { const int dl = (int)rf - (int)pr; if (dl > 1) upi += (uint32_t)dl; else if (dl < -1) dni += (uint32_t)(-dl); }
        else runs++;
        if (my_event && dB > 0) { sa += dB; na++; }
    }
    // ---- the column's sums.  f64 through the LDS transposition: lane (k = lane >> 3, h = lane & 7) adds octet h of row k, lane ^ 1 completes a column
    red[0 * 64 + lane] = sc; red[1 * 64 + lane] = sM; red[2 * 64 + lane] = sT; red[3 * 64 + lane] = sK; red[4 * 64 + lane] = sKpos;
    wsync();
    {
        double s = 0;
        if ((lane >> 3) < 5) {
            const lds_d* r = red + (lane >> 3) * 64 + (lane & 7) * 8;
            s = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
        }
        s += dpp_f64_perm<0xB1>(s);
        // the event sum: inclusive scan inside the row of 16 (the row's last lane holds the column's sum)
        sa += dpp_f64_row<0x111>(sa); sa += dpp_f64_row<0x112>(sa); sa += dpp_f64_row<0x114>(sa); sa += dpp_f64_row<0x118>(sa);
        wsync();
        if ((lane >> 3) < 5 && (lane & 7) < 6 && !(lane & 1)) col[((lane & 7) >> 1) * 8 + (lane >> 3)] = s;
        if (t == 15 && n < 3) col[n * 8 + 5] = sa;
    }
    const double sw = row_allsum_u32(swi), up = row_allsum_u32(upi), dn = row_allsum_u32(dni);
    const double m = row_allsum_u32(cnt), nruns = row_allsum_u32(runs), nkp = row_allsum_u32(nKpos);
    const uint32_t na_t = row_allsum_u32(na);
    wsync();
    const int nc = n < 3 ? n : 0;
    const double c_sKpos = col[nc * 8 + 4], c_sa = col[nc * 8 + 5];
    double ma = 0, mk = 0, vw = 0, vk = 0, va = 0;
    const bool on = nruns > 0;
    if (on) {
        const double mw = sw / m;
        mk = c_sKpos / nkp;
        if (nA > 0) ma = c_sa / (double)na_t;
        if (valid) {
            const double d1 = (double)rf - mw, d2 = dB - mk;
            vw += d1 * d1; vk += d2 * d2;
            if (my_event) { const double d3 = dB - ma; va += d3 * d3; }
        }
    }
    wsync();
    red[0 * 64 + lane] = vw; red[1 * 64 + lane] = vk; red[2 * 64 + lane] = va;
    wsync();
    {
        double s = 0;
        if ((lane >> 3) < 3) {
            const lds_d* r = red + (lane >> 3) * 64 + (lane & 7) * 8;
            s = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
        }
        s += dpp_f64_perm<0xB1>(s);
        wsync();
        if ((lane >> 3) < 3 && (lane & 7) < 6 && !(lane & 1)) col[((lane & 7) >> 1) * 8 + 5 + (lane >> 3)] = s;      // [n][5 .. 7] = vw, vk, va (the event sum has been read)
    }
    wsync();
    if (t == 0 && n < 3) {
        lds_d* f = fx + n * FEAT_FX;
        const lds_d* c = col + n * 8;
        f[0] = c[0]; f[1] = c[3]; f[2] = on ? c[5] : 0.0; f[3] = m; f[4] = on ? c[6] : 0.0; f[5] = c[2]; f[6] = c[1]; f[7] = ma; f[8] = on ? c[7] : 0.0; f[9] = (double)nA; f[10] = nruns;
        f[11] = up; f[12] = dn; f[13] = mk; f[14] = (double)a; f[15] = 0.0; f[16] = 1.0;
    }
}

// The same feature computation for frames that live in LDS (the usual case; `fr` must be derived from a __shared__
// array so that the compiler emits ds_ reads).  Differences from the version above: the energy peak-then-halve state
// machine does not re-read the frames one by one through memory — lane t already holds frame t's energy, so the wave
// walks the valid frames of a 64-frame block with v_readlane and each lane notes whether its frame is an event
// (bit b of `myev` for block b: slices of up to 2048 frames) — and the event statistics are wave sums over those lanes.
// `red` = an LDS scratch of FEAT_SCRATCH doubles: the f64 reductions go through it (wave_sums_f64_lds) and the sixteen results of ALL THREE columns are
// evaluated together at the end — lane 16 n + q takes result q of column n: one division, one dependent division and one square root for the 48 of them
// instead of that block once per column; every value is the same IEEE operation on the same operands either way.
constexpr int FEAT_SCRATCH = 8 * 64 + 3 * FEAT_FX;   // doubles
__device__ __forceinline__ void formant_features_lds(const float* fr, int a, double ctx_max, double* x, int lane, double* red, bool packed = false, bool no_walk = false) {
    double* const fx = red + 8 * 64;
    // ---- energy peak-then-halve events of inputs of at most 128 frames (every segment finalize_fast takes, every syllable): lanes 0 .. 2 walk the frames of
    //      columns 0 .. 2 one after the other — the reference's own walk, three columns at a time, ~16 instructions per frame for all of them — and keep the event
    //      frames as bit masks (evw0: frames 0 .. 63, evw1: 64 .. 127).  energy_events_block (a max-scan, a ballot and a branch per run and per event of ONE
    //      column: ~1 400 instructions for the three columns of a 48-frame segment against ~800 here) serves the longer ones.  Same fp32 comparisons, same events.
    uint32_t evq0 = 0, evq1 = 0, evq2 = 0, evq3 = 0;
    const bool walk = a <= 128 && !packed && !no_walk;          // (no_walk: WSA_DBG bit 131072, the equivalence test of the two event implementations)
    if (walk && lane < 3) {
        const float* c = fr + 3 * lane;
        bool prev = false; float L = 0.f;
        auto group = [&](int t0) __attribute__((always_inline)) -> uint32_t {      // frames t0 .. t0 + 31, four at a time: the reads of four frames ahead of their updates
            uint32_t bits = 0;
            const int lim = min(32, a - t0);
            auto four = [&](int q0, bool tail) __attribute__((always_inline)) {
                float r_[4], e_[4];
#pragma unroll
                for (int k = 0; k < 4; k++) { const int t = tail ? min(t0 + q0 + k, a - 1) : t0 + q0 + k; r_[k] = c[9 * t]; e_[k] = c[9 * t + 1]; }
                uint32_t b4 = 0;
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    const bool valid = (!tail || q0 + k < lim) && r_[k] > 0.f && e_[k] > 0.f;
                    const bool eff = valid && prev;                              // the run's first frame only opens it (ref `if (prev) {...} prev = true`)
                    const bool ev = eff && e_[k] < L * 0.5f;                     // S == 1 exactly when L > 0, and then E < L / 2 excludes E > L
                    if (ev && L > 10.f) b4 |= 1u << k;
                    float mx; asm("v_max_f32 %0, %1, %2" : "=v"(mx) : "v"(L), "v"(e_[k]));      // (fmaxf would quiet both operands first: neither can be a NaN here)
                    L = (eff && !ev) ? mx : 0.f;                                 // (a frame that opens a run finds L == 0 and leaves it there)
                    prev = valid;
                }
                bits |= b4 << q0;
            };
            int q0 = 0;
#pragma unroll 1
            for (; q0 + 4 <= lim; q0 += 4) four(q0, false);
            if (q0 < lim) four(q0, true);
            return bits;
        };
        evq0 = group(0);
        if (a > 32) evq1 = group(32);
        if (a > 64) evq2 = group(64);
        if (a > 96) evq3 = group(96);
    }
    if (packed) formant_columns_packed(fr, a, lane, red);
    else
#pragma unroll 1
    for (int n = 0; n < 3; n++) {
        double sc = 0, sM = 0, sT = 0, sK = 0, sKpos = 0, sa = 0;
        uint32_t myev = 0;
        uint32_t swi = 0, udi = 0;                           // sums of bins; of upward | downward << 20 bin differences: small integers (a lane's share of either stays below 2^12 over 32 blocks)
        int cnt = 0, runs = 0, nKpos = 0, na = 0;            // counts of lanes: ballots and scalar popcounts, not wave sums
        int carry_valid = 0, nA = 0; float carry_r = 0.f;
        float evL = 0.f;                                     // running maximum L of the reference's scan (uniform across the wave)
        double dB_first = 0;                                 // dB of this lane's frame in the first block: the second pass reuses it (most segments are one block)
#pragma unroll 1
        for (int base = 0, b = 0; base < a; base += 64, b++) {
            const int t = base + lane;
            float rf = 0.f, Ef = 0.f, wf = 0.f;
            if (t < a) { rf = fr[9 * t + 3 * n]; Ef = fr[9 * t + 3 * n + 1]; wf = fr[9 * t + 3 * n + 2]; }
            const bool valid = t < a && rf > 0.f && Ef > 0.f;
            const uint64_t vm = __ballot(valid);
            // the lane before (DPP wave_shr:1; lane 0: the block before)
            int pv = (int)((vm << 1) >> lane) & 1;
            float pr = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, rf), 0x138, 0xf, 0xf, false));
            if (lane == 0) { pv = carry_valid; pr = carry_r; }
            // ---- energy peak-then-halve events of this block
            uint64_t ev;
            if (walk) {
                const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(b == 0 ? evq0 : evq2), n), hi = (uint32_t)__builtin_amdgcn_readlane((int)(b == 0 ? evq1 : evq3), n);
                ev = ((uint64_t)hi << 32) | lo;
            } else {
                const float Ep = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, Ef), 0x138, 0xf, 0xf, false));
                ev = energy_events_block(vm, Ef, Ep, carry_valid != 0, evL, lane);
            }
            nA += __popcll(ev);
            const bool my_event = ((ev >> lane) & 1ull) != 0ull;
            if (my_event) myev |= 1u << (b & 31);
            carry_valid = (int)(vm >> 63); carry_r = __builtin_bit_cast(float, read_lane_i32(__builtin_bit_cast(int, rf), 63));
            bool kpos = false, run0 = false, evpos = false;
            if (valid) {
                const double r = rf, E = Ef, wd = wf, dB = 20 * jsm::log10_fin(E);      // (valid: an fp32 energy above zero — positive, finite, normal as a double)
                if (b == 0) dB_first = dB;
                sc += r * dB; swi += (uint32_t)rf; sM += wd * dB; sT += E; sK += dB;
                kpos = dB > 0;
                if (kpos) sKpos += dB;
                if (pv) { const int dl = (int)rf - (int)pr; if (dl > 1) udi += (uint32_t)dl; else if (dl < -1) udi += (uint32_t)(-dl) << 20; }
                else run0 = true;
                evpos = my_event && kpos;
                if (evpos) sa += dB;
            }
            cnt += __popcll(vm); runs += __popcll(__ballot(run0)); nKpos += __popcll(__ballot(kpos)); na += __popcll(__ballot(evpos));
        }
        // the two integer sums ride along with the five f64 sums through the LDS transposition (integers far below 2^53: exact in any order)
        double sw, up, dn;
        {
            double r7[7] = {sc, sM, sT, sK, sKpos, (double)swi, (double)udi};
            wave_sums_f64_lds(r7, red, lane);
            sc = r7[0]; sM = r7[1]; sT = r7[2]; sK = r7[3]; sKpos = r7[4]; sw = r7[5];
            const unsigned long long ud = (unsigned long long)r7[6];
            up = (double)(uint32_t)(ud & 0xfffffull); dn = (double)(uint32_t)(ud >> 20);
        }
        const double m = cnt, nruns = runs, nkp = nKpos;
        // lane q < 16 collects result q of this column (one coalesced store at the end).  The column's nine quotients and three square
        // roots are not evaluated one after the other by the whole wave: lane q takes the operands of ITS result, and one division,
        // one dependent division (the two-step results 4, 5, 14) and one square root serve all of them — each value is the same
        // IEEE operation on the same operands as before.
        double ma = 0, vw = 0, vk = 0, va = 0;
        double mk = 0;
        if (nruns > 0) {
            const double mw = sw / m;
            mk = sKpos / nkp;
            if (nA > 0) { sa = wave_sum_f64(sa); ma = sa / (double)na; }
#pragma unroll 1
            for (int base = 0, b = 0; base < a; base += 64, b++) {
                const int t = base + lane;
                if (t < a) {
                    const float rf = fr[9 * t + 3 * n], Ef = fr[9 * t + 3 * n + 1];
                    if (rf > 0.f && Ef > 0.f) {
                        double dB = dB_first;
                        if (b != 0) dB = 20 * jsm::log10_fin((double)Ef);
                        const double d1 = (double)rf - mw, d2 = dB - mk;
                        vw += d1 * d1; vk += d2 * d2;
                        if ((myev >> (b & 31)) & 1u) { const double d3 = dB - ma; va += d3 * d3; }
                    }
                }
            }
            { double r3[3] = {vw, vk, va}; wave_sums_f64_lds(r3, red, lane); vw = r3[0]; vk = r3[1]; va = r3[2]; }      // (va = 0 without events)
        }
        {
            // the column's operands wait in LDS for the common evaluation below
            if (lane == 0) {
                double* f = fx + n * FEAT_FX;
                f[0] = sc; f[1] = sK; f[2] = vw; f[3] = m; f[4] = vk; f[5] = sT; f[6] = sM; f[7] = ma; f[8] = va; f[9] = (double)nA; f[10] = nruns; f[11] = up; f[12] = dn; f[13] = mk;
                f[14] = (double)a; f[15] = 0.0; f[16] = 1.0;
            }
        }
    }
    {
        wsync();
        const int n = lane >> 4, q = lane & 15;
        if (n < 3) {
            const double* f = fx + n * FEAT_FX;
            // numerator / denominator slot of result q (15: zero, 16: one)
            const int ni = (q == 0 ? 0 : q == 1 ? 2 : q == 3 ? 4 : (q == 4 || q == 5) ? 5 : q == 6 ? 6 : q == 13 ? 8 : q == 14 ? 1 : q == 15 ? 3 : 15);
            const int di = q == 0 ? 1 : (q == 1 || q == 3 || q == 5 || q == 14) ? 3 : (q == 4 || q == 15) ? 14 : q == 6 ? 1 : q == 13 ? 9 : 16;
            const double nruns = f[10], nA = f[9], m = f[3], ma = f[7], mk = f[13];
            const bool on = nruns > 0, ev_on = on && nA > 0;
            double num = f[ni], den = f[di];
            if (q == 15) num = 100 * num;                                                    // 100 m / a
            if ((!on && q != 15) || (!ev_on && (q == 13 || q == 14))) { num = 0; den = 1; }   // what the per-column form leaves at 0 / 1
            const double q1 = num / den;
            double num2 = 0, den2 = 1;
            if (q == 4 || q == 5) { num2 = q1 * 100; den2 = ctx_max; }                        // sT / a * 100 / ctx_max, sT / m * 100 / ctx_max
            if (q == 14) { num2 = ma; den2 = q1; }                                            // ma / (sK / m)
            const double q2 = num2 / den2;
            const double sq = sqrt(q1);
            double mine = 0;
            if (q == 0 || q == 6) mine = on ? q1 : 0.0;
            if (q == 1 || q == 3) mine = on ? sq : 0.0;
            if (q == 2) mine = on ? mk : 0.0;
            if (q == 4 || q == 5) mine = on ? q2 : 0.0;
            if (q == 7) mine = m;
            if (q == 8) mine = nruns;
            if (q == 9) mine = f[11];
            if (q == 10) mine = f[12];
            if (q == 11) mine = on ? nA : 0.0;
            if (q == 12) mine = ev_on ? ma : 0.0;
            if (q == 13) mine = ev_on ? sq : 0.0;
            if (q == 14) mine = ev_on ? 100 * (q2 - 1) : 0.0;
            if (q == 15) mine = q1;
            x[5 + lane] = mine;
        }
        wsync();
    }
}

// RAW = output_level 3: the points carry their extra words and the span ends in the raw-track export instead of a finalize
// (its own instantiation: the usual kernels do not pay registers for it)
// ST = incremental streaming (one wave per stream and step, tracker state carried in HBM between steps; see the ST block below)
// PAIR = two spans per wave, one per half-wave, tracked in lock step (see the PAIR block below); finalize stays wave-wide per span
// SPLIT = 1 (with PAIR): accumulate only — tracks and points go to the span's region of p.pool, a header per span is left in p.span_hdr;
// SPLIT = 2: finalize only, one span per wave and turn, out of those regions and headers (tracker_kernel_finalize)
// GW (with PAIR): lanes per span — 32: two spans per wave (halves), 16: four (the DPP rows; SPLIT = 1 only)
template <int AC, bool RAW, bool ST, bool PAIR = false, int SPLIT = 0, int GW = 32>
__device__ __forceinline__ void tracker_body(const TrParams& p) {
    // One LDS block, carved by hand so that finalize can have ALL of it.  First part, two lives: while a span is tracked it
    // holds the active tracks (ref `l`, the live part, in track order); at finalize the tracks are dead and the same bytes hold
    // the ranking scratch and the straightened formant frames, so that finalize works out of LDS, not HBM.  Behind it the per-frame
    // scratch of accumulate_fm (dead at finalize as well: finalize_fast runs over the whole block).
    constexpr int SCRATCH = MAXC * (4 + 4 + 8 + 8) + 64 * 8 + MAXC * 8 + MAXC * 4;
    constexpr int LDS_ONE = AC * 52 + SCRATCH;
    constexpr int LDS_ALL = (PAIR && GW == 16 && 4 * QUAD_GSZ > LDS_ONE) ? 4 * QUAD_GSZ : LDS_ONE;
    __shared__ __attribute__((aligned(16))) unsigned char s_big[LDS_ALL];
    static_assert((AC * 52) % 16 == 0, "the scratch arrays start 16-byte aligned");
    // accepted peaks of the current frame, compacted (lane o <-> peak o)
    double* const s_plo = reinterpret_cast<double*>(s_big + AC * 52);
    double* const s_phi = s_plo + MAXC;
    // per-peak arg-max scratch and the (track, peak) pairs of one scoring pass
    unsigned long long* const s_best = reinterpret_cast<unsigned long long*>(s_phi + MAXC);
    uint32_t* const s_pk = reinterpret_cast<uint32_t*>(s_best + MAXC);
    uint32_t* const s_amp = s_pk + MAXC;
    int32_t* const s_pr_j = reinterpret_cast<int32_t*>(s_amp + MAXC);
    int32_t* const s_pr_o = s_pr_j + 64;
    int32_t* const s_asg = s_pr_o + 64;
    double* const a_vel = reinterpret_cast<double*>(s_big);
    double* const a_sumE = a_vel + AC;
    double* const a_sumEbin = a_sumE + AC;
    unsigned long long* const a_mmask = reinterpret_cast<unsigned long long*>(a_sumEbin + AC);   // peaks assigned to the track this frame
    int32_t* const a_last_frame = reinterpret_cast<int32_t*>(a_mmask + AC);
    int32_t* const a_len = a_last_frame + AC;
    int32_t* const a_gid = a_len + AC;
    uint32_t* const a_bins = reinterpret_cast<uint32_t*>(a_gid + AC);          // last bin | P[h-2] << 8 | P[h-3] << 16
    uint32_t* const a_amp = a_bins + AC;
    // finalize view: q_mb[AC] f64 | q_idx[AC] | sorted[AC] | fr[FRCAP][9] f32 | sm[FRCAP] f32
    constexpr int FRCAP = (AC * 52 - AC * 16) / 40;
    double* const f_qmb = reinterpret_cast<double*>(s_big);
    int32_t* const f_qidx = reinterpret_cast<int32_t*>(f_qmb + AC);
    int32_t* const f_sorted = f_qidx + AC;
    float* const f_fr = reinterpret_cast<float*>(f_sorted + AC);
    float* const f_sm = f_fr + FRCAP * 9;

    const int lane = threadIdx.x;
    Ws W = carve_ws(SPLIT ? p.pool : p.ws + (uint64_t)blockIdx.x * (PAIR ? 2 : 1) * p.ws_stride, p.tcap, p.pcap, p.fcap, RAW ? p.pcap : 0, nullptr);
    int gen = 0;
    int vz; asm volatile("v_mov_b32 %0, 0" : "=v"(vz));          // a zero the compiler cannot see through (see load_hdr)
    int aev_stride = p.fcap + 2;
    if (!ST && !SPLIT) { for (int d = lane; d < p.fcap + 2; d += 64) W.d_gen[d] = 0; }
    if (PAIR && !SPLIT) { const Ws W1 = carve_ws(p.ws + ((uint64_t)blockIdx.x * 2 + 1) * p.ws_stride, p.tcap, p.pcap, p.fcap, 0, nullptr); for (int d = lane; d < p.fcap + 2; d += 64) W1.d_gen[d] = 0; }
    wsync();

    uint32_t item = (!ST && p.order) ? 0u : blockIdx.x;
    bool gen_once = false;
    for (;;) {
        // ---- next span.  Spans = (clip, segment) pairs, dealt out statically: item i -> clip i % n_clips, segment
        //      i / n_clips, wave w takes items w, w + waves, ...  (A work queue costs a device-wide atomic per span on one
        //      address, served at ~30 ns a piece on this chip: with all waves pulling together the last one got its first
        //      span ~90 us into the kernel, and the queue line also slowed every other access to its memory channel.)
        uint32_t k_seg = 0, clip = 0;
        uint64_t pair_idx = 0; uint32_t pair_total = 0;
        if (ST) { if (gen_once) break; gen_once = true; clip = blockIdx.x; k_seg = 0; }      // streams: wave = stream, one pass
        else if (SPLIT == 2) {
            // finalize kernel: the spans in the tracker's order (longest first), wave w takes entries w, w + waves, ...
            const uint32_t total = p.counters[p.order_cnt];
            const uint64_t idx = (uint64_t)item * gridDim.x + blockIdx.x;
            if (idx >= total) break;
            item++;
            const uint2 e = p.order[idx];
            clip = e.x; k_seg = e.y;
        }
        else if (PAIR) {
            // pairs of neighbours in the length-sorted list (entries 2 i and 2 i + 1: spans of nearly the same number of frames), dealt out in snake order
            pair_total = p.counters[p.order_cnt];
            const uint32_t npairs = (pair_total + (uint32_t)(64 / GW) - 1u) / (uint32_t)(64 / GW), W_ = gridDim.x, r = item;
            if ((uint64_t)r * W_ >= npairs) break;
            item++;
            pair_idx = (uint64_t)r * W_ + ((r & 1u) ? W_ - 1u - blockIdx.x : blockIdx.x);
            if (pair_idx >= npairs) continue;
        }
        else if (p.order) {
            // spans sorted by their number of frames, longest first (span_order_kernel), dealt out in snake order — round r hands
            // wave w entry r W + w (r even) or r W + W-1-w (r odd) — so that every wave gets a long and a short one: with the
            // (clip, segment) enumeration the busiest wave of the 1024-clip batch worked 1.5x the mean.  (First span static, the rest
            // from an atomic queue — longest-processing-time-first proper — was slower: 0.62 vs 0.44 ms, profiles/r02_notes.md.)
            const uint32_t total = p.counters[p.order_cnt], W_ = gridDim.x, r = item;       // `item` counts the rounds here
            if ((uint64_t)r * W_ >= total) break;
            item++;
            const uint64_t idx = (uint64_t)r * W_ + ((r & 1u) ? W_ - 1u - blockIdx.x : blockIdx.x);
            if (idx >= total) continue;
            const uint2 e = p.order[idx];
            clip = e.x; k_seg = e.y;
        } else {
            k_seg = item / p.n_clips; clip = item - k_seg * p.n_clips;
            if (k_seg >= p.counters[0]) break;
            item += gridDim.x;
            if (k_seg >= p.seg_count[clip]) continue;
        }
        int my_seg = (int)k_seg;
        int32_t* sg = p.seg_i + ((uint64_t)clip * p.seg_cap + my_seg) * 8;
        int start = 0, len = 0, c_ci = 0; uint32_t f_begin = 0, f_end = 0; double ctx_max = 0, floor_ = 0;
        if (!ST && !PAIR) {
            start = sg[SEG_START]; len = sg[SEG_LEN]; c_ci = sg[SEG_CCI];
            f_begin = (uint32_t)sg[SEG_FBEGIN]; f_end = (uint32_t)sg[SEG_FEND];
            ctx_max = p.seg_d[((uint64_t)clip * p.seg_cap + my_seg) * 2];
            floor_ = p.seg_d[((uint64_t)clip * p.seg_cap + my_seg) * 2 + 1];
        }
        uint32_t foff = p.frame_off[clip];

        const unsigned long long tk0 = (WSA_TUNE(16)) ? __builtin_readcyclecounter() : 0ull;
        unsigned long long tk1 = tk0;
        // the two running sums of accumulate_fm (ref @B35952: `S += g; S -= E; C += E` per updated track): all terms are integers below
        // 2^40, so any order is exact — accG collects the g's (uniform), accL this lane's share of the E's, and the two totals are
        // formed where they are read (finalize, the trace, a stream's saved state) instead of by a wave reduction on every frame
        double accG = 0, accL = 0;
        auto acc_totals = [&](double& S, double& C) __attribute__((always_inline)) { C = wave_sum_f64(accL); S = accG - C; };
        int n_tr = 0, n_pt = 0, n_act = 0, stale_d = -1, stale_p1 = 0;
        bool overflow = false, act_overflow = false;
        if (!ST) gen++;

        // the result part of finalize O(e) (ref @B27190-): gate.hip has already pushed segments_ci
        unsigned long long ph[4] = {0, 0, 0, 0};
        unsigned long long acp[5] = {0, 0, 0, 0, 0}, act = 0;       // tuning (WSA_DBG bit 9): cycles per accumulate phase
#define WSA_ACP(k_) do { if (WSA_TUNE(512)) { const unsigned long long now_ = __builtin_readcyclecounter(); acp[k_] += now_ - act; act = now_; } } while (0)

        // rows go to a pool in completion order; K3 (compaction) restores (clip, segment, syllable) order
        auto take_rows = [&](int n) __attribute__((always_inline)) -> long long {
            uint32_t r0 = 0;
            if (lane == 0) r0 = atomicAdd(&p.clip_rows[clip], (uint32_t)n);      // one counter per clip: no two waves queue up on it
            r0 = (uint32_t)read_lane_i32((int)r0, 0);
            if ((uint64_t)r0 + (uint32_t)n > p.row_cap) { overflow = true; return -1; }
            return (long long)clip * p.row_cap + r0;
        };

        // ---- the same finalize out of LDS (the usual case): track keys, the ranking scratch, the points of the span
        //      (key | bin | width, energy) and the straightened frames all fit the block the dead active table leaves
        //      behind, every pointer below is a plain LDS pointer (ds_ instructions, no flat accesses), and the two
        //      inherently sequential steps of the slow version — slot assignment and the energy-event scan — run on
        //      ballots / v_readlane.  Returns false (nothing touched) when the span does not fit; finalize_slow then runs.
        constexpr int BIG = LDS_ALL;
        // GFR (third form, the batch finalize kernel only): the straightened frames do not fit the block and live in the span's region of the pool (W.fr, W.sm1) — keys,
        // ranking scratch and 4-byte points stay in LDS, straighten takes the selection loop (its slots are registers, stored once per frame), the feature sums read
        // the frames through flat loads.  Holds spans of up to ~330 frames (4.8 points per frame); what the generic path cost such a span: profiles/r06_notes.md section 4.
        auto finalize_fast_impl = [&](auto GFR) __attribute__((always_inline)) -> bool {
            constexpr bool G = decltype(GFR)::value;
            const int off_u = (int)align16((size_t)2 * n_tr);                           // union starts behind the track keys
            const int rank_bytes = 16 * n_tr, fr_bytes = G ? 0 : (int)align16((size_t)40 * len);
            const int off_pt = off_u + fr_bytes;
            // (behind the straightened frames the block also has to hold the scratch of the feature reductions: a span of more than ~130 frames takes the generic path)
            // A point costs the block 12 bytes (band energy f64 + packed bin / width / key) — or 4 where that does not fit: the energies then stay in the span's
            // region of the pool and straighten reads them from there (an 8-byte load per applied point out of lines the copy loop below has just touched).
            // At the library's 25 ms step every span of the bench batch fits the 12-byte form; at the application's 15 ms step (segments 1.67 x as long in frames)
            // 29 % of the spans did not and took the generic path in HBM, which made the finalize kernel 3.5 x as long (profiles/r06_notes.md section 4).
            const bool pe_lds = !G && off_pt + 12 * n_pt <= BIG;
            const int ppb = pe_lds ? 12 : 4;
            if (n_tr > 8000 || n_pt > 60000 || off_u + rank_bytes > BIG || off_pt + ppb * n_pt > BIG || off_pt + FEAT_SCRATCH * 8 > BIG) return false;
            int16_t* const trk_key = reinterpret_cast<int16_t*>(s_big);               // per track id: rank << 2 | slot, or -1
            double* const qmb = reinterpret_cast<double*>(s_big + off_u);              // ranking scratch (dies before fr / points are written)
            int32_t* const qt = reinterpret_cast<int32_t*>(s_big + off_u + 8 * n_tr);
            int32_t* const srt = qt + n_tr;
            float* const fr = G ? W.fr : reinterpret_cast<float*>(s_big + off_u);      // [len][9]
            float* const smv = G ? W.sm1 : fr + 9 * len;                               // [len]
            double* const pE = reinterpret_cast<double*>(s_big + off_pt);              // [n_pt] band energy (pe_lds)
            uint32_t* const pkb = reinterpret_cast<uint32_t*>(s_big + off_pt + (pe_lds ? 8 * n_pt : 0));      // [n_pt] bin | width << 8 | key15 << 17 (0x7fff: no part)
            auto energy_of = [&](int q) __attribute__((always_inline)) -> double { return pe_lds ? pE[q] : reinterpret_cast<const double*>(W.pt + q)[1]; };      // (a point record's .z / .w are the f64's words)
            if (WSA_TUNE(16)) ph[0] = ph[1] = ph[2] = ph[3] = __builtin_readcyclecounter();
            // ---- get_ranked_formants (ref @B35670): count >= 2 and mean bin >= 7, stable ascending
            int nq = 0;
            for (int base = 0; base < n_tr; base += 64) {
                const int t = base + lane;
                bool q = false; double mb = 0;
                if (t < n_tr) {
                    trk_key[t] = -1;
                    const int tl = W.tr_len[t]; const double sb = W.tr_sumEbin[t], se = W.tr_sumE[t];      // one round trip, not two
                    if (tl >= 2) { mb = sb / se; q = mb >= 7; }
                }
                const uint64_t mask = __ballot(q);
                if (q) { const int pos = nq + __popcll(mask & lanemask_lt(lane)); qmb[pos] = mb; qt[pos] = t; }
                nq += __popcll(mask);
            }
            wsync();
            for (int base = 0; base < nq; base += 64) {
                const int qi = base + lane;
                if (qi < nq) {
                    const double mb = qmb[qi];
                    int rank = 0;
                    for (int u = 0; u < nq; u++) { const double o = qmb[u]; rank += (o < mb || (o == mb && u < qi)) ? 1 : 0; }
                    srt[rank] = qi;
                }
            }
            wsync();
            // ---- slot assignment of straighten_formants (ref @B35074, first loop header): walking the ranked tracks,
            //      `if |mb - last| > 20: last = mb, slot++, stop at slot 3`.  Lane = rank; each jump is found by a ballot.
            int n_part = 0;                      // ranked tracks that got a slot = ranks 0 .. n_part - 1
            {
                double last = 0; int slot = 0; bool stopped = false;
                for (int base = 0; base < nq && !stopped; base += 64) {
                    const int r = base + lane;
                    double mb = 0; int t = 0;
                    if (r < nq) { const int qi = srt[r]; mb = qmb[qi]; t = qt[qi]; }
                    uint64_t todo = __ballot(r < nq);
                    int myslot = -1;
                    while (todo) {
                        const uint64_t jm = __ballot(((todo >> lane) & 1ull) && fabs(mb - last) > 20);
                        if (jm == 0ull) { if ((todo >> lane) & 1ull) myslot = slot; break; }
                        const int j = __ffsll((long long)jm) - 1;
                        if (((todo >> lane) & 1ull) && lane < j) myslot = slot;
                        last = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(mb), j), __builtin_amdgcn_readlane(__double2loint(mb), j));
                        slot++;
                        if (slot >= 3) { stopped = true; break; }
                        todo &= ~lanemask_lt(j);
                    }
                    if (myslot >= 0) trk_key[t] = (int16_t)((r << 2) | myslot);
                    n_part += __popcll(__ballot(myslot >= 0));
                }
            }
            wsync();
            // ---- straighten applies a frame's points in (track rank, arrival) order.  The tracks that take part are the first n_part
            //      of the ranking (the slot walk above stops at the fourth jump), and a track files at most one point per index — except
            //      that the span's first frame is usually filed under a stale index (quirk 1), which a later frame may carry as
            //      well: its points get a row of their own.  So the points go into a table [filing index][rank] (the filing index
            //      travels in the point record) next to a 64-bit map of the ranks present per index, and a frame's lane walks the set
            //      bits of its map instead of searching its points for the next key over and over (the selection loop below).  More than
            //      64 ranks or no room in the block: the selection loop.
            const int off_tbl = (int)align16((size_t)off_pt + (size_t)ppb * (size_t)n_pt);
            const int tbl_bytes = 4 * (len + 1) + 2 * (len + 1) * n_part;
            const bool use_tbl = !G && n_part <= 32 && c_ci + 1 < 0x7fff && stale_d < 0x7fff && off_tbl + tbl_bytes <= BIG && !(p.dbg & 32768);
            uint32_t* const tblm = reinterpret_cast<uint32_t*>(s_big + off_tbl);                         // [len + 1]: ranks present at index d (32 of them: more take the selection loop); [len]: in the stale row
            uint16_t* const tbl = reinterpret_cast<uint16_t*>(tblm + len + 1);                           // [len + 1][n_part]: point index + 1; row len = the stale row
            if (use_tbl) for (int q = lane; q <= len; q += 64) tblm[q] = 0u;
            wsync();
            // ---- the points of the span move into LDS with their application key: (rank of the track) << 2 | slot
            bool bad = false;
            {
                int4 nxt4 = lane < n_pt ? W.pt[lane] : make_int4(0, 0, 0, 0);
                for (int q = lane; q < n_pt; q += 64) {
                    const int4 rec4 = nxt4;
                    if (q + 64 < n_pt) nxt4 = W.pt[q + 64];
                    const int key = trk_key[rec4.x];
                    if (pe_lds) pE[q] = __hiloint2double(rec4.w, rec4.z);
                    pkb[q] = ((uint32_t)rec4.y & 0x1ffffu) | ((key < 0 ? 0x7fffu : (uint32_t)key) << 17);
                    if (use_tbl && key >= 0) {
                        const int d = (int)((uint32_t)rec4.y >> 17);
                        // a point of a processed track filed at an index >= len makes the reference throw (below)
                        if (d >= len) bad = true;
                        else {
                            const int row = (q < stale_p1 && stale_d >= 0) ? len : d;        // the first frame's points when it was filed under a stale index
                            tbl[row * n_part + (key >> 2)] = (uint16_t)(q + 1);
                            atomicOr(&tblm[row], 1u << (key >> 2));
                        }
                    }
                }
            }
            wsync();
            if (WSA_TUNE(16)) ph[0] = __builtin_readcyclecounter();
            // ---- a point of a processed track filed at an index >= len makes the reference throw
            //      (r[d] undefined, ref @B35484): segments_ci keeps the entry, nothing else is stored
            if (!use_tbl) bad = false;
            for (int base = len; base <= (use_tbl ? -1 : c_ci + 1); base += 64) {
                const int d = base + lane;
                if (d <= c_ci + 1 && W.d_gen[d] == gen)
                    for (int q = W.d_p0[d]; q < W.d_p1[d]; q++) if ((pkb[q] >> 17) != 0x7fffu) bad = true;
            }
            if (stale_d >= len && lane == 0)
                for (int q = 0; q < stale_p1; q++) if ((pkb[q] >> 17) != 0x7fffu) bad = true;
            if (__ballot(bad) != 0ull) { if (lane == 0) { sg[SEG_FLAG] = -1; sg[SEG_NROWS] = 0; } return true; }
            // ---- straighten body, lane = frame index d: apply this frame's points in (track rank, arrival) order
            if (use_tbl) {
                const uint32_t stale_m = tblm[len];
                // The slots live in the frame's row of `fr` (LDS) while the points are applied: a point reads the one float it compares with and writes its
                // three — the nine selects of a register copy cost more than the round trip.  The slots hold bins (small integers, or 0) as floats, so the
                // reference's `cur > floor && cur < f` (f64) is decided in fp32: cur > floor <=> cur >= floor(floor) + 1 (clamped to 256: no bin reaches it;
                // a negative floor admits every bin, -0 admits cur > 0; a NaN floor admits none, as there), cur < f exactly.
                float thr_f;
                { const double t0 = floor_ < 0 ? 0.0 : floor(floor_) + 1.0; thr_f = (float)(t0 > 256.0 ? 256.0 : t0); }
                for (int base = 0; base < ((WSA_TUNE(8)) ? 0 : len); base += 64) {
                    const int d = base + lane;
                    float* const row = fr + 9 * (d < len ? d : 0);
                    if (d < len) {
#pragma unroll
                        for (int q = 0; q < 9; q++) row[q] = 0.f;
                    }
                    float sm = 0.f;
                    auto apply = [&](int q) __attribute__((always_inline)) {
                        const uint32_t w = pkb[q];
                        int l = (int)((w >> 17) & 3u);
                        const double E = energy_of(q);
                        const float ff = (float)(w & 0xffu), cur = row[3 * l];
                        if (cur >= thr_f && cur < ff && l < 2) l++;
                        row[3 * l] = ff; row[3 * l + 1] = (float)E; row[3 * l + 2] = (float)((w >> 8) & 0x1ffu);
                        sm = (float)((double)sm + E);
                    };
                    const uint32_t m_main = d < len ? tblm[d] : 0u, m_st = (d < len && d == stale_d) ? stale_m : 0u;
                    uint32_t mm = m_main | m_st;
                    while (mm) {                                   // ranks in ascending order; of one rank the stale frame's point first (it arrived first)
                        const int r = __ffs((int)mm) - 1; mm &= mm - 1u;
                        if ((m_st >> r) & 1u) apply((int)tbl[len * n_part + r] - 1);
                        if ((m_main >> r) & 1u) apply((int)tbl[d * n_part + r] - 1);
                    }
                    if (d < len) smv[d] = sm;
                }
            } else
            for (int base = 0; base < ((WSA_TUNE(8)) ? 0 : len); base += 64) {
                const int d = base + lane;
                if (d < len) {
                    float f9[9];
#pragma unroll
                    for (int q = 0; q < 9; q++) f9[q] = 0.f;
                    float sm = 0.f;
                    const int a1 = (stale_d == d) ? stale_p1 : 0;
                    const int dg = W.d_gen[d], dp0 = W.d_p0[d], dp1 = W.d_p1[d];                         // one round trip, not two
                    const bool has_main = dg == gen;
                    const int b0 = has_main ? dp0 : 0, b1 = has_main ? dp1 : 0;
                    int last_key = -1;
                    for (;;) {
                        int best_key = 0x7fffffff, best_q = -1;
                        for (int q = 0; q < a1; q++) {
                            const uint32_t w = pkb[q];
                            const int key = (int)(((w >> 19) << 16) | (uint32_t)q);            // rank << 16 | arrival
                            if ((w >> 17) != 0x7fffu && key > last_key && key < best_key) { best_key = key; best_q = q; }
                        }
                        for (int q = b0; q < b1; q++) {
                            const uint32_t w = pkb[q];
                            const int key = (int)(((w >> 19) << 16) | (uint32_t)q);
                            if ((w >> 17) != 0x7fffu && key > last_key && key < best_key) { best_key = key; best_q = q; }
                        }
                        if (best_q < 0) break;
                        last_key = best_key;
                        const uint32_t w = pkb[best_q];
                        int l = (int)((w >> 17) & 3u);
                        const double f = w & 0xffu, wd = (w >> 8) & 0x1ffu, E = energy_of(best_q);
                        const float cur = l == 0 ? f9[0] : (l == 1 ? f9[3] : f9[6]);
                        if ((double)cur > floor_ && (double)cur < f && l < 2) l++;
                        const float ff = (float)f, Ef = (float)E, wf = (float)wd;
                        if (l == 0) { f9[0] = ff; f9[1] = Ef; f9[2] = wf; }
                        else if (l == 1) { f9[3] = ff; f9[4] = Ef; f9[5] = wf; }
                        else { f9[6] = ff; f9[7] = Ef; f9[8] = wf; }
                        sm = (float)((double)sm + E);
                    }
#pragma unroll
                    for (int q = 0; q < 9; q++) fr[9 * d + q] = f9[q];
                    smv[d] = sm;
                }
            }
            wsync();
            if (WSA_TUNE(16)) ph[1] = __builtin_readcyclecounter();
            // levels 4 / 10 hand out the straightened frames themselves (ref @B28124, @B27713)
            if (p.formants && (p.level == 4 || p.level == 10)) {
                // (frame index & ring_mask: a batch's mask is all ones, a stream keeps the frames in its ring like the frame records)
                for (int q = lane; q < 9 * len; q += 64) { const int d = q / 9; p.formants[((uint64_t)foff + (((uint32_t)start + (uint32_t)d) & p.ring_mask)) * 9 + (uint32_t)(q - 9 * d)] = fr[q]; }
                if (p.sums) { for (int q = lane; q < len; q += 64) p.sums[(uint64_t)foff + (((uint32_t)start + (uint32_t)q) & p.ring_mask)] = smv[q]; }
            }
            double accS, accC; acc_totals(accS, accC);
            const double cs = accC / accS;
            const double lg_ctx = jsm::log10(ctx_max);
            // scratch of the feature reductions: what the points and the straighten table occupied (dead by now), when it is large enough
            double* const red = reinterpret_cast<double*>(s_big + off_pt);
            if (p.level == 4 || p.level == 5) {
                const long long r0 = take_rows(1);
                if (r0 < 0) return true;
                double* x = p.row_feat + (uint64_t)r0 * WSA_NFEAT;
                if (WSA_TUNE(16)) ph[2] = __builtin_readcyclecounter();
                if (p.level == 5) {
                    if (!(WSA_TUNE(4))) formant_features_lds(fr, len, ctx_max, x, lane, red, !G && len <= 15 && !(p.dbg & 65536), G || (p.dbg & 131072) != 0);
                    if (WSA_TUNE(16)) ph[3] = __builtin_readcyclecounter();
                    if (lane == 0) { x[0] = len; x[1] = sqrt((double)len); x[2] = cs; x[3] = lg_ctx; x[4] = floor_; }
                } else if (lane < WSA_NFEAT) x[lane] = 0;
                if (lane == 0) {
                    int32_t* m = p.row_meta + (uint64_t)r0 * 8;
                    m[0] = (int32_t)clip; m[1] = 0; m[2] = 0; m[3] = 0; m[4] = my_seg; m[5] = 0; m[6] = start; m[7] = len;
                    sg[SEG_FLAG] = 1; sg[SEG_NROWS] = 1; sg[SEG_ROW0] = (int32_t)r0;
                }
                return true;
            }
            // ---- levels 10 / 13: sep_syllables (ref @B34757), then one feature row per syllable.  Lane k keeps
            //      syllable k (the 65th and later ones of a very long segment go through the global scratch).
            int nsyl = 0, my_si = 0, my_sl = 0;
            {
                int si = -1, cc = 0, uu = 0;
                for (int base = 0; base < len; base += 64) {
                    const int dd = base + lane;
                    const float smq = dd < len ? smv[dd] : 0.f;
                    const int lim = min(64, len - base);
                    for (int j = 0; j < lim; j++) {
                        const int e2 = base + j;
                        const double v = __builtin_bit_cast(float, read_lane_i32(__builtin_bit_cast(int, smq), j));
                        if (v > floor_) { cc = 0; uu++; if (si < 0) si = e2; } else cc++;
                        if ((uu > 20 && cc > 0) || (uu > 10 && cc > 1) || (uu > 0 && cc > 4) || (e2 >= len - 1 && uu > 4)) {
                            const int t = e2 - cc;
                            if (t - si > 1) {
                                if (nsyl < 64) { if (lane == nsyl) { my_si = si; my_sl = t - si; } }
                                else if (lane == 0) { W.q_idx[2 * nsyl] = si; W.q_idx[2 * nsyl + 1] = t - si; }
                                nsyl++;
                                si = -1; uu = 0;
                            }
                        }
                    }
                }
            }
            wsync();
            long long r0 = 0;
            if (nsyl > 0) { r0 = take_rows(nsyl); if (r0 < 0) return true; }
            for (int k = 0; k < nsyl; k++) {
                const int si = k < 64 ? read_lane_i32(my_si, k) : W.q_idx[2 * k], sl = k < 64 ? read_lane_i32(my_sl, k) : W.q_idx[2 * k + 1];
                double* x = p.row_feat + (uint64_t)(r0 + k) * WSA_NFEAT;
                if (p.level == 13) {
                    if (!(WSA_TUNE(4))) formant_features_lds(fr + 9 * si, sl, ctx_max, x, lane, red, !G && sl <= 15 && !(p.dbg & 65536), G || (p.dbg & 131072) != 0);
                    if (lane == 0) { x[0] = sl; x[1] = sqrt((double)sl); x[2] = cs; x[3] = lg_ctx; x[4] = floor_; }
                } else if (lane < WSA_NFEAT) x[lane] = 0;
                if (lane == 0) {
                    int32_t* m = p.row_meta + (uint64_t)(r0 + k) * 8;
                    m[0] = (int32_t)clip; m[1] = 0; m[2] = si; m[3] = sl; m[4] = my_seg; m[5] = k; m[6] = start + si; m[7] = sl;
                }
            }
            if (lane == 0) { sg[SEG_FLAG] = nsyl > 0 ? 1 : 0; sg[SEG_NROWS] = nsyl; sg[SEG_ROW0] = (int32_t)r0; }
            return true;
        };
        auto finalize_fast = [&]() __attribute__((always_inline)) -> bool {
            if (finalize_fast_impl(std::false_type{})) return true;
            if constexpr (SPLIT == 2) { if (!(p.dbg & 262144)) return finalize_fast_impl(std::true_type{}); }      // (WSA_DBG bit 262144, tests: the third form off)
            return false;
        };
        auto finalize_slow = [&]() __attribute__((always_inline)) {
            if (WSA_TUNE(16)) ph[0] = ph[1] = ph[2] = ph[3] = __builtin_readcyclecounter();
            // ---- get_ranked_formants (ref @B35670): count >= 2 and mean bin >= 7, stable ascending
            int nq = 0;
            for (int base = 0; base < n_tr; base += 64) {
                const int t = base + lane;
                bool q = false; double mb = 0;
                if (t < n_tr) {
                    W.tr_slot[t] = -1;
                    if (W.tr_len[t] >= 2) { mb = W.tr_sumEbin[t] / W.tr_sumE[t]; q = mb >= 7; }
                }
                const uint64_t mask = __ballot(q);
                if (q) { const int pos = nq + __popcll(mask & lanemask_lt(lane)); W.q_idx[pos] = t; W.q_mb[pos] = mb; }
                nq += __popcll(mask);
            }
            wsync();
            // ranking scratch: LDS when the qualified tracks fit (they almost always do), else the
            // global arrays; generic pointers serve both
            const bool q_lds = nq <= AC;
            double* qmb = W.q_mb; int32_t* qidx = W.q_idx; int32_t* sorted = W.sorted;
            if (q_lds) {
                for (int qi = lane; qi < nq; qi += 64) { f_qmb[qi] = W.q_mb[qi]; f_qidx[qi] = W.q_idx[qi]; }
                qmb = f_qmb; qidx = f_qidx; sorted = f_sorted;
                wsync();
            }
            for (int base = 0; base < nq; base += 64) {
                const int qi = base + lane;
                if (qi < nq) {
                    const double mb = qmb[qi];
                    int rank = 0;
                    for (int u = 0; u < nq; u++) { const double o = qmb[u]; rank += (o < mb || (o == mb && u < qi)) ? 1 : 0; }
                    sorted[rank] = qi;
                }
            }
            wsync();
            // ---- slot assignment of straighten_formants (ref @B35074, first loop header): walking the ranked tracks, `if |mb - last| > 20: last = mb, slot++,
            //      stop at slot 3`.  Lane = rank, each jump found by a ballot (as in finalize_fast): one lane walking the ranks was a chain of three dependent
            //      global loads per rank wherever the ranking scratch does not fit LDS (a 266-frame segment has ~180 qualified tracks: 0.4 ms of its finalize).
            //      A track's key goes into ONE word, rank << 2 | slot (-1: takes no part), so that a point needs one gather, not two.
            {
                double last = 0; int slot = 0; bool stopped = false;
                for (int base = 0; base < nq && !stopped; base += 64) {
                    const int r = base + lane;
                    double mb = 0; int t = 0;
                    if (r < nq) { const int qi = sorted[r]; mb = qmb[qi]; t = qidx[qi]; }
                    uint64_t todo = __ballot(r < nq);
                    int myslot = -1;
                    while (todo) {
                        const uint64_t jm = __ballot(((todo >> lane) & 1ull) && fabs(mb - last) > 20);
                        if (jm == 0ull) { if ((todo >> lane) & 1ull) myslot = slot; break; }
                        const int j = __ffsll((long long)jm) - 1;
                        if (((todo >> lane) & 1ull) && lane < j) myslot = slot;
                        last = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(mb), j), __builtin_amdgcn_readlane(__double2loint(mb), j));
                        slot++;
                        if (slot >= 3) { stopped = true; break; }
                        todo &= ~lanemask_lt(j);
                    }
                    if (myslot >= 0) W.tr_slot[t] = (r << 2) | myslot;
                }
            }
            wsync();
            // every point gets its application key once: (rank of its track) << 2 | slot, or -1 when the
            // track takes no part (lane = point; the frame lanes below then read keys, not track tables); the next round's track ids are on their way
            // while this round's keys are gathered
            {
                int t_nxt = lane < n_pt ? W.pt[lane].x : 0;
                for (int q = lane; q < n_pt; q += 64) {
                    const int t = t_nxt;
                    if (q + 64 < n_pt) t_nxt = W.pt[q + 64].x;
                    W.pt_key[q] = W.tr_slot[t];
                }
            }
            wsync();
            if (WSA_TUNE(16)) ph[0] = __builtin_readcyclecounter();
            // ---- a point of a processed track filed at an index >= len makes the reference throw
            //      (r[d] undefined, ref @B35484): segments_ci keeps the entry, nothing else is stored
            bool bad = false;
            for (int base = len; base <= c_ci + 1; base += 64) {
                const int d = base + lane;
                if (d <= c_ci + 1 && W.d_gen[d] == gen)
                    for (int q = W.d_p0[d]; q < W.d_p1[d]; q++) if (W.pt_key[q] >= 0) bad = true;
            }
            if (stale_d >= len && lane == 0)
                for (int q = 0; q < stale_p1; q++) if (W.pt_key[q] >= 0) bad = true;
            if (__ballot(bad) != 0ull) { if (lane == 0) { sg[SEG_FLAG] = -1; sg[SEG_NROWS] = 0; } return; }
            // ---- straighten body, lane = frame index d: apply this frame's points in
            //      (track rank, arrival) order
            // the q_* scratch is dead from here on; fr / sm of the segment go to LDS when they fit
            float* const fr = len <= FRCAP ? f_fr : W.fr;
            float* const smv_ = len <= FRCAP ? f_sm : W.sm1;
            // the features of a span whose frames live in HBM (more frames than the block holds): the block is free then, and the wave-parallel reductions of the
            // LDS path run on it with the frames read through flat loads — formant_features_wave walks the energy events frame by frame on ONE lane, a dependent
            // global round trip per frame, which made a 266-frame segment's finalize 600 us and with it the whole kernel (the application's settings at 48 kHz)
            auto slow_features = [&](const float* f, int a, double* x) __attribute__((always_inline)) {
                if (len > FRCAP) formant_features_lds(f, a, ctx_max, x, lane, reinterpret_cast<double*>(s_big), false, true);
                else formant_features_wave(f, a, ctx_max, x, W.Aev, aev_stride, lane);
            };
            for (int base = 0; base < ((WSA_TUNE(8)) ? 0 : len); base += 64) {
                const int d = base + lane;
                if (d < len) {
                    float f9[9];
#pragma unroll
                    for (int q = 0; q < 9; q++) f9[q] = 0.f;
                    float sm = 0.f;
                    const int a0 = 0, a1 = (stale_d == d) ? stale_p1 : 0;
                    const bool has_main = W.d_gen[d] == gen;
                    const int b0 = has_main ? W.d_p0[d] : 0, b1 = has_main ? W.d_p1[d] : 0;
                    long long last_key = -1;
                    for (;;) {
                        long long best_key = 0x7fffffffffffffffLL; int best_q = -1;
                        for (int part = 0; part < 2; part++) {
                            const int q0 = part ? b0 : a0, q1 = part ? b1 : a1;
                            for (int q = q0; q < q1; q++) {
                                const int pk = W.pt_key[q];
                                if (pk < 0) continue;
                                const long long key = (long long)(pk >> 2) * (long long)(p.pcap + 1) + q;
                                if (key > last_key && key < best_key) { best_key = key; best_q = q; }
                            }
                        }
                        if (best_q < 0) break;
                        last_key = best_key;
                        int l = W.pt_key[best_q] & 3;
                        const int4 rec4 = W.pt[best_q];
                        const int bw = rec4.y;
                        const double f = bw & 0xff, wd = (bw >> 8) & 0x1ff, E = __hiloint2double(rec4.w, rec4.z);
                        const float cur = l == 0 ? f9[0] : (l == 1 ? f9[3] : f9[6]);
                        if ((double)cur > floor_ && (double)cur < f && l < 2) l++;
                        const float ff = (float)f, Ef = (float)E, wf = (float)wd;
                        if (l == 0) { f9[0] = ff; f9[1] = Ef; f9[2] = wf; }
                        else if (l == 1) { f9[3] = ff; f9[4] = Ef; f9[5] = wf; }
                        else { f9[6] = ff; f9[7] = Ef; f9[8] = wf; }
                        sm = (float)((double)sm + E);
                    }
#pragma unroll
                    for (int q = 0; q < 9; q++) fr[9 * d + q] = f9[q];
                    smv_[d] = sm;
                }
            }
            wsync();
            if (WSA_TUNE(16)) ph[1] = __builtin_readcyclecounter();
            // levels 4 / 10 hand out the straightened frames themselves (ref @B28124, @B27713): the segment's
            // [len][9] fp32 frames go to formants[frame_off[clip] + start + d] (segments never overlap)
            if (p.formants && (p.level == 4 || p.level == 10)) {
                for (int q = lane; q < 9 * len; q += 64) { const int d = q / 9; p.formants[((uint64_t)foff + (((uint32_t)start + (uint32_t)d) & p.ring_mask)) * 9 + (uint32_t)(q - 9 * d)] = fr[q]; }
                if (p.sums) { for (int q = lane; q < len; q += 64) p.sums[(uint64_t)foff + (((uint32_t)start + (uint32_t)q) & p.ring_mask)] = smv_[q]; }
            }
            double accS, accC; acc_totals(accS, accC);
            const double cs = accC / accS;
            const double lg_ctx = jsm::log10(ctx_max);
            if (p.level == 4 || p.level == 5) {
                const long long r0 = take_rows(1);
                if (r0 < 0) return;
                double* x = p.row_feat + (uint64_t)r0 * WSA_NFEAT;
                if (WSA_TUNE(16)) ph[2] = __builtin_readcyclecounter();
                if (p.level == 5) {
                    if (!(WSA_TUNE(4))) slow_features(fr, len, x);
                    if (WSA_TUNE(16)) ph[3] = __builtin_readcyclecounter();
                    if (lane == 0) { x[0] = len; x[1] = sqrt((double)len); x[2] = cs; x[3] = lg_ctx; x[4] = floor_; }
                } else if (lane < WSA_NFEAT) x[lane] = 0;
                if (lane == 0) {
                    int32_t* m = p.row_meta + (uint64_t)r0 * 8;
                    m[0] = (int32_t)clip; m[1] = 0; m[2] = 0; m[3] = 0; m[4] = my_seg; m[5] = 0; m[6] = start; m[7] = len;
                    sg[SEG_FLAG] = 1; sg[SEG_NROWS] = 1; sg[SEG_ROW0] = (int32_t)r0;
                }
                return;
            }
            // ---- levels 10 / 13: sep_syllables (ref @B34757), then one feature row per syllable.
            // pass 1 finds the syllables (sequential scan over the frame sums), pass 2 fills the rows.
            int nsyl = 0;
            {
                int si = -1, cc = 0, uu = 0;
                for (int base = 0; base < len; base += 64) {
                    const int dd = base + lane;
                    const float smv = dd < len ? smv_[dd] : 0.f;
                    const int lim = min(64, len - base);
                    for (int j = 0; j < lim; j++) {
                        const int e2 = base + j;
                        const double v = __builtin_bit_cast(float, read_lane_i32(__builtin_bit_cast(int, smv), j));
                        if (v > floor_) { cc = 0; uu++; if (si < 0) si = e2; } else cc++;
                        if ((uu > 20 && cc > 0) || (uu > 10 && cc > 1) || (uu > 0 && cc > 4) || (e2 >= len - 1 && uu > 4)) {
                            const int t = e2 - cc;
                            if (t - si > 1) {
                                if (lane == 0) { W.q_idx[2 * nsyl] = si; W.q_idx[2 * nsyl + 1] = t - si; }   // q_idx is free again here
                                nsyl++;
                                si = -1; uu = 0;
                            }
                        }
                    }
                }
            }
            wsync();
            long long r0 = 0;
            if (nsyl > 0) { r0 = take_rows(nsyl); if (r0 < 0) return; }
            for (int k = 0; k < nsyl; k++) {
                const int si = W.q_idx[2 * k], sl = W.q_idx[2 * k + 1];
                double* x = p.row_feat + (uint64_t)(r0 + k) * WSA_NFEAT;
                if (p.level == 13) {
                    if (!(WSA_TUNE(4))) slow_features(fr + 9 * si, sl, x);
                    if (lane == 0) { x[0] = sl; x[1] = sqrt((double)sl); x[2] = cs; x[3] = lg_ctx; x[4] = floor_; }
                } else if (lane < WSA_NFEAT) x[lane] = 0;
                if (lane == 0) {
                    int32_t* m = p.row_meta + (uint64_t)(r0 + k) * 8;
                    m[0] = (int32_t)clip; m[1] = 0; m[2] = si; m[3] = sl; m[4] = my_seg; m[5] = k; m[6] = start + si; m[7] = sl;
                }
            }
            if (lane == 0) { sg[SEG_FLAG] = nsyl > 0 ? 1 : 0; sg[SEG_NROWS] = nsyl; sg[SEG_ROW0] = (int32_t)r0; }
        };

        // ---- frames of the span.  Per frame gate.hip left: info (filing index | stale << 30, or -1 when
        //      accumulate_fm is not called), v (acceptance floor), fl (floor handed to accumulate_fm).
        //      Everything of frame f+1 is requested before frame f is processed.
        // The header words are the same for all lanes, but they are loaded through a lane-dependent zero offset
        // (vz) so that the compiler treats them as ordinary vector data: knowing them uniform it wants them in
        // SGPRs the moment they are loaded (v_readfirstlane behind an s_waitcnt), which turned every header load
        // into an exposed memory round trip.  They become scalars (uni_*) only where they are consumed.
        struct Hdr { int info; double v, fl; uint4 h; };                 // h = the frame's record header, as loaded (decoded where it is consumed)
        struct Pre { int info; double v, fl, g; int n; uint32_t pk, amp, plo, phi, hi; };
        auto uni_i = [](int x) __attribute__((always_inline)) { return __builtin_amdgcn_readfirstlane(x); };
        auto uni_d = [](double x) __attribute__((always_inline)) {
            return __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(x)), __builtin_amdgcn_readfirstlane(__double2loint(x)));
        };
        auto load_hdr = [&](uint32_t f, Hdr& q) __attribute__((always_inline)) {      // branch-free: frames past the span read its last frame
            const uint32_t fi = (min(f, f_end - 1) & p.ring_mask) + (uint32_t)vz;
            q.info = p.fr_info[foff + fi]; q.v = p.fr_v[foff + fi]; q.fl = p.fr_fl[foff + fi];
            q.h = p.rec.hdr[foff + fi];
        };
        auto load_ent = [&](uint32_t f, const Hdr& h, Pre& q) __attribute__((always_inline)) {
            const int hy = uni_i((int)h.h.y);
            q.info = f < f_end ? uni_i(h.info) : -1; q.v = uni_d(h.v); q.fl = uni_d(h.fl);
            q.g = (double)(hy & 0xff) * 4294967296.0 + (double)(uint32_t)uni_i((int)h.h.x);      // exact: g < 2^40
            q.n = (hy >> 8) & 0xff; q.pk = q.amp = q.plo = q.phi = q.hi = 0;
            if (q.info >= 0 && lane < q.n && !(WSA_TUNE(32))) {           // only frames accumulate_fm sees, only the entries they hold
                const uint32_t c = (uint32_t)uni_i((int)h.h.w) + (uint32_t)lane;
                const uint4 e4 = p.rec.ent[c];
                q.amp = p.rec.amp[c]; q.pk = e4.x; q.plo = e4.y; q.phi = e4.z; q.hi = e4.w;
            }
        };
        // ---- accumulate_fm for one frame (ref @B35952); `cur` = the frame's header words and this lane's candidate entry
        // `refill` runs exactly once, as soon as this lane's candidate entry of `cur` is no longer needed (behind the compaction
        // of the accepted peaks): the batch path requests a later frame's entry into the same registers there
        auto accumulate = [&](const Pre& cur, auto&& refill) __attribute__((always_inline)) {
            const int info = cur.info;
            if (!(info >= 0 && !(WSA_TUNE(2)))) refill();
            else {
                {
                    const int ncand = cur.n;
                    const double g = cur.g, v = cur.v;
                    const uint32_t pkw = cur.pk, amp = cur.amp;
                    // exact prefix sums P[i-1], P[s] (< 2^40) from their low words and high bytes
                    const double plo = (double)(cur.hi & 0xffu) * 4294967296.0 + (double)cur.plo, phi = (double)((cur.hi >> 8) & 0xffu) * 4294967296.0 + (double)cur.phi;
                    const bool reset_this_frame = (info >> 30) & 1;
                    const int t_idx = info & 0x3fffffff;
                    // accepted peaks (ref @B25827: `e[l] > v`), lane = candidate
                    const bool acc = lane < ncand && (double)amp > v;
                    const uint64_t amask = __ballot(acc);
                    const int n = __popcll(amask);
                    if (n < 1) refill();
                    // ---- accumulate_fm(e, peaks, t_idx, g, floor_) (ref @B35952)
                    if (n >= 1) {
                        if (WSA_TUNE(512)) act = __builtin_readcyclecounter();
                        const int nfile = t_idx;
                        const double fl = cur.fl;
                        accG += g;
                        // compact the accepted peaks: lane o < n owns peak o
                        const int my_o = __popcll(amask & lanemask_lt(lane));
                        if (acc) { s_pk[my_o] = pkw; s_amp[my_o] = amp; s_plo[my_o] = plo; s_phi[my_o] = phi; }
                        refill();
                        wsync();
                        int pk_i = 0, pk_s = 0, pk_l = -1000; uint32_t pk_amp = 0; double pk_plo = 0, pk_phi = 0;
                        if (lane < n) {
                            const uint32_t w = s_pk[lane];
                            pk_i = w & 0xff; pk_s = (w >> 8) & 0xff; pk_l = (w >> 16) & 0xff;
                            pk_amp = s_amp[lane]; pk_plo = s_plo[lane]; pk_phi = s_phi[lane];
                        }
                        WSA_ACP(0);
                        // 1. retire tracks whose last filing index is 4 or more behind (gap only grows); most frames retire
                        //    nothing from a block of 64, which then stays as it is
                        {
                            int kept = 0;
                            for (int base = 0; base < n_act; base += 64) {
                                const int j = base + lane;
                                const bool valid = j < n_act;
                                const int lf = valid ? a_last_frame[j] : 0;
                                const bool keep = valid && (nfile - lf) < 4;
                                const uint64_t km = __ballot(keep);
                                if (kept == base && km == __ballot(valid)) { kept += __popcll(km); continue; }
                                int ln = 0, gi = 0; uint32_t bn = 0, am = 0; double ve = 0, se = 0, sb = 0;
                                if (valid) { ln = a_len[j]; gi = a_gid[j]; bn = a_bins[j]; am = a_amp[j]; ve = a_vel[j]; se = a_sumE[j]; sb = a_sumEbin[j]; }
                                if (valid && !keep) { W.tr_len[gi] = ln; W.tr_sumE[gi] = se; W.tr_sumEbin[gi] = sb; }   // the summary finalize ranks by
                                wsync();
                                if (keep) {
                                    const int q = kept + __popcll(km & lanemask_lt(lane));
                                    a_last_frame[q] = lf; a_len[q] = ln; a_gid[q] = gi; a_bins[q] = bn; a_amp[q] = am; a_vel[q] = ve; a_sumE[q] = se; a_sumEbin[q] = sb;
                                }
                                kept += __popcll(km);
                                wsync();
                            }
                            n_act = kept;
                        }
                        WSA_ACP(1);
                        // 2. score every (track, peak) pair inside the track's search window; per peak keep
                        //    the best score > 1, the EARLIER track on ties (ref: `i>1&&i>d[o]` in track order)
                        int asg = -1; double best = 0;
                        for (int tbase = 0; tbase < n_act; tbase += 64) {
                            const int j = tbase + lane;
                            const bool valid = j < n_act;
                            int gap = -1, bin = 0;
                            if (valid) { gap = nfile - a_last_frame[j]; bin = (int)(a_bins[j] & 0xff); a_mmask[j] = 0ull; }
                            const bool live = valid && gap >= 0 && gap < 4;
                            const int win = gap == 0 ? 3 : (gap == 1 ? 4 : (gap == 2 ? 6 : 9));      // ref @B32325
                            int o_lo = 0, o_hi = 0;
                            for (int o = 0; o < n; o++) {
                                const int lo = __builtin_amdgcn_readlane(pk_l, o);
                                o_lo += (lo <= bin - win) ? 1 : 0;
                                o_hi += (lo < bin + win) ? 1 : 0;
                            }
                            const int cnt = live ? o_hi - o_lo : 0;
                            const int incl = (int)wave_incl_scan_u32((uint32_t)cnt);
                            const int off = incl - cnt;
                            const int M = __builtin_amdgcn_readlane(incl, 63);
                            const int maxc = (int)wave_max_u32((uint32_t)cnt);
                            for (int base = 0; base < M; base += 64) {
                                if (lane < MAXC) { s_best[lane] = 0ull; s_asg[lane] = 0x7fffffff; }
                                for (int c = 0; c < maxc; c++) {
                                    const int slot = off + c - base;
                                    if (c < cnt && slot >= 0 && slot < 64) { s_pr_j[slot] = j; s_pr_o[slot] = o_lo + c; }
                                }
                                wsync();
                                const bool pv = base + lane < M;
                                int jj = 0, oo = 0; double sc = 0;
                                if (pv) {
                                    jj = s_pr_j[lane]; oo = s_pr_o[lane];
                                    const int tb = (int)(a_bins[jj] & 0xff), tg = nfile - a_last_frame[jj];
                                    const int pl = (int)((s_pk[oo] >> 16) & 0xff);
                                    sc = match_score(tg, (double)abs(tb - pl), (double)a_len[jj], (double)tb, (double)pl,
                                                     (double)a_amp[jj], (double)s_amp[oo], a_vel[jj]);
                                    if (sc > 1) atomicMax(&s_best[oo], (unsigned long long)__double_as_longlong(sc));
                                }
                                wsync();
                                if (pv && sc > 1 && (unsigned long long)__double_as_longlong(sc) == s_best[oo]) atomicMin(&s_asg[oo], jj);
                                wsync();
                                if (lane < n) {
                                    const int cj = s_asg[lane];
                                    if (cj != 0x7fffffff) {
                                        const double cs = __longlong_as_double((long long)s_best[lane]);
                                        if (cs > best) { best = cs; asg = cj; }
                                    }
                                }
                                wsync();
                            }
                        }
                        WSA_ACP(2);
                        // 3. hand each matched track the set of its peaks
                        if (lane < n && asg >= 0) atomicOr(&a_mmask[asg], 1ull << lane);
                        wsync();
                        const int p_begin = n_pt;
                        // 4. matched tracks update themselves (lane = track), points in track order
                        for (int tbase = 0; tbase < n_act; tbase += 64) {
                            const int j = tbase + lane;
                            const unsigned long long mm = j < n_act ? a_mmask[j] : 0ull;
                            bool upd = false; int pb = 0, st = 0, en = 0; uint32_t a0 = 0; double be = 0;
                            if (mm) {
                                const int first = __ffsll((long long)mm) - 1;
                                const uint32_t w0 = s_pk[first];
                                pb = (w0 >> 16) & 0xff;
                                a0 = s_amp[first];                       // amplitude of the FIRST assigned peak (quirk 3)
                                if ((double)a0 > fl) {
                                    upd = true;
                                    st = w0 & 0xff; en = (w0 >> 8) & 0xff;
                                    double lo_sum = s_plo[first], hi_sum = s_phi[first];
                                    uint32_t pb_amp = a0;
                                    unsigned long long rest = mm & (mm - 1ull);       // (the first assigned peak is where st / en / pb start from)
                                    while (rest) {
                                        const int o = __ffsll((long long)rest) - 1; rest &= rest - 1;
                                        const uint32_t w = s_pk[o];
                                        const int oi = w & 0xff, os = (w >> 8) & 0xff, ol = (w >> 16) & 0xff;
                                        if (os > en) { en = os; hi_sum = s_phi[o]; }
                                        if (oi < st) { st = oi; lo_sum = s_plo[o]; }
                                        if (s_amp[o] > pb_amp) { pb = ol; pb_amp = s_amp[o]; }
                                    }
                                    be = hi_sum - lo_sum;                // sum e[st..en], exact
                                }
                            }
                            const uint64_t um = __ballot(upd);
                            const int nu = __popcll(um);
                            if (n_pt + nu > p.pcap) { overflow = true; }
                            else if (upd) {
                                const int q = n_pt + __popcll(um & lanemask_lt(lane));
                                const int hlen = a_len[j];
                                const uint32_t bn = a_bins[j];
                                const int P1 = bn & 0xff, P2 = (bn >> 8) & 0xff, P3 = (bn >> 16) & 0xff;
                                double vel = a_vel[j];
                                if (hlen >= 3) {      // x / 3, correctly rounded: q = x * (1/3), r = x - 3q (exact), q + r * (1/3)
                                    const double xv = (double)((pb - P1) + (P2 - P1) + (P3 - P2)), third = 1.0 / 3.0;
                                    const double q0 = xv * third;
                                    vel = __builtin_fma(__builtin_fma(-3.0, q0, xv), third, q0);
                                }
                                else if (hlen == 2) vel = (double)((pb - P1) + (P2 - P1)) / 2;
                                else if (hlen == 1) vel = (double)(pb - P1);
                                const double se = a_sumE[j] + be, sb = a_sumEbin[j] + be * pb;
                                a_vel[j] = vel; a_bins[j] = (uint32_t)pb | ((uint32_t)P1 << 8) | ((uint32_t)P2 << 16);
                                a_amp[j] = a0; a_last_frame[j] = nfile; a_len[j] = hlen + 1; a_sumE[j] = se; a_sumEbin[j] = sb;
                                const int t = a_gid[j];
                                W.pt[q] = make_int4(t, pb | ((en - st + 1) << 8) | (min(nfile, 0x7fff) << 17), __double2loint(be), __double2hiint(be));
                                if (RAW) W.ptx[q] = make_int4(st, (int)a0, nfile, en);
                            }
                            if (upd) accL += be;                     // integer-valued: exact in any order
                            if (!overflow) n_pt += nu;
                        }
                        WSA_ACP(3);
                        // 5. unassigned peaks above the floor open new tracks, in peak order (lane = peak)
                        const bool mk = lane < n && asg == -1 && (double)pk_amp > fl;
                        const uint64_t nm = __ballot(mk);
                        const int nnew = __popcll(nm);
                        // (WSA_DBG bit 10, tests: the LDS table of the default variant pretends to hold 12 tracks, so that the rerun path runs on ordinary input)
                        if (n_act + nnew > ((p.dbg & 1024) && AC < AC_MAX ? 12 : AC)) { act_overflow = true; overflow = true; }
                        if (n_tr + nnew > p.tcap || n_pt + nnew > p.pcap) overflow = true;
                        if (overflow) {}
                        else if (mk) {
                            const int r = __popcll(nm & lanemask_lt(lane));
                            const int t = n_tr + r, q = n_pt + r, j = n_act + r;
                            const double be = pk_phi - pk_plo;
                            a_last_frame[j] = nfile; a_len[j] = 1; a_gid[j] = t; a_bins[j] = (uint32_t)pk_l; a_amp[j] = pk_amp;
                            a_vel[j] = 0; a_sumE[j] = be; a_sumEbin[j] = be * pk_l;
                            W.pt[q] = make_int4(t, pk_l | ((pk_s - pk_i + 1) << 8) | (min(nfile, 0x7fff) << 17), __double2loint(be), __double2hiint(be));
                            if (RAW) W.ptx[q] = make_int4(pk_i, (int)pk_amp, nfile, pk_s);
                        }
                        if (!overflow) { n_tr += nnew; n_pt += nnew; n_act += nnew; }
                        WSA_ACP(4);
                        // file this frame's point range under its (possibly stale) index
                        if (reset_this_frame) { stale_d = nfile; stale_p1 = n_pt; }
                        else if (lane == 0 && nfile < p.fcap + 2) { W.d_p0[nfile] = p_begin; W.d_p1[nfile] = n_pt; W.d_gen[nfile] = gen; }
                        wsync();
                    }
                }
            }
        };
        // ---- end of a span: the live tracks hand their summaries over, then the result part of finalize (or the level-3 export)
        auto finish_span = [&]() __attribute__((always_inline)) {
            // tracks still in the table hand their summaries over as well (the paired variant has done that for both halves already)
            if constexpr (!PAIR) { for (int j = lane; j < n_act; j += 64) { const int gi = a_gid[j]; W.tr_len[gi] = a_len[j]; W.tr_sumE[gi] = a_sumE[j]; W.tr_sumEbin[gi] = a_sumEbin[j]; } }
            wsync();
            if constexpr (RAW) {
                // ---- level 3 hands out the ranked raw tracks themselves (ref @B28273 `s.push(i)`, i = get_ranked_formants() @B35670):
                //      the span's points (arrival order) and the ranked track ids go to a pool behind the span's first frame
                //      (a frame brings at most MAXC points / tracks); the host rebuilds the 18-field records from them
                int nq = 0;
                for (int base = 0; base < n_tr; base += 64) {
                    const int t = base + lane;
                    bool q = false; double mb = 0;
                    if (t < n_tr && W.tr_len[t] >= 2) { mb = W.tr_sumEbin[t] / W.tr_sumE[t]; q = mb >= 7; }
                    const uint64_t mask = __ballot(q);
                    if (q) { const int pos = nq + __popcll(mask & lanemask_lt(lane)); W.q_idx[pos] = t; W.q_mb[pos] = mb; }
                    nq += __popcll(mask);
                }
                wsync();
                // batch: the pool entries of a span start behind its first frame's slot.  Streams: the slots are the stream's ring, so the
                // entries run modulo the ring (the host unwraps them, wsa_stream_collect); the span's first frame is what the gate noted
                const uint64_t pbase = (uint64_t)foff * MAXC;
                const uint64_t pmask = ST ? (uint64_t)(p.ring_mask + 1u) * MAXC - 1ull : ~0ull;
                const uint64_t poff = (ST ? (uint64_t)((uint32_t)sg[SEG_FBEGIN] & p.ring_mask) : (uint64_t)f_begin) * MAXC;
                const uint64_t pool0 = pbase + poff;
                auto slot = [&](uint64_t q) __attribute__((always_inline)) -> uint64_t { return pbase + ((poff + q) & pmask); };
                for (int qi = lane; qi < nq; qi += 64) {
                    const double mb = W.q_mb[qi];
                    int rank = 0;
                    for (int u = 0; u < nq; u++) { const double o = W.q_mb[u]; rank += (o < mb || (o == mb && u < qi)) ? 1 : 0; }
                    p.trk_rank[slot((uint64_t)rank)] = W.q_idx[qi];
                }
                for (int q = lane; q < n_pt; q += 64) { int4 v = W.pt[q]; v.y &= 0x1ffff; p.trk_pts[2 * slot((uint64_t)q)] = v; p.trk_pts[2 * slot((uint64_t)q) + 1] = W.ptx[q]; }   // (the filing index also sits in ptx.z)
                if (lane == 0) {
                    int32_t* ts = p.trk_seg + ((uint64_t)clip * p.seg_cap + my_seg) * 4;
                    ts[0] = (int32_t)(pool0 & 0xffffffffu); ts[1] = n_pt; ts[2] = nq; ts[3] = (int32_t)(pool0 >> 32);
                }
            } else
            if (!(WSA_TUNE(1))) { if ((p.dbg & 256) || !finalize_fast()) finalize_slow(); }
        };
        if constexpr (SPLIT == 2) {
            // ---- finalize only: the span's state comes from its header, its tracks and points from its region of the pool
            const double* hd = p.span_hdr + ((uint64_t)clip * p.seg_cap + k_seg) * 8;
            const double h0 = hd[0], h1 = hd[1], h2 = hd[2], h3 = hd[3], h4 = hd[4], h5 = hd[5], h6 = hd[6];
            const int flag = (int)h6;
            if (flag == 0) continue;                                            // on the redo list: the one-span kernel does the whole span
            if (flag & 2) { if (lane == 0) atomicOr(&p.shared[1], 1u); continue; }
            const int F = (int)(f_end - f_begin);
            W = carve_ws(p.pool + (uint64_t)(foff + f_begin) * p.pool_bpf, MAXC * F, MAXC * F, F, 0, nullptr);
            aev_stride = F + 2;
            n_tr = (int)h0; n_pt = (int)h1; n_act = 0; stale_d = (int)h2; stale_p1 = (int)h3;
            accG = h4; accL = lane == 0 ? h5 : 0.0;
            gen = 1;
            const unsigned long long tf0 = WSA_TUNE(16) ? __builtin_readcyclecounter() : 0ull;
            finish_span();
            if (WSA_TUNE(16) && lane == 0 && p.trace) {      // tuning: per-span finalize cycles and phases into the trace buffer (tools/fin_probe.py)
                double* tr = p.trace + (uint64_t)atomicAdd(&p.shared[0], 1u) * 12;
                tr[0] = 0; tr[1] = (double)(__builtin_readcyclecounter() - tf0); tr[2] = len; tr[3] = F; tr[4] = n_tr; tr[5] = n_pt; tr[6] = blockIdx.x;
                tr[7] = (double)(ph[0] - tf0); tr[8] = (double)(ph[1] - ph[0]); tr[9] = (double)(ph[2] - ph[1]); tr[10] = (double)(ph[3] - ph[2]);
            }
            if (overflow && lane == 0) atomicOr(&p.shared[1], 1u);
            wsync();
            continue;
        } else
        if constexpr (PAIR) {
            // ---- two spans per wave.  accumulate_fm keeps ~10 of a wave's 64 lanes busy (ten peaks, ten-odd live tracks), and the kernel is
            //      bound by instruction issue, so the halves of the wave track two spans in lock step: every instruction below serves both.
            //      Each half has its own active-track table (PAIR_AC entries), peak scratch and work space; quantities that are scalars in the
            //      one-span code are vector registers that hold one value per half.  A frame brings at most 32 accepted peaks here and a span
            //      at most PAIR_AC live tracks; a span that needs more is put on the redo list and tracked by the one-span kernel afterwards.
            //      Retired tracks only leave the table every fourth frame (a dead track never matches: its gap only grows), the window
            //      counts come from a bit map of the accepted peaks' bins instead of a loop over the peaks.  Finalize then runs for one
            //      span after the other with the whole wave, out of the same LDS block (both tables are dead by then).
            static_assert(GW == 32 || (GW == 16 && SPLIT == 1), "four spans per wave only as the accumulate half of the split tracker");
            constexpr int ACG = GW == 32 ? PAIR_AC : QUAD_AC, NGR = 64 / GW, GSZ = GW == 32 ? PAIR_GSZ : QUAD_GSZ;
            const int g = lane / GW, gl = lane % GW;
            const uint32_t below = (1u << gl) - 1u;
            unsigned char* const gb = s_big + g * GSZ;
            double* const t_vel = reinterpret_cast<double*>(gb);
            double* const t_sumE = t_vel + ACG;
            double* const t_sumEbin = t_sumE + ACG;
            uint32_t* const t_mmask = reinterpret_cast<uint32_t*>(t_sumEbin + ACG);
            int32_t* const t_lf = reinterpret_cast<int32_t*>(t_mmask + ACG);
            int32_t* const t_len = t_lf + ACG;
            int32_t* const t_gid = t_len + ACG;
            uint32_t* const t_bins = reinterpret_cast<uint32_t*>(t_gid + ACG);
            uint32_t* const t_amp = t_bins + ACG;
            uint32_t* const q_pk = t_amp + ACG;               // accepted peaks of the half's frame, compacted: entry word, amplitude, low words of P[i-1] / P[s], their high bytes
            uint32_t* const q_amp = q_pk + GW;
            uint32_t* const q_plo = q_amp + GW;
            uint32_t* const q_phi = q_plo + GW;
            uint32_t* const q_hi = q_phi + GW;
            unsigned long long* const q_best = reinterpret_cast<unsigned long long*>(q_hi + GW);
            int32_t* const q_asg = reinterpret_cast<int32_t*>(q_best + GW);
            int32_t* const q_prj = q_asg + GW;
            int32_t* const q_pro = q_prj + GW;
            uint32_t* const q_map = reinterpret_cast<uint32_t*>(q_pro + GW);     // {0, bins 0..31, 32..63, 64..95, 96..127, 0}: which bins hold an accepted peak
            static_assert(NGR * GSZ <= LDS_ALL && GSZ % 16 == 0 && GSZ >= ACG * 48 + GW * 40 + 24 && (ACG * 24) % 8 == 0 && (ACG * 48 + GW * 20) % 8 == 0, "group layout fits the block");
            const uint32_t e_idx = (uint32_t)(NGR * pair_idx) + (uint32_t)g;
            const bool has = e_idx < pair_total;
            const uint2 oe = has ? p.order[e_idx] : make_uint2(0u, 0u);
            const uint32_t g_clip = oe.x, g_seg = oe.y;
            const int32_t* gsg = p.seg_i + ((uint64_t)g_clip * p.seg_cap + g_seg) * 8;
            const uint32_t g_fb = has ? (uint32_t)gsg[SEG_FBEGIN] : 0u, g_fe = has ? (uint32_t)gsg[SEG_FEND] : 0u;
            const uint32_t g_foff = p.frame_off[g_clip];
            const int g_F = (int)(g_fe - g_fb);
            const int g_tcap = SPLIT ? MAXC * g_F : p.tcap, g_fcap = SPLIT ? g_F : p.fcap;            // split finalize: the span's own region, 64 tracks / points per frame
            const Ws Wg = SPLIT ? carve_ws(p.pool + (uint64_t)(g_foff + g_fb) * p.pool_bpf, g_tcap, g_tcap, g_fcap, 0, nullptr)
                                : carve_ws(p.ws + ((uint64_t)blockIdx.x * 2 + (uint32_t)g) * p.ws_stride, p.tcap, p.pcap, p.fcap, 0, nullptr);
            if (SPLIT) { if (has) for (int d = gl; d < g_F + 2; d += GW) Wg.d_gen[d] = 0; }
            int g_ntr = 0, g_npt = 0, g_nact = 0, g_stale_d = -1, g_stale_p1 = 0;
            double g_accG = 0, g_accL = 0;
            bool g_ovf = false, g_redo = SPLIT && has && g_F < 1;
            if (gl == 0) { q_map[0] = 0u; q_map[5] = 0u; }
            auto dbl40 = [](uint32_t lo, uint32_t hi8) __attribute__((always_inline)) { return (double)(hi8 & 0xffu) * 4294967296.0 + (double)lo; };
            // per frame: what gate.hip left (info, v, fl), the record header, the first 32 candidate entries; two / one frame(s) ahead
            struct FH { int info; double v, fl; uint4 h; };
            struct FC { uint4 e; uint32_t amp; };
            auto load_fh = [&](uint32_t k, FH& q) __attribute__((always_inline)) {
                const uint32_t f = g_fb + k, fi = g_foff + (g_fe > g_fb ? min(f, g_fe - 1u) : 0u);
                // 32-bit byte offsets off the (uniform) table bases: `global_load v, v_off, s[base]` instead of a 64-bit address per table (a batch holds fewer
                // than 2^28 frames: wsa_batch_create)
                auto at = [](const auto* base, uint32_t byte_off) __attribute__((always_inline)) { return *reinterpret_cast<decltype(base)>(reinterpret_cast<const char*>(base) + byte_off); };
                q.info = at(p.fr_info, fi << 2); q.v = at(p.fr_v, fi << 3); q.fl = at(p.fr_fl, fi << 3); q.h = at(p.rec.hdr, fi << 4);
                if (f >= g_fe) q.info = -1;
            };
            auto load_fc = [&](const FH& h, FC& q) __attribute__((always_inline)) {
                q.e = make_uint4(0u, 0u, 0u, 0u); q.amp = 0u;
                if (h.info >= 0 && gl < (int)((h.h.y >> 8) & 0xffu)) { const uint32_t c = h.h.w + (uint32_t)gl; q.e = p.rec.ent[c]; q.amp = p.rec.amp[c]; }
            };
            const int nsteps = groups_max_i32<GW>((int)(g_fe - g_fb));
            unsigned long long pcy[5] = {0, 0, 0, 0, 0}, pt0 = 0; int pn_chunk2 = 0, pn_pass = 0, pn_on = 0;      // tuning (WSA_DBG bit 16): cycles per phase, steps with two track chunks, pair passes
#define WSA_PCY(k_) do { if (WSA_TUNE(16)) { const unsigned long long now_ = __builtin_readcyclecounter(); pcy[k_] += now_ - pt0; pt0 = now_; } } while (0)
            const unsigned long long ptk0 = WSA_TUNE(16) ? __builtin_readcyclecounter() : 0ull;
            FH h0, h1, h2; FC c0, c1;
            load_fh(0u, h0); load_fh(1u, h1); load_fc(h0, c0);
            for (int step = 0; step < nsteps; step++) {
                load_fh((uint32_t)step + 2u, h2);
                load_fc(h1, c1);
                const bool act = h0.info >= 0 && !g_redo && !WSA_TUNE(2);      // (WSA_DBG bit 2, TUNING builds: the what-if "no accumulate" — the spans are walked, nothing is tracked)
                if (__ballot(act) != 0ull) {
                    const int info = h0.info, nfile = info & 0x3fffffff;
                    const bool rst = ((info >> 30) & 1) != 0;
                    const int ncand = (int)((h0.h.y >> 8) & 0xffu);
                    const double v = h0.v, fl = h0.fl;
                    if (WSA_TUNE(16)) pt0 = __builtin_readcyclecounter();
                    if (gl < 4) q_map[1 + gl] = 0u;
                    wsync();
                    // ---- accepted peaks (ref @B25827: `e[l] > v`), compacted per half; their bins into the bit map
                    int n = 0;
                    const int ncmax = groups_max_i32<GW>(act ? ncand : 0);
                    for (int cb = 0; cb < ncmax; cb += GW) {
                        uint4 e4 = c0.e; uint32_t am = c0.amp;
                        const bool hasc = act && cb + gl < ncand;
                        if (cb > 0) { e4 = make_uint4(0u, 0u, 0u, 0u); am = 0u; if (hasc) { const uint32_t c = h0.h.w + (uint32_t)(cb + gl); e4 = p.rec.ent[c]; am = p.rec.amp[c]; } }
                        const bool acc = hasc && (double)am > v;
                        const uint32_t m = group_ballot<GW>(acc, lane);
                        const int pos = n + __popc(m & below);
                        if (acc && pos < GW) {
                            q_pk[pos] = e4.x; q_amp[pos] = am; q_plo[pos] = e4.y; q_phi[pos] = e4.z; q_hi[pos] = e4.w;
                            const uint32_t lb = (e4.x >> 16) & 0x7fu;
                            atomicOr(&q_map[1 + (lb >> 5)], 1u << (lb & 31u));
                        }
                        n += __popc(m);
                    }
                    if (WSA_TUNE(16) && act && n > GW && !g_redo && gl == 0) atomicAdd(&p.shared[10], 1u);      // tuning: spans declined for their peaks ...
                    if (act && n > GW) g_redo = true;                       // more peaks than the group of lanes holds: the one-span kernel takes the span
                    const bool on = act && n >= 1 && n <= GW;
                    if (on) g_accG += (double)(h0.h.y & 0xffu) * 4294967296.0 + (double)h0.h.x;          // g < 2^40, exact
                    wsync();
                    if (__ballot(on) != 0ull) {
                        const bool ispk = on && gl < n;
                        // (reads without a lane test where the index stays inside the group's arrays: what a lane without a peak / a track reads is never used —
                        //  every conditional block costs the wave an exec save, a branch and a restore, and this kernel is bound by its instruction count)
                        const uint32_t pkw = q_pk[gl], pamp = q_amp[gl];
                        const int pk_i = pkw & 0xff, pk_s = (pkw >> 8) & 0xff, pk_l = (pkw >> 16) & 0xff;
                        const uint32_t m0 = q_map[1], m1 = q_map[2], m2 = q_map[3], m3 = q_map[4];
                        const int pc1 = __popc(m0), pc2 = pc1 + __popc(m1), pc3 = pc2 + __popc(m2);
                        WSA_PCY(0); pn_on++;
                        // ---- 1. retired tracks leave the table (stable compaction): every fourth frame, or when the frame's new tracks might not fit
                        const bool compact = on && ((step & 3) == 0 || g_nact + n > ACG);
                        if (__ballot(compact) != 0ull) {
                            int kept = 0;
                            const int na_max = groups_max_i32<GW>(compact ? g_nact : 0);
                            for (int tb = 0; tb < na_max; tb += GW) {
                                const int j = tb + gl;
                                const bool valid = compact && j < g_nact;
                                const int jr = GW == 32 ? j : min(j, ACG - 1);      // (two chunks of 32 are the 64 entries; a third chunk of 16 would reach past 38)
                                const int lf = t_lf[jr], ln = t_len[jr], gi = t_gid[jr]; const uint32_t bn = t_bins[jr], am = t_amp[jr]; const double ve = t_vel[jr], se = t_sumE[jr], sb = t_sumEbin[jr];
                                const bool keep = valid && (nfile - lf) < 4;
                                const uint32_t km = group_ballot<GW>(keep, lane);
                                if (valid && !keep) { Wg.tr_len[gi] = ln; Wg.tr_sumE[gi] = se; Wg.tr_sumEbin[gi] = sb; }   // the summary finalize ranks by
                                wsync();
                                if (keep) {
                                    const int q = kept + __popc(km & below);
                                    t_lf[q] = lf; t_len[q] = ln; t_gid[q] = gi; t_bins[q] = bn; t_amp[q] = am; t_vel[q] = ve; t_sumE[q] = se; t_sumEbin[q] = sb;
                                }
                                kept += __popc(km);
                                wsync();
                            }
                            if (compact) g_nact = kept;
                        }
                        // ---- 2. score every (track, peak) pair inside the track's search window; per peak the best score > 1, the EARLIER
                        //         track on ties (ref: `i>1&&i>d[o]` in track order)
                        WSA_PCY(1);
                        int asg = -1; double best = 0;
                        const int na_max = groups_max_i32<GW>(on ? g_nact : 0);
                        if (na_max > GW) pn_chunk2++;
                        for (int tb = 0; tb < na_max; tb += GW) {
                            const int j = tb + gl;
                            const bool valid = on && j < g_nact;
                            const int jr = GW == 32 ? j : min(j, ACG - 1);
                            const int gap = nfile - t_lf[jr], bin = (int)(t_bins[jr] & 0xffu);
                            if (valid) t_mmask[j] = 0u;
                            const bool live = valid && gap >= 0 && gap < 4;
                            const int win = (int)((0x9643u >> (4 * (gap & 3))) & 0xfu);               // [3, 4, 6, 9][gap], ref @B32325 (gap in 0 .. 3 wherever the value is used)
                            // peaks with bin - win < l < bin + win: the map's bits [lo, bin + win); o_lo = peaks below lo (the peaks are in bin order)
                            const int lo = max(bin - win + 1, 0), width = bin + win - lo;              // width in 3 .. 17
                            const int w0 = lo >> 5, sh = lo & 31;
                            const uint32_t wa = w0 == 0 ? m0 : (w0 == 1 ? m1 : (w0 == 2 ? m2 : m3));
                            const uint32_t wb = w0 == 0 ? m1 : (w0 == 1 ? m2 : (w0 == 2 ? m3 : 0u));
                            const uint32_t wnd = (uint32_t)(((((unsigned long long)wb) << 32) | wa) >> sh) & ((1u << width) - 1u);
                            const int o_lo = (w0 == 0 ? 0 : (w0 == 1 ? pc1 : (w0 == 2 ? pc2 : pc3))) + __popc(wa & ((1u << sh) - 1u));
                            const int cnt = live ? __popc(wnd) : 0;
                            const int incl = (int)group_incl_scan_u32<GW>((uint32_t)cnt);
                            const int off = incl - cnt;
                            const int M = (int)group_last_u32<GW>((uint32_t)incl, lane);
                            const int M_max = groups_max_i32<GW>(M);
                            for (int base = 0; base < M_max; base += GW) {
                                pn_pass++;
                                q_best[gl] = 0ull; q_asg[gl] = 0x7fffffff;
                                for (int c = 0; __ballot(c < cnt) != 0ull; c++) {
                                    const int slot = off + c - base;
                                    if (c < cnt && slot >= 0 && slot < GW) { q_prj[slot] = j; q_pro[slot] = o_lo + c; }
                                }
                                wsync();
                                const bool pv = base + gl < M;
                                // (a lane without a pair scores whatever its list slot holds, clamped into the tables, and keeps the result to itself)
                                const int jj = (int)min((uint32_t)q_prj[gl], (uint32_t)(ACG - 1)), oo = (int)min((uint32_t)q_pro[gl], (uint32_t)(GW - 1));
                                const int tbn = (int)(t_bins[jj] & 0xffu), tg = nfile - t_lf[jj];
                                const int pl = (int)((q_pk[oo] >> 16) & 0xffu);
                                const double sc = match_score(tg, (double)abs(tbn - pl), (double)t_len[jj], (double)tbn, (double)pl,
                                                              (double)t_amp[jj], (double)q_amp[oo], t_vel[jj]);
                                const bool cand = pv && sc > 1;
                                if (cand) atomicMax(&q_best[oo], (unsigned long long)__double_as_longlong(sc));
                                wsync();
                                if (cand && (unsigned long long)__double_as_longlong(sc) == q_best[oo]) atomicMin(&q_asg[oo], jj);
                                wsync();
                                {
                                    const int cj = q_asg[gl];
                                    const double cs = __longlong_as_double((long long)q_best[gl]);
                                    if (ispk && cj != 0x7fffffff && cs > best) { best = cs; asg = cj; }
                                }
                                wsync();
                            }
                        }
                        WSA_PCY(2);
                        // ---- 3. hand each matched track the set of its peaks
                        if (ispk && asg >= 0) atomicOr(&t_mmask[asg], 1u << gl);
                        wsync();
                        const int p_begin = g_npt;
                        // ---- 4. matched tracks update themselves (lane = track)
                        for (int tb = 0; tb < na_max; tb += GW) {
                            const int j = tb + gl;
                            const uint32_t mm = (on && j < g_nact) ? t_mmask[j] : 0u;
                            // the first assigned peak is where st / en / pb start from (a track without one reads peak 0: not used)
                            const int first = mm ? __ffs((int)mm) - 1 : 0;
                            const uint32_t w0_ = q_pk[first], a0 = q_amp[first], hb = q_hi[first];      // a0: amplitude of the FIRST assigned peak (quirk 3)
                            const uint32_t plo0 = q_plo[first], phi0 = q_phi[first];
                            const bool upd = mm != 0u && (double)a0 > fl;
                            int pb = (w0_ >> 16) & 0xff, st = w0_ & 0xff, en = (w0_ >> 8) & 0xff;
                            // P[i-1] and P[s] as 40-bit integers {low word, high byte}: the band sum is one 64-bit subtraction, converted once
                            uint32_t lo_l = plo0, lo_h = hb & 0xffu, hi_l = phi0, hi_h = (hb >> 8) & 0xffu;
                            {
                                uint32_t pb_amp = a0;
                                uint32_t rest = upd ? mm & (mm - 1u) : 0u;
                                while (rest) {
                                    const int o = __ffs((int)rest) - 1; rest &= rest - 1u;
                                    const uint32_t w = q_pk[o];
                                    const int oi = w & 0xff, os = (w >> 8) & 0xff, ol = (w >> 16) & 0xff;
                                    const uint32_t hbo = q_hi[o], ao = q_amp[o], plo_o = q_plo[o], phi_o = q_phi[o];
                                    if (os > en) { en = os; hi_l = phi_o; hi_h = (hbo >> 8) & 0xffu; }
                                    if (oi < st) { st = oi; lo_l = plo_o; lo_h = hbo & 0xffu; }
                                    if (ao > pb_amp) { pb = ol; pb_amp = ao; }
                                }
                            }
                            // sum e[st..en] = P[en] - P[st-1], exact (below 2^40)
                            const unsigned long long be_i = (((unsigned long long)hi_h << 32) | hi_l) - (((unsigned long long)lo_h << 32) | lo_l);
                            const double be = upd ? (double)(uint32_t)(be_i >> 32) * 4294967296.0 + (double)(uint32_t)be_i : 0.0;
                            const uint32_t um = group_ballot<GW>(upd, lane);
                            const int nu = __popc(um);
                            // (the split tracker's span regions hold 64 points and tracks per frame of the span and a frame adds at most GW <= 32 of either: they cannot overflow)
                            if (!SPLIT && g_npt + nu > g_tcap) g_ovf = true;
                            else if (upd) {
                                const int q = g_npt + __popc(um & below);
                                const int hlen = t_len[j];
                                const uint32_t bn = t_bins[j];
                                const int P1 = bn & 0xff, P2 = (bn >> 8) & 0xff, P3 = (bn >> 16) & 0xff;
                                // velocity (ref @B36624): all three forms evaluated, one selected (three nested branches cost more than the two extra conversions)
                                // x / 3, correctly rounded: q = x * (1/3), r = x - 3q (exact), q + r * (1/3); x / 2 = x * 0.5 exactly
                                const double xv = (double)((pb - P1) + (P2 - P1) + (P3 - P2)), third = 1.0 / 3.0;
                                const double q0 = xv * third;
                                const double v3 = __builtin_fma(__builtin_fma(-3.0, q0, xv), third, q0);
                                const double v2 = (double)((pb - P1) + (P2 - P1)) * 0.5, v1 = (double)(pb - P1);
                                const double vel = hlen >= 3 ? v3 : (hlen == 2 ? v2 : (hlen == 1 ? v1 : t_vel[j]));
                                const double se = t_sumE[j] + be, sb = t_sumEbin[j] + be * pb;
                                t_vel[j] = vel; t_bins[j] = (uint32_t)pb | ((uint32_t)P1 << 8) | ((uint32_t)P2 << 16);
                                t_amp[j] = a0; t_lf[j] = nfile; t_len[j] = hlen + 1; t_sumE[j] = se; t_sumEbin[j] = sb;
                                Wg.pt[q] = make_int4(t_gid[j], pb | ((en - st + 1) << 8) | (min(nfile, 0x7fff) << 17), __double2loint(be), __double2hiint(be));
                            }
                            if (upd) g_accL += be;                   // integer-valued: exact in any order
                            if (!g_ovf) g_npt += nu;
                        }
                        WSA_PCY(3);
                        // ---- 5. unassigned peaks above the floor open new tracks, in peak order (lane = peak)
                        const bool mk = ispk && asg == -1 && (double)pamp > fl;
                        const uint32_t nm = group_ballot<GW>(mk, lane);
                        const int nnew = __popc(nm);
                        // (WSA_DBG bits 1024 / 16384, tests: the table pretends to hold 12 tracks, so that the redo list is used on ordinary input)
                        if (WSA_TUNE(16) && on && g_nact + nnew > ACG && !g_redo && gl == 0) atomicAdd(&p.shared[11], 1u);      // ... and for their live tracks
                        if (on && g_nact + nnew > ((p.dbg & (1024 | 16384)) ? 12 : ACG)) g_redo = true;           // more live tracks than the half's table holds
                        if (!SPLIT && on && (g_ntr + nnew > g_tcap || g_npt + nnew > g_tcap)) g_ovf = true;
                        const bool grow = on && !g_ovf && !g_redo;
                        if (grow && mk) {
                            const int r = __popc(nm & below);
                            const int t = g_ntr + r, q = g_npt + r, j = g_nact + r;
                            const uint32_t hb = q_hi[gl];
                            const double be = dbl40(q_phi[gl], hb >> 8) - dbl40(q_plo[gl], hb);
                            t_lf[j] = nfile; t_len[j] = 1; t_gid[j] = t; t_bins[j] = (uint32_t)pk_l; t_amp[j] = pamp;
                            t_vel[j] = 0; t_sumE[j] = be; t_sumEbin[j] = be * pk_l;
                            Wg.pt[q] = make_int4(t, pk_l | ((pk_s - pk_i + 1) << 8) | (min(nfile, 0x7fff) << 17), __double2loint(be), __double2hiint(be));
                        }
                        if (grow) { g_ntr += nnew; g_npt += nnew; g_nact += nnew; }
                        // file this frame's point range under its (possibly stale) index
                        if (on) {
                            if (rst) { g_stale_d = nfile; g_stale_p1 = g_npt; }
                            else if (gl == 0 && nfile < g_fcap + 2) { Wg.d_p0[nfile] = p_begin; Wg.d_p1[nfile] = g_npt; Wg.d_gen[nfile] = SPLIT ? 1 : gen; }
                        }
                        wsync();
                        WSA_PCY(4);
                    }
                }
                h0 = h1; h1 = h2; c0 = c1;
            }
#undef WSA_PCY
            const unsigned long long ptk1 = WSA_TUNE(16) ? __builtin_readcyclecounter() : 0ull;
            // ---- both spans are through: the live tracks hand their summaries over, then one finalize after the other with the whole wave
            for (int j = gl; j < g_nact; j += GW) { const int gi = t_gid[j]; Wg.tr_len[gi] = t_len[j]; Wg.tr_sumE[gi] = t_sumE[j]; Wg.tr_sumEbin[gi] = t_sumEbin[j]; }
            wsync();
            if constexpr (SPLIT == 1) {
                // ---- split finalize: a header per span for the finalize kernel (sum E of the half: integer-valued terms, exact in any order)
                double cg[NGR];
#pragma unroll
                for (int q = 0; q < NGR; q++) cg[q] = g == q ? g_accL : 0.0;
                wave_sums_f64(cg);
                double c_mine = cg[0];
#pragma unroll
                for (int q = 1; q < NGR; q++) c_mine = g == q ? cg[q] : c_mine;
                if (has && gl == 0) {
                    if (g_redo) { const uint32_t k = atomicAdd(p.redo_count, 1u); p.redo[k] = make_uint2(g_clip, g_seg); }
                    double* hd = p.span_hdr + ((uint64_t)g_clip * p.seg_cap + g_seg) * 8;
                    hd[0] = g_ntr; hd[1] = g_npt; hd[2] = g_stale_d; hd[3] = g_stale_p1; hd[4] = g_accG; hd[5] = c_mine;
                    hd[6] = g_redo ? 0.0 : (g_ovf ? 2.0 : 1.0);
                }
            } else
            for (int gs = 0; gs < (GW == 32 ? 2 : 0); gs++) {
                const int src = gs * 32;
                if (!read_lane_i32((int)has, src)) continue;
                clip = (uint32_t)read_lane_i32((int)g_clip, src); k_seg = (uint32_t)read_lane_i32((int)g_seg, src); my_seg = (int)k_seg;
                if (read_lane_i32((int)g_redo, src)) {
                    if (lane == 0) { const uint32_t k = atomicAdd(p.redo_count, 1u); p.redo[k] = make_uint2(clip, k_seg); }
                    continue;
                }
                sg = p.seg_i + ((uint64_t)clip * p.seg_cap + my_seg) * 8;
                start = sg[SEG_START]; len = sg[SEG_LEN]; c_ci = sg[SEG_CCI];
                f_begin = (uint32_t)sg[SEG_FBEGIN]; f_end = (uint32_t)sg[SEG_FEND];
                ctx_max = p.seg_d[((uint64_t)clip * p.seg_cap + my_seg) * 2];
                floor_ = p.seg_d[((uint64_t)clip * p.seg_cap + my_seg) * 2 + 1];
                foff = p.frame_off[clip];
                n_tr = read_lane_i32(g_ntr, src); n_pt = read_lane_i32(g_npt, src); n_act = 0;
                stale_d = read_lane_i32(g_stale_d, src); stale_p1 = read_lane_i32(g_stale_p1, src);
                accG = __hiloint2double(read_lane_i32(__double2hiint(g_accG), src), read_lane_i32(__double2loint(g_accG), src));
                accL = g == gs ? g_accL : 0.0;
                overflow = read_lane_i32((int)g_ovf, src) != 0; act_overflow = false;
                W = carve_ws(p.ws + ((uint64_t)blockIdx.x * 2 + (uint32_t)gs) * p.ws_stride, p.tcap, p.pcap, p.fcap, 0, nullptr);
                if (!overflow) finish_span();
                if (overflow && lane == 0) atomicOr(&p.shared[1], 1u);
                wsync();
            }
            if (WSA_TUNE(16) && lane == 0 && p.trace) {      // tuning: per-pair cycle counts into the trace buffer
                double* tr = p.trace + (uint64_t)atomicAdd(&p.shared[0], 1u) * 12;
                tr[0] = (double)(ptk1 - ptk0); tr[1] = (double)(__builtin_readcyclecounter() - ptk1); tr[2] = nsteps; tr[3] = pn_on; tr[4] = pn_chunk2; tr[5] = pn_pass; tr[6] = blockIdx.x;
                tr[7] = (double)pcy[0]; tr[8] = (double)pcy[1]; tr[9] = (double)pcy[2]; tr[10] = (double)pcy[3]; tr[11] = (double)pcy[4];
            }
            gen++;
            continue;
        } else
        if constexpr (ST) {
            // ---- incremental streaming: this wave owns stream `clip`.  Its tracker state (counters, accumulators, the active
            //      table; the track / point arrays live in the stream's work space anyway) comes from HBM, the frames of this step
            //      are accumulated one by one, a segment the gate closed in this step is finalized right behind its last frame,
            //      and the state goes back.  Every reset_segment of the reference clears the tracker: gate.hip notes for each
            //      accumulate call the span it belongs to (fr_span) and a change of span clears the state here.
            int32_t* stt = p.st_state + (uint64_t)clip * TR_STATE_WORDS;
            double* std_ = reinterpret_cast<double*>(stt + 8);
            n_tr = stt[0]; n_pt = stt[1]; n_act = stt[2]; stale_d = stt[3]; stale_p1 = stt[4]; int my_span = stt[5]; gen = stt[6];
            accG = std_[0] + std_[1]; accL = lane == 0 ? std_[1] : 0.0;
            char* ab = p.st_act + (uint64_t)clip * TR_ACT_BYTES;
            double* const g_vel = reinterpret_cast<double*>(ab); double* const g_sumE = g_vel + AC; double* const g_sumEbin = g_sumE + AC;
            int32_t* const g_lf = reinterpret_cast<int32_t*>(g_sumEbin + AC); int32_t* const g_len = g_lf + AC; int32_t* const g_gid = g_len + AC;
            uint32_t* const g_bins = reinterpret_cast<uint32_t*>(g_gid + AC); uint32_t* const g_amp = g_bins + AC;
            for (int j = lane; j < n_act; j += 64) {
                a_vel[j] = g_vel[j]; a_sumE[j] = g_sumE[j]; a_sumEbin[j] = g_sumEbin[j];
                a_last_frame[j] = g_lf[j]; a_len[j] = g_len[j]; a_gid[j] = g_gid[j]; a_bins[j] = g_bins[j]; a_amp[j] = g_amp[j];
            }
            wsync();
            auto clear_state = [&](int span) __attribute__((always_inline)) { n_tr = n_pt = n_act = 0; stale_d = -1; stale_p1 = 0; accG = accL = 0; gen++; my_span = span; };
            const uint32_t nfr = p.n_frames_step[clip];
            const uint32_t fbase = (uint32_t)p.gate_state[(uint64_t)clip * GATE_STATE] - nfr;       // the gate has counted this step's frames already
            const int nseg = (int)p.seg_count[clip];
            f_begin = fbase; f_end = fbase + nfr;                 // the record fetchers clamp to [f_begin, f_end)
            int ks = 0;
            auto close_segments = [&](uint32_t f_next, bool all) __attribute__((always_inline)) {
                while (ks < nseg) {
                    int32_t* sgk = p.seg_i + ((uint64_t)clip * p.seg_cap + ks) * 8;
                    if (!all && (uint32_t)sgk[SEG_FEND] != f_next) break;
                    if (sgk[SEG_FBEGIN] != my_span) clear_state(sgk[SEG_FBEGIN]);      // no frame of the span reached accumulate_fm
                    my_seg = ks; sg = sgk; start = sg[SEG_START]; len = sg[SEG_LEN]; c_ci = sg[SEG_CCI];
                    ctx_max = p.seg_d[((uint64_t)clip * p.seg_cap + ks) * 2]; floor_ = p.seg_d[((uint64_t)clip * p.seg_cap + ks) * 2 + 1];
                    finish_span();
                    clear_state(-2);                                                    // every finalize is followed by a reset_segment
                    ks++;
                }
            };
            for (uint32_t f = fbase; f < fbase + nfr; f++) {
                Hdr h; load_hdr(f, h);
                Pre cur; load_ent(f, h, cur);
                if (cur.info >= 0) {
                    const int sp = p.fr_span[foff + (f & p.ring_mask)];
                    if (sp != my_span) clear_state(sp);
                    accumulate(cur, [] {});
                }
                close_segments(f + 1, false);
            }
            close_segments(0, true);
            double accS, accC; acc_totals(accS, accC);
            if (lane == 0) { stt[0] = n_tr; stt[1] = n_pt; stt[2] = n_act; stt[3] = stale_d; stt[4] = stale_p1; stt[5] = my_span; stt[6] = gen; std_[0] = accS; std_[1] = accC; }
            for (int j = lane; j < n_act; j += 64) {
                g_vel[j] = a_vel[j]; g_sumE[j] = a_sumE[j]; g_sumEbin[j] = a_sumEbin[j];
                g_lf[j] = a_last_frame[j]; g_len[j] = a_len[j]; g_gid[j] = a_gid[j]; g_bins[j] = a_bins[j]; g_amp[j] = a_amp[j];
            }
        } else {
            // What gate.hip left per frame (info, v, fl) and the record header are fetched for 64 frames at a time, lane j = frame
            // blk + j, one block ahead; a frame gets its values by v_readlane.  The candidate entries (lane = candidate) of frame
            // f + PFD are requested while frame f is processed — as soon as f's own entry has been copied out of its registers —
            // so that neither fetch is waited for (before: groups of 4 frames paid one memory round trip each, ~900 cycles a frame).
            constexpr int PFD = 4;
            struct Blk { int info; double v, fl; uint4 h; };
            struct Ent { uint32_t pk, amp, plo, phi, hi; };
            auto load_blk = [&](uint32_t fb, Blk& q) __attribute__((always_inline)) {
                const uint32_t f = fb + (uint32_t)lane, fi = foff + min(f, f_end - 1);
                q.info = p.fr_info[fi]; q.v = p.fr_v[fi]; q.fl = p.fr_fl[fi]; q.h = p.rec.hdr[fi];
                if (f >= f_end) q.info = -1;
            };
            Blk bc, bn;
            auto rl = [](int x, int j) __attribute__((always_inline)) { return __builtin_amdgcn_readlane(x, j); };
            auto rl_d = [&](double x, int j) __attribute__((always_inline)) { return __hiloint2double(rl(__double2hiint(x), j), rl(__double2loint(x), j)); };
            // entry of the frame at position j of the current block (j >= 64: of the next block)
            auto request = [&](int j, Ent& e) __attribute__((always_inline)) {
                const int info_ = j < 64 ? rl(bc.info, j & 63) : rl(bn.info, j & 63);
                const int hy = j < 64 ? rl((int)bc.h.y, j & 63) : rl((int)bn.h.y, j & 63);
                const uint32_t cb = (uint32_t)(j < 64 ? rl((int)bc.h.w, j & 63) : rl((int)bn.h.w, j & 63));
                e.pk = e.amp = e.plo = e.phi = e.hi = 0u;
                if (info_ >= 0 && lane < ((hy >> 8) & 0xff) && !(WSA_TUNE(32))) {           // only frames accumulate_fm sees, only the entries they hold
                    const uint32_t c = cb + (uint32_t)lane;
                    const uint4 e4 = p.rec.ent[c];
                    e.amp = p.rec.amp[c]; e.pk = e4.x; e.plo = e4.y; e.phi = e4.z; e.hi = e4.w;
                }
            };
            load_blk(f_begin, bc);
            bn = bc;
            if (f_begin + 64 < f_end) load_blk(f_begin + 64, bn);
            Ent ring[PFD];
    #pragma unroll
            for (int k = 0; k < PFD; k++) request(k, ring[k]);
            for (uint32_t blk = f_begin; blk < f_end; blk += 64) {
              const int nb = (int)min(64u, f_end - blk);
              for (int j0 = 0; j0 < nb; j0 += PFD) {
    #pragma unroll
                for (int k = 0; k < PFD; k++) {
                  const int j = j0 + k;
                  if (j >= nb) break;
                  Pre cur;
                  const int hy = rl((int)bc.h.y, j);
                  cur.info = rl(bc.info, j); cur.v = rl_d(bc.v, j); cur.fl = rl_d(bc.fl, j);
                  cur.g = (double)(hy & 0xff) * 4294967296.0 + (double)(uint32_t)rl((int)bc.h.x, j);      // exact: g < 2^40
                  cur.n = (hy >> 8) & 0xff;
                  cur.pk = ring[k].pk; cur.amp = ring[k].amp; cur.plo = ring[k].plo; cur.phi = ring[k].phi; cur.hi = ring[k].hi;
                  accumulate(cur, [&]() __attribute__((always_inline)) { request(j + PFD, ring[k]); });
                  if (p.trace && !(WSA_TUNE(16))) { double accS, accC; acc_totals(accS, accC); if (lane == 0) { double* tr = p.trace + ((uint64_t)foff + blk + (uint32_t)j) * 12; tr[10] = accS; tr[11] = accC; } }
                }
              }
              bc = bn;
              if (blk + 128 < f_end && !(WSA_TUNE(64))) load_blk(blk + 128, bn);
            }
            tk1 = (WSA_TUNE(16)) ? __builtin_readcyclecounter() : 0ull;
            finish_span();
        }
        if ((WSA_TUNE(16)) && lane == 0 && p.trace) {      // tuning: per-span cycle counts into the trace buffer
            double* tr = p.trace + (uint64_t)atomicAdd(&p.shared[0], 1u) * 12;      // shared[0] is otherwise unused
            tr[0] = (double)(tk1 - tk0); tr[1] = (double)(__builtin_readcyclecounter() - tk1); tr[2] = len; tr[3] = (double)(f_end - f_begin); tr[4] = n_tr; tr[5] = n_pt; tr[6] = blockIdx.x;
            if (WSA_TUNE(512)) { tr[7] = (double)acp[0]; tr[8] = (double)acp[1]; tr[9] = (double)acp[2]; tr[10] = (double)acp[3]; tr[11] = (double)acp[4]; }
            else { tr[7] = (double)(ph[0] - tk1); tr[8] = (double)(ph[1] - ph[0]); tr[9] = (double)(ph[2] - ph[1]); tr[10] = (double)(ph[3] - ph[2]); }
        }
        // bit0: an arena overflowed (results invalid); bit1: it was (only) the LDS active-track table of
        // the fast variant — the host then reruns the back end with the full-size variant
        if (overflow && lane == 0) atomicOr(&p.shared[1], act_overflow && AC < AC_MAX ? 2u : 1u);
        wsync();
    }
}

// The fast variant is held to 128 VGPRs (4 waves per SIMD; ~80 registers spill to scratch) and its active-track table to
// AC_FAST = 140 entries (10 096 B of LDS: 8 of the CU's 1 280-byte allocation units, 16 waves per CU).  The kernel is bound by
// instruction issue with most lanes idle, so waves in flight beat spill traffic: 3 per SIMD (168 VGPRs, 192 entries) 0.409 ms,
// 4 per SIMD 0.381 ms, 5 per SIMD (96 VGPRs, 100 entries, 241 spills) 0.643 ms on the 1024-clip batch (profiles/r02_notes.md).
// Mind the allocation unit: 16 bytes of LDS more than 12 800 cost the 3-per-SIMD variant a twelfth wave per CU and 50 % of its speed.
// The full-table variant is LDS-limited to 8 waves per CU and keeps its registers.
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(4, 4))) void tracker_kernel_fast(TrParams p) { tracker_body<AC_FAST, false, false>(p); }
// two spans per wave (half-waves in lock step): 3 waves per SIMD (168 VGPRs) are all the pairs of a 1024-clip batch need
// (held to 128 registers / 4 waves it spills 78 of them: 1.02 -> 1.12 ms per batch alone, 0.73 -> 0.75 ms pipelined)
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(3, 4))) void tracker_kernel_pair(TrParams p) {
    __builtin_amdgcn_s_setprio(3);      // the dependent chains of this kernel go first, the front end of the next batch fills what they leave (0.686 -> 0.680 ms per pipelined step)
    tracker_body<AC_FAST, false, false, true>(p);
}
// split finalize: the paired accumulate on its own (its waves end with the tracking: fewer registers, shorter lives) and the finalize of every span, one span per
// wave and turn, out of the span regions (TrParams::pool); rows bit for bit those of the kernel above (tests/test_gpu_parity.py)
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(4, 4))) void tracker_kernel_pair_acc(TrParams p) {
    __builtin_amdgcn_s_setprio(3);
    tracker_body<AC_FAST, false, false, true, 1>(p);
}
// four spans per wave: the quarters of a wave (its four DPP rows) track four neighbours of the length-sorted span list in lock step; a frame brings a
// span at most 16 accepted peaks here and its table holds 38 live tracks (4 % of the spans need more: redo list).  Every instruction serves four frames
// instead of two; the loops over the tracks take two chunks of 16 where the halves took one of 32
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(4, 4))) void tracker_kernel_quad_acc(TrParams p) {
    __builtin_amdgcn_s_setprio(3);
    tracker_body<AC_FAST, false, false, true, 1, 16>(p);
}
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(4, 4))) void tracker_kernel_finalize(TrParams p) {
    __builtin_amdgcn_s_setprio(3);
    tracker_body<AC_FAST, false, false, false, 2>(p);
}
__global__ __launch_bounds__(64) void tracker_kernel_full(TrParams p) { tracker_body<AC_MAX, false, false>(p); }
__global__ __launch_bounds__(64) void tracker_kernel_raw(TrParams p) { tracker_body<AC_MAX, true, false>(p); }
__global__ __launch_bounds__(64) void tracker_kernel_stream(TrParams p) { tracker_body<AC_MAX, false, true>(p); }
__global__ __launch_bounds__(64) void tracker_kernel_stream_raw(TrParams p) { tracker_body<AC_MAX, true, true>(p); }      // level 3 for streams

// ---- span order: all (clip, segment) pairs the gate kernel produced, sorted by span length (frames between the resets that
// bound the span: what the tracker's time goes with), longest first — a counting sort whose counting the gate kernel has done already
// (GateParams::span_hist / span_key: every finalized segment took a rank inside its length bucket).  Here: an exclusive scan over the
// buckets (one workgroup) and the scatter (one thread per clip).  The order inside a bucket is whatever the atomics made it — the
// tracker's results do not depend on the order the spans are processed in, rows are put back in callback order by K3.
// counters[1] = number of spans.
// One kernel: every workgroup forms the exclusive scan of the histogram itself (2 048 counts: 8 KB out of L2, a few microseconds) and keeps the
// offsets in LDS, then scatters the spans of its clips — a scan kernel in front was one more dependent launch on the run's chain (~15 us of
// kernel + the launch gap, for a 1024-clip batch) to save each of a handful of workgroups that scan.  The histogram itself stays as counted.
constexpr int ORDER_T = 256;
__global__ __launch_bounds__(ORDER_T) void span_order_kernel(const uint32_t* hist, const uint2* key, const uint32_t* seg_count, uint32_t n_clips, int seg_cap, uint2* order, uint32_t* counters) {
    __shared__ uint32_t offs[SPAN_BUCKETS], part[ORDER_T];
    const int tid = threadIdx.x;
    constexpr int PER = SPAN_BUCKETS / ORDER_T;
    uint32_t mine[PER], sum = 0;
#pragma unroll
    for (int q = 0; q < PER; q++) { mine[q] = hist[tid * PER + q]; sum += mine[q]; }
    part[tid] = sum;
    __syncthreads();
    for (int d = 1; d < ORDER_T; d <<= 1) {
        const uint32_t add = tid >= d ? part[tid - d] : 0u;
        __syncthreads();
        part[tid] += add;
        __syncthreads();
    }
    uint32_t run = part[tid] - sum;
#pragma unroll
    for (int q = 0; q < PER; q++) { offs[tid * PER + q] = run; run += mine[q]; }
    if (blockIdx.x == 0 && tid == ORDER_T - 1) counters[1] = part[tid];
    __syncthreads();
    const uint32_t clip = blockIdx.x * ORDER_T + tid;
    if (clip >= n_clips) return;
    const uint32_t ns = seg_count[clip];
    for (uint32_t k = 0; k < ns; k++) {
        const uint2 e = key[(uint64_t)clip * seg_cap + k];
        order[offs[e.x] + e.y] = make_uint2(clip, k);
    }
}

void launch_span_order(const TrParams& p, uint32_t* span_hist, const uint2* span_key, uint2* order, uint32_t* counters, hipStream_t s) {
    if (p.n_clips == 0) return;
    hipLaunchKernelGGL(span_order_kernel, dim3((p.n_clips + ORDER_T - 1) / ORDER_T), dim3(ORDER_T), 0, s, span_hist, span_key, p.seg_count, p.n_clips, p.seg_cap, order, counters);
}

void launch_tracker_stream(const TrParams& p, uint32_t n_streams, hipStream_t s) {
    if (n_streams == 0) return;
    static_assert(AC_MAX == TR_ACT_MAX, "the streams' saved active table is sized for the full variant");
    if (p.level == 3) hipLaunchKernelGGL(tracker_kernel_stream_raw, dim3(n_streams), dim3(64), 0, s, p);
    else hipLaunchKernelGGL(tracker_kernel_stream, dim3(n_streams), dim3(64), 0, s, p);
}

void launch_tracker(const TrParams& p, int n_waves, bool full_table, bool pair, hipStream_t s) {
    if (n_waves <= 0) return;
    if (p.level == 3) hipLaunchKernelGGL(tracker_kernel_raw, dim3(n_waves), dim3(64), 0, s, p);
    else if (full_table) hipLaunchKernelGGL(tracker_kernel_full, dim3(n_waves), dim3(64), 0, s, p);
    else if (pair && p.order && p.redo && (!p.trace || (p.dbg & 16))) {
        // two spans per wave; what the paired variant declines goes through the one-span kernel right behind it (usually nothing: its waves find an empty list)
        if (p.pool && p.span_hdr) {
            if (p.quad) hipLaunchKernelGGL(tracker_kernel_quad_acc, dim3(p.quad_waves > 0 ? p.quad_waves : n_waves), dim3(64), 0, s, p);
            else hipLaunchKernelGGL(tracker_kernel_pair_acc, dim3(n_waves), dim3(64), 0, s, p);
            hipLaunchKernelGGL(tracker_kernel_finalize, dim3(p.fin_waves > 0 ? p.fin_waves : 2 * n_waves), dim3(64), 0, s, p);
        }
        else hipLaunchKernelGGL(tracker_kernel_pair, dim3(n_waves), dim3(64), 0, s, p);
        TrParams r = p; r.order = p.redo; r.order_cnt = 2; r.pool = nullptr; r.span_hdr = nullptr;
        // (its waves find an empty list almost always — the paired kernel declines 1 span in 10^4, the quad kernel 1 in 10^2 — and every wave costs its set-up)
        const int redo_waves = p.quad ? 1024 : 128;
        hipLaunchKernelGGL(tracker_kernel_fast, dim3(n_waves < redo_waves ? n_waves : redo_waves), dim3(64), 0, s, r);
    }
    else hipLaunchKernelGGL(tracker_kernel_fast, dim3(n_waves), dim3(64), 0, s, p);
}

// ---- K3 compaction: segment table + row pool -> dense tables in (clip, si, syllable) order, the order
// in which the reference's dispatcher P() (ref @B28869) would have invoked the callback.
// per-clip row counts (a thread per clip walks its segment table: chains of dependent loads, so as many clips at once as the chip takes),
// then one workgroup turns the per-clip row / segment counts into offsets
__global__ void compact_count_kernel(CompactParams p) {
    const uint32_t c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= p.n_clips) return;
    const uint32_t ns = p.seg_count[c];
    const int32_t* sg = p.seg_i + (uint64_t)c * p.seg_cap * 8;
    uint32_t r = 0;
    for (uint32_t k = 0; k < ns; k++) if (sg[8 * k + SEG_FLAG] >= 0) r += (uint32_t)sg[8 * k + SEG_NROWS];
    p.clip_row_off[c] = r;                // count for now; compact_scan_kernel turns it into an offset
}
constexpr int CSCAN_T = 1024;
template <bool COUNT>          // COUNT: small launches (streams, up to two clips per thread) count their rows here and save a kernel
__global__ __launch_bounds__(CSCAN_T) void compact_scan_kernel(CompactParams p) {
    // single block: exclusive scan of the per-clip row / segment counts (a thread owns a run of consecutive clips)
    __shared__ uint32_t s_rows[CSCAN_T], s_segs[CSCAN_T];
    const int tid = threadIdx.x;
    const uint32_t per = (p.n_clips + CSCAN_T - 1) / CSCAN_T;
    const uint32_t c0 = min(p.n_clips, tid * per), c1 = min(p.n_clips, c0 + per);
    uint32_t rs = 0, ss = 0;
    for (uint32_t c = c0; c < c1; c++) {
        if (COUNT) {
            const uint32_t ns = p.seg_count[c];
            const int32_t* sg = p.seg_i + (uint64_t)c * p.seg_cap * 8;
            uint32_t r = 0;
            for (uint32_t k = 0; k < ns; k++) if (sg[8 * k + SEG_FLAG] >= 0) r += (uint32_t)sg[8 * k + SEG_NROWS];
            p.clip_row_off[c] = r;
        }
        ss += p.seg_count[c]; rs += p.clip_row_off[c];
    }
    s_rows[tid] = rs; s_segs[tid] = ss;
    __syncthreads();
    for (int d = 1; d < CSCAN_T; d <<= 1) {          // inclusive scan of both columns
        const uint32_t ar_ = tid >= d ? s_rows[tid - d] : 0u, as_ = tid >= d ? s_segs[tid - d] : 0u;
        __syncthreads();
        s_rows[tid] += ar_; s_segs[tid] += as_;
        __syncthreads();
    }
    if (tid == CSCAN_T - 1) {
        p.totals[0] = s_rows[tid]; p.totals[1] = s_segs[tid];
        p.clip_row_off[p.n_clips] = s_rows[tid]; p.clip_seg_off[p.n_clips] = s_segs[tid];
    }
    uint32_t ar = s_rows[tid] - rs, as = s_segs[tid] - ss;
    for (uint32_t c = c0; c < c1; c++) {
        const uint32_t r = p.clip_row_off[c];
        p.clip_row_off[c] = ar; p.clip_seg_off[c] = as; as += p.seg_count[c]; ar += r;
    }
}

// FUSED (batches of up to a few thousand clips): no scan kernel in front — the wave of clip c sums the row / segment counts of the clips before it
// itself (the per-clip row count is the clip's row counter, which the tracker bumped once per result), writes its two offsets, and the wave of
// clip 0 also forms the totals and hands the run's result counters to the host (p.host: mapped pinned words).  Three dependent launches less at
// the end of every run: with several batches in flight a stream's run is as long as the chain of its kernels, and these three were ~90 us of it
// in which the stream kept next to nothing of the GPU busy.
template <bool FUSED>
__global__ __launch_bounds__(64) void compact_gather_kernel(CompactParams p) {
    const uint32_t clip = blockIdx.x;
    const int lane = threadIdx.x;
    const uint32_t nseg = p.seg_count[clip];
    if (FUSED) {
        uint32_t rs = 0, ss = 0;
        for (uint32_t i = lane; i < clip; i += 64) { rs += p.clip_rows[i]; ss += p.seg_count[i]; }
        rs = wave_sum_u32(rs); ss = wave_sum_u32(ss);
        if (lane == 0) { p.clip_row_off[clip] = rs; p.clip_seg_off[clip] = ss; }
        if (clip == 0) {
            uint32_t rt = 0, st = 0;
            for (uint32_t i = lane; i < p.n_clips; i += 64) { rt += p.clip_rows[i]; st += p.seg_count[i]; }
            rt = wave_sum_u32(rt); st = wave_sum_u32(st);
            if (lane == 0) {
                p.totals[0] = rt; p.totals[1] = st; p.clip_row_off[p.n_clips] = rt; p.clip_seg_off[p.n_clips] = st;
                if (p.host) { p.host[0] = rt; p.host[1] = st; p.host[2] = p.flags[0]; p.host[3] = p.totals[3]; __threadfence_system(); }
            }
            if (p.clr_counters) {
                // every kernel that looks at the run's counters, totals or span histogram is through (stream order; the other waves of this one read none
                // of them): cleared here, the next run of the batch needs no clear kernel in front — one dependent launch less on its chain
                wsync();
                if (lane < 16) p.clr_counters[lane] = 0u;
                if (lane < 4) p.totals[lane] = 0u;
                if (p.clr_hist) for (int b = lane; b < SPAN_BUCKETS; b += 64) p.clr_hist[b] = 0u;
            }
        }
        wsync();
    }
    const uint32_t so = p.clip_seg_off[clip];
    const int32_t* sg = p.seg_i + (uint64_t)clip * p.seg_cap * 8;
    for (uint32_t i = lane; i < nseg; i += 64) {
        int32_t* o = p.seg_out + (uint64_t)(so + i) * 4;
        o[0] = (int32_t)clip; o[1] = sg[8 * i + SEG_START]; o[2] = sg[8 * i + SEG_LEN]; o[3] = sg[8 * i + SEG_FLAG];
    }
    // results in segment order; si = index among the segments that produced a result entry
    uint32_t ro = p.clip_row_off[clip];
    int si = 0;
    // streaming: segments / results of earlier steps (the table holds this step's segments only)
    int32_t* cy = p.carry ? p.carry + (uint64_t)clip * CARRY_WORDS : nullptr;
    const int seg_before = cy ? cy[0] : 0, res_before = cy ? cy[1] : 0;
    bool lost = false;
    for (uint32_t k = 0; k < nseg; k++) {
        const int flag = sg[8 * k + SEG_FLAG];
        if (flag < 0) continue;                       // straighten threw: segments_ci entry without a result
        const int nr = sg[8 * k + SEG_NROWS];
        const uint32_t r0 = (uint32_t)sg[8 * k + SEG_ROW0];
        // the dispatcher indexes segments_ci with the RESULT index (ref @B29138 / @B29622): after a
        // dropped segment the timestamps come from the wrong entry — reproduced, not repaired
        const int gsi = res_before + si;              // result index since the launch
        int32_t ts, tl;
        if (gsi >= seg_before) { ts = sg[8 * (gsi - seg_before) + SEG_START]; tl = sg[8 * (gsi - seg_before) + SEG_LEN]; }
        else {                                        // an entry of an earlier step
            if (seg_before - gsi > CARRY_HIST) lost = true;
            ts = cy[2 + 2 * (gsi % CARRY_HIST)]; tl = cy[3 + 2 * (gsi % CARRY_HIST)];
        }
        for (int i = lane; i < nr; i += 64) {
            const int32_t* m = p.row_meta_in + (uint64_t)(r0 + i) * 8;
            int32_t* o = p.row_meta_out + (uint64_t)(ro + i) * 8;
            o[0] = m[0]; o[1] = gsi; o[4] = seg_before + m[4]; o[5] = m[5]; o[6] = m[6]; o[7] = m[7];
            if (p.level == 10 || p.level == 13) { o[2] = ts + m[2]; o[3] = m[3]; }     // syllable row (ref @B31114)
            else { o[2] = ts; o[3] = tl; }                                            // segment row (ref @B31504)
        }
        for (int i = lane; i < nr * WSA_NFEAT; i += 64) p.row_feat_out[(uint64_t)ro * WSA_NFEAT + i] = p.row_feat_in[(uint64_t)r0 * WSA_NFEAT + i];
        ro += (uint32_t)nr;
        si++;
    }
    if (cy) {
        wsync();
        if (lane == 0) {
            for (uint32_t k = 0; k < nseg; k++) {
                const int g = seg_before + (int)k;
                cy[2 + 2 * (g % CARRY_HIST)] = sg[8 * k + SEG_START]; cy[3 + 2 * (g % CARRY_HIST)] = sg[8 * k + SEG_LEN];
            }
            cy[0] = seg_before + (int)nseg; cy[1] = res_before + si;
            if (lost) atomicOr(&p.totals[2], 1u);
        }
    }
}

void launch_compact(const CompactParams& p, hipStream_t s) {
    if (p.n_clips == 0) return;
    if (p.fused && !p.carry && p.clip_rows && p.n_clips <= 4096) { hipLaunchKernelGGL(compact_gather_kernel<true>, dim3(p.n_clips), dim3(64), 0, s, p); return; }
    if (p.n_clips <= 2 * CSCAN_T) hipLaunchKernelGGL(compact_scan_kernel<true>, dim3(1), dim3(CSCAN_T), 0, s, p);
    else {
        hipLaunchKernelGGL(compact_count_kernel, dim3((p.n_clips + 255) / 256), dim3(256), 0, s, p);
        hipLaunchKernelGGL(compact_scan_kernel<false>, dim3(1), dim3(CSCAN_T), 0, s, p);
    }
    hipLaunchKernelGGL(compact_gather_kernel<false>, dim3(p.n_clips), dim3(64), 0, s, p);
}
bool compact_is_fused(const CompactParams& p) { return p.fused && !p.carry && p.clip_rows && p.n_clips <= 4096 && p.n_clips > 0; }

}  // namespace wsa
