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
#include "wave_ops.hpp"
#include <cstdlib>

namespace wsa {

constexpr int PK_TILE = 16;                 // bins per LDS tile: 64 bytes per frame row per load (half a cache line: with 32-byte tiles every line
                                            // was fetched four times, a wave's 64 rows outliving their stay in L2 between tiles: 2.1x the spectrum read,
                                            // 0.30 ms; 16 bins 0.9x (as counted), 0.22 ms; 32 bins 0.5x but too much LDS / registers: 0.24 ms)
constexpr int PK_RING = 32;                // bins of history kept in LDS per row (two tiles)
constexpr int PK_RS = PK_RING + 1;          // row stride in words: conflict-free lane-per-row walks

__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(4, 4))) void peaks_kernel(PkParams p) {
    // one lane = one frame.  Rows are staged PK_TILE bins at a time through LDS: the wave reads 64 rows
    // x 64 B (4 lanes per row, 16 B per lane) and each lane then walks its own row segment out of LDS
    // (row stride PK_RING + 1 words: conflict free).  Tiles of 16 bins keep the ring at 8.4 KB per wave; with the parked record pieces (below) the wave's LDS is exactly 10 240 B = 16 waves per CU.
    // Candidates leave as entries of the structure-of-arrays table behind the frame's header (wsa_internal.hpp).
    __shared__ uint32_t tile[64 * PK_RS];     // bin t of row r lives at r * PK_RS + (t & (PK_RING - 1))
    __shared__ uint4 park_ent[64];            // record stores: the entry / amplitudes waiting for their sector to fill (WSA_STORE)
    __shared__ uint32_t park_amp[64 * 3];
    const int lane = threadIdx.x;
    const uint32_t f0 = p.frame0 + blockIdx.x * 64u;
    const uint32_t nf = min(64u, p.frame0 + p.total_frames - f0);
    const int B = p.bands;
    const uint32_t f = f0 + lane;
    bool live = (uint32_t)lane < nf;
    const uint32_t* e = p.spec + (uint64_t)(live ? f : f0) * (uint32_t)B;       // own row (shoulder re-reads)
    uint64_t slot = live ? f : f0;
    if (p.stream_state) {                       // streaming: records live in per-stream rings
        const uint32_t sidx = (live ? f : f0) / p.step_frames, j = (live ? f : f0) - sidx * p.step_frames;
        live = live && j < p.n_frames[sidx];
        const uint32_t seen = (uint32_t)p.stream_state[(uint64_t)sidx * GATE_STATE];
        slot = (uint64_t)sidx * p.ring + ((seen + j) & (p.ring - 1));
    }
    const uint32_t cbase = (uint32_t)slot * (uint32_t)CAND_CAP;      // this frame's own CAND_CAP entries of the candidate table
    int n = 0, i = 0, l = 0, s = 0;
    // direction u in {1, -1, 0} and flat counter c in {0, 1, 2} of the reference's scan as LANE MASKS in scalar registers (bit = lane =
    // frame): U1 / UM = lanes with u == 1 / u == -1, C1 / C2 = lanes with c == 1 / c == 2.  Everything that only combines them is a
    // scalar instruction for all 64 frames at once; a mask becomes a per-lane condition again through inverse_ballot (free: the
    // mask register IS the condition of v_cndmask / exec)
    uint64_t U1 = 0, UM = 0, C1 = 0, C2 = 0, PEND = 0;
    const uint64_t LIVE = __ballot(live);            // only live lanes ever emit
    // exact running prefix sums P[x] = sum e[0..x]: tot = P[a], t1 = P[a-1], t2 = P[a-2] (P[-1] = 0)
    uint64_t tot = 0, t1 = 0, t2 = 0;
    uint32_t e0 = 0;
    // thr = e[l]/10 in the reference; e[x] < e[l]/10  <=>  10 e[x] < e[l] for u32 values
    // (e[l]/10 differs from an integer by 0 or >= 0.1, far more than a double ulp).
    // bit 24 marks the end-of-spectrum emission, which the reference adds to n and d but never
    // lets update h / p (ref @B26383: no `e[l]>h&&(h=e[l],p=l)` in that arm).
    // Each emission also records the exact prefix sums at its (shrunk) shoulders, so that the tracker
    // gets any band energy sum e[st..en] (ref @B36500 `for(t=a;t<=f;t++)d+=e[t]`) by one subtraction.
    // p_i = P[i-1] and p_s = P[s] are carried along with i and s (set where the scan sets i / s, adjusted by the very
    // elements the shoulder shrink looks at).
    uint64_t p_i = 0, p_s = 0;
    uint32_t e_l = 0;                           // e[l]
    // Emission.  A lane emits a candidate every ~10 bins, but with 64 lanes SOME lane emits at almost every bin, and
    // an emission body inside the scan is paid by the whole wave each time.  So (batch variant) a lane only parks its
    // candidate in registers; the wave runs the emission body — /10 shoulder shrink out of the LDS ring, entry store —
    // when some lane needs its parking slot again (every ~4 bins), for all parked candidates at once.
    // 10 e[x] < e[l]  <=>  e[x] < ceil(e[l] / 10): one 32-bit compare per shoulder bin.  Shoulder bins are re-read
    // from the LDS ring (current and previous tile); older ones from the row in global memory.
    // the largest candidate that may become h / p in the gate (the end-of-spectrum one never does), first one on ties:
    // independent of the noise floor, so it is found here, one lane per frame, instead of by a wave reduction per frame there
    uint32_t mx_amp = 0, mx_bin = 0;
    int qi = 0, qs = 0, ql = 0; uint32_t qe = 0, qlast = 0; uint64_t qpi = 0, qps = 0;
    // Record stores.  A lane's candidates fill its 64 table slots one by one, ~20 per frame, over the whole life of the wave: written as
    // they come (one 4-byte amplitude, one 16-byte entry) almost every 32-byte sector left L2 half-written and was written again —
    // 395 MB of stores for 170 MB of records.  So the even-numbered entry and three amplitudes of four wait in LDS (1.75 KB per wave:
    // with the ring exactly the 10 240 bytes 16 waves per CU allow) and leave together with the next one(s): whole sectors, written once.
#define WSA_STORE(ci_, cs_, cl_, ce_, cpi_, cps_, clast_) do { \
        if (n >= CAND_CAP) { atomicOr(p.flags, 1u); } else { /* a frame holds CAND_CAP candidates: all a spectrum of <= 128 bands can have */ \
        const uint32_t c_ = cbase + (uint32_t)n; \
        const uint4 ent_ = make_uint4((uint32_t)(ci_) | ((uint32_t)(cs_) << 8) | ((uint32_t)(cl_) << 16) | ((uint32_t)(clast_) << 24), (uint32_t)(cpi_), (uint32_t)(cps_), \
                                      (uint32_t)((uint64_t)(cpi_) >> 32) | ((uint32_t)((uint64_t)(cps_) >> 32) << 8)); \
        if (n & 1) { const uint4 prev_ = park_ent[lane]; p.rec.ent[c_ - 1u] = prev_; p.rec.ent[c_] = ent_; } else park_ent[lane] = ent_; \
        if ((n & 3) == 3) { const uint32_t* pa_ = park_amp + 3 * lane; *reinterpret_cast<uint4*>(p.rec.amp + (c_ - 3u)) = make_uint4(pa_[0], pa_[1], pa_[2], (ce_)); } \
        else park_amp[3 * lane + (n & 3)] = (ce_); \
        n++; \
        if (!(clast_) && (ce_) > mx_amp) { mx_amp = (ce_); mx_bin = (uint32_t)(cl_); } } } while (0)
#define WSA_IB(m_) __builtin_amdgcn_inverse_ballot_w64(m_)
    // Shoulder shrink: bins of the current and the previous tile come out of the LDS ring with plain ds_read; only a candidate wider than
    // that reaches back into its row in global memory — in loops of their own, so that the usual path holds no global / flat load (a flat
    // load, which is what one loop over "ring or row" compiles to, waits on vmcnt AND lgkmcnt and would drain the next tile's prefetch
    // at every flush).
#define WSA_FLUSH(a_now) do { if (WSA_IB(PEND)) { \
        const int lo_valid_ = ((a_now) & ~(PK_TILE - 1)) - PK_TILE;   /* ring holds this tile and the one before */ \
        const uint32_t thr_ = qe / 10u + (qe % 10u != 0u ? 1u : 0u); \
        bool stop_ = false; \
        if (__builtin_expect(qi < lo_valid_, 0)) while (qi < ql && qi < lo_valid_) { const uint32_t x_ = e[qi]; if (!(x_ < thr_)) { stop_ = true; break; } qpi += x_; qi++; } \
        if (!stop_) while (qi < ql) { const uint32_t x_ = myrow[qi & (PK_RING - 1)]; if (!(x_ < thr_)) break; qpi += x_; qi++; } \
        stop_ = false; \
        while (qs > ql && qs >= lo_valid_) { const uint32_t x_ = myrow[qs & (PK_RING - 1)]; if (!(x_ < thr_)) { stop_ = true; break; } qps -= x_; qs--; } \
        if (__builtin_expect(!stop_ && qs > ql, 0)) while (qs > ql) { const uint32_t x_ = e[qs]; if (!(x_ < thr_)) break; qps -= x_; qs--; } \
        WSA_STORE(qi, qs, ql, qe, qpi, qps, qlast); } PEND = 0; } while (0)
    // EM = lanes that emit at this bin: park [i, s, l] (flushing first when one of them still holds a parked candidate)
#define WSA_EMIT(EM, last, a_now) do { \
        if ((EM) & PEND) WSA_FLUSH(a_now); \
        const bool em_ = WSA_IB(EM); \
        qi = em_ ? i : qi; qs = em_ ? s : qs; ql = em_ ? l : ql; qe = em_ ? e_l : qe; qpi = em_ ? p_i : qpi; qps = em_ ? p_s : qps; qlast = em_ ? (last) : qlast; \
        PEND |= (EM); } while (0)
    // one bin step (ref @B25827, restated in oracle/backend.c) for the 64 frames of the wave.
    // GUARD = the first bins, where e[a-2] / e[a-3] do not exist yet (ref `(a<2||...)&&(a<3||...)`).
#define WSA_STEP(a, ea, GUARD) do { \
        tot = t1 + (ea); \
        const uint64_t r1_ = __ballot((ea) > e1), f1_ = __ballot((ea) < e1); \
        const uint64_t r2_ = ((GUARD) && (a) < 2) ? ~0ull : __ballot((ea) > e2), f2_ = ((GUARD) && (a) < 2) ? ~0ull : __ballot((ea) < e2); \
        const uint64_t r3_ = ((GUARD) && (a) < 3) ? ~0ull : __ballot((ea) > e3), f3_ = ((GUARD) && (a) < 3) ? ~0ull : __ballot((ea) < e3); \
        const uint64_t R_ = r1_ & r2_ & r3_, F_ = f1_ & f2_ & f3_; \
        const uint64_t FLAT_ = ~R_ & ~F_ & UM, TRIG_ = FLAT_ & C2; \
        const uint64_t EM_ = ((R_ & UM) | TRIG_) & __ballot(i <= l && l < s) & LIVE; \
        if (EM_) WSA_EMIT(EM_, 0u, a); \
        const uint64_t NC1_ = (FLAT_ & ~C1 & ~C2) | (~FLAT_ & C1), NC2_ = (FLAT_ & C1) | (~FLAT_ & C2);   /* c: 0 -> 1 -> 2 -> (trigger) 0 */ \
        C1 = NC1_; C2 = NC2_; \
        const bool set_i_ = WSA_IB(R_ & ~U1); \
        i = set_i_ ? (a) - 1 : i; p_i = set_i_ ? t2 : p_i; \
        const bool set_l_ = WSA_IB(R_ | (~F_ & U1 & r1_)); \
        l = set_l_ ? (a) : l; e_l = set_l_ ? (ea) : e_l; \
        const uint64_t SETS_ = F_ & (U1 | UM); \
        const bool set_s_ = WSA_IB(SETS_); \
        s = set_s_ ? (a) : s; p_s = set_s_ ? tot : p_s; \
        const uint64_t NU1_ = R_ | (U1 & ~SETS_), NUM_ = ~R_ & (SETS_ | (UM & ~TRIG_));   /* u = rise ? 1 : (set_s ? -1 : (trig ? 0 : u)) */ \
        U1 = NU1_; UM = NUM_; \
        t2 = t1; t1 = tot; \
        e3 = e2; e2 = e1; e1 = (ea); } while (0)
    uint32_t e1 = 0, e2 = 0, e3 = 0;            // e[a-1], e[a-2], e[a-3]
    const uint32_t* myrow = tile + lane * PK_RS;
    const uint32_t* src = p.spec + (uint64_t)f0 * (uint32_t)B;
    // full tiles of a 16-byte-aligned row travel global -> registers -> LDS, the next loads being issued before the current tile is
    // walked (the walk hides their latency).  When a row is a whole number of 128-byte lines (B % 32 == 0) the two tiles that share a
    // line are requested TOGETHER, every second tile, and the second one waits in registers: a line requested half by half, a tile
    // walk apart, had been evicted from L2 in between more often than not (the wave's 64 rows x 16 waves x 32 CUs is all an XCD's
    // L2 holds) and was fetched twice — 1.8x the spectrum read.
    constexpr int LPR = PK_TILE / 4, RPI = 64 / LPR, NLD = 64 / RPI;
    const bool vec = (B & 3) == 0;
    const bool pair = (B % (2 * PK_TILE)) == 0;
    uint4 nxt[NLD], nxt2[NLD];
    auto fetch = [&](int t0, uint4 (&dst)[NLD]) __attribute__((always_inline)) {
#pragma unroll
        for (int k = 0; k < NLD; k++) {
            const int r = RPI * k + lane / LPR, q = lane % LPR;
            dst[k] = make_uint4(0u, 0u, 0u, 0u);
            if ((uint32_t)r < nf) dst[k] = *reinterpret_cast<const uint4*>(src + (uint64_t)r * (uint32_t)B + t0 + 4 * q);
        }
    };
    // PK_TILE/4 lanes x 16 B cover one row's tile; 256/PK_TILE rows per load instruction
    auto to_lds = [&](int t0, const uint4 (&v)[NLD]) __attribute__((always_inline)) {
#pragma unroll
        for (int k = 0; k < NLD; k++) {
            const int r = RPI * k + lane / LPR, q = lane % LPR;
            if ((uint32_t)r < nf) {
                uint32_t* d = tile + r * PK_RS + ((t0 + 4 * q) & (PK_RING - 1));
                d[0] = v[k].x; d[1] = v[k].y; d[2] = v[k].z; d[3] = v[k].w;
            }
        }
    };
    // every lane walks its row, live or not (rows past the launch's last frame hold whatever the ring held: their lanes never emit):
    // the lane masks stay uniform values in scalar registers only as long as no divergent branch encloses their updates
    auto walk = [&](int t0, int tw) __attribute__((always_inline)) {
        const uint32_t* seg = myrow + (t0 & (PK_RING - 1));      // PK_TILE divides PK_RING: the tile is contiguous in the ring
        if (t0 == 0) {
            for (int q = 0; q < tw; q++) {
                const uint32_t ea = seg[q];
                if (q == 0) { e1 = ea; e0 = ea; t1 = ea; tot = ea; } else WSA_STEP(q, ea, true);
            }
        } else if (tw == PK_TILE) {
#pragma unroll
            for (int q = 0; q < PK_TILE; q++) { const uint32_t ea = seg[q]; WSA_STEP(t0 + q, ea, false); }
        } else {
            for (int q = 0; q < tw; q++) { const uint32_t ea = seg[q]; WSA_STEP(t0 + q, ea, false); }
        }
    };
    // (the workgroup is one wave: its LDS accesses execute in order, so only the compiler has to be kept from moving them across —
    //  __syncthreads() would also drain vmcnt, i.e. wait for the loads issued just above it, and nothing would be in flight while
    //  the current tile is walked)
    if (pair) {
        fetch(0, nxt); fetch(PK_TILE, nxt2);
        for (int t0 = 0; t0 < B; t0 += 2 * PK_TILE) {
            wsync();
            to_lds(t0, nxt);
            wsync();
            walk(t0, PK_TILE);
            wsync();
            to_lds(t0 + PK_TILE, nxt2);
            if (t0 + 4 * PK_TILE <= B) { fetch(t0 + 2 * PK_TILE, nxt); fetch(t0 + 3 * PK_TILE, nxt2); }
            wsync();
            walk(t0 + PK_TILE, PK_TILE);
        }
    } else {
        if (vec && B >= PK_TILE) fetch(0, nxt);
        for (int t0 = 0; t0 < B; t0 += PK_TILE) {
            const int tw = min(PK_TILE, B - t0);
            wsync();
            if (tw == PK_TILE && vec) {
                to_lds(t0, nxt);
                if (t0 + 2 * PK_TILE <= B) fetch(t0 + PK_TILE, nxt);
            } else {
                for (int idx = lane; idx < (int)nf * tw; idx += 64) { const int r = idx / tw, q = idx - r * tw; tile[r * PK_RS + ((t0 + q) & (PK_RING - 1))] = src[(uint64_t)r * (uint32_t)B + t0 + q]; }
            }
            wsync();
            walk(t0, tw);
        }
    }
    {
        // end of spectrum (ref @B26383): a peak still rising at the last bin is closed there
        const bool last_ = B > 1 && WSA_IB(U1);
        if (last_) { s = B - 1; p_s = tot; l = B - 1; e_l = e1; }
        const uint64_t EL_ = __ballot(last_ && i < l && l <= s) & LIVE;
        if (EL_) WSA_EMIT(EL_, 1u, B - 1);
        WSA_FLUSH(B - 1);
    }
#undef WSA_STEP
#undef WSA_EMIT
#undef WSA_FLUSH
#undef WSA_STORE
#undef WSA_IB
    if (live) {
        // what still waits in LDS: the last entry of an odd count, the last n % 4 amplitudes
        if ((n & 1) && n <= CAND_CAP) p.rec.ent[cbase + (uint32_t)n - 1u] = park_ent[lane];
        for (int k = 0; k < (n & 3); k++) p.rec.amp[cbase + (uint32_t)(n & ~3) + (uint32_t)k] = park_amp[3 * lane + k];
        const uint64_t g = tot - (uint64_t)e0;                              // g = sum e[1..B-1]
        p.rec.hdr[slot] = make_uint4((uint32_t)g, (uint32_t)(g >> 32) | ((uint32_t)n << 8) | (mx_bin << 16), mx_amp, cbase);
    }
}


// ---- the same scan with one WAVE per frame, for launches too small to fill lanes with frames (stream steps: a few hundred
// frames, where the lane-per-frame kernel is eight waves walking 127 bins one after the other).  The rising / falling / creeping
// tests of all bins are evaluated at once (lane = bin, neighbours by wave shifts) and become three bit masks; an inclusive
// wave scan gives every prefix sum; the direction / flat-run state machine (ref @B25827) then runs over the masks in scalar
// registers — five small integers per bin instead of the whole bookkeeping — and looks amplitudes and prefix sums up with
// v_readlane only where a candidate is emitted.  Same records, bit for bit (tests/test_gpu_stream.py).  Up to 128 bands.
__device__ __forceinline__ double wave_incl_scan_f64(double v) {
    v += dpp_f64_or_zero<0x111, 0xf>(v);
    v += dpp_f64_or_zero<0x112, 0xf>(v);
    v += dpp_f64_or_zero<0x114, 0xf>(v);
    v += dpp_f64_or_zero<0x118, 0xf>(v);
    v += dpp_f64_or_zero<0x142, 0xa>(v);
    v += dpp_f64_or_zero<0x143, 0xc>(v);
    return v;
}

__global__ __launch_bounds__(64) void peaks_wave_kernel(PkParams p) {
    const int lane = threadIdx.x;
    const int B = p.bands;
    const uint32_t f = p.frame0 + blockIdx.x;
    uint64_t slot = f;
    if (p.stream_state) {
        const uint32_t sidx = f / p.step_frames, j = f - sidx * p.step_frames;
        if (j >= p.n_frames[sidx]) return;
        const uint32_t seen = (uint32_t)p.stream_state[(uint64_t)sidx * GATE_STATE];
        slot = (uint64_t)sidx * p.ring + ((seen + j) & (p.ring - 1));
    }
    const uint32_t cbase = (uint32_t)slot * (uint32_t)CAND_CAP;
    const uint32_t* e = p.spec + (uint64_t)f * (uint32_t)B;
    // bins a = lane (half 0) and a = lane + 64 (half 1)
    const uint32_t x0 = lane < B ? e[lane] : 0u, x1 = lane + 64 < B ? e[lane + 64] : 0u;
    auto up1 = [&](uint32_t lo, uint32_t hi, uint32_t& slo, uint32_t& shi) __attribute__((always_inline)) {   // value of bin a - 1 at bin a
        const uint32_t carry = (uint32_t)__builtin_amdgcn_readlane((int)lo, 63);
        slo = (uint32_t)__shfl_up((int)lo, 1, 64); shi = (uint32_t)__shfl_up((int)hi, 1, 64);
        if (lane == 0) { slo = 0u; shi = carry; }
    };
    uint32_t a1, b1, a2, b2, a3, b3;
    up1(x0, x1, a1, b1); up1(a1, b1, a2, b2); up1(a2, b2, a3, b3);
    auto classify = [&](int a, uint32_t ea, uint32_t e1, uint32_t e2, uint32_t e3, bool& rise, bool& fall, bool& creep) __attribute__((always_inline)) {
        const bool in = a >= 1 && a < B;
        rise = in && ea > e1 && (a < 2 || ea > e2) && (a < 3 || ea > e3);
        fall = in && ea < e1 && (a < 2 || ea < e2) && (a < 3 || ea < e3);
        creep = in && ea > e1;
    };
    bool r0, f0, g0, r1, f1, g1;
    classify(lane, x0, a1, a2, a3, r0, f0, g0); classify(lane + 64, x1, b1, b2, b3, r1, f1, g1);
    const uint64_t R0 = __ballot(r0), R1 = __ballot(r1), F0 = __ballot(f0), F1 = __ballot(f1), G0 = __ballot(g0), G1 = __ballot(g1);
    // inclusive prefix sums P[a] = sum e[0..a] (integers below 2^40: exact in double)
    const double P0 = wave_incl_scan_f64((double)x0);
    const double tot0 = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(P0), 63), __builtin_amdgcn_readlane(__double2loint(P0), 63));
    const double P1 = wave_incl_scan_f64((double)x1) + tot0;
    auto amp_at = [&](int a) __attribute__((always_inline)) -> uint32_t {
        return a < 64 ? (uint32_t)__builtin_amdgcn_readlane((int)x0, a) : (uint32_t)__builtin_amdgcn_readlane((int)x1, a - 64);
    };
    auto prefix_at = [&](int a) __attribute__((always_inline)) -> double {      // P[a]; P[-1] = 0
        if (a < 0) return 0.0;
        return a < 64 ? __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(P0), a), __builtin_amdgcn_readlane(__double2loint(P0), a))
                      : __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(P1), a - 64), __builtin_amdgcn_readlane(__double2loint(P1), a - 64));
    };
    int n = 0, i = 0, l = 0, s = 0, c = 0, u = 0;
    uint32_t mx_amp = 0, mx_bin = 0;
    bool too_many = false;
    auto emit = [&](int ci, int cs, int cl, uint32_t last) __attribute__((always_inline)) {
        const uint32_t qe = amp_at(cl);
        const uint32_t thr = qe / 10u + (qe % 10u != 0u ? 1u : 0u);        // e[x] < e[l] / 10  <=>  e[x] < ceil(e[l] / 10)
        int qi = ci, qs = cs;
        while (qi < cl && amp_at(qi) < thr) qi++;
        while (qs > cl && amp_at(qs) < thr) qs--;
        if (n >= CAND_CAP) { too_many = true; return; }
        if (lane == 0) {
            const uint32_t c = cbase + (uint32_t)n;
            const uint64_t plo = (uint64_t)prefix_at(qi - 1), phi = (uint64_t)prefix_at(qs);      // exact integers below 2^40
            p.rec.amp[c] = qe;
            p.rec.ent[c] = make_uint4((uint32_t)qi | ((uint32_t)qs << 8) | ((uint32_t)cl << 16) | (last << 24), (uint32_t)plo, (uint32_t)phi, (uint32_t)(plo >> 32) | ((uint32_t)(phi >> 32) << 8));
        }
        n++;
        if (!last && qe > mx_amp) { mx_amp = qe; mx_bin = (uint32_t)cl; }
    };
    for (int a = 1; a < B; a++) {
        const uint64_t bit = 1ull << (a & 63);
        const bool rise = ((a < 64 ? R0 : R1) & bit) != 0ull, fall = ((a < 64 ? F0 : F1) & bit) != 0ull, creep = ((a < 64 ? G0 : G1) & bit) != 0ull;
        const bool flat = !rise && !fall && u == -1;
        c += flat ? 1 : 0;
        const bool trig = flat && c > 2;
        if (((rise && u == -1) || trig) && i <= l && l < s) emit(i, s, l, 0u);
        if (rise && u != 1) i = a - 1;
        if (rise || (!fall && u == 1 && creep)) l = a;
        const bool set_s = fall && u != 0;
        if (set_s) s = a;
        u = rise ? 1 : (set_s ? -1 : (trig ? 0 : u));
        if (trig) c = 0;
    }
    // end of spectrum (ref @B26383): a peak still rising at the last bin is closed there
    if (B > 1 && u == 1) { s = B - 1; l = B - 1; if (i < l && l <= s) emit(i, s, l, 1u); }
    if (lane == 0) {
        const uint64_t g = (uint64_t)(prefix_at(B - 1) - (double)amp_at(0));        // g = sum e[1..B-1]
        p.rec.hdr[slot] = make_uint4((uint32_t)g, (uint32_t)(g >> 32) | ((uint32_t)n << 8) | (mx_bin << 16), mx_amp, cbase);
        if (too_many) atomicOr(p.flags, 1u);
    }
}

void launch_peaks(const PkParams& p, hipStream_t s) {
    if (p.total_frames == 0) return;
    // few frames (stream steps): one wave per frame instead of one lane per frame (WSA_PEAKS_LANES=1 keeps the lane kernel: test hook)
    const bool lanes_only = std::getenv("WSA_PEAKS_LANES") != nullptr;
    if (p.total_frames <= 4096u && p.bands <= 128 && !lanes_only) hipLaunchKernelGGL(peaks_wave_kernel, dim3(p.total_frames), dim3(64), 0, s, p);
    else hipLaunchKernelGGL(peaks_kernel, dim3((p.total_frames + 63) / 64), dim3(64), 0, s, p);
}

}  // namespace wsa
