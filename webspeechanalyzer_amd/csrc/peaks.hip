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

#ifdef WSA_TUNING
#define WSA_PKT(bits_) (p.dbg & (bits_))
#else
#define WSA_PKT(bits_) false
#endif

namespace wsa {

// The kernel's geometry is a template parameter: PK_W bins per round (at most one word of the per-lane bit masks: 32, or 16), an LDS ring
// of PK_RB = 2 PK_W bins (this round's tile and the one before) and a candidate list of PK_LIST = 8 PK_W entries (a round of 32 bins
// yields ~230 candidates, one of 16 ~115).  <32>: 20 480 B of LDS per wave = 16 allocation units, 8 waves per CU; <16>: 11 136 B = 9 units,
// 14 waves per CU — and 8 instead of 4 beside the two front-end workgroups of another batch (profiles/r04_notes.md).
constexpr int PK_RS = 65;                  // words per bin row of the ring ([bin & (PK_RB - 1)][frame]): the odd stride keeps the walks (lane = frame) free of bank conflicts

// mask = mask * 2 + (a > b) / (a < b): the compare's lane mask IS the carry-in of v_addc (two instructions per bin and mask;
// the first bin of a round ends up in the highest bit, v_bfrev turns the word round once per round)
#define WSA_PUSH_GT(m_, a_, b_) asm("v_cmp_gt_u32 vcc, %1, %2\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc" : "+v"(m_) : "v"(a_), "v"(b_) : "vcc")
#define WSA_PUSH_LT(m_, a_, b_) asm("v_cmp_lt_u32 vcc, %1, %2\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc" : "+v"(m_) : "v"(a_), "v"(b_) : "vcc")
// p += x (low word of the running prefix sum); cm = cm * 2 + carry: where the sum crosses a multiple of 2^32
#define WSA_ADD_CARRY(p_, cm_, x_) asm("v_add_co_u32 %0, vcc, %0, %2\n\tv_addc_co_u32 %1, vcc, %1, %1, vcc" : "+v"(p_), "+v"(cm_) : "v"(x_) : "vcc")

template <int PK_W>
__global__ __launch_bounds__(64) void peaks_kernel(PkParams p) {
    constexpr int PK_RB = 2 * PK_W, PK_LIST = 8 * PK_W;
    // One lane = one frame for the scan, one lane = one CANDIDATE for what follows it; per round of 32 bins:
    //  1. the tile travels global -> registers (coalesced: 8 lanes x 16 B per row) -> LDS, transposed to [bin][frame];
    //  2. every lane walks its own 32 bins once: running prefix sum P (low word written back over the tile: what the shoulder
    //     shrink and the band sums need later; the carries into bit 32 as a bit mask) and the three predicates of the reference's
    //     scan — rising / falling against the three bins before, e[a] > e[a-1] — as one bit per bin in three mask registers;
    //  3. the direction / flat-run state machine (ref @B25827) does not visit bins, it jumps from event to event with
    //     find-first-bit on the masks: next rise, next fall, third flat bin of a falling stretch.  One iteration of the (branch-free)
    //     loop is one peak cycle — close the falling stretch, rise, fall — for all 64 frames; a candidate [i, s, l] it closes is
    //     appended to a list the wave keeps in LDS;
    //  4. the list is worked off with one lane per candidate, 64 at a time whatever frame they belong to: /10 shoulder shrink and the
    //     exact prefix sums out of the LDS ring, entry + amplitude stores, the frame's largest candidate by an LDS atomic.
    // The state machine's cost goes with the number of peak cycles (~23 per frame, 34 iterations per wave: the slowest lane of each
    // round), the emission's with the number of candidates (~14 per frame) at full lanes — not with the number of bins (r02: one bin
    // step of the wave cost ~140 instructions; profiles/r03_notes.md).
    __builtin_amdgcn_s_setprio(3);               // as in the paired tracker: chains first, the front end of the next batch fills the gaps
    __shared__ uint32_t ringP[PK_RB * PK_RS];
    __shared__ uint32_t lstA[PK_LIST], lstB[PK_LIST];     // i | s << 8 | l << 16 | last << 24;  frame lane | ordinal in the frame << 8
    __shared__ unsigned long long mxk[64];               // per frame: amplitude << 32 | (63 - ordinal) << 8 | bin of the largest candidate so far
    __shared__ uint32_t cbl[64];                         // per frame: first entry of its candidate table
    __shared__ uint32_t cmw[2][64], hib[2][64];          // per frame and ring half: carry mask of the tile's 32 bins, high byte of P before the tile
    const int lane = threadIdx.x;
    const uint32_t f0 = p.frame0 + blockIdx.x * 64u;
    const uint32_t nf = min(64u, p.frame0 + p.total_frames - f0);
    const int B = p.bands;
    const uint32_t f = f0 + lane;
    bool live = (uint32_t)lane < nf;
    uint64_t slot = live ? f : f0;
    if (p.stream_state) {                       // streaming: records live in per-stream rings
        const uint32_t sidx = (live ? f : f0) / p.step_frames, j = (live ? f : f0) - sidx * p.step_frames;
        live = live && j < p.n_frames[sidx];
        const uint32_t seen = (uint32_t)p.stream_state[(uint64_t)sidx * GATE_STATE];
        slot = (uint64_t)sidx * p.ring + ((seen + j) & (p.ring - 1));
    }
    const uint32_t cbase = (uint32_t)slot * (uint32_t)CAND_CAP;      // this frame's own CAND_CAP entries of the candidate table
    const uint32_t* src = p.spec + (uint64_t)f0 * (uint32_t)B;
    const bool vec = (B & 3) == 0 && (reinterpret_cast<uintptr_t>(p.spec) & 15u) == 0;
    cbl[lane] = cbase; mxk[lane] = 0ull;

    // ---- state of the reference's scan: u (0 idle, 1 rising, 2 falling = the reference's -1), flat counter c, [i, s, l]
    int u = 0, c = 0, i = 0, l = 0, s = 0, n = 0;
    uint32_t plo = 0, phi = 0;                  // P[a] = sum e[0..a]: low word, high byte
    uint32_t e0 = 0, e1 = 0, e2 = 0, e3 = 0;    // e[0]; e[a-1], e[a-2], e[a-3]
    int nlist = 0;                              // candidates waiting in the list (uniform)
    bool too_many = false;

    // ---- emission, lane = list entry (ref EMIT: `thr = e[l] / 10; while (i < l && e[i] < thr) i++; while (s > l && e[s] < thr) s--`):
    // e[x] < e[l] / 10  <=>  10 e[x] < e[l]  <=>  e[x] < ceil(e[l] / 10) for u32 values (e[l] / 10 differs from an integer by 0 or >= 0.1,
    // far more than a double ulp).  e[x] = P[x] - P[x-1] in the low words; the walk leaves P[i-1] and P[s] of the shrunk shoulders
    // behind, which the tracker turns into any band sum e[st..en] (ref @B36500) by one subtraction.  Bit 24 marks the end-of-spectrum
    // emission, which the reference adds to n and d but never lets update h / p (ref @B26383).
    // lo_valid = first bin the ring still holds; for a candidate that starts before it (a rise of more than PK_W bins: 6 in 10^5 candidates of
    // speech-like input at 32, 6 in 10^3 at 16) the few bins below come from its frame's row in global memory.  any_hi: some frame's sum has passed 2^32 (uniform; else every high byte is 0).
    // all = false (the list is full in the middle of a round): only whole groups of 64 are worked off, the rest moves to the front
    auto flush = [&](int lo_valid, bool any_hi, bool all) __attribute__((always_inline)) {
        wsync();
        const int nwork = all ? nlist : (nlist & ~63);
        for (int j0 = 0; j0 < (WSA_PKT(1) ? 0 : nwork); j0 += 64) {
            const int j = j0 + lane;
            if (j < nwork) {
                const uint32_t wa = lstA[j], wb = lstB[j];
                const int fl = wb & 63, ord = wb >> 8;
                const int ci = wa & 0xff, cs = (wa >> 8) & 0xff, cl = (wa >> 16) & 0xff;
                int qi = ci, qs = cs;
                uint32_t qe, pil, pih = 0, psl, psh = 0;
                const uint32_t* rp = ringP + fl;
                auto at = [&](int x) __attribute__((always_inline)) -> uint32_t { return rp[(x & (PK_RB - 1)) * PK_RS]; };
                auto hi_at = [&](int x) __attribute__((always_inline)) -> uint32_t {        // high byte of P[x], x >= lo_valid
                    const int h = (x / PK_W) & 1;
                    return hib[h][fl] + (uint32_t)__popc(cmw[h][fl] & (0xffffffffu >> (31 - (x & (PK_W - 1)))));
                };
                if (__builtin_expect((ci > 0 ? ci - 1 : 0) < lo_valid && !WSA_PKT(32), 0)) {
                    // the ring holds P[lo_valid ..]: what lies below is P[lo_valid] minus the bins between, read from the frame's row
                    const uint32_t* e = src + (uint64_t)fl * (uint32_t)B;
                    qe = e[cl];
                    const uint32_t thr = qe / 10u + (qe % 10u != 0u ? 1u : 0u);
                    while (qi < cl) { const uint32_t x = e[qi]; if (!(x < thr)) break; qi++; }
                    while (qs > cl) { const uint32_t x = e[qs]; if (!(x < thr)) break; qs--; }
                    auto p64 = [&](int x) __attribute__((always_inline)) -> uint64_t { return ((uint64_t)(any_hi ? hi_at(x) : 0u) << 32) | at(x); };
                    uint64_t acc = 0;
                    if (qi > 0) { const int base = max(qi - 1, lo_valid); acc = p64(base); for (int x = base; x >= qi; x--) acc -= e[x]; }
                    pil = (uint32_t)acc; pih = (uint32_t)(acc >> 32);
                    { const int base = max(qs, lo_valid); acc = p64(base); for (int x = base; x > qs; x--) acc -= e[x]; }
                    psl = (uint32_t)acc; psh = (uint32_t)(acc >> 32);
                } else {
                    // the shoulders shrink by 0.8 / 1.0 bins on average but by 6 / 4 for the slowest of 64 candidates: walking bin by bin the
                    // wave paid an LDS round trip per step.  Both shoulders advance together, two bins per round trip each (1 / 2 / 3 / 4 bins per
                    // trip: 161 / 156 / 158 / 159 us for the kernel — the later trips serve a few lanes, so their instruction count matters too).
                    const uint32_t v_l = at(cl), v_l1 = at(cl - 1);                      // cl >= 1
                    uint32_t cur = at(ci - 1), cur2 = at(cs);
                    const uint32_t l0 = at(ci), r0 = at(cs - 1);                         // cs > cl >= 1
                    if (ci == 0) cur = 0u;
                    qe = v_l - v_l1;
                    const uint32_t thr = qe / 10u + (qe % 10u != 0u ? 1u : 0u);
                    // the first bin of either shoulder rides along with the reads above (55 % / 36 % of the shoulders do not shrink at all, 85 % / 73 % by
                    // at most one bin); what goes on after that advances two bins per LDS round trip on both sides
                    bool goL = true, goR = true;
#define WSA_STEP_L(v_) do { goL = goL && qi < cl && (v_) - cur < thr; cur = goL ? (v_) : cur; qi += goL ? 1 : 0; } while (0)
#define WSA_STEP_R(v_) do { goR = goR && qs > cl && cur2 - (v_) < thr; cur2 = goR ? (v_) : cur2; qs -= goR ? 1 : 0; } while (0)
                    WSA_STEP_L(l0); WSA_STEP_R(r0);
                    while (goL || goR) {
                        const uint32_t l1 = at(qi), l2 = at(qi + 1);
                        const uint32_t r1 = at(qs - 1), r2 = at(qs - 2);
                        WSA_STEP_L(l1); WSA_STEP_L(l2);
                        WSA_STEP_R(r1); WSA_STEP_R(r2);
                    }
#undef WSA_STEP_L
#undef WSA_STEP_R
                    pil = cur; psl = cur2;
                    if (any_hi) {
                        pih = qi > 0 ? hi_at(qi - 1) : 0u;
                        psh = hi_at(qs);
                    }
                }
                const uint32_t c_ = cbl[fl] + (uint32_t)ord;
                if (!WSA_PKT(8)) p.rec.ent[c_] = make_uint4((uint32_t)qi | ((uint32_t)qs << 8) | (wa & 0xffff0000u), pil, psl, pih | (psh << 8));
                if (!WSA_PKT(8)) p.rec.amp[c_] = qe; else if (qe == 0x12345u && pil == 77u && psl == 99u) p.rec.amp[c_] = qe + (uint32_t)qi + (uint32_t)qs + pih + psh;
                if (!(wa >> 24) && !WSA_PKT(16)) atomicMax(&mxk[fl], ((unsigned long long)qe << 32) | (unsigned long long)(((63u - (uint32_t)ord) << 8) | (uint32_t)cl));
            }
        }
        wsync();
        if (nwork < nlist) {
            const int rest = nlist - nwork;
            uint32_t a = 0, b = 0;
            if (lane < rest) { a = lstA[nwork + lane]; b = lstB[nwork + lane]; }
            wsync();
            if (lane < rest) { lstA[lane] = a; lstB[lane] = b; }
            nlist = rest;
            wsync();
        } else nlist = 0;
    };
    // append the candidates of the lanes in `cand` (their [ci, cs, cl]) to the list
    auto append = [&](bool cand, int ci, int cs, int cl, uint32_t last, int lo_valid, bool any_hi) __attribute__((always_inline)) {
        if (cand && n >= CAND_CAP) { too_many = true; cand = false; }      // a frame holds CAND_CAP candidates: all a spectrum of <= 128 bands can have
        const uint64_t m = __ballot(cand);
        if (m) {
            if (nlist + 64 > PK_LIST) flush(lo_valid, any_hi, false);
            if (cand) {
                const int pos = nlist + __popcll(m & lanemask_lt(lane));
                lstA[pos] = (uint32_t)ci | ((uint32_t)cs << 8) | ((uint32_t)cl << 16) | (last << 24);
                lstB[pos] = (uint32_t)lane | ((uint32_t)n << 8);
                n++;
            }
            nlist += __popcll(m);
        }
    };

    // ---- tile staging.  LPR lanes x 16 B cover one row's tile (128 bytes = one cache line at 32 bins), 64 / LPR rows per load instruction; the
    //      next tile is requested before the current one is walked.  Rows past the launch's last frame read as zeros (their lanes never emit).
    constexpr int LPR = PK_W / 4, RPI = 64 / LPR, NLD = LPR;
    uint4 nxt[NLD];
    auto fetch = [&](int t0) __attribute__((always_inline)) {
#pragma unroll
        for (int k = 0; k < NLD; k++) {
            const int r = RPI * k + lane / LPR, b = t0 + 4 * (lane % LPR);
            nxt[k] = make_uint4(0u, 0u, 0u, 0u);
            if ((uint32_t)r < nf && b < B) nxt[k] = *reinterpret_cast<const uint4*>(src + (uint64_t)r * (uint32_t)B + b);
        }
    };
    auto to_lds = [&](int t0) __attribute__((always_inline)) {
#pragma unroll
        for (int k = 0; k < NLD; k++) {
            const int r = RPI * k + lane / LPR;
            uint32_t* d = ringP + ((t0 + 4 * (lane % LPR)) & (PK_RB - 1)) * PK_RS + r;
            d[0] = nxt[k].x; d[PK_RS] = nxt[k].y; d[2 * PK_RS] = nxt[k].z; d[3 * PK_RS] = nxt[k].w;
        }
    };

    if (vec) fetch(0);
    bool any_hi = false;
    for (int t0 = 0; t0 < B; t0 += PK_W) {
        const int tw = min(PK_W, B - t0);
        // (the workgroup is one wave: its LDS accesses execute in order, so only the compiler has to be kept from moving them across —
        //  __syncthreads() would also drain vmcnt, i.e. wait for the loads issued just above it)
        wsync();
        if (vec) { to_lds(t0); if (t0 + PK_W < B) fetch(t0 + PK_W); }
        else for (int idx = lane; idx < 64 * tw; idx += 64) { const int r = idx / tw, q = idx - r * tw; ringP[((t0 + q) & (PK_RB - 1)) * PK_RS + r] = (uint32_t)r < nf ? src[(uint64_t)r * (uint32_t)B + t0 + q] : 0u; }
        wsync();
        // ---- pass 2: prefix sums and predicate masks of this lane's 32 bins
        uint32_t mR = 0, mF = 0, mG = 0, cm = 0;
        {
            uint32_t* colP = ringP + (t0 & (PK_RB - 1)) * PK_RS + lane;
            // by the letter (the first bins, where e[a-2] / e[a-3] do not exist yet: ref `(a<2||...)&&(a<3||...)`; partial tiles)
            auto step_slow = [&](int q) __attribute__((always_inline)) {
                const int a = t0 + q;
                const uint32_t ea = colP[q * PK_RS];
                WSA_ADD_CARRY(plo, cm, ea);
                colP[q * PK_RS] = plo;
                bool R = false, F = false, G = false;
                if (a == 0) e0 = ea;
                else {
                    R = ea > e1 && (a < 2 || ea > e2) && (a < 3 || ea > e3);
                    F = ea < e1 && (a < 2 || ea < e2) && (a < 3 || ea < e3);
                    G = ea > e1;
                }
                mR = (mR << 1) | (R ? 1u : 0u); mF = (mF << 1) | (F ? 1u : 0u); mG = (mG << 1) | (G ? 1u : 0u);
                e3 = e2; e2 = e1; e1 = ea;
            };
            auto step_fast = [&](int q) __attribute__((always_inline)) {
                const uint32_t ea = colP[q * PK_RS];
                WSA_ADD_CARRY(plo, cm, ea);
                colP[q * PK_RS] = plo;
                uint32_t hi3, lo3;
                asm("v_max3_u32 %0, %1, %2, %3" : "=v"(hi3) : "v"(e1), "v"(e2), "v"(e3));
                asm("v_min3_u32 %0, %1, %2, %3" : "=v"(lo3) : "v"(e1), "v"(e2), "v"(e3));
                WSA_PUSH_GT(mR, ea, hi3); WSA_PUSH_LT(mF, ea, lo3); WSA_PUSH_GT(mG, ea, e1);
                e3 = e2; e2 = e1; e1 = ea;
            };
            if (WSA_PKT(4)) {}
            else if (tw == PK_W && t0 > 0) {
#pragma unroll
                for (int q = 0; q < PK_W; q++) step_fast(q);
            } else if (tw == PK_W) {
                step_slow(0); step_slow(1); step_slow(2);
#pragma unroll
                for (int q = 3; q < PK_W; q++) step_fast(q);
            } else {
                for (int q = 0; q < tw; q++) step_slow(q);
            }
            const int sh = 32 - tw;
            mR = __builtin_bitreverse32(mR) >> sh; mF = __builtin_bitreverse32(mF) >> sh; mG = __builtin_bitreverse32(mG) >> sh;
            cm = __builtin_bitreverse32(cm) >> sh;
            const int h = (t0 / PK_W) & 1;
            cmw[h][lane] = cm; hib[h][lane] = phi;
            phi += (uint32_t)__popc(cm);
            any_hi = __ballot(phi != 0u) != 0ull;
        }
        // ---- pass 3: the state machine over this word's events, every lane in every step (selects, no branches)
        {
            const uint32_t mN = ~(mR | mF);
            uint32_t rem = tw == 32 ? ~0u : ((1u << tw) - 1u);         // bins of this word not yet visited
            if (t0 == 0) rem &= ~1u;                                   // the scan starts at bin 1
            if (!live || WSA_PKT(2)) rem = 0u;
            const int lo_valid = max(0, t0 - PK_W);
            while (__ballot(rem != 0u) != 0ull) {
                // idle or falling: on to the next rise — through the flat bins of a falling stretch, which end it at the third
                const bool actA = rem != 0u && u != 1, falling = actA && u == 2;
                const uint32_t rm = mR & rem, rm1 = rm - 1u;
                const uint32_t span = rem & rm1 & ~rm;                         // bins before the next rise (all of rem when there is none)
                const uint32_t nm = mN & span;
                const int tot = c + __popc(nm);
                const bool to = falling && tot >= 3;                           // c reaches 3 at the (3 - c)th flat bin: u = 0 there (ref `c>2&&(c=0,...,u=0)`)
                const uint32_t tmA = nm & (nm - 1u), tmB = tmA & (tmA - 1u);
                const uint32_t tm = c == 2 ? nm : (c == 1 ? tmA : tmB);
                const uint32_t tbit = tm & (0u - tm), tlow = tbit - 1u;
                const uint32_t fm = mF & (to ? span & tlow : span);             // the falls that still move s
                s = (falling && fm != 0u) ? t0 + 31 - __clz((int)fm) : s;
                const bool rise = actA && rm != 0u && !to;
                const bool cand = falling && (to || rise) && i <= l && l < s;
                append(cand, i, s, l, 0u, lo_valid, any_hi);
                const int r = t0 + __ffs((int)rm) - 1;
                i = rise ? r - 1 : i; l = rise ? r : l;
                c = falling ? (to ? 0 : tot) : c;
                rem = actA ? (to ? rem & ~(tbit | tlow) : (rise ? rem & ~(rm ^ rm1) : 0u)) : rem;
                u = actA ? (rise ? 1 : (to ? 0 : u)) : u;
                // rising: l follows every bin above its predecessor up to the next fall, which sets s
                const bool actB = rem != 0u && u == 1;
                const uint32_t fm2 = mF & rem, f21 = fm2 - 1u;
                const uint32_t gm = mG & rem & f21 & ~fm2;
                l = (actB && gm != 0u) ? t0 + 31 - __clz((int)gm) : l;
                const bool fall = actB && fm2 != 0u;
                s = fall ? t0 + __ffs((int)fm2) - 1 : s;
                u = fall ? 2 : u;
                rem = actB ? (fall ? rem & ~(fm2 ^ f21) : 0u) : rem;
            }
            // end of spectrum (ref @B26383): a peak still rising at the last bin is closed there
            if (t0 + PK_W >= B) {
                const bool last = live && B > 1 && u == 1;
                if (last) { s = B - 1; l = B - 1; }
                append(last && i < l, i, s, l, 1u, lo_valid, any_hi);
            }
            flush(lo_valid, any_hi, true);
        }
    }
    if (live) {
        const uint64_t g = (((uint64_t)phi << 32) | plo) - (uint64_t)e0;        // g = sum e[1..B-1]
        const unsigned long long k = mxk[lane];
        p.rec.hdr[slot] = make_uint4((uint32_t)g, (uint32_t)(g >> 32) | ((uint32_t)n << 8) | (((uint32_t)k & 0xffu) << 16), (uint32_t)(k >> 32), cbase);
    }
    if (__ballot(too_many) != 0ull && lane == 0) atomicOr(p.flags, 1u);
}
#undef WSA_PUSH_GT
#undef WSA_PUSH_LT
#undef WSA_ADD_CARRY


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
    // The direction / flat-run state machine does not visit bins: as in the lane-per-frame kernel it jumps from event to event with find-first-bit
    // on the masks — next rise, next fall, third flat bin of a falling stretch — here on 64-bit words in scalar registers with real branches
    // (~23 peak cycles per frame instead of 127 bin steps: the stream step's peak scan 31 -> 12 us).
    for (int t0 = 0; t0 < B; t0 += 64) {
        const int tw = min(64, B - t0);
        const uint64_t mR = t0 ? R1 : R0, mF = t0 ? F1 : F0, mG = t0 ? G1 : G0, mN = ~(mR | mF);
        uint64_t rem = tw == 64 ? ~0ull : ((1ull << tw) - 1ull);          // bins of this word not yet visited
        if (t0 == 0) rem &= ~1ull;                                        // the scan starts at bin 1
        while (rem != 0ull) {
            if (u != 1) {
                // idle or falling: on to the next rise — through the flat bins of a falling stretch, which end it at the third
                const bool falling = u == -1;
                const uint64_t rm = mR & rem, rm1 = rm - 1ull;
                const uint64_t span = rem & rm1 & ~rm;                     // bins before the next rise (all of rem when there is none)
                const uint64_t nm = mN & span;
                const int tot = c + __popcll(nm);
                const bool to = falling && tot >= 3;                       // c reaches 3 at the (3 - c)th flat bin: u = 0 there (ref `c>2&&(c=0,...,u=0)`)
                uint64_t tm = nm;
                if (c < 2) tm &= tm - 1ull;
                if (c < 1) tm &= tm - 1ull;
                const uint64_t tbit = tm & (0ull - tm), tlow = tbit - 1ull;
                const uint64_t fm = mF & (to ? span & tlow : span);       // the falls that still move s
                if (falling && fm != 0ull) s = t0 + 63 - __clzll((long long)fm);
                const bool rise = rm != 0ull && !to;
                if (falling && (to || rise) && i <= l && l < s) emit(i, s, l, 0u);
                if (rise) { const int r = t0 + __ffsll((long long)rm) - 1; i = r - 1; l = r; }
                if (falling) c = to ? 0 : tot;
                rem = to ? rem & ~(tbit | tlow) : (rise ? rem & ~(rm ^ rm1) : 0ull);
                u = rise ? 1 : (to ? 0 : u);
            }
            if (rem != 0ull && u == 1) {
                // rising: l follows every bin above its predecessor up to the next fall, which sets s
                const uint64_t fm2 = mF & rem, f21 = fm2 - 1ull;
                const uint64_t gm = mG & rem & f21 & ~fm2;
                if (gm != 0ull) l = t0 + 63 - __clzll((long long)gm);
                if (fm2 != 0ull) { s = t0 + __ffsll((long long)fm2) - 1; u = -1; rem &= ~(fm2 ^ f21); }
                else rem = 0ull;
            }
        }
    }
    // end of spectrum (ref @B26383): a peak still rising at the last bin is closed there
    if (B > 1 && u == 1) { s = B - 1; l = B - 1; if (i < l && l <= s) emit(i, s, l, 1u); }
    if (lane == 0) {
        const uint64_t g = (uint64_t)(prefix_at(B - 1) - (double)amp_at(0));        // g = sum e[1..B-1]
        p.rec.hdr[slot] = make_uint4((uint32_t)g, (uint32_t)(g >> 32) | ((uint32_t)n << 8) | (mx_bin << 16), mx_amp, cbase);
        if (too_many) atomicOr(p.flags, 1u);
    }
}

// mode 0: by size (below); 1: one lane per frame; 2: one wave per frame; 3 / 4: one lane per frame in rounds of 16 / 32 bins (tests: wsa_debug_peaks)
void launch_peaks_mode(const PkParams& p0, int mode, hipStream_t s) {
    if (p0.total_frames == 0) return;
    PkParams p = p0;
    if (mode == 3 || mode == 4) { p.round_bins = mode == 3 ? 16 : 32; mode = 1; }
    // few frames (stream steps): one wave per frame instead of one lane per frame (WSA_PEAKS_LANES=1 keeps the lane kernel: test hook)
    const bool wave = mode == 2 || (mode == 0 && p.total_frames <= 4096u && !p.lanes_only);
    // co-residency experiments (Tuning::peaks_wpc): dynamic LDS on top of the kernel's static block so that only `wpc` waves fit a CU
    const bool w16 = p.round_bins == 16;
    const size_t own = w16 ? 11520 : 20480;
    size_t pad = 0;
    if (p.wpc >= 1 && p.wpc < 14) { const size_t per = ((size_t)163840 / (size_t)(p.wpc + 1) + 256) & ~(size_t)255; pad = per > own ? per - own : 0; }
    if (wave && p.bands <= 128) hipLaunchKernelGGL(peaks_wave_kernel, dim3(p.total_frames), dim3(64), 0, s, p);
    else if (w16) hipLaunchKernelGGL(peaks_kernel<16>, dim3((p.total_frames + 63) / 64), dim3(64), pad, s, p);
    else hipLaunchKernelGGL(peaks_kernel<32>, dim3((p.total_frames + 63) / 64), dim3(64), pad, s, p);
}
void launch_peaks(const PkParams& p, hipStream_t s) { launch_peaks_mode(p, 0, s); }

}  // namespace wsa
