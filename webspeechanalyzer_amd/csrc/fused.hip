// fused.hip — K1 + K1b in one launch for the 1024-point geometry (BASELINE configs 2 - 4): PCM -> Hann -> FFT -> mel -> u32
// frame -> peak candidates, without the u32 spectra ever leaving the chip.
//
// Stands in for the reference's "spectrum-processor" worklet (source not in the reference tree, ref dist/main.js:2 @B6480,
// output consumed @B8568; arithmetic = specification FE-1, DESIGN.md, bit-exact with oracle/frontend.c) followed by the
// candidate half of the frame loop D() (ref @B25717, scan @B25827; see peaks.hip for what is and is not decided here).
//
// One workgroup of 12 wavefronts per CU walks a contiguous range of the batch's frames in ROUNDS of 64 frames:
//   * eleven waves transform the round's frames, one wavefront per frame (the mapping of frontend.hip, fe_kernel_r8), and
//     leave the u32 frame as a row of the round's LDS buffer (row stride 129 words);
//   * the twelfth wave meanwhile scans the PREVIOUS round's rows, one LANE per frame: the rising / falling / flat state
//     machine of the reference (direction and flat counter held as lane masks in scalar registers), the running prefix sum of
//     the row written back over the row itself (e[x] = P[x] - P[x-1] stays recoverable, wrap-arounds of the low word are
//     noted in a 128-bit mask per frame), raw candidates [i, s, l] into a per-frame list; then, one lane per CANDIDATE, the
//     /10 shoulder shrink and the exact prefix sums at the shrunk shoulders, and the records go to HBM packed back to back.
//   The scanning role rotates (round r: wave r mod 12), rows are double buffered, one workgroup barrier per round.
// HBM traffic per frame: 1600 B of PCM in, 16 B header + 20 B per candidate out (~300 B).
#include "fe_common.hpp"
#include <cstdlib>

namespace wsa {

constexpr int FW = 12;                     // wavefronts per workgroup
constexpr int RND = 64;                    // frames per round
constexpr int ROWS = 129;                  // row stride in words (conflict-free for lane-per-row and lane-per-band access)
constexpr int LCAP = 32;                   // raw candidates per frame kept in LDS; later ones go through the global overflow list

struct RoundBuf { uint32_t* rows; uint32_t* list; uint32_t* wrap; };
constexpr size_t RB_ROWS = (size_t)RND * ROWS * 4, RB_LIST = (size_t)RND * LCAP * 4, RB_WRAP = (size_t)RND * 4 * 4;
constexpr size_t RB_BYTES = RB_ROWS + RB_LIST + RB_WRAP;

size_t fused_lds_bytes(int bands) { return (size_t)((bands + 3) & ~3) * 4 + (size_t)FW * XBUF * 8 + 2 * RB_BYTES + 64; }

template <int AZ>
__global__ __launch_bounds__(FW * 64) void fe_scan_kernel(FeParams p, FusedParams q) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // ---- LDS carve-up: emphasis table | per wave X (the power spectrum P of a frame reuses its wave's X) | two round buffers | cursor
    float* s_emph = reinterpret_cast<float*>(smem);
    char* base = smem + (size_t)((p.bands + 3) & ~3) * 4;
    v2f* X = reinterpret_cast<v2f*>(base) + (size_t)wave * XBUF;
    float* P = reinterpret_cast<float*>(X);
    char* rb0 = base + (size_t)FW * XBUF * 8;
    auto round_buf = [&](int r) __attribute__((always_inline)) -> RoundBuf {        // rounds alternate between the two buffers
        char* b = rb0 + (size_t)(r & 1) * RB_BYTES;
        RoundBuf o; o.rows = reinterpret_cast<uint32_t*>(b); o.list = reinterpret_cast<uint32_t*>(b + RB_ROWS); o.wrap = reinterpret_cast<uint32_t*>(b + RB_ROWS + RB_LIST);
        return o;
    };
    uint32_t* s_cursor = reinterpret_cast<uint32_t*>(rb0 + 2 * RB_BYTES);
    for (int i = threadIdx.x; i < p.bands; i += FW * 64) s_emph[i] = p.emph[i];
    if (threadIdx.x == 0) *s_cursor = 0u;

    // ---- this workgroup's frames [F0, F1) of the batch's frame index space, R rounds
    const uint32_t F0 = min(q.total_frames, blockIdx.x * q.frames_per_block), F1 = min(q.total_frames, F0 + q.frames_per_block);
    const int R = (int)((F1 - F0 + RND - 1) / RND);
    if (R == 0) return;
    const uint32_t region0 = F0 * (uint32_t)CAND_CAP;           // the workgroup's candidates are packed from here on

    // ---- loop-invariant per-lane constants of the transform (registers; see fe_kernel_r8)
    v2f tw1[8], tw2[8];
#pragma unroll
    for (int k = 1; k < 8; k++) { tw1[k] = to_v2f(p.tw_n2[lane * k]); tw2[k] = to_v2f(p.tw_64[(lane & 7) * k]); }
    v2f wn[AZ];
#pragma unroll
    for (int a = 0; a < AZ; a++) {
        const int n = 2 * (64 * a + lane);
        wn[a].x = n < p.win ? p.window[n] : 0.0f;
        wn[a].y = n + 1 < p.win ? p.window[n + 1] : 0.0f;
    }
    v2f ss; ss.x = 0.70710678118654752440f; ss.y = 0.70710678118654752440f;
    const int hi3 = lane >> 3, lo3 = lane & 7;
    const int k0 = hi3 + 8 * lo3;
    const int k0p = (64 - k0) & 63;
    const int partner = ((k0p & 7) << 3) | (k0p >> 3);
    const int nrow = p.kmax / 64 + 1;
    v2f tws[9];
#pragma unroll
    for (int c = 0; c < 9; c++) {
        const int k = k0 + 64 * c;
        tws[c] = to_v2f((k <= p.kmax) ? p.tw_nfft[k] : make_float2(0.f, 0.f));
    }
    float mw[2][MELW]; int mk[2];
#pragma unroll
    for (int h = 0; h < 2; h++) {
        const int m = lane + 64 * h;
        mk[h] = 0;
#pragma unroll
        for (int j = 0; j < MELW; j++) mw[h][j] = 0.f;
        if (m < p.bands) {
            mk[h] = p.mel_k0[m];
            const int cnt = p.mel_cnt[m], off = p.mel_off[m];
#pragma unroll
            for (int j = 0; j < MELW; j++) if (j < cnt) mw[h][j] = p.mel_w[off + j];       // the host launches this kernel only when every band has <= MELW taps
        }
    }
    const int pmax = p.kmax;
    int ld_idx[AZ]; bool ld_v0[AZ], ld_v1[AZ], ld_odd[AZ];
#pragma unroll
    for (int a = 0; a < AZ; a++) {
        const int n = 2 * (64 * a + lane);
        ld_idx[a] = min(n, p.win - 2);
        ld_v0[a] = n < p.win; ld_v1[a] = n + 1 < p.win; ld_odd[a] = n == p.win - 1;
    }
    // ---- frame -> clip: a wave's frames come in ascending order, so the clip only ever advances (uniform scalars)
    uint32_t clip = 0;
    {   // clip of the workgroup's first frame: the last clip whose offset is <= F0
        uint32_t lo = 0, hi = q.n_clips;                // frame_off[lo] <= F0 < frame_off[hi] (frame_off[n_clips] = total > F0)
        while (hi - lo > 1) { const uint32_t mid = (lo + hi) >> 1; if (p.frame_off[mid] <= F0) lo = mid; else hi = mid; }
        clip = lo;
    }
    uint32_t clip_hi = p.frame_off[clip + 1];
    auto frame_pcm = [&](uint32_t gf) __attribute__((always_inline)) -> const float* {
        while (gf >= clip_hi) { clip++; clip_hi = p.frame_off[clip + 1]; }
        const uint32_t f = gf - p.frame_off[clip];
        return p.pcm + (uint64_t)clip * p.clip_stride + (uint64_t)f * (uint32_t)p.hop;
    };
    auto load_pcm = [&](const float* fr, v2f (&x)[AZ]) __attribute__((always_inline)) {
#pragma unroll
        for (int a = 0; a < AZ; a++) {
            const pcm2 t = *reinterpret_cast<const pcm2*>(fr + ld_idx[a]);
            x[a].x = t.x; x[a].y = t.y;
        }
    };
    __syncthreads();

    v2f xin[AZ];
    uint32_t have = 0xffffffffu;               // frame whose samples sit in xin
    for (int r = 0; r <= R; r++) {
        const int scanner = r % FW;
        if (wave != scanner && r < R) {
            // =============================== transform the frames j = k, k + 11, ... of round r
            const uint32_t rf0 = F0 + (uint32_t)r * RND;
            const int nr = (int)min((uint32_t)RND, F1 - rf0);
            const int k = (wave - scanner - 1 + FW) % FW;
            uint32_t* rows = round_buf(r).rows;
            for (int j = k; j < nr; j += FW - 1) {
                const uint32_t gf = rf0 + (uint32_t)j;
                if (have != gf) load_pcm(frame_pcm(gf), xin);
                // ---- window (F1-F3)
                v2f v[8];
#pragma unroll
                for (int a = 0; a < 8; a++) { v[a].x = 0.f; v[a].y = 0.f; }
#pragma unroll
                for (int a = 0; a < AZ; a++) {
                    v2f x;
                    x.x = ld_v0[a] ? (ld_odd[a] ? xin[a].y : xin[a].x) : 0.f;
                    x.y = ld_v1[a] ? xin[a].y : 0.f;
                    v[a] = pk_mul(x, wn[a]);
                }
                // the wave's next frame: in this round, or (if it does not scan then) its first one of the next round
                {
                    uint32_t nx = 0xffffffffu;
                    if (j + FW - 1 < nr) nx = gf + (uint32_t)(FW - 1);
                    else if (r + 1 < R) {
                        const int sc2 = (r + 1) % FW;
                        if (wave != sc2) {
                            const uint32_t rf2 = rf0 + RND;
                            const uint32_t k2 = (uint32_t)((wave - sc2 - 1 + FW) % FW);
                            if (rf2 + k2 < F1) nx = rf2 + k2;
                        }
                    }
                    have = nx;
                    if (nx != 0xffffffffu) load_pcm(frame_pcm(nx), xin);
                }
                // ---- pass 1: radix 8 over a, twiddle W_512^{m a'}
                radix8_pk<AZ>(v, ss);
#pragma unroll
                for (int c = 1; c < 8; c++) v[c] = pk_cmul(v[c], tw1[c]);
#pragma unroll
                for (int c = 0; c < 8; c++) X[c * XROW + lane] = v[c];
                wave_lds_sync();
#pragma unroll
                for (int b = 0; b < 8; b++) v[b] = X[hi3 * XROW + 8 * b + lo3];
                wave_lds_sync();
                // ---- pass 2: radix 8 over b, twiddle W_64^{c b'}
                radix8_pk<8>(v, ss);
#pragma unroll
                for (int c = 1; c < 8; c++) v[c] = pk_cmul(v[c], tw2[c]);
#pragma unroll
                for (int c = 0; c < 8; c++) X[hi3 * XROW + c * 9 + lo3] = v[c];
                wave_lds_sync();
#pragma unroll
                for (int c = 0; c < 8; c++) v[c] = X[hi3 * XROW + lo3 * 9 + c];
                wave_lds_sync();
                // ---- pass 3: radix 8 over c -> v[c'] = Z[k0 + 64 c']
                radix8_pk<8>(v, ss);
                // ---- real-FFT split + 4x power (F4); P takes the place of X
#pragma unroll
                for (int c = 0; c < 9; c++) {
                    if (c < nrow) {
                        v2f zb;
                        if (c < 8) {
                            const v2f src = v[7 - c];
                            zb.x = __shfl(src.x, partner, 64);
                            zb.y = __shfl(src.y, partner, 64);
                        } else { zb.x = 0.f; zb.y = 0.f; }
                        if (k0 == 0) zb = v[(8 - c) & 7];
                        const v2f za = v[c & 7];
                        const v2f e = pk_add_conj(za, zb), o = pk_sub_conj(za, zb);
                        const v2f t = pk_cmul(o, tws[c]);
                        const v2f xx = pk_add_mi(e, t);
                        const int kk = k0 + 64 * c;
                        if (kk <= p.kmax) P[kk] = __builtin_fmaf(xx.x, xx.x, xx.y * xx.y);
                    }
                }
                wave_lds_sync();
                // ---- bands (F5-F8) -> row j of the round buffer (and the spectra array when the caller wants the frames)
                uint32_t* row = rows + (size_t)j * ROWS;
#pragma unroll
                for (int h = 0; h < 2; h++) {
                    const int m = lane + 64 * h;
                    float pv[MELW];
#pragma unroll
                    for (int t = 0; t < MELW; t++) { const int kk = mk[h] + t; pv[t] = P[kk <= pmax ? kk : pmax]; }
                    float e = 0.f;
#pragma unroll
                    for (int t = 0; t < MELW; t++) e = __builtin_fmaf(mw[h][t], pv[t], e);
                    e = e * s_emph[m < p.bands ? m : 0];
                    e = e * p.gain;
                    const uint32_t u = to_u32(e);
                    if (m < p.bands) { row[m] = u; if (p.spec) p.spec[(uint64_t)gf * (uint32_t)p.bands + m] = u; }
                }
                wave_lds_sync();
            }
        } else if (wave == scanner && r > 0) {
            // =============================== scan round r - 1: lane = frame
            const uint32_t rf0 = F0 + (uint32_t)(r - 1) * RND;
            const int nr = (int)min((uint32_t)RND, F1 - rf0);
            const RoundBuf B = round_buf(r - 1);
            const int Bn = p.bands;
            const bool live = lane < nr;
            uint32_t* row = B.rows + (size_t)lane * ROWS;
            uint32_t* lst = B.list + (size_t)lane * LCAP;
            const uint32_t gf = rf0 + (uint32_t)lane;
            const int lcap = q.lcap;                      // LCAP, or less under the WSA_FUSED_LCAP test hook (exercises the overflow list)
            uint32_t* glst = q.glist + (uint64_t)gf * CAND_CAP;
            // the reference's scan (ref @B25827; restated in oracle/backend.c): direction u in {1, -1, 0} as the lane masks
            // u1 / um, flat counter c in {0, 1, 2} as c1 / c2
            bool u1 = false, um = false, c1 = false, c2 = false;
            int ci = 0, cl = 0, cs = 0, n = 0;
            uint32_t e1 = 0, e2 = 0, e3 = 0, e_l = 0, mx_amp = 0, mx_bin = 0;
            uint32_t plo = 0, phi = 0, wcur = 0, e0 = 0;
            auto emit = [&](uint32_t last) __attribute__((always_inline)) {
                const uint32_t w = (uint32_t)ci | ((uint32_t)cs << 8) | ((uint32_t)cl << 16) | (last << 24);
                if (n < lcap) lst[n] = w; else if (n < CAND_CAP) __hip_atomic_store(&glst[n - lcap], w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (!last && e_l > mx_amp) { mx_amp = e_l; mx_bin = (uint32_t)cl; }
                n++;
            };
            if (live) {
                e0 = row[0]; plo = e0; e1 = e0;                    // P[0] = e[0] stays in place
#pragma unroll 1
                for (int a0 = 0; a0 < Bn; a0 += 32) {
                    wcur = 0;
                    const int a_end = min(a0 + 32, Bn);
#pragma unroll 4
                    for (int a = max(a0, 1); a < a_end; a++) {
                        const uint32_t ea = row[a];
                        const uint32_t np = plo + ea;
                        const bool carry = np < ea;
                        plo = np; phi += carry ? 1u : 0u;
                        wcur |= (carry ? 1u : 0u) << (a & 31);
                        row[a] = np;
                        const bool g2 = a < 2, g3 = a < 3;
                        const bool rise = ea > e1 && (g2 || ea > e2) && (g3 || ea > e3);
                        const bool fall = ea < e1 && (g2 || ea < e2) && (g3 || ea < e3);
                        const bool creep = ea > e1;
                        const bool flat = !rise && !fall && um;
                        const bool trig = flat && c2;
                        if (((rise && um) || trig) && ci <= cl && cl < cs) emit(0u);
                        const bool nc1 = flat ? (!c1 && !c2) : c1, nc2 = flat ? c1 : c2;
                        c1 = nc1; c2 = nc2;
                        if (rise && !u1) ci = a - 1;
                        if (rise || (!fall && u1 && creep)) { cl = a; e_l = ea; }
                        const bool set_s = fall && (u1 || um);
                        if (set_s) cs = a;
                        const bool nu1 = rise || (u1 && !set_s), num = !rise && (set_s || (um && !trig));
                        u1 = nu1; um = num;
                        e3 = e2; e2 = e1; e1 = ea;
                    }
                    B.wrap[lane * 4 + (a0 >> 5)] = wcur;
                }
                // end of spectrum (ref @B26383): a peak still rising at the last bin is closed there
                if (Bn > 1 && u1) { cs = Bn - 1; cl = Bn - 1; e_l = e1; if (ci < cl && cl <= cs) emit(1u); }
            }
            if (n > CAND_CAP) { n = CAND_CAP; atomicOr(q.flags, 1u); }
            // ---- where the round's candidates go: packed behind the workgroup's earlier ones
            const uint32_t incl = wave_incl_scan_u32((uint32_t)n);
            const uint32_t excl = incl - (uint32_t)n;
            const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
            const uint32_t cur = *s_cursor;
            const uint32_t cbase = region0 + cur + excl;
            if (live) {
                const uint64_t g = (((uint64_t)phi << 32) | plo) - e0;                  // g = sum e[1..B-1]
                q.rec.hdr[gf] = make_uint4((uint32_t)g, (uint32_t)(g >> 32) | ((uint32_t)n << 8) | (mx_bin << 16), mx_amp, cbase);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // overflow-list stores of this wave before its own loads of them below
            wave_lds_sync();
            // ---- one lane per candidate: shoulder shrink (ref: `e[i] < e[l] / 10`), exact prefix sums at the shrunk shoulders
            for (uint32_t c0 = 0; c0 < total; c0 += 64) {
                const uint32_t qi_ = c0 + (uint32_t)lane;
                const bool on = qi_ < total;
                // frame of candidate qi_: the last j with excl[j] <= qi_ (branch-free binary search over the lanes' offsets)
                int j = 0;
#pragma unroll
                for (int step = 32; step >= 1; step >>= 1) {
                    const int t = j + step;
                    const uint32_t ex = (uint32_t)__shfl((int)excl, t & 63, 64);
                    if (t < 64 && ex <= qi_) j = t;
                }
                // (frames without candidates share their offset with the next frame: the search lands on the LAST such j,
                //  which is the frame that owns the candidate, because empty frames have excl[j] == excl[j+1] <= qi_ too — step back over
                //  them is not needed: the last j with excl[j] <= qi_ always has incl[j] > qi_)
                const uint32_t exj = (uint32_t)__shfl((int)excl, j, 64);
                const uint32_t kq = qi_ - exj;
                if (on) {
                    const uint32_t* rw = B.rows + (size_t)j * ROWS;
                    const uint32_t w = kq < (uint32_t)lcap ? B.list[(size_t)j * LCAP + kq]
                                                : __hip_atomic_load(&q.glist[(uint64_t)(rf0 + (uint32_t)j) * CAND_CAP + (kq - (uint32_t)lcap)], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    int ci2 = (int)(w & 0xff), cs2 = (int)((w >> 8) & 0xff);
                    const int cl2 = (int)((w >> 16) & 0xff);
                    auto ev = [&](int x) __attribute__((always_inline)) -> uint32_t { return x > 0 ? rw[x] - rw[x - 1] : rw[0]; };
                    const uint32_t amp = ev(cl2);
                    const uint32_t thr = amp / 10u + (amp % 10u != 0u ? 1u : 0u);         // e[x] < e[l] / 10  <=>  e[x] < ceil(e[l] / 10)
                    while (ci2 < cl2 && ev(ci2) < thr) ci2++;
                    while (cs2 > cl2 && ev(cs2) < thr) cs2--;
                    const uint32_t* wr = B.wrap + (size_t)j * 4;
                    auto hi_at = [&](int x) __attribute__((always_inline)) -> uint32_t {     // number of low-word wrap-arounds in bins 0..x
                        uint32_t h = 0;
#pragma unroll
                        for (int t = 0; t < 4; t++) {
                            const uint32_t m = wr[t];
                            const int lim = x - 32 * t;                                        // bits 0..lim of this word count
                            h += lim >= 31 ? __popc(m) : (lim >= 0 ? __popc(m & (0xffffffffu >> (31 - lim))) : 0);
                        }
                        return h;
                    };
                    uint32_t lo_w = 0, lo_h = 0;
                    if (ci2 > 0) { lo_w = rw[ci2 - 1]; lo_h = hi_at(ci2 - 1); }
                    const uint32_t hi_w = rw[cs2], hi_h = hi_at(cs2);
                    const uint32_t c = region0 + cur + qi_;
                    q.rec.amp[c] = amp;
                    q.rec.ent[c] = make_uint4((uint32_t)ci2 | ((uint32_t)cs2 << 8) | ((uint32_t)cl2 << 16) | (w & 0x01000000u), lo_w, hi_w, lo_h | (hi_h << 8));
                }
            }
            if (lane == 0) *s_cursor = cur + total;
        }
        __syncthreads();
    }
}

bool fused_supported(const FeParams& p, int R, int three, const std::vector<int32_t>& mel_cnt) {
    if (three || R != 8 || p.spec_type != 1 || p.bands > 128 || p.bands < 2 || p.pcm_off) return false;
    for (int32_t c : mel_cnt) if (c > MELW) return false;
    return true;
}

void launch_fused(const FeParams& p, const FusedParams& q0, int n_cu, hipStream_t s) {
    if (q0.total_frames == 0) return;
    FusedParams q = q0;
    q.lcap = LCAP;
    if (const char* e = std::getenv("WSA_FUSED_LCAP")) { const int v = std::atoi(e); if (v >= 0 && v < LCAP) q.lcap = v; }      // test hook
    const uint32_t rounds = (q.total_frames + RND - 1) / RND;
    uint32_t grid = (uint32_t)(n_cu > 0 ? n_cu : 256);
    if (grid > rounds) grid = rounds;
    const uint32_t per = (rounds + grid - 1) / grid;
    q.frames_per_block = per * RND;
    grid = (rounds + per - 1) / per;
    const size_t lds = fused_lds_bytes(p.bands);
    const int az = (p.win + 127) / 128;
    if (az <= 2) hipLaunchKernelGGL(fe_scan_kernel<2>, dim3(grid), dim3(FW * 64), lds, s, p, q);
    else if (az <= 4) hipLaunchKernelGGL(fe_scan_kernel<4>, dim3(grid), dim3(FW * 64), lds, s, p, q);
    else hipLaunchKernelGGL(fe_scan_kernel<8>, dim3(grid), dim3(FW * 64), lds, s, p, q);
}

}  // namespace wsa
