// wsa_oracle.js — ORACLE in plain JavaScript (test infrastructure / CPU baseline only; never loaded by
// the product in webspeechanalyzer_amd/).
//
// The whole path under Node with no GPU: PCM -> FE-1 front end (our specification, see
// oracle/frontend.c and DESIGN.md §3; Math.fround reproduces every fp32 rounding, Math.fround of a
// double fma is NOT a fused fmaf, so fmaf() below uses the exact-product trick) -> back end restating
// formantanalyzer@1.1.6 (ref /root/reference/dist/main.js:2, inner modules 3, 4, 0 — same citations as
// oracle/backend.c, which this file mirrors function by function).  Checked in tests/ against the C
// oracle (u32 frames bit-exact) and against the reference fixtures (indices / features bit-exact).
// Math.log10 / Math.pow are the engine's own here — this IS the reference's arithmetic under Node.
'use strict';

const fr = Math.fround;

// ---- exact fp32 fused multiply-add: a*b is exact in double (24+24 bits), the sum with c is rounded
// once to double (53 bits) and once more to float.  Double rounding can differ from a true fmaf only
// when the double result sits exactly on a float rounding boundary; split the error term to fix it.
function fmaf(a, b, c) {
  const p = a * b;                 // exact
  const s = p + c;                 // rounded to double
  // error of the double addition (TwoSum), exact
  const bb = s - p;
  const err = (p - (s - bb)) + (c - bb);
  let r = fr(s);
  if (err !== 0 && r === s) return r;          // s is itself a float: inexact tail only matters on ties
  if (err !== 0) {
    // is s exactly halfway between two adjacent floats?  then the tail decides the direction
    const lo = r < s ? r : prevFloat(r), hi = nextFloat(lo);
    if (s - lo === hi - s) r = err > 0 ? hi : lo;
  }
  return r;
}
const f32 = new Float32Array(1), u32v = new Uint32Array(f32.buffer);
function nextFloat(x) { f32[0] = x; if (x >= 0) u32v[0]++; else u32v[0]--; return f32[0]; }
function prevFloat(x) { f32[0] = x; if (x > 0) u32v[0]--; else if (x < 0) u32v[0]++; else { u32v[0] = 0x80000001; } return f32[0]; }

// =====================================================================================================
// Front end FE-1
// =====================================================================================================
function twiddle(j, N) {
  j %= N;
  const q = Math.floor(j * 8 / N), r8 = j * 8 - q * N, h = 0.70710678118654752440;
  let c, s;
  if (r8 === 0) {
    c = [1, h, 0, -h, -1, -h, 0, h][q]; s = [0, h, 1, h, 0, -h, -1, -h][q];
  } else {
    const num = (q & 1) ? N - r8 : r8, t = num / N * (Math.PI / 4), ca = Math.cos(t), sa = Math.sin(t);
    switch (q) {
      case 0: c = ca; s = sa; break; case 1: c = sa; s = ca; break; case 2: c = -sa; s = ca; break;
      case 3: c = -ca; s = sa; break; case 4: c = -ca; s = -sa; break; case 5: c = -sa; s = -ca; break;
      case 6: c = sa; s = -ca; break; default: c = ca; s = -sa;
    }
  }
  return [fr(c), fr(-s)];
}
const melOf = (f) => 2595.0 * Math.log10(1.0 + f / 700.0);
const hzOf = (m) => 700.0 * (Math.pow(10.0, m / 2595.0) - 1.0);

class FrontEnd {
  constructor(cfg) {            // cfg: fs, spec_type, f_min, f_max, N_fft_bins, N_mel_bins, window_width, window_step, pre_norm_gain, high_f_emph
    this.cfg = cfg;
    const fs = cfg.fs;
    this.win = Math.floor(fs * cfg.window_width / 1000 + 0.5);
    this.hop = Math.floor(fs * cfg.window_step / 1000 + 0.5);
    let need = Math.ceil(fs * cfg.N_fft_bins / cfg.f_max);
    if (this.win > need) need = this.win;
    let n = 256;                                   // FE-1 F2: the smallest of {2^k, 3 * 2^k} >= need, >= 256
    for (;;) { if (n >= need) break; if (n / 2 * 3 >= need) { n = n / 2 * 3; break; } n <<= 1; }
    this.nfft = n; this.n2 = n >> 1;
    this.three = this.n2 % 3 === 0; this.M = this.three ? this.n2 / 3 : this.n2; this.R = this.M / 64;
    this.kmax = Math.min(this.n2, Math.floor(cfg.f_max * n / fs));
    this.bands = cfg.spec_type === 1 ? cfg.N_mel_bins : cfg.N_fft_bins;
    this.window = new Float32Array(this.win);
    for (let i = 0; i < this.win; i++) this.window[i] = 0.5 - 0.5 * Math.cos(2.0 * Math.PI * i / this.win);
    const mk = (N, count) => { const t = new Float32Array(2 * count); for (let j = 0; j < count; j++) { const w = twiddle(j, N); t[2 * j] = w[0]; t[2 * j + 1] = w[1]; } return t; };
    this.tw_n2 = mk(this.n2, this.n2); this.tw_m = this.three ? mk(this.M, this.M) : this.tw_n2; this.tw_64 = mk(64, 64); this.tw_nfft = mk(this.nfft, this.kmax + 1);
    this.gain = fr(cfg.pre_norm_gain);
    this.emph = new Float32Array(this.bands);
    for (let m = 0; m < this.bands; m++) this.emph[m] = 1.0 + m * cfg.high_f_emph;
    this.bins_hz = new Float64Array(this.bands);
    if (cfg.spec_type === 1) {
      const M = this.bands, df = fs / n, mlo = melOf(cfg.f_min), mhi = melOf(cfg.f_max);
      const pts = []; for (let j = 0; j < M + 2; j++) pts.push(hzOf(mlo + (mhi - mlo) * j / (M + 1)));
      this.mel_k0 = new Int32Array(M); this.mel_cnt = new Int32Array(M); this.mel_off = new Int32Array(M);
      const w = [];
      for (let m = 0; m < M; m++) {
        const lo = pts[m], ce = pts[m + 1], hi = pts[m + 2];
        this.bins_hz[m] = ce;
        let k0 = -1, cnt = 0; const off = w.length;
        for (let k = 0; k <= this.kmax; k++) {
          const fk = k * df; let wt = 0;
          if (fk > lo && fk < hi) { const up = (fk - lo) / (ce - lo), dn = (hi - fk) / (hi - ce); wt = up < dn ? up : dn; }
          if (wt > 0) { if (k0 < 0) k0 = k; while (w.length < off + (k - k0) + 1) w.push(0); w[off + (k - k0)] = fr(fr(wt) * 0.25); cnt = k - k0 + 1; }
        }
        if (k0 < 0) {
          const pos = ce / df; k0 = Math.floor(pos); let f = pos - k0;
          if (k0 >= this.kmax) { k0 = this.kmax; f = 0; }
          w.push(fr(fr(1.0 - f) * 0.25)); cnt = 1;
          if (f > 0) { w.push(fr(fr(f) * 0.25)); cnt = 2; }
        }
        this.mel_k0[m] = k0; this.mel_cnt[m] = cnt; this.mel_off[m] = off;
      }
      this.mel_w = Float32Array.from(w);
    } else for (let m = 0; m < this.bands; m++) this.bins_hz[m] = m * fs / n;
    this.zr = new Float32Array(this.n2); this.zi = new Float32Array(this.n2);
    this.vr = new Float32Array(64); this.vi = new Float32Array(64);
    this.P = new Float32Array(this.kmax + 2);
  }
  n_frames(n) { return n < this.win ? 0 : Math.floor((n - this.win) / this.hop) + 1; }
  butterfly(r) {
    const vr = this.vr, vi = this.vi, tw = this.tw_64, s = fr(0.70710678118654752440);
    for (let h = r >> 1; h >= 1; h >>= 1) for (let blk = 0; blk < r; blk += 2 * h) for (let j = 0; j < h; j++) {
      const a = blk + j, b = a + h, ur = vr[a], ui = vi[a], wr = vr[b], wi = vi[b];
      vr[a] = ur + wr; vi[a] = ui + wi;
      const tr = fr(ur - wr), ti = fr(ui - wi);
      if (j === 0) { vr[b] = tr; vi[b] = ti; }
      else if (2 * j === h) { vr[b] = ti; vi[b] = -tr; }
      else if (4 * j === h) { vr[b] = fr(s * fr(tr + ti)); vi[b] = fr(s * fr(ti - tr)); }
      else if (4 * j === 3 * h) { vr[b] = fr(s * fr(ti - tr)); vi[b] = -fr(s * fr(tr + ti)); }
      else { const k = 2 * (j * 32 / h); vr[b] = fmaf(-ti, tw[k + 1], fr(tr * tw[k])); vi[b] = fmaf(ti, tw[k], fr(tr * tw[k + 1])); }
    }
  }
  dif_pass(base, N, r, tw) {
    const M = N / r, p = Math.round(Math.log2(r)), zr = this.zr, zi = this.zi, vr = this.vr, vi = this.vi;
    const rev = (x) => { let o = 0; for (let i = 0; i < p; i++) o |= ((x >> i) & 1) << (p - 1 - i); return o; };
    const yr = new Float32Array(r), yi = new Float32Array(r);
    for (let m = 0; m < M; m++) {
      for (let a = 0; a < r; a++) { vr[a] = zr[base + a * M + m]; vi[a] = zi[base + a * M + m]; }
      this.butterfly(r);
      for (let k = 0; k < r; k++) { yr[k] = vr[rev(k)]; yi[k] = vi[rev(k)]; }
      for (let k = 0; k < r; k++) {
        let a = yr[k], b = yi[k];
        if (tw && k > 0 && m > 0) { const wr = tw[2 * m * k], wi = tw[2 * m * k + 1]; const na = fmaf(-b, wi, fr(a * wr)); b = fmaf(b, wr, fr(a * wi)); a = na; }
        zr[base + k * M + m] = a; zi[base + k * M + m] = b;
      }
    }
  }
  fft_m(base) {                   // M-point complex FFT, radices [R, 8, 8], digit-reversed result
    const R = this.R;
    if (R > 1) this.dif_pass(base, this.M, R, this.tw_m);
    for (let a = 0; a < R; a++) { this.dif_pass(base + a * 64, 64, 8, this.tw_64); for (let b = 0; b < 8; b++) this.dif_pass(base + a * 64 + b * 8, 8, 8, null); }
  }
  power4(pcm, off) {
    const N2 = this.n2, R = this.R, M = this.M, zr = this.zr, zi = this.zi, P = this.P;
    zr.fill(0); zi.fill(0);
    for (let n = 0; n < this.win; n++) { const xw = fr(pcm[off + n] * this.window[n]); if (n & 1) zi[n >> 1] = xw; else zr[n >> 1] = xw; }
    if (this.three) {
      // radix-3 DIF stage over the thirds (oracle/frontend.c radix3()), twiddle W_N2^{n k3}, then three M-point FFTs
      const c3 = fr(0.86602540378443864676), tw = this.tw_n2;
      for (let n = 0; n < M; n++) {
        const x0r = zr[n], x0i = zi[n], x1r = zr[n + M], x1i = zi[n + M], x2r = zr[n + 2 * M], x2i = zi[n + 2 * M];
        const tr = fr(x1r + x2r), ti = fr(x1i + x2i);
        const mr = fmaf(-0.5, tr, x0r), mi = fmaf(-0.5, ti, x0i);
        const sr = fr(c3 * fr(x1r - x2r)), si = fr(c3 * fr(x1i - x2i));
        const yr = [fr(x0r + tr), fr(mr + si), fr(mr - si)], yi = [fr(x0i + ti), fr(mi - sr), fr(mi + sr)];
        for (let k3 = 1; k3 < 3; k3++) if (n > 0) {
          const wr = tw[2 * n * k3], wi = tw[2 * n * k3 + 1], a = yr[k3], b = yi[k3];
          yr[k3] = fmaf(-b, wi, fr(a * wr)); yi[k3] = fmaf(b, wr, fr(a * wi));
        }
        for (let k3 = 0; k3 < 3; k3++) { zr[k3 * M + n] = yr[k3]; zi[k3 * M + n] = yi[k3]; }
      }
      for (let k3 = 0; k3 < 3; k3++) this.fft_m(k3 * M);
    } else this.fft_m(0);
    const zsub = (k) => (k % R) * 64 + (Math.floor(k / R) % 8) * 8 + Math.floor(k / (R * 8));
    const zpos = this.three ? ((k) => (k % 3) * M + zsub(Math.floor(k / 3))) : zsub;
    for (let k = 0; k <= this.kmax; k++) {
      const pa = zpos(k % N2), pb = zpos((N2 - k) % N2);
      const ar = zr[pa], ai = zi[pa], br = zr[pb], bi = -zi[pb];
      const er = fr(ar + br), ei = fr(ai + bi), or_ = fr(ar - br), oi = fr(ai - bi);
      const wr = this.tw_nfft[2 * k], wi = this.tw_nfft[2 * k + 1];
      const tr = fmaf(-oi, wi, fr(or_ * wr)), ti = fmaf(oi, wr, fr(or_ * wi));
      const xr = fr(er + ti), xi = fr(ei - tr);
      P[k] = fmaf(xr, xr, fr(xi * xi));
    }
    return P;
  }
  frame(pcm, off, out, oo) {
    const P = this.power4(pcm, off);
    for (let m = 0; m < this.bands; m++) {
      let e;
      if (this.cfg.spec_type === 1) { e = 0; const o = this.mel_off[m], k0 = this.mel_k0[m]; for (let j = 0; j < this.mel_cnt[m]; j++) e = fmaf(this.mel_w[o + j], P[k0 + j], e); }
      else { e = fr(0.25 * P[m]); if (this.cfg.spec_type === 3) e = fr(Math.sqrt(e)); }
      e = fr(e * this.emph[m]); e = fr(e * this.gain);
      out[oo + m] = !(e > 0) ? 0 : (e >= 4294967296 ? 0xffffffff : Math.trunc(e));
    }
  }
  run(pcm) {
    const nf = this.n_frames(pcm.length), out = new Uint32Array(nf * this.bands);
    for (let k = 0; k < nf; k++) this.frame(pcm, k * this.hop, out, k * this.bands);
    return out;
  }
}

// =====================================================================================================
// Back end (u32 frames -> segments / syllables / 53 features); mirrors oracle/backend.c
// =====================================================================================================
const WIN = [3, 4, 6, 9];                                                        // ref @B32325
function matchScore(gap, dist, n, tbin, pbin, tamp, pamp, vel) {                 // ref @B37340
  let s;
  if (tamp >= pamp) s = pamp / tamp; else { if (!(pamp > 0)) return 0; s = tamp / pamp; }
  if (gap === 0) return s > .1 ? 300 * s / dist : 0;
  if (s < .001) return 0;
  if (s >= 1) s = 10; else if (s < .1) s = 1; else s *= 10;
  let t = 10 - Math.abs(pbin - tbin - vel);
  if (t < 0) return 0;
  if (t < 1) t = 1;
  return 10 / gap * (t * t + Math.min(n, 10) * s);
}
const meanNZ = (v) => { let t = 0, c = 0; for (const x of v) if (x > 0) { t += x; c++; } return t / c; };
const stdNZ = (v, m) => { let t = 0; for (const x of v) t += (x - m) * (x - m); return Math.sqrt(t / v.length); };
const arrSum = (v) => { let t = 0; for (const x of v) t += x; return t; };

function formantFeatures(frs, ctx_max, floor, cs) {                              // ref @B32369
  const a = frs.length, x = new Array(53).fill(0);
  for (let n = 0; n < 3; n++) {
    const b = 5 + 16 * n, c = [], w = [], M = [], T = [], K = [], A = [];
    let prev = false, S = 0, L = 0, cnt = 0, runs = 0, up = 0, dn = 0;
    for (let t = 0; t < a; t++) {
      const r = frs[t][3 * n], E = frs[t][3 * n + 1];
      if (r > 0 && E > 0) {
        const wd = frs[t][3 * n + 2], dB = 20 * Math.log10(E);
        c.push(r * dB); w.push(r); M.push(wd * dB); T.push(E); K.push(dB);
        if (prev) {
          const dl = r - frs[t - 1][3 * n];
          if (dl > 1) up += dl; else if (dl < -1) dn += -1 * dl;
          if (E > L) { L = E; S = 1; } else if (S === 1 && E < L / 2) { if (L > 10) A.push(dB); L = 0; S = -1; }
        }
        if (!prev) runs += 1;
        prev = true; cnt += 1;
      } else { prev = false; S = 0; L = 0; }
    }
    if (runs > 0) {
      const sT = arrSum(T), sK = arrSum(K);
      x[b + 4] = sT / a * 100 / ctx_max; x[b + 5] = sT / cnt * 100 / ctx_max;
      x[b + 0] = arrSum(c) / sK; x[b + 1] = stdNZ(w, meanNZ(w)); x[b + 6] = arrSum(M) / sK;
      const mk = meanNZ(K); x[b + 2] = mk; x[b + 3] = stdNZ(K, mk);
      x[b + 11] = A.length;
      if (A.length > 0) { const ma = meanNZ(A); x[b + 12] = ma; x[b + 13] = stdNZ(A, ma); x[b + 14] = 100 * (ma / (sK / K.length) - 1); }
    }
    x[b + 7] = cnt; x[b + 8] = runs; x[b + 9] = up; x[b + 10] = dn; x[b + 15] = 100 * cnt / a;
  }
  x[0] = a; x[1] = Math.sqrt(a); x[2] = cs; x[3] = Math.log10(ctx_max); x[4] = floor;
  return x;
}

// ---- level 12: make_coeffs h(e) / f(e,t,n,i) (ref @B34150, @B33793) with the slice of numeric.js 1.2.6 they use
// (ref inner module 5: dotVV @B48151, inv @B55496, gradient @B89174, uncmin @B89779), restated loop for loop
const NEPS = 2220446049250313e-31;
function dotVV(x, y) { const n = x.length; let r = x[n - 1] * y[n - 1], i = n - 2; for (; i >= 1; i -= 2) r += x[i] * y[i] + x[i - 1] * y[i - 1]; if (i === 0) r += x[0] * y[0]; return r; }
const colOf = (b, j) => b.map((row) => row[j]);
const dotMM = (a, b) => a.map((row) => b[0].map((_, j) => dotVV(row, colOf(b, j))));
const dotMV = (a, x) => a.map((row) => dotVV(row, x));
const transposeM = (a) => a[0].map((_, j) => a.map((row) => row[j]));
const identityM = (n) => Array.from({ length: n }, (_, i) => Array.from({ length: n }, (_, j) => (i === j ? 1 : 0)));
function invM(a) {
  const m = a.length, n = a[0].length, A = a.map((r) => r.slice()), I = identityM(m);
  for (let j = 0; j < n; ++j) {
    let i0 = -1, v0 = -1;
    for (let i = j; i !== m; ++i) { const k = Math.abs(A[i][j]); if (k > v0) { i0 = i; v0 = k; } }
    const Aj = A[i0]; A[i0] = A[j]; A[j] = Aj;
    const Ij = I[i0]; I[i0] = I[j]; I[j] = Ij;
    let x = Aj[j];
    for (let k = j; k !== n; ++k) Aj[k] /= x;
    for (let k = n - 1; k !== -1; --k) Ij[k] /= x;
    for (let i = m - 1; i !== -1; --i) if (i !== j) {
      const Ai = A[i], Ii = I[i]; x = Ai[j];
      for (let k = j + 1; k !== n; ++k) Ai[k] -= Aj[k] * x;
      for (let k = n - 1; k !== -1; --k) Ii[k] -= Ij[k] * x;
    }
  }
  return I;
}
const norm2 = (x) => { let acc = 0; for (let i = x.length - 1; i >= 0; i--) acc += x[i] * x[i]; return Math.sqrt(acc); };
function gradientN(f, x) {
  const n = x.length, f0 = f(x);
  if (isNaN(f0)) throw new Error('gradient: f(x) is a NaN!');
  const x0 = x.slice(), J = Array(n); let it = 0;
  for (let i = 0; i < n; i++) for (let h = Math.max(1e-6 * f0, 1e-8); ;) {
    if (++it > 20) throw new Error('Numerical gradient fails');
    x0[i] = x[i] + h; const f1 = f(x0); x0[i] = x[i] - h; const f2 = f(x0); x0[i] = x[i];
    if (isNaN(f1) || isNaN(f2)) { h /= 16; continue; }
    J[i] = (f1 - f2) / (2 * h);
    const t0 = x[i] - h, t1 = x[i], t2 = x[i] + h, d1 = (f1 - f0) / h, d2 = (f0 - f2) / h;
    const N = Math.max(Math.abs(J[i]), Math.abs(f0), Math.abs(f1), Math.abs(f2), Math.abs(t0), Math.abs(t1), Math.abs(t2), 1e-8);
    const errest = Math.min(Math.max(Math.abs(d1 - J[i]), Math.abs(d2 - J[i]), Math.abs(d1 - d2)) / N, h / N);
    if (errest > 1e-3) h /= 16; else break;
  }
  return J;
}
function uncminN(f, x0) {
  const tol = Math.max(1e-8, NEPS), maxit = 1000, n = x0.length;
  x0 = x0.slice(); let f0 = f(x0);
  if (isNaN(f0)) throw new Error('uncmin: f(x0) is a NaN!');
  let H1 = identityM(n), it = 0, g0 = gradientN(f, x0);
  const fin = (v) => v.every((t) => isFinite(t));
  while (it < maxit) {
    if (!fin(g0)) break;
    const step = dotMV(H1, g0).map((t) => -t);
    if (!fin(step)) break;
    const nstep = norm2(step);
    if (nstep < tol) break;
    let t = 1, x1 = x0, s = null, f1 = f0; const df0 = dotVV(g0, step);
    while (it < maxit) {
      if (t * nstep < tol) break;
      s = step.map((p) => p * t); x1 = x0.map((a, k) => a + s[k]); f1 = f(x1);
      if (f1 - f0 >= 0.1 * t * df0 || isNaN(f1)) { t *= 0.5; ++it; continue; }
      break;
    }
    if (t * nstep < tol) break;
    if (it === maxit) break;
    const g1 = gradientN(f, x1), y = g1.map((a, k) => a - g0[k]), ys = dotVV(y, s), Hy = dotMV(H1, y), c = (ys + dotVV(y, Hy)) / (ys * ys);
    H1 = H1.map((row, i) => row.map((h, j) => (h + c * (s[i] * s[j])) - (Hy[i] * s[j] + s[i] * Hy[j]) / ys));
    x0 = x1; f0 = f1; g0 = g1; ++it;
  }
  return x0;
}
function polyfit(rows, col, order, log) {
  const xs = [], ys = [], X = []; let first = -1;
  for (let r = 0; r < rows.length; r++) if (rows[r][col] > 0) {
    if (first === -1) first = r;
    xs.push(r - first); ys.push(log ? 10 * Math.log10(rows[r][col]) : rows[r][col]);
    const row = []; for (let e = 0; e <= order; e++) row.push(1 * Math.pow(r, e)); X.push(row);
  }
  if (xs.length > 2) {
    const Xt = transposeM(X), c0 = Array.from(new Float32Array(dotMM(invM(dotMM(Xt, X)), dotMM(Xt, transposeM([ys]))).map((v) => v[0])));
    const cost = (c) => { let t = 0; for (let n = 0; n < xs.length; ++n) { let p = 0; for (let k = 0; k < c.length; k++) p += c[k] * Math.pow(xs[n], k); const a = p - ys[n]; t += a * a; } return t; };
    const sol = uncminN(cost, c0);
    return sol.concat([Math.sqrt(cost(sol)) / xs.length, xs.length]);
  }
  return new Array(order + 1).fill(0).concat([0, xs.length]);
}
const syllableCoeffs = (fr, sm) => [].concat(polyfit(sm, 1, 4, true), polyfit(fr, 0, 3, false), polyfit(fr, 3, 3, false), polyfit(fr, 6, 1, false));

// get_utterance_features(e, t) of inner module 7 (ref @B107902): 15 histograms, each divided by the total of ALL
// its properties (a NaN or negative index creates a property outside the array part that only the total sees)
function utteranceFeatures(segs, results) {
  const mk = (n) => new Array(n).fill(0);
  const H = { i: mk(10), o: mk(10), l: mk(10), s: mk(10), c: mk(20), u: mk(40), f: mk(40), d: mk(24), h: mk(24), p: mk(8), m: mk(8), g: mk(10), y: mk(10), v: mk(20), x: mk(20) };
  let prevEnd = segs[0].start;
  for (let r = 0; r < results.length; r++) {
    const segLen = segs[r].len, syl = results[r].syl, frs = results[r].frs;
    let osum = 0;
    for (const y of syl) {
      let a = 0, i = 0, l = 0, s = 0, c = 0, u = 0, f = 0, d = 0, h = 0, p = 0;
      for (let o = 0; o < y.len; o++) {
        const F = frs[y.start + o], G = o > 0 ? frs[y.start + o - 1] : null;
        if (F[0] > 0) { c++; a += F[0]; i += F[1]; l += F[2]; if (o > 0) s += F[0] - G[0]; }
        if (F[3] > 0) { p++; u += F[3]; f += F[4]; d += F[5]; if (o > 0) h += F[3] - G[3]; }
      }
      a /= c; i /= c; l /= c; u /= p; f /= p; d /= p;
      const e = y.len;
      let T = parseInt(e / 2); if (T >= 20) T = 19; H.c[T]++;
      let k = parseInt(a / 2); if (k >= 40) k = 39; H.u[k]++;
      let M = parseInt(u / 2); if (M >= 40) M = 39; H.f[M]++;
      let A = parseInt(3 * Math.log10(i)); if (A >= 24) A = 23; H.d[A]++;
      let S = parseInt(4 * Math.log10(f)); if (S >= 24) S = 23; H.h[S]++;
      let L = parseInt(l / 2); if (L >= 8) L = 7; H.p[L]++;
      let D = parseInt(d / 2); if (D >= 8) D = 7; H.m[D]++;
      let O = parseInt(10 * (e - c) / e); if (O >= 10) O = 9; H.g[O]++;
      let C = parseInt(10 * (e - p) / e); if (C >= 10) C = 9; H.y[C]++;
      let P = parseInt(20 * (s + 50) / 100); if (P >= 20) P = 19; if (P < 0) P = 0; H.v[P]++;
      let I = parseInt(20 * (h + 50) / 100); if (I >= 20) I = 19; if (I < 0) I = 0; H.x[I]++;
      osum += e;
    }
    let q = parseInt(10 * segLen / 150); if (q >= 10) q = 9; H.i[q]++;
    let n = syl.length; if (n >= 10) n = 9; H.o[n]++;
    let g = parseInt(10 * (segs[r].start - prevEnd) / 150); if (g >= 10) g = 9; H.l[g]++;
    let t = parseInt(2 * (osum / segLen - .3) * 10); if (t >= 10) t = 9; if (t < 0) t = 0; H.s[t]++;
    prevEnd = segs[r].start + segs[r].len;
  }
  const out = [];
  for (const key of 'iolscufdhpmgyvx') {
    const a = H[key]; let tot = 0;
    for (const n in a) tot += a[n];
    const b = a.slice();
    if (tot > 0) for (let j = 0; j < b.length; j++) b[j] /= tot;
    out.push(...b);
  }
  return out;
}

class Segmenter {
  constructor(c) {          // c: level, bands, window_step, pause_length, min_seg_length, auto_noise_gate, voiced_max_dB, voiced_min_dB
    this.c = c; this.B = c.bands;
    this.maxVoiced = Math.trunc(0.7 * c.bands);
    this.breaker = c.pause_length > 2 * c.window_step ? c.pause_length / c.window_step : 250 / c.window_step;
    this.minFrames = Math.trunc(c.min_seg_length / c.window_step);
    this.curFrame = 0; this.noFm = 0; this.cci = 0; this.started = -1;
    if (c.auto_noise_gate) { this.ctxMax = 50; this.floor = 2; } else { this.ctxMax = Math.pow(10, c.voiced_max_dB / 20); this.floor = Math.pow(10, c.voiced_min_dB / 20); }
    this.lastMax = this.ctxMax; this.lastFloor = this.floor; this.w = 0; this.T = 0; this.k = 0;
    this.tracks = []; this.accS = 0; this.accC = 0;
    this.segs = [];           // {start, len, flag, feat | syl: [{start,len,feat}]}
  }
  reset(x) { this.cci = 0; this.started = x; this.noFm = 0; this.tracks = []; this.accS = 0; this.accC = 0; }
  accumulate(e, pk, n, energy, floor) {                                           // ref @B35952
    const U = pk.length; if (U < 1) return;
    const asg = new Array(U).fill(-1), best = new Array(U).fill(0);
    this.accS += energy;
    const tr = this.tracks, ntr = tr.length;
    for (let r = 0; r < ntr; r++) {
      const t = tr[r], gap = n - t.lastFrame;
      if (gap >= 0 && gap < 4) for (let o = 0; o < U; o++) {
        const dist = Math.abs(t.lastBin - pk[o][2]);
        if (dist < WIN[gap]) { const sc = matchScore(gap, dist, t.frames.length, t.lastBin, pk[o][2], t.lastAmp, e[pk[o][2]], t.vel); if (sc > 1 && sc > best[o]) { best[o] = sc; asg[o] = r; } }
      }
    }
    for (let r = 0; r < ntr; r++) {
      const ids = []; for (let o = 0; o < U; o++) if (asg[o] === r) ids.push(o);
      if (!ids.length) continue;
      const t = tr[r]; let pb = pk[ids[0]][2]; const amp = e[pb];
      if (amp > floor) {
        let st = pk[ids[0]][0], en = pk[ids[0]][1];
        for (const o of ids) { if (pk[o][1] > en) en = pk[o][1]; if (pk[o][0] < st) st = pk[o][0]; if (e[pk[o][2]] > e[pb]) pb = pk[o][2]; }
        let be = 0; for (let q = st; q <= en; q++) be += e[q];
        const P = t.bins, h = P.length;
        if (h >= 3) t.vel = (pb - P[h - 1] + (P[h - 2] - P[h - 1]) + (P[h - 3] - P[h - 2])) / 3;
        else if (h === 2) t.vel = (pb - P[h - 1] + (P[h - 2] - P[h - 1])) / 2;
        else if (h === 1) t.vel = pb - P[h - 1];
        t.lastFrame = n; t.lastBin = pb; t.lastAmp = amp;
        t.frames.push(n); t.starts.push(st); t.ends.push(en); t.bins.push(pb); t.amps.push(amp); t.energies.push(be);
        t.sumE += be; t.count += 1; t.sumEbin += be * pb; t.sumW += en - st + 1;
        this.accS -= be; this.accC += be;
      }
    }
    for (let o = 0; o < U; o++) if (asg[o] === -1) {
      const pb = pk[o][2], amp = e[pb];
      if (amp > floor) {
        const st = pk[o][0], en = pk[o][1]; let be = 0; for (let q = st; q <= en; q++) be += e[q];
        tr.push({ lastFrame: n, vel: 0, lastBin: pb, lastAmp: amp, frames: [n], starts: [st], ends: [en], bins: [pb], amps: [amp], energies: [be], sumE: be, count: 1, sumEbin: be * pb, sumW: en - st + 1 });
      }
    }
  }
  finalize(eArg) {                                                                // ref @B27088
    const len = eArg - this.noFm;
    if (!(len > this.minFrames && this.started >= 2)) return;
    const seg = { start: this.curFrame - len, len, flag: 0, feat: null, syl: [] };
    this.segs.push(seg);
    const level = this.c.level;
    const ranked = [];
    for (const t of this.tracks) if (t.count >= 2) {
      const mb = t.sumEbin / t.sumE;
      if (mb >= 7) { let r = 0; while (r < ranked.length && !(ranked[r].sumEbin / ranked[r].sumE > mb)) r++; ranked.splice(r, 0, t); }
    }
    if (level === 3) {          // ref @B28273: the ranked track records themselves are the result (18 fields, SURVEY.md App. A)
      seg.flag = 1;
      seg.tracks = ranked.map((t) => [t.starts[t.count - 1], t.ends[t.count - 1], t.lastFrame, t.lastFrame, t.vel, t.lastBin, t.lastAmp,
        t.frames.slice(), t.starts.slice(), t.ends.slice(), t.bins.slice(), t.amps.slice(), t.energies.slice(), t.sumE, t.count, t.sumEbin, 0, t.sumW]);
      return;
    }
    const frs = [], sm = [];
    for (let i = 0; i < len; i++) { frs.push(new Float32Array(9)); sm.push(new Float32Array(3)); }
    let last = 0, slot = 0;
    for (const t of ranked) {
      const mb = t.sumEbin / t.sumE;
      if (Math.abs(mb - last) > 20) { last = mb; slot++; if (slot >= 3) break; }
      for (let i = 0; i < t.count; i++) {
        let l = slot; const f = t.bins[i];
        if (f > 0) {
          const E = t.energies[i], d = t.frames[i], wd = t.ends[i] - t.starts[i] + 1;
          if (d >= len) { seg.flag = -1; return; }            // the reference throws here (r[d] undefined): entry kept, no result
          if (frs[d][3 * l] > this.floor && frs[d][3 * l] < f && l < 2) l++;
          frs[d][3 * l] = f; frs[d][3 * l + 1] = E; frs[d][3 * l + 2] = wd;
          sm[d][0] += f * E; sm[d][1] += E; sm[d][2] += wd * E;
        }
      }
    }
    const cs = this.accC / this.accS;
    if (level === 5) { seg.feat = formantFeatures(frs, this.ctxMax, this.floor, cs); seg.flag = 1; }
    if (level === 4) seg.flag = 1;
    if (level === 10 || level === 11 || level === 12 || level === 13) {
      seg.frs = frs; seg.sms = sm;                                         // levels 10 / 11 keep the straightened frames (ref @B27713)
      let i = -1, c = 0, u = 0;
      for (let e2 = 0; e2 < len; e2++) {
        if (sm[e2][1] > this.floor) { c = 0; u++; if (i < 0) i = e2; } else c++;
        if ((u > 20 && c > 0) || (u > 10 && c > 1) || (u > 0 && c > 4) || (e2 >= len - 1 && u > 4)) {
          const t = e2 - c;
          if (t - i > 1) { seg.syl.push({ start: i, len: t - i, feat: level === 13 ? formantFeatures(frs.slice(i, t), this.ctxMax, this.floor, cs)
            : null }); i = -1; u = 0; }
        }
      }
      if (level === 12) {                                  // ref make_coeffs h(e) @B34150: rows until numeric throws
        seg.coef = [];
        try { for (const y of seg.syl) if (y.len > 1) seg.coef.push(syllableCoeffs(frs.slice(y.start, y.start + y.len), sm.slice(y.start, y.start + y.len))); } catch (err) { /* console.error(e) in the reference */ }
      }
      seg.flag = seg.syl.length > 0 ? 1 : 0;
    }
  }
  gate(h) {                                                                       // ref @B28506
    this.w++;
    if (h > this.ctxMax || (this.w > 40 && h > 2 * this.floor)) {
      if (h >= this.ctxMax) { this.w = 0; this.lastMax = this.ctxMax = h; }
      else if (h > this.lastMax / 100) { this.ctxMax -= Math.trunc(this.ctxMax / 8); this.w = 35; }
      const y = this.ctxMax, t = Math.log10(y);
      const v = t > 7 ? Math.trunc(Math.pow(10, t - 3) / 20) : t > 6 ? Math.trunc(Math.pow(10, t - 3) / 2) : t > 4 ? Math.trunc(Math.pow(10, t - 2) / 2)
        : t > 2 ? Math.trunc(Math.pow(10, t / 3)) : t > 1 ? Math.trunc(y / 10) : 1;
      this.floor = v; this.lastFloor = v;
      if (this.k > 0 && this.T / this.k < 30 * v) { this.reset(0); this.k = 0; this.T = 0; }
      this.T += this.ctxMax; this.k += 1;
    } else if (this.floor > 10 && this.floor > this.lastFloor / 10 && this.w > 20) {
      this.floor -= Math.trunc(this.lastFloor / 20);
      if (this.floor < 10) this.floor = 10;
    }
  }
  push(e, eo) {                                                                   // ref @B30392 + D() @B25717; e = Uint32Array view, eo = offset
    const B = this.B; this.curFrame++;
    const t = this.cci, v = this.floor;
    let n = 0, i = 0, l = 0, s = 0, c = 0, u = 0, p = 0, d = 0, h = 2 * v, g = 0;
    const pk = [], E = eo === 0 && e.length === B ? e : e.subarray(eo, eo + B);
    const emit = (upd) => { if (upd && E[l] > h) { h = E[l]; p = l; } const thr = E[l] / 10; while (i < l && E[i] < thr) i++; while (s > l && E[s] < thr) s--; pk.push([i, s, l]); n++; d += E[l]; };
    for (let a = 1; a < B; a++) {
      const ea = E[a]; g += ea;
      if (ea > E[a - 1] && (a < 2 || ea > E[a - 2]) && (a < 3 || ea > E[a - 3])) {
        if (u === -1 || u === 0) { if (u === -1 && E[l] > v && i <= l && l < s) emit(true); i = a - 1; l = a; } else if (u === 1) l = a;
        u = 1;
      } else if (ea < E[a - 1] && (a < 2 || ea < E[a - 2]) && (a < 3 || ea < E[a - 3])) { if (u === 1 || u === -1) { s = a; u = -1; } }
      else if (u === -1) { c++; if (c > 2) { c = 0; if (E[l] > v && i <= l && l < s) emit(true); u = 0; } }
      else if (u === 1 && ea > E[a - 1]) l = a;
      if (a === B - 1 && u === 1) { s = a; l = a; if (E[l] > v && i < l && l <= s) emit(false); }
    }
    if (this.started < 0) {
      const r = d > h ? h * (n - 1) / (d - h) : 0;
      if (n > 0 && p > 7 && p < this.maxVoiced && n > 4 && r > 4) this.reset(0); else this.noFm++;
    }
    let doReset = false;
    if (this.started >= 0) {
      if (n === 0 || p < 7 || p >= this.maxVoiced || (n > 3 && d / (g - d) < .1)) {
        this.noFm++;
        if (this.started < 2) this.started--;
        else if (this.noFm >= this.breaker) { this.finalize(this.cci + 1); doReset = true; }
        else if (this.c.auto_noise_gate) this.gate(h);
      } else {
        if (this.c.auto_noise_gate) this.gate(h);
        this.accumulate(E, pk, t, g, this.floor);
        if (this.started < 2) this.started++; else this.noFm = 0;
      }
    }
    this.cci++;
    if (doReset) this.reset(-1);
  }
  finish() { this.finalize(this.cci); this.reset(1); }                             // ref @B30757
  // the callback sequence of the reference's dispatcher P() (ref @B28869), incl. the misaligned
  // timestamps after a segment whose straighten step threw
  callbacks() {
    const step = this.c.window_step / 1e3, out = [], res = this.segs.filter((s) => s.flag >= 0);
    res.forEach((s, k) => {
      const u = this.segs[k];
      if (this.c.level === 5) out.push([k, [], [u.start * step, (u.len + 1) * step], s.feat]);
      else if (this.c.level === 3) { if (s.tracks.length > 0) out.push([k, [], s.tracks]); }                    // ref @B30132: three arguments
      else if (this.c.level === 13 && s.syl.length > 0) out.push([k, [], s.syl.map((y) => [((u.start + y.start) * step).toFixed(3), ((y.len + 1) * step).toFixed(3)]), s.syl.map((y) => y.feat)]);
      else if (this.c.level === 12 && s.coef && s.coef.length > 0) out.push([k, [], s.syl.map((y) => [((u.start + y.start) * step).toFixed(3), ((y.len + 1) * step).toFixed(3)]), s.coef]);
      else if (this.c.level === 11) {
        // ref dispatcher P() @B28869: (0, label, Y(), get_utterance_features(u, h)) after every new result, over
        // everything so far; u = the entries pushed up to then (own one included), indexed by RESULT index
        const own = this.segs.indexOf(s);
        let tsum = 0; for (let j = 0; j <= own; j++) tsum += this.segs[j].len;
        out.push([0, [], [this.segs[0].start * step, (tsum + 1) * step], utteranceFeatures(this.segs, res.slice(0, k + 1))]);
      }
    });
    return out;
  }
}

function runBackend(spectra, frames, cfg) {
  const sg = new Segmenter(cfg);
  for (let f = 0; f < frames; f++) sg.push(spectra, f * cfg.bands);
  sg.finish();
  return sg;
}

// whole path for one clip: {pcm: Float32Array, fs} + reference-style settings -> Segmenter
// spec RS-1 (oracle/resample.c, DESIGN.md): the sample-rate conversion the browser's decodeAudioData does in front of the reference's
// offline path (always to 48 kHz, ref dist/main.js:2 @B18769) — 32-tap windowed-sinc kernels at 32 sub-sample offsets, linear
// interpolation between two neighbouring kernels, fp32 fmaf chains, fp64 blend.  cos / sin of the table are the engine's (the table
// is compared with the C oracle's in tests/test_oracle_js.py)
function resample(x, fs_in, fs_out) {
  const TAPS = 32, OFFS = 32, ratio = fs_in / fs_out, scale = (ratio > 1.0 ? 1.0 / ratio : 1.0) * 0.9;
  const K = new Float32Array((OFFS + 1) * TAPS);
  for (let o = 0; o <= OFFS; o++) {
    const s = o / OFFS;
    for (let i = 0; i < TAPS; i++) {
      const pre = Math.PI * ((i - TAPS / 2) - s), xx = (i - s) / TAPS;
      const w = 0.42 - 0.5 * Math.cos(2.0 * Math.PI * xx) + 0.08 * Math.cos(4.0 * Math.PI * xx);
      K[o * TAPS + i] = w * (pre === 0.0 ? scale : Math.sin(scale * pre) / pre);
    }
  }
  const n_in = x.length, n_out = Math.floor(n_in / ratio), out = new Float32Array(n_out);
  for (let n = 0; n < n_out; n++) {
    const pos = n * ratio, src = Math.floor(pos), vo = (pos - src) * OFFS, o = Math.floor(vo), f = vo - o;
    let s1 = 0, s2 = 0;
    for (let i = 0; i < TAPS; i++) {
      const q = src + i - TAPS / 2, xv = (q >= 0 && q < n_in) ? x[q] : 0;
      s1 = fmaf(xv, K[o * TAPS + i], s1); s2 = fmaf(xv, K[(o + 1) * TAPS + i], s2);
    }
    out[n] = (1.0 - f) * s1 + f * s2;
  }
  return out;
}

function analyze(pcm, fs, settings) {
  if (settings.resample_to > 0 && settings.resample_to !== fs) { pcm = resample(pcm, fs, settings.resample_to); fs = settings.resample_to; }
  const fe = new FrontEnd(Object.assign({ fs }, settings));
  const spec = fe.run(pcm);
  const sg = runBackend(spec, fe.n_frames(pcm.length), { level: settings.output_level, bands: fe.bands, window_step: settings.window_step,
    pause_length: settings.pause_length, min_seg_length: settings.min_seg_length, auto_noise_gate: settings.auto_noise_gate,
    voiced_max_dB: settings.voiced_max_dB, voiced_min_dB: settings.voiced_min_dB });
  return { fe, spec, sg };
}

const DEFAULTS = { spec_type: 1, output_level: 5, f_min: 50, f_max: 4000, N_fft_bins: 256, N_mel_bins: 128, window_width: 25, window_step: 25,
  pause_length: 200, min_seg_length: 50, auto_noise_gate: true, voiced_max_dB: 100, voiced_min_dB: 10, pre_norm_gain: 1000, high_f_emph: 0 };

module.exports = { FrontEnd, Segmenter, runBackend, analyze, resample, formantFeatures, matchScore, fmaf, DEFAULTS };
