"""ctypes binding of the ORACLE (oracle/libwsa_oracle.so) — test infrastructure only.

May be imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg; never by the
product package (webspeechanalyzer_amd/).  Builds the library with `make -C oracle` on first use.
"""
import ctypes
import os
import subprocess
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


class Cfg(ctypes.Structure):
    _fields_ = [("level", ctypes.c_int32), ("bands", ctypes.c_int32),
                ("window_step", ctypes.c_double), ("pause_length", ctypes.c_double),
                ("min_seg_length", ctypes.c_double), ("auto_noise_gate", ctypes.c_int32),
                ("voiced_max_dB", ctypes.c_double), ("voiced_min_dB", ctypes.c_double)]


class FeCfg(ctypes.Structure):
    _fields_ = [("fs", ctypes.c_double), ("spec_type", ctypes.c_int32), ("f_min", ctypes.c_double),
                ("f_max", ctypes.c_double), ("n_fft_bins", ctypes.c_int32), ("n_mel_bins", ctypes.c_int32),
                ("window_width", ctypes.c_double), ("window_step", ctypes.c_double),
                ("pre_norm_gain", ctypes.c_double), ("high_f_emph", ctypes.c_double)]


def build():
    subprocess.run(["make", "-s", "-C", _HERE], check=True)


def lib():
    global _LIB
    if _LIB is not None:
        return _LIB
    so = os.path.join(_HERE, "libwsa_oracle.so")
    srcs = [os.path.join(_HERE, f) for f in os.listdir(_HERE) if f.endswith((".c", ".h"))]
    if not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs):
        build()
    L = ctypes.CDLL(so)
    d, i32, vp = ctypes.c_double, ctypes.c_int32, ctypes.c_void_p
    for name in ("wsa_or_log", "wsa_or_log10"):
        getattr(L, name).restype = d
        getattr(L, name).argtypes = [d]
    L.wsa_or_pow.restype = d
    L.wsa_or_pow.argtypes = [d, d]
    L.wsa_or_match_score.restype = d
    L.wsa_or_match_score.argtypes = [d] * 8
    L.wsa_or_run_clip.restype = vp
    L.wsa_or_run_clip.argtypes = [ctypes.POINTER(Cfg), vp, i32]
    L.wsa_or_seg_new.restype = vp
    L.wsa_or_seg_new.argtypes = [ctypes.POINTER(Cfg)]
    L.wsa_or_seg_push.argtypes = [vp, vp]
    L.wsa_or_seg_finish.argtypes = [vp]
    L.wsa_or_enable_trace.argtypes = [vp, i32]
    L.wsa_or_seg_free.argtypes = [vp]
    for name in ("wsa_or_n_segments", "wsa_or_n_syllables", "wsa_or_trace_len"):
        getattr(L, name).restype = i32
        getattr(L, name).argtypes = [vp]
    L.wsa_or_segment.argtypes = [vp, i32, ctypes.POINTER(i32)]
    L.wsa_or_syllable.argtypes = [vp, i32, ctypes.POINTER(i32)]
    for name in ("wsa_or_segment_features", "wsa_or_syllable_features"):
        getattr(L, name).restype = ctypes.POINTER(d)
        getattr(L, name).argtypes = [vp, i32]
    L.wsa_or_segment_formants.restype = ctypes.POINTER(ctypes.c_float)
    L.wsa_or_segment_formants.argtypes = [vp, i32]
    L.wsa_or_segment_sums.restype = ctypes.POINTER(ctypes.c_float)
    L.wsa_or_segment_sums.argtypes = [vp, i32]
    L.wsa_or_resample_length.restype = ctypes.c_uint64
    L.wsa_or_resample_length.argtypes = [ctypes.c_uint64, ctypes.c_double, ctypes.c_double]
    L.wsa_or_resample.argtypes = [vp, ctypes.c_uint64, ctypes.c_double, ctypes.c_double, vp]
    L.wsa_or_resample.restype = None
    L.wsa_or_segment_tracks.restype = ctypes.POINTER(ctypes.c_double)
    L.wsa_or_segment_tracks.argtypes = [vp, i32, ctypes.POINTER(ctypes.c_int32)]
    L.wsa_or_trace.restype = ctypes.POINTER(d)
    L.wsa_or_trace.argtypes = [vp]
    L.wsa_or_formant_features.argtypes = [vp, i32, d, d, d, vp]
    L.wsa_or_fe_new.restype = vp
    L.wsa_or_fe_new.argtypes = [ctypes.POINTER(FeCfg)]
    L.wsa_or_fe_free.argtypes = [vp]
    for name in ("wsa_or_fe_bands", "wsa_or_fe_nfft", "wsa_or_fe_win", "wsa_or_fe_hop", "wsa_or_fe_kmax"):
        getattr(L, name).restype = i32
        getattr(L, name).argtypes = [vp]
    L.wsa_or_fe_n_frames.restype = i32
    L.wsa_or_fe_n_frames.argtypes = [vp, ctypes.c_int64]
    L.wsa_or_fe_bins_hz.restype = ctypes.POINTER(d)
    L.wsa_or_fe_bins_hz.argtypes = [vp]
    L.wsa_or_fe_table.restype = ctypes.POINTER(ctypes.c_float)
    L.wsa_or_fe_table.argtypes = [vp, i32, ctypes.POINTER(i32)]
    L.wsa_or_fe_power4.argtypes = [vp, vp, vp]
    L.wsa_or_fe_frame.argtypes = [vp, vp, vp]
    L.wsa_or_fe_run.restype = i32
    L.wsa_or_fe_run.argtypes = [vp, vp, ctypes.c_int64, vp]
    _LIB = L
    return L


def fe_cfg(fs=16000.0, **kw):
    c = dict(fs=fs, spec_type=1, f_min=50.0, f_max=4000.0, n_fft_bins=256, n_mel_bins=128,
             window_width=25.0, window_step=25.0, pre_norm_gain=1000.0, high_f_emph=0.0)
    c.update(kw)
    return FeCfg(**c)


class FrontEnd:
    """Oracle front end FE-1 (oracle/frontend.c)."""

    def __init__(self, cfg):
        self.L = lib()
        self.cfg = cfg
        self.h = self.L.wsa_or_fe_new(ctypes.byref(cfg))
        if not self.h:
            raise ValueError("invalid front-end configuration")
        for k in ("bands", "nfft", "win", "hop", "kmax"):
            setattr(self, k, getattr(self.L, "wsa_or_fe_" + k)(self.h))

    def __del__(self):
        if getattr(self, "h", None):
            self.L.wsa_or_fe_free(self.h)
            self.h = None

    def n_frames(self, n):
        return self.L.wsa_or_fe_n_frames(self.h, n)

    def bins_hz(self):
        return np.ctypeslib.as_array(self.L.wsa_or_fe_bins_hz(self.h), shape=(self.bands,)).copy()

    def table(self, which):
        n = ctypes.c_int32()
        p = self.L.wsa_or_fe_table(self.h, which, ctypes.byref(n))
        return np.ctypeslib.as_array(p, shape=(n.value,)).copy() if n.value else np.zeros(0, np.float32)

    def power4(self, frame):
        frame = np.ascontiguousarray(frame, dtype=np.float32)
        assert frame.shape[0] >= self.win
        P = np.zeros(self.kmax + 2, np.float32)
        self.L.wsa_or_fe_power4(self.h, frame.ctypes.data, P.ctypes.data)
        return P[: self.kmax + 1]

    def run(self, pcm):
        pcm = np.ascontiguousarray(pcm, dtype=np.float32)
        nf = self.n_frames(pcm.shape[0])
        out = np.zeros((nf, self.bands), np.uint32)
        if nf:
            self.L.wsa_or_fe_run(self.h, pcm.ctypes.data, pcm.shape[0], out.ctypes.data)
        return out


def default_cfg(level=5, bands=128, **kw):
    c = dict(level=level, bands=bands, window_step=25.0, pause_length=200.0, min_seg_length=50.0,
             auto_noise_gate=1, voiced_max_dB=100.0, voiced_min_dB=10.0)
    c.update(kw)
    c["auto_noise_gate"] = int(bool(c["auto_noise_gate"]))
    return Cfg(**c)


def run_backend(spectra, cfg, trace=False):
    """spectra: (frames, bands) uint32.  Returns dict mirroring tests/golden/gen/ref_driver.js output:
    segments_ci [[start,len]], syllables_ci [[[start,len]...]] (levels 10/13), features
    (level 5: [53] per segment; level 13: [[53]...] per segment), formants (level>=4)."""
    L = lib()
    spectra = np.ascontiguousarray(spectra, dtype=np.uint32)
    frames, bands = spectra.shape
    assert bands == cfg.bands
    want_level = cfg.level
    if cfg.level == 12:              # level 12 = level 10's products + polynomial fits per syllable (ref make_coeffs @B34150)
        cfg = Cfg(**{k: getattr(cfg, k) for k, _ in Cfg._fields_})
        cfg.level = 10
    if cfg.level == 11:              # level 11 = level 10's products + a reduction at dispatch time (ref @B27713, @B28869)
        cfg = Cfg(**{k: getattr(cfg, k) for k, _ in Cfg._fields_})
        cfg.level = 10
    h = L.wsa_or_seg_new(ctypes.byref(cfg))
    try:
        L.wsa_or_enable_trace(h, int(trace))
        for f in range(frames):
            L.wsa_or_seg_push(h, spectra[f].ctypes.data)
        L.wsa_or_seg_finish(h)
        out = {"segments_ci": [], "syllables_ci": [], "features": [], "formants": [], "flags": [], "sums": []}
        info = (ctypes.c_int32 * 5)()
        sy = (ctypes.c_int32 * 3)()
        for i in range(L.wsa_or_n_segments(h)):
            L.wsa_or_segment(h, i, info)
            start, ln, syl0, nsyl, has = list(info)
            out["segments_ci"].append([start, ln])
            out["flags"].append(has)
            if has < 0:          # straighten threw in the reference: no result entry
                out["sums"].append(None)
                out["formants"].append(None)
                out["features"].append(None)
                out["syllables_ci"].append(None)
                continue
            if cfg.level == 3:
                nn = ctypes.c_int32(0)
                tp = L.wsa_or_segment_tracks(h, i, ctypes.byref(nn))
                out.setdefault("tracks", []).append(unflatten_tracks(np.ctypeslib.as_array(tp, shape=(nn.value,)).copy()))
            if cfg.level >= 4:
                fp = L.wsa_or_segment_formants(h, i)
                out["formants"].append(np.ctypeslib.as_array(fp, shape=(ln, 9)).copy())
                out["sums"].append(np.ctypeslib.as_array(L.wsa_or_segment_sums(h, i), shape=(ln, 3)).copy())
            if cfg.level == 5:
                out["features"].append(np.ctypeslib.as_array(L.wsa_or_segment_features(h, i), shape=(53,)).copy())
            if cfg.level in (10, 13):
                ci, ft = [], []
                for j in range(syl0, syl0 + nsyl):
                    L.wsa_or_syllable(h, j, sy)
                    ci.append([sy[1], sy[2]])
                    if cfg.level == 13:
                        ft.append(np.ctypeslib.as_array(L.wsa_or_syllable_features(h, j), shape=(53,)).copy())
                out["syllables_ci"].append(ci)
                if cfg.level == 13:
                    out["features"].append(ft)
        if want_level in (11, 12):
            cfg = Cfg(**{k: getattr(cfg, k) for k, _ in Cfg._fields_})
            cfg.level = want_level
        out["callbacks"] = callbacks(out, cfg)
        if trace:
            n = L.wsa_or_trace_len(h)
            out["trace"] = np.ctypeslib.as_array(L.wsa_or_trace(h), shape=(n, 10)).copy() if n else np.zeros((0, 10))
        return out
    finally:
        L.wsa_or_seg_free(h)


def resample(x, fs_in, fs_out):
    """spec RS-1 (oracle/resample.c): float32 clip at fs_in -> float32 clip at fs_out."""
    L = lib()
    x = np.ascontiguousarray(x, dtype=np.float32)
    y = np.zeros(int(L.wsa_or_resample_length(len(x), float(fs_in), float(fs_out))), np.float32)
    L.wsa_or_resample(x.ctypes.data, len(x), float(fs_in), float(fs_out), y.ctypes.data)
    return y


def _num(x):
    return int(x) if float(x).is_integer() else float(x)


def unflatten_tracks(flat):
    """wsa_or_segment_tracks layout -> the reference's 18-field track records (SURVEY.md App. A field map)."""
    n, w, tracks = int(flat[0]), 1, []
    for _ in range(n):
        st, en, lf, vel, lb, la, se, cnt, seb, sw = flat[w:w + 10]; w += 10
        c = int(cnt)
        arrs = [[_num(v) for v in flat[w + k * c:w + (k + 1) * c]] for k in range(6)]; w += 6 * c
        tracks.append([_num(st), _num(en), _num(lf), _num(lf), _num(vel), _num(lb), _num(la)] + arrs + [_num(se), c, _num(seb), 0, _num(sw)])
    return tracks


def callbacks(out, cfg):
    """The callback sequence the reference's dispatcher P() (dist/main.js:2 @B28869) would deliver.

    A segment whose straighten step throws (track frame index >= segment length, reachable when
    unvoiced frames interleave the first voiced ones) is already in segments_ci but has no result
    entry (@B27240: `u.push` precedes the throw), so from then on result k is reported with the
    timestamps of segments_ci[k] — reproduced here, not repaired."""
    step = cfg.window_step / 1e3
    segs = out["segments_ci"]
    res = [i for i, f in enumerate(out["flags"]) if f >= 0]
    cbs = []
    for k, i in enumerate(res):
        u = segs[k]
        if cfg.level == 5:
            cbs.append([k, [], [u[0] * step, (u[1] + 1) * step], out["features"][i]])
        elif cfg.level == 13:
            ci, ft = out["syllables_ci"][i], out["features"][i]
            if len(ft) > 0:
                tm = [["%.3f" % ((u[0] + c[0]) * step), "%.3f" % ((c[1] + 1) * step)] for c in ci]
                cbs.append([k, [], tm, ft])
        elif cfg.level == 4:
            cbs.append([k, [], [u[0] * step, (u[1] + 1) * step], out["formants"][i]])
        elif cfg.level == 3:                       # ref @B30132: `s[e].length > 0 && b(e, label, s[e])` — three arguments
            if len(out["tracks"][i]) > 0:
                cbs.append([k, [], out["tracks"][i]])
        elif cfg.level == 12:                      # ref @B27240 (12 == process_level): make_coeffs(sep_syllables(...))
            ci = out["syllables_ci"][i]
            ft = []
            try:                                   # ref h(e) @B34150: `try { for (...) r.push(...) } catch (e) { console.error(e) } return r`
                for c in ci:
                    if c[1] > 1:
                        ft.append(syllable_coeffs(out["formants"][i][c[0]:c[0] + c[1]], out["sums"][i][c[0]:c[0] + c[1]]))
            except ValueError:                     # numeric threw (NaN cost after a singular normal matrix, gradient fails):
                pass                               # the rows collected so far are what the segment reports
            if len(ft) > 0:
                tm = [["%.3f" % ((u[0] + c[0]) * step), "%.3f" % ((c[1] + 1) * step)] for c in ci]
                cbs.append([k, [], tm, ft])
        elif cfg.level == 11:
            # ref @B28869: `b(0, label, Y(), get_utterance_features(u, h))` after every new result, over everything so
            # far; u = segments_ci pushed up to then (own entry included, dropped ones too), indexed by RESULT index
            res_syl = [out["syllables_ci"][j] for j in res[:k + 1]]
            res_fr = [out["formants"][j] for j in res[:k + 1]]
            upto = segs[:i + 1]
            tsum = sum(x[1] for x in upto)
            cbs.append([0, [], [upto[0][0] * step, (tsum + 1) * step], utterance_features(segs, res_syl, res_fr)])
        elif cfg.level == 10:                      # ref @B27713: per syllable its slice of the straightened frames
            ci = out["syllables_ci"][i]
            if len(ci) > 0:
                tm = [["%.3f" % ((u[0] + c[0]) * step), "%.3f" % ((c[1] + 1) * step)] for c in ci]
                cbs.append([k, [], tm, [out["formants"][i][c[0]:c[0] + c[1]] for c in ci]])
    return cbs


def _polyfit(rows, col, order, log):
    """f(e, t, n, i) of ref inner module 4 (@B33793): least-squares polynomial of column `col` over the rows where it
    is positive (design matrix in the ROW INDEX r, residuals later in r - first: the reference's own inconsistency),
    Float32 start, numeric.uncmin refinement, then [coefficients..., rms error, points]."""
    from oracle import numeric_js as nj
    L = lib()
    xs, ys, X = [], [], []
    first = -1
    for r in range(len(rows)):
        v = float(rows[r][col])
        if v > 0:
            if first == -1:
                first = r
            xs.append(float(r - first))
            ys.append(10 * L.wsa_or_log10(v) if log else v)
            X.append([1 * L.wsa_or_pow(float(r), float(e)) for e in range(order + 1)])
    if len(xs) > 2:
        Y = nj.transpose([ys])
        Xt = nj.transpose(X)
        XtX = nj.dotMM(Xt, X)
        Inv = nj.inv(XtX)
        XtY = nj.dotMM(Xt, Y)
        c0 = [float(np.float32(v[0])) for v in nj.dotMM(Inv, XtY)]              # new Float32Array(...)

        def cost(c):
            t = 0.0
            for n in range(len(xs)):
                p = 0.0
                for k in range(len(c)):
                    p += c[k] * L.wsa_or_pow(xs[n], float(k))                     # solve_poly @B1521
                a = p - ys[n]
                t += a * a
            return t
        sol = nj.uncmin(cost, c0)
        return list(sol) + [np.sqrt(cost(sol)) / len(xs), float(len(xs))]
    return [0.0] * (order + 1) + [0.0, float(len(xs))]


def syllable_coeffs(fr, sm):
    """one row of make_coeffs h(e) (ref @B34150): 23 numbers for a syllable with frames fr [n, 9] and sums sm [n, 3]."""
    return np.array(_polyfit(sm, 1, 4, True) + _polyfit(fr, 0, 3, False) + _polyfit(fr, 3, 3, False) + _polyfit(fr, 6, 1, False))


def utterance_features(segs, syl_ci, frames):
    """get_utterance_features(e, t) of inner module 7 (ref @B107902): 15 histograms over the syllables of the
    results so far, each normalised by its total.  segs = segments_ci (indexed by result index, as the reference
    does), syl_ci[r] = [[start, len]...] and frames[r] = [seg_len, 9] float32 of result r.  `a[idx]++` with an index
    that is NaN or negative creates a property outside the array part whose value is NaN (undefined + 1); the
    normalisation's `for (n in e) t += e[n]` then yields NaN, `t > 0` is false and that histogram stays RAW COUNTS."""
    L = lib()
    sizes = dict(i=10, o=10, l=10, s=10, c=20, u=40, f=40, d=24, h=24, p=8, m=8, g=10, y=10, v=20, x=20)
    H = {k: [0.0] * n for k, n in sizes.items()}
    ghost = {k: False for k in sizes}         # histogram holds a NaN-valued property

    def bump(k, idx, hi_clamp=True, lo_clamp=False):
        n = sizes[k]
        if idx != idx:                          # parseInt(NaN) -> property "NaN" = NaN
            ghost[k] = True
            return
        idx = int(idx)                          # parseInt of a finite Number in these ranges = truncation
        if idx >= n:
            idx = n - 1
        if idx < 0:
            if lo_clamp:
                idx = 0
            else:
                ghost[k] = True
                return
        H[k][idx] += 1

    def trunc(x):
        if x != x or x in (float("inf"), float("-inf")):
            return float("nan")
        return float(int(x))

    def log10(x):
        return L.wsa_or_log10(float(x)) if x == x else float("nan")

    def div(a, b):
        if b == 0:
            return float("nan") if a == 0 or a != a else (float("inf") if a > 0 else float("-inf"))
        return a / b

    prev_end = segs[0][0]
    for r in range(len(syl_ci)):
        seg_len = segs[r][1]
        osum = 0
        for e, (st, sl) in enumerate(syl_ci[r]):
            fr = np.asarray(frames[r][st:st + sl], dtype=np.float64)
            a = i_ = l_ = s_ = c_ = u_ = f_ = d_ = h_ = p_ = 0.0
            for o in range(sl):
                F = fr[o]
                if F[0] > 0:
                    c_ += 1; a += F[0]; i_ += F[1]; l_ += F[2]
                    if o > 0:
                        s_ += F[0] - fr[o - 1][0]
                if F[3] > 0:
                    p_ += 1; u_ += F[3]; f_ += F[4]; d_ += F[5]
                    if o > 0:
                        h_ += F[3] - fr[o - 1][3]
            a, i_, l_, u_, f_, d_ = div(a, c_), div(i_, c_), div(l_, c_), div(u_, p_), div(f_, p_), div(d_, p_)
            bump("c", trunc(sl / 2))
            bump("u", trunc(a / 2))
            bump("f", trunc(u_ / 2))
            bump("d", trunc(3 * log10(i_)))
            bump("h", trunc(4 * log10(f_)))
            bump("p", trunc(l_ / 2))
            bump("m", trunc(d_ / 2))
            bump("g", trunc(10 * (sl - c_) / sl))
            bump("y", trunc(10 * (sl - p_) / sl))
            bump("v", trunc(20 * (s_ + 50) / 100), lo_clamp=True)
            bump("x", trunc(20 * (h_ + 50) / 100), lo_clamp=True)
            osum += sl
        bump("i", trunc(10 * seg_len / 150))
        bump("o", float(len(syl_ci[r])))
        bump("l", trunc(10 * (segs[r][0] - prev_end) / 150))
        bump("s", trunc(2 * (div(osum, seg_len) - .3) * 10), lo_clamp=True)
        prev_end = segs[r][0] + segs[r][1]
    out = []
    for k in "iolscufdhpmgyvx":
        tot = sum(H[k])
        out += [x / tot for x in H[k]] if (tot > 0 and not ghost[k]) else list(H[k])
    return np.array(out, dtype=np.float64)


def formant_features(fr9, ctx_max, floor, cs):
    L = lib()
    fr9 = np.ascontiguousarray(fr9, dtype=np.float32)
    out = np.zeros(53)
    L.wsa_or_formant_features(fr9.ctypes.data, fr9.shape[0], ctx_max, floor, cs, out.ctypes.data)
    return out
