"""ctypes binding of include/wsa.h (libwsa.so).  No CPU fallback: if the HIP library is missing or no
gfx950 device is present, construction raises."""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
NFEAT = 53


class WsaError(RuntimeError):
    pass


class _Config(ctypes.Structure):
    _fields_ = [("spec_type", ctypes.c_int32), ("output_level", ctypes.c_int32),
                ("f_min", ctypes.c_double), ("f_max", ctypes.c_double),
                ("N_fft_bins", ctypes.c_int32), ("N_mel_bins", ctypes.c_int32),
                ("window_width", ctypes.c_double), ("window_step", ctypes.c_double),
                ("pause_length", ctypes.c_double), ("min_seg_length", ctypes.c_double),
                ("auto_noise_gate", ctypes.c_int32),
                ("voiced_max_dB", ctypes.c_double), ("voiced_min_dB", ctypes.c_double),
                ("pre_norm_gain", ctypes.c_double), ("high_f_emph", ctypes.c_double)]


class _Geometry(ctypes.Structure):
    _fields_ = [(k, ctypes.c_int32) for k in ("nfft", "win", "hop", "bands", "kmax")]


class _DeviceResult(ctypes.Structure):
    _fields_ = [("n_clips", ctypes.c_uint32), ("n_rows", ctypes.c_uint32), ("n_segments", ctypes.c_uint32),
                ("n_frames_total", ctypes.c_uint32), ("status_flags", ctypes.c_uint32),
                ("d_row_meta", ctypes.c_void_p), ("d_row_feat", ctypes.c_void_p), ("d_segments", ctypes.c_void_p),
                ("d_clip_row_off", ctypes.c_void_p), ("d_clip_seg_off", ctypes.c_void_p),
                ("d_spectra", ctypes.c_void_p), ("d_clip_frame_off", ctypes.c_void_p), ("d_formants", ctypes.c_void_p),
                ("n_utterance_rows", ctypes.c_uint32), ("d_utt_meta", ctypes.c_void_p), ("d_utt_feat", ctypes.c_void_p),
                ("d_clip_utt_off", ctypes.c_void_p)]


def _track_records(seg_off, pts, rk, n_segments):
    """the reference's 18-field track records per segment from the per-point entries (layout: include/wsa.h wsa_batch_copy_tracks)."""
    num = lambda x: int(x) if float(x).is_integer() else float(x)
    out = []
    for k in range(n_segments):
        p0, p1 = int(seg_off[k][0]), int(seg_off[k + 1][0])
        r0, r1 = int(seg_off[k][1]), int(seg_off[k + 1][1])
        per = {}
        for q in pts[p0:p1]:
            per.setdefault(int(q[0]), []).append(q)
        seg = []
        for t in rk[r0:r1]:
            P = per[int(t)]
            frames = [int(q[6]) for q in P]; starts = [int(q[4]) for q in P]; ends = [int(q[7]) for q in P]
            bins = [int(q[1]) & 0xff for q in P]; amps = [int(np.uint32(q[5])) for q in P]
            en = [float(np.array([q[2], q[3]], np.int32).view(np.float64)[0]) for q in P]
            h = len(P) - 1                                   # the velocity of the last update (ref @B36624), from the bins before it
            pb = bins[-1]
            vel = 0.0 if h == 0 else (pb - bins[0] if h == 1 else (((pb - bins[1]) + (bins[0] - bins[1])) / 2 if h == 2
                                      else ((pb - bins[h - 1]) + (bins[h - 2] - bins[h - 1]) + (bins[h - 3] - bins[h - 2])) / 3))
            sE = 0.0; sEb = 0.0; sW = 0
            for b_, e_, st_, en_ in zip(bins, en, starts, ends):
                sE += e_; sEb += e_ * b_; sW += en_ - st_ + 1
            seg.append([starts[-1], ends[-1], frames[-1], frames[-1], num(vel), pb, amps[-1], frames, starts, ends, bins, amps,
                        [num(e_) for e_ in en], num(sE), len(P), num(sEb), 0, sW])
        out.append(seg)
    return out


class _StreamRows(ctypes.Structure):
    _fields_ = [("n_rows", ctypes.c_uint32), ("n_segments", ctypes.c_uint32), ("status_flags", ctypes.c_uint32),
                ("row_meta", ctypes.c_void_p), ("row_feat", ctypes.c_void_p), ("segments", ctypes.c_void_p), ("stream_cuts", ctypes.c_void_p),
                ("formants", ctypes.c_void_p), ("row_formant_off", ctypes.c_void_p),
                ("n_utterance_rows", ctypes.c_uint32), ("utt_meta", ctypes.c_void_p), ("utt_feat", ctypes.c_void_p),
                ("n_track_points", ctypes.c_uint64), ("n_track_ranked", ctypes.c_uint64),
                ("track_off", ctypes.c_void_p), ("track_points", ctypes.c_void_p), ("track_ranked", ctypes.c_void_p)]


class _BatchInfo(ctypes.Structure):
    _fields_ = [("n_clips", ctypes.c_uint32), ("n_frames_total", ctypes.c_uint32),
                ("max_frames_per_clip", ctypes.c_uint32), ("bands", ctypes.c_uint32),
                ("rows_cap", ctypes.c_uint32), ("segments_cap", ctypes.c_uint32),
                ("workspace_bytes", ctypes.c_uint64)]


# every symbol include/wsa.h declares (checked by tests/test_abi.py)
ABI_VERSION = 5            # WSA_ABI_VERSION of include/wsa.h this binding's structures follow
ABI_SYMBOLS = ["wsa_config_default", "wsa_abi_version", "wsa_create", "wsa_destroy", "wsa_last_error",
               "wsa_geometry_for", "wsa_bins_hz", "wsa_batch_create", "wsa_batch_destroy", "wsa_batch_run",
               "wsa_batch_run_host", "wsa_batch_result", "wsa_batch_copy_rows", "wsa_batch_copy_spectra",
               "wsa_batch_get_info", "wsa_batch_stage_ms", "wsa_batch_enable_timing", "wsa_batch_run_frontend",
               "wsa_batch_run_backend", "wsa_batch_enable_trace", "wsa_batch_copy_trace", "wsa_batch_copy_formants", "wsa_batch_copy_utterance",
               "wsa_batch_tracks_info", "wsa_batch_copy_tracks", "wsa_batch_create_resampled", "wsa_resample_length", "wsa_batch_copy_pcm",
               "wsa_stream_create", "wsa_stream_destroy", "wsa_stream_samples_per_step", "wsa_stream_step",
               "wsa_stream_host_input", "wsa_stream_step_host", "wsa_stream_collect", "wsa_stream_enable_graph",
               "wsa_batch_keep_spectra", "wsa_batch_backend_reruns", "wsa_stream_time_steps", "wsa_batch_run_host_i16",
               "wsa_gather_create", "wsa_gather_destroy", "wsa_gather_rows", "wsa_gather_copy_rows", "wsa_host_alloc", "wsa_host_free", "wsa_queue_create", "wsa_queue_destroy"]

_LIB = None


def library_path():
    # WSA_LIB_DIR: another build of the same sources (A/B timing of two builds in one GPU session, tools/README.md); never a fallback
    return os.path.join(os.environ.get("WSA_LIB_DIR") or os.path.join(_HERE, "lib"), "libwsa.so")


def build_library():
    """hipcc cross-compiles for gfx950 without a GPU (used by __graft_entry__.build())."""
    subprocess.run(["make", "-s", "-j4", "-C", os.path.join(_HERE, "csrc")], check=True)
    return library_path()


def lib():
    global _LIB
    if _LIB is not None:
        return _LIB
    path = library_path()
    if not os.path.exists(path):
        raise WsaError(f"{path} is missing: build it with `make -C webspeechanalyzer_amd/csrc` "
                       "(python -c 'import __graft_entry__ as g; g.build()'). There is no CPU fallback.")
    # PyTorch-ROCm bundles its own HIP runtime (soname libamdhip64.so.7).  Device pointers and streams
    # handed to libwsa come from torch, so both must share ONE runtime: load torch's first, then the
    # loader satisfies libwsa's DT_NEEDED libamdhip64.so.7 from the copy already in the process.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    L = ctypes.CDLL(path)
    vp, i32, u32, u64, dbl = ctypes.c_void_p, ctypes.c_int32, ctypes.c_uint32, ctypes.c_uint64, ctypes.c_double
    L.wsa_abi_version.restype = ctypes.c_int
    if L.wsa_abi_version() != ABI_VERSION:
        raise WsaError(f"{path} has ABI version {L.wsa_abi_version()}, this binding is written against {ABI_VERSION} (include/wsa.h): rebuild the library")
    L.wsa_config_default.argtypes = [ctypes.POINTER(_Config)]
    L.wsa_create.argtypes = [ctypes.POINTER(_Config), i32, ctypes.POINTER(vp)]
    L.wsa_destroy.argtypes = [vp]
    L.wsa_last_error.restype = ctypes.c_char_p
    L.wsa_last_error.argtypes = [vp]
    L.wsa_geometry_for.argtypes = [vp, dbl, ctypes.POINTER(_Geometry)]
    L.wsa_bins_hz.argtypes = [vp, dbl, vp, i32]
    L.wsa_batch_create.argtypes = [vp, u32, vp, dbl, ctypes.POINTER(vp)]
    L.wsa_batch_create_resampled.argtypes = [vp, u32, vp, dbl, dbl, ctypes.POINTER(vp)]
    L.wsa_resample_length.argtypes = [u64, dbl, dbl]
    L.wsa_resample_length.restype = ctypes.c_uint64
    L.wsa_batch_copy_pcm.argtypes = [vp, vp, vp, u64]
    L.wsa_batch_destroy.argtypes = [vp]
    L.wsa_batch_run.argtypes = [vp, vp, u64, vp]
    L.wsa_batch_run_host.argtypes = [vp, vp, vp]
    L.wsa_batch_run_host_i16.argtypes = [vp, vp, vp, vp]
    L.wsa_batch_run_frontend.argtypes = [vp, vp, u64, vp]
    L.wsa_batch_run_backend.argtypes = [vp, vp, vp]
    L.wsa_batch_result.argtypes = [vp, vp, ctypes.POINTER(_DeviceResult)]
    L.wsa_batch_copy_rows.argtypes = [vp, vp, vp, vp, u32, vp, u32, vp, vp]
    L.wsa_batch_copy_spectra.argtypes = [vp, vp, vp, u64, vp]
    L.wsa_batch_get_info.argtypes = [vp, ctypes.POINTER(_BatchInfo)]
    L.wsa_batch_stage_ms.argtypes = [vp, vp]
    L.wsa_batch_enable_timing.argtypes = [vp, i32]
    L.wsa_batch_enable_trace.argtypes = [vp, i32]
    L.wsa_batch_keep_spectra.argtypes = [vp, i32]
    L.wsa_batch_backend_reruns.argtypes = [vp, vp]
    L.wsa_batch_copy_trace.argtypes = [vp, vp, vp, u64]
    L.wsa_batch_copy_formants.argtypes = [vp, vp, vp, u64]
    L.wsa_batch_tracks_info.argtypes = [vp, vp, vp]
    L.wsa_batch_copy_tracks.argtypes = [vp, vp, vp, vp, u64, vp, u64]
    L.wsa_batch_copy_utterance.argtypes = [vp, vp, vp, vp, u32, vp]
    L.wsa_stream_create.argtypes = [vp, u32, dbl, u32, u32, ctypes.POINTER(vp)]
    L.wsa_stream_destroy.argtypes = [vp]
    L.wsa_stream_samples_per_step.argtypes = [vp]
    L.wsa_stream_samples_per_step.restype = u32
    L.wsa_stream_step.argtypes = [vp, vp, u64, vp, vp]
    L.wsa_stream_host_input.argtypes = [vp]
    L.wsa_stream_host_input.restype = ctypes.POINTER(ctypes.c_float)
    L.wsa_stream_step_host.argtypes = [vp, vp, vp]
    L.wsa_stream_collect.argtypes = [vp, vp, ctypes.POINTER(_StreamRows)]
    L.wsa_stream_enable_graph.argtypes = [vp, i32]
    L.wsa_stream_time_steps.argtypes = [vp, u32, vp, u32, vp, vp, vp]
    L.wsa_gather_create.argtypes = [vp, i32, i32, vp]
    L.wsa_gather_destroy.argtypes = [vp]
    L.wsa_gather_rows.argtypes = [vp, vp, vp, vp]
    L.wsa_gather_copy_rows.argtypes = [vp, vp, vp, u32]
    for name in ABI_SYMBOLS:
        if name not in ("wsa_abi_version", "wsa_last_error", "wsa_config_default", "wsa_destroy", "wsa_batch_destroy", "wsa_resample_length",
                        "wsa_stream_destroy", "wsa_stream_samples_per_step", "wsa_stream_host_input", "wsa_gather_destroy", "wsa_host_free"):
            getattr(L, name).restype = ctypes.c_int
    _LIB = L
    return L


class Config(dict):
    """The reference's settings object (defaults dist/main.js:2 @B2965, output_level 4 = Segment Formants)."""

    def __init__(self, **kw):
        c = _Config()
        lib().wsa_config_default(ctypes.byref(c))
        super().__init__({k: getattr(c, k) for k, _ in _Config._fields_})
        for k, v in kw.items():
            if k not in self:
                raise KeyError(k)
            self[k] = v

    def c_struct(self):
        c = _Config()
        for k, t in _Config._fields_:
            setattr(c, k, int(self[k]) if t is ctypes.c_int32 else float(self[k]))
        return c


class Analyzer:
    """One configured context on one GPU (wsa_ctx)."""

    def __init__(self, config=None, device=0):
        self.L = lib()
        self.config = config or Config()
        self.h = ctypes.c_void_p()
        cs = self.config.c_struct()
        st = self.L.wsa_create(ctypes.byref(cs), device, ctypes.byref(self.h))
        if st != 0:
            raise WsaError(f"wsa_create failed ({st}): {self.L.wsa_last_error(None).decode()}")
        self.device = device

    def _check(self, st):
        if st != 0:
            raise WsaError(f"libwsa error {st}: {self.L.wsa_last_error(self.h).decode()}")

    def geometry(self, fs):
        g = _Geometry()
        self._check(self.L.wsa_geometry_for(self.h, float(fs), ctypes.byref(g)))
        return {k: getattr(g, k) for k, _ in _Geometry._fields_}

    def bins_hz(self, fs):
        n = self.geometry(fs)["bands"]
        out = np.zeros(n)
        self._check(self.L.wsa_bins_hz(self.h, float(fs), out.ctypes.data, n))
        return out

    def batch(self, n_samples, fs, resample_to=None):
        """resample_to: analysis rate when the clips handed to run* are at `fs` and are to be converted first (spec RS-1)."""
        return Batch(self, n_samples, fs, resample_to)

    def streams(self, n_streams, fs, frames_per_step=1, max_span_frames=1024):
        return Streams(self, n_streams, fs, frames_per_step, max_span_frames)

    def close(self):
        if self.h:
            self.L.wsa_destroy(self.h)
            self.h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Batch:
    """A planned batch shape (wsa_batch).  `run*` take raw device pointers (e.g. torch data_ptr())."""

    def __init__(self, an, n_samples, fs, resample_to=None):
        self.an, self.L = an, an.L
        self.n_samples = np.ascontiguousarray(n_samples, dtype=np.uint32)
        self.fs = float(fs)
        self.h = ctypes.c_void_p()
        if resample_to:
            self.n_samples_in, self.fs_in, self.fs = self.n_samples, self.fs, float(resample_to)
            an._check(self.L.wsa_batch_create_resampled(an.h, len(self.n_samples_in), self.n_samples_in.ctypes.data, self.fs_in, self.fs, ctypes.byref(self.h)))
            self.n_samples = np.array([self.L.wsa_resample_length(int(n), self.fs_in, self.fs) for n in self.n_samples_in], np.uint32)
        else:
            an._check(self.L.wsa_batch_create(an.h, len(self.n_samples), self.n_samples.ctypes.data, self.fs, ctypes.byref(self.h)))
        info = _BatchInfo()
        an._check(self.L.wsa_batch_get_info(self.h, ctypes.byref(info)))
        self.info = {k: getattr(info, k) for k, _ in _BatchInfo._fields_}

    def run(self, d_pcm, clip_stride, stream=0):
        self.an._check(self.L.wsa_batch_run(self.h, d_pcm, int(clip_stride), stream))

    def run_frontend(self, d_pcm, clip_stride, stream=0):
        self.an._check(self.L.wsa_batch_run_frontend(self.h, d_pcm, int(clip_stride), stream))

    def run_backend(self, d_spectra, stream=0):
        self.an._check(self.L.wsa_batch_run_backend(self.h, d_spectra, stream))

    def converted_pcm(self, stream=0):
        """Resampling batch, after a run: the clips at the analysis rate, [n_clips, longest] float32 (zero padded)."""
        stride = max(int(self.n_samples.max()) if len(self.n_samples) else 0, 1)
        out = np.zeros((len(self.n_samples), stride), np.float32)
        self.an._check(self.L.wsa_batch_copy_pcm(self.h, stream, out.ctypes.data, stride))
        return out

    def run_host(self, clips, stream=0):
        clips = [np.ascontiguousarray(c, dtype=np.float32) for c in clips]
        ptrs = (ctypes.c_void_p * len(clips))(*[c.ctypes.data for c in clips])
        self.an._check(self.L.wsa_batch_run_host(self.h, ptrs, stream))

    def run_host_i16(self, clips, channels=None, stream=0):
        """16-bit PCM in host memory (clip i interleaved over channels[i] channels, channel 0 analysed); converted on the device."""
        clips = [np.ascontiguousarray(c, dtype=np.int16) for c in clips]
        ptrs = (ctypes.c_void_p * len(clips))(*[c.ctypes.data for c in clips])
        ch = None if channels is None else (ctypes.c_uint32 * len(clips))(*[int(c) for c in channels])
        self.an._check(self.L.wsa_batch_run_host_i16(self.h, ptrs, ch, stream))

    def enable_timing(self, on):
        self.an._check(self.L.wsa_batch_enable_timing(self.h, int(on)))

    def keep_spectra(self, on=True):
        """Kept for older hosts: the u32 frames are always stored (spectra())."""
        self.an._check(self.L.wsa_batch_keep_spectra(self.h, int(on)))
        return self

    def backend_reruns(self):
        n = ctypes.c_uint32(0)
        self.an._check(self.L.wsa_batch_backend_reruns(self.h, ctypes.byref(n)))
        return n.value

    def enable_trace(self, on=True):
        self.an._check(self.L.wsa_batch_enable_trace(self.h, int(on)))

    def trace(self, stream=0):
        n = self.info["n_frames_total"]
        out = np.zeros((n, 12))
        self.an._check(self.L.wsa_batch_copy_trace(self.h, stream, out.ctypes.data, max(n, 1)))
        return out

    def device_result(self, stream=0):
        r = _DeviceResult()
        self.an._check(self.L.wsa_batch_result(self.h, stream, ctypes.byref(r)))
        return r

    def stage_ms(self):
        out = np.zeros(4, np.float32)
        self.an._check(self.L.wsa_batch_stage_ms(self.h, out.ctypes.data))
        return out

    def rows(self, stream=0):
        """Host copies: dict(meta [n,8] i32, feat [n,53] f64, segments [m,4] i32, row_off, seg_off)."""
        r = self.device_result(stream)
        n, m, nc = r.n_rows, r.n_segments, r.n_clips
        meta = np.zeros((n, 8), np.int32)
        feat = np.zeros((n, NFEAT), np.float64)
        segs = np.zeros((m, 4), np.int32)
        roff = np.zeros(nc + 1, np.uint32)
        soff = np.zeros(nc + 1, np.uint32)
        self.an._check(self.L.wsa_batch_copy_rows(self.h, stream, meta.ctypes.data, feat.ctypes.data, max(n, 1),
                                                   segs.ctypes.data, max(m, 1), roff.ctypes.data, soff.ctypes.data))
        return dict(meta=meta, feat=feat, segments=segs, row_off=roff, seg_off=soff)

    def spectra(self, stream=0):
        words = self.info["n_frames_total"] * self.info["bands"]
        out = np.zeros((self.info["n_frames_total"], self.info["bands"]), np.uint32)
        foff = np.zeros(self.info["n_clips"] + 1, np.uint32)
        self.an._check(self.L.wsa_batch_copy_spectra(self.h, stream, out.ctypes.data, max(words, 1), foff.ctypes.data))
        return out, foff

    def formants(self, stream=0):
        """levels 4 / 10: the [n_frames_total, 9] float32 table of straightened frames."""
        n = self.info["n_frames_total"]
        out = np.zeros((n, 9), np.float32)
        self.an._check(self.L.wsa_batch_copy_formants(self.h, stream, out.ctypes.data, max(n, 1)))
        return out

    def utterance(self, stream=0):
        """level 11: dict(meta [n,4] i32 = {clip, result index, t_start, t_sum}, feat [n,264] f64, off [n_clips+1])."""
        r = self.device_result(stream)
        n = r.n_utterance_rows
        meta = np.zeros((n, 4), np.int32)
        feat = np.zeros((n, 264), np.float64)
        off = np.zeros(len(self.n_samples) + 1, np.uint32)
        self.an._check(self.L.wsa_batch_copy_utterance(self.h, stream, meta.ctypes.data, feat.ctypes.data, max(n, 1), off.ctypes.data))
        return dict(meta=meta, feat=feat, off=off)

    def tracks(self, stream=0):
        """Level 3: per segment (in d_segments order) the ranked raw tracks as the reference's 18-field records
        (ref accumulate_fm @B35952; field map SURVEY.md App. A), rebuilt from the per-point entries the device hands out:
        [0] start [1] end [2],[3] last frame [4] velocity [5] last bin [6] last amp [7] frames [8] starts [9] ends [10] bins
        [11] amps [12] energies [13] sum E [14] count [15] sum E*bin [16] 0 [17] sum width."""
        class _TI(ctypes.Structure):
            _fields_ = [("n_segments", ctypes.c_uint32), ("n_points", ctypes.c_uint64), ("n_ranked", ctypes.c_uint64)]
        ti = _TI()
        self.an._check(self.L.wsa_batch_tracks_info(self.h, stream, ctypes.byref(ti)))
        seg_off = np.zeros((ti.n_segments + 1, 2), np.uint64)
        pts = np.zeros((max(int(ti.n_points), 1), 8), np.int32)
        rk = np.zeros(max(int(ti.n_ranked), 1), np.int32)
        self.an._check(self.L.wsa_batch_copy_tracks(self.h, stream, seg_off.ctypes.data, pts.ctypes.data, int(ti.n_points), rk.ctypes.data, int(ti.n_ranked)))
        return _track_records(seg_off, pts, rk, ti.n_segments)

    def callbacks(self, stream=0):
        """Per clip, the callback sequence of the reference's dispatcher (dist/main.js:2 @B28869) in the
        same shape tests/golden/gen/ref_driver.js records: [si, label, seg_time, features]."""
        r = self.rows(stream)
        level = int(self.an.config["output_level"])
        step = float(self.an.config["window_step"]) / 1e3
        utt = self.utterance(stream) if level == 11 else None
        trk = self.tracks(stream) if level == 3 else None
        fm = self.formants(stream) if level in (4, 10) else None
        foff = None
        if fm is not None:
            foff = np.zeros(len(self.n_samples) + 1, np.uint32)
            self.an._check(self.L.wsa_batch_copy_spectra(self.h, stream, None, 0, foff.ctypes.data))
        out = []
        for c in range(len(self.n_samples)):
            a, b = int(r["row_off"][c]), int(r["row_off"][c + 1])
            meta, feat = r["meta"][a:b], r["feat"][a:b]
            cbs = []
            payload = (lambda m, f: f[:23].copy() if level == 12 else f.copy()) if fm is None else (lambda m, f: fm[int(foff[c]) + m[6]: int(foff[c]) + m[6] + m[7]].copy())
            if level == 11:
                for k in range(int(utt["off"][c]), int(utt["off"][c + 1])):
                    m = utt["meta"][k]
                    cbs.append([0, [], [m[2] * step, (m[3] + 1) * step], utt["feat"][k].copy()])
            elif level in (4, 5):
                for m, f in zip(meta, feat):
                    cbs.append([int(m[1]), [], [m[2] * step, (m[3] + 1) * step], payload(m, f)])
            elif level in (10, 12, 13):
                i = 0
                while i < len(meta):
                    j = i
                    while j < len(meta) and meta[j][1] == meta[i][1]:
                        j += 1
                    tm = [["%.3f" % (m[2] * step), "%.3f" % ((m[3] + 1) * step)] for m in meta[i:j]]
                    rows_ = list(zip(meta[i:j], feat[i:j]))
                    if level == 12:                  # numeric threw on a syllable: the reference keeps the rows before it
                        cut = next((q for q, (_, f) in enumerate(rows_) if f[23] != 0), len(rows_))
                        rows_ = rows_[:cut]
                    if level != 12 or rows_:
                        cbs.append([int(meta[i][1]), [], tm, [payload(m, f) for m, f in rows_]])
                    i = j
            sa, sb = int(r["seg_off"][c]), int(r["seg_off"][c + 1])
            if level == 3:                           # ref @B30132: `s[e].length > 0 && b(e, label, s[e])` (three arguments)
                cbs = [[k, [], trk[sa + k]] for k in range(sb - sa) if len(trk[sa + k]) > 0]
            out.append(dict(callbacks=cbs, segments_ci=[[int(s[1]), int(s[2])] for s in r["segments"][sa:sb]],
                            flags=[int(s[3]) for s in r["segments"][sa:sb]], meta=meta))
        return out

    def close(self):
        if self.h:
            self.L.wsa_batch_destroy(self.h)
            self.h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class _GatherResult(ctypes.Structure):
    _fields_ = [("n_ranks", ctypes.c_uint32), ("n_rows", ctypes.c_uint32), ("rows_per_rank", ctypes.POINTER(ctypes.c_uint32)),
                ("d_row_meta", ctypes.c_void_p), ("d_row_feat", ctypes.c_void_p)]


class Gather:
    """wsa_gather: the feature rows of the batches of several contexts (one per GPU, this process) collected on the root's device
    with one grouped RCCL exchange (include/wsa.h)."""

    def __init__(self, analyzers, root=0):
        self.L = analyzers[0].L
        self.ans = list(analyzers)
        self.h = ctypes.c_void_p()
        arr = (ctypes.c_void_p * len(self.ans))(*[a.h for a in self.ans])
        st = self.L.wsa_gather_create(arr, len(self.ans), int(root), ctypes.byref(self.h))
        if st != 0:
            raise WsaError(f"wsa_gather_create failed ({st}): {self.L.wsa_last_error(self.ans[root].h).decode()}")
        self.root = root

    def rows(self, batches, streams=None):
        """-> (rows_per_rank, meta [n, 8] i32, feat [n, 53] f64) on the host, rank after rank"""
        n = len(self.ans)
        ba = (ctypes.c_void_p * n)(*[b.h for b in batches])
        sa = (ctypes.c_void_p * n)(*[int(s) for s in streams]) if streams is not None else None
        r = _GatherResult()
        self.ans[self.root]._check(self.L.wsa_gather_rows(self.h, ba, sa, ctypes.byref(r)))
        per = [int(r.rows_per_rank[i]) for i in range(n)]
        meta = np.zeros((r.n_rows, 8), np.int32)
        feat = np.zeros((r.n_rows, NFEAT), np.float64)
        self.ans[self.root]._check(self.L.wsa_gather_copy_rows(self.h, meta.ctypes.data, feat.ctypes.data, max(int(r.n_rows), 1)))
        return per, meta, feat

    def close(self):
        if self.h:
            self.L.wsa_gather_destroy(self.h)
            self.h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


ACTIVE, START, STOP = 1, 2, 4        # WSA_STREAM_* control bits


class Streams:
    """n concurrent launches advancing in lock step (wsa_stream): the reference's online path — one
    spectrum_push per frame with carried state, callbacks as segments close (dist/main.js:2 @B8752, @B28869)."""

    def __init__(self, an, n_streams, fs, frames_per_step=1, max_span_frames=1024):
        self.an, self.L = an, an.L
        self.n = int(n_streams)
        self.h = ctypes.c_void_p()
        an._check(self.L.wsa_stream_create(an.h, self.n, float(fs), int(frames_per_step), int(max_span_frames), ctypes.byref(self.h)))
        self.samples_per_step = int(self.L.wsa_stream_samples_per_step(self.h))

    def enable_graph(self, on=True):
        self.an._check(self.L.wsa_stream_enable_graph(self.h, int(on)))

    def host_input(self):
        """The pinned [n, samples_per_step] float32 input buffer of step_host (a numpy view)."""
        p = self.L.wsa_stream_host_input(self.h)
        return np.ctypeslib.as_array(p, shape=(self.n, self.samples_per_step))

    @staticmethod
    def _ctl(ctl):
        if ctl is None:
            return None, None
        a = np.ascontiguousarray(ctl, dtype=np.uint8)
        return a, a.ctypes.data

    def step(self, d_pcm, stream_stride, ctl=None, stream=0):
        keep, ptr = self._ctl(ctl)
        self.an._check(self.L.wsa_stream_step(self.h, d_pcm, int(stream_stride), ptr, stream))

    def step_host(self, ctl=None, stream=0):
        keep, ptr = self._ctl(ctl)
        self.an._check(self.L.wsa_stream_step_host(self.h, ptr, stream))

    def time_steps(self, n_steps, feed=None, stream=0):
        """n_steps steps timed inside the library (step_host + collect, microseconds each); feed = [k, n, samples_per_step] float32
        blocks copied into the pinned input before each step (cycled), or None.  Returns (us array, rows produced)."""
        out = np.zeros(n_steps)
        rows = ctypes.c_uint64(0)
        fptr, fk = None, 0
        if feed is not None:
            feed = np.ascontiguousarray(feed, dtype=np.float32)
            assert feed.ndim == 3 and feed.shape[1:] == (self.n, self.samples_per_step)
            fptr, fk = feed.ctypes.data, feed.shape[0]
        self.an._check(self.L.wsa_stream_time_steps(self.h, int(n_steps), fptr, int(fk), stream, out.ctypes.data, ctypes.byref(rows)))
        return out, rows.value

    def collect(self, stream=0):
        """Rows of the last step: dict(meta [n,8] i32, feat [n,53] f64, segments [m,4] i32) (copies)."""
        r = _StreamRows()
        self.an._check(self.L.wsa_stream_collect(self.h, stream, ctypes.byref(r)))
        n, m = r.n_rows, r.n_segments
        meta = np.ctypeslib.as_array(ctypes.cast(r.row_meta, ctypes.POINTER(ctypes.c_int32)), shape=(n, 8)).copy() if n else np.zeros((0, 8), np.int32)
        feat = np.ctypeslib.as_array(ctypes.cast(r.row_feat, ctypes.POINTER(ctypes.c_double)), shape=(n, NFEAT)).copy() if n else np.zeros((0, NFEAT))
        segs = np.ctypeslib.as_array(ctypes.cast(r.segments, ctypes.POINTER(ctypes.c_int32)), shape=(m, 4)).copy() if m else np.zeros((0, 4), np.int32)
        cuts = np.ctypeslib.as_array(ctypes.cast(r.stream_cuts, ctypes.POINTER(ctypes.c_uint32)), shape=(self.n,)).copy()
        out = dict(meta=meta, feat=feat, segments=segs, cuts=cuts, flags=int(r.status_flags))
        if r.utt_feat:                                  # level 11: one 264-vector per result of the step
            nu = int(r.n_utterance_rows)
            out["utt_meta"] = np.ctypeslib.as_array(ctypes.cast(r.utt_meta, ctypes.POINTER(ctypes.c_int32)), shape=(nu, 4)).copy() if nu else np.zeros((0, 4), np.int32)
            out["utt_feat"] = np.ctypeslib.as_array(ctypes.cast(r.utt_feat, ctypes.POINTER(ctypes.c_double)), shape=(nu, 264)).copy() if nu else np.zeros((0, 264))
        if r.track_off:                                 # level 3: ranked raw tracks of the step's segments (the layout of Batch.tracks())
            npt, nrk = int(r.n_track_points), int(r.n_track_ranked)
            out["track_off"] = np.ctypeslib.as_array(ctypes.cast(r.track_off, ctypes.POINTER(ctypes.c_uint64)), shape=(m + 1, 2)).copy()
            out["track_points"] = np.ctypeslib.as_array(ctypes.cast(r.track_points, ctypes.POINTER(ctypes.c_int32)), shape=(npt, 8)).copy() if npt else np.zeros((0, 8), np.int32)
            out["track_ranked"] = np.ctypeslib.as_array(ctypes.cast(r.track_ranked, ctypes.POINTER(ctypes.c_int32)), shape=(nrk,)).copy() if nrk else np.zeros((0,), np.int32)
            out["tracks"] = _track_records(out["track_off"], out["track_points"], out["track_ranked"], m)
        if r.formants and r.row_formant_off:           # levels 4 / 10: frames of row k = formants[formant_off[k]:formant_off[k + 1]]
            off = np.ctypeslib.as_array(ctypes.cast(r.row_formant_off, ctypes.POINTER(ctypes.c_uint32)), shape=(r.n_rows + 1,)).copy()
            out["formant_off"] = off
            out["formants"] = (np.ctypeslib.as_array(ctypes.cast(r.formants, ctypes.POINTER(ctypes.c_float)), shape=(int(off[-1]), 9)).copy()
                               if off[-1] else np.zeros((0, 9), np.float32))
        return out

    def close(self):
        if self.h:
            self.L.wsa_stream_destroy(self.h)
            self.h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
