"""Streams (wsa_stream_*, BASELINE config "streaming"): feeding a signal step by step gives exactly the
rows of one batch clip holding the whole signal, and those equal the oracle's callbacks."""
import numpy as np
import pytest
import torch

from tests.util import callbacks_equal

pytestmark = pytest.mark.gpu


def _stream():
    return torch.cuda.current_stream().cuda_stream


@pytest.fixture(scope="module")
def wsa():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    import webspeechanalyzer_amd as w
    return w


def _per_stream_callbacks(rows_per_step, n, level, step_s):
    """[(meta, feat)] of every step -> per stream the callback list in the shape Batch.callbacks() gives."""
    out = [[] for _ in range(n)]
    seen = [0] * n
    for r in rows_per_step:
        if level == 3:                   # (segment index since START, label, ranked raw tracks) for every segment with tracks
            for sg, tr in zip(r["segments"], r["tracks"]):
                if len(tr) > 0:
                    out[int(sg[0])].append([seen[int(sg[0])], [], tr])
                seen[int(sg[0])] += 1
            continue
        if level == 11:                  # one callback per result: (0, label, Y(), 264 utterance features over everything so far)
            for m, f in zip(r["utt_meta"], r["utt_feat"]):
                out[int(m[0])].append([0, [], [m[2] * step_s, (m[3] + 1) * step_s], f.copy()])
            continue
        meta, feat = r["meta"], r["feat"]
        i = 0
        while i < len(meta):
            j = i
            while j < len(meta) and meta[j][0] == meta[i][0] and meta[j][1] == meta[i][1]:
                j += 1
            s = int(meta[i][0])
            frames = (lambda k: r["formants"][int(r["formant_off"][k]):int(r["formant_off"][k + 1])].copy()) if "formants" in r else None
            if level in (5, 4):
                for k, (m, f) in enumerate(zip(meta[i:j], feat[i:j])):
                    out[s].append([int(m[1]), [], [m[2] * step_s, (m[3] + 1) * step_s], f.copy() if level == 5 else frames(i + k)])
            elif level == 12:            # 23 numbers per syllable; a syllable on which numeric threw (slot 23 set) ends the segment's list
                tm = [["%.3f" % (m[2] * step_s), "%.3f" % ((m[3] + 1) * step_s)] for m in meta[i:j]]
                cut = next((q for q, f in enumerate(feat[i:j]) if f[23] != 0), j - i)
                if cut:
                    out[s].append([int(meta[i][1]), [], tm, [f[:23].copy() for f in feat[i:i + cut]]])
            else:
                tm = [["%.3f" % (m[2] * step_s), "%.3f" % ((m[3] + 1) * step_s)] for m in meta[i:j]]
                out[s].append([int(meta[i][1]), [], tm, [f.copy() for f in feat[i:j]] if level == 13 else [frames(k) for k in range(i, j)]])
            i = j
    return out


def _run_streams(wsa, pcm, fs, level, F, graph, host_in, cfg_kw=None, max_span=1024):
    """pcm [n, ns] on the GPU -> (per-stream callbacks, per-stream segments) fed F frames per step."""
    n, ns = pcm.shape
    an = wsa.Analyzer(wsa.Config(output_level=level, **(cfg_kw or {})))
    g = an.geometry(fs)
    st = an.streams(n, fs, frames_per_step=F, max_span_frames=max_span)
    st.enable_graph(graph)
    sps = st.samples_per_step
    assert sps == F * g["hop"]
    nsteps = ns // sps
    rows, segs = [], [[] for _ in range(n)]
    buf = torch.zeros((n, sps), device="cuda", dtype=torch.float32)
    hin = st.host_input() if host_in else None
    for k in range(nsteps):
        ctl = np.full(n, wsa.ACTIVE, np.uint8)
        if k == 0:
            ctl |= wsa.START
        if k == nsteps - 1:
            ctl |= wsa.STOP
        chunk = pcm[:, k * sps:(k + 1) * sps]
        if host_in:
            hin[:] = chunk.cpu().numpy()
            st.step_host(ctl, _stream())
        else:
            buf.copy_(chunk)
            st.step(buf.data_ptr(), buf.stride(0), ctl, _stream())
        r = st.collect(_stream())
        rows.append(r)
        for sg in r["segments"]:
            segs[int(sg[0])].append([int(sg[1]), int(sg[2])])
    st.close(); an.close()
    step_s = float(wsa.Config(**(cfg_kw or {}))["window_step"]) / 1e3
    return _per_stream_callbacks(rows, n, level, step_s), segs, nsteps * sps


@pytest.mark.parametrize("level,F,graph,host_in", [(5, 1, True, False), (5, 4, False, False), (13, 1, True, True), (13, 7, True, False), (5, 40, True, True),
                                                     (4, 1, True, False), (4, 5, False, True), (10, 1, True, True), (10, 16, True, False), (12, 1, True, False), (12, 9, True, True), (11, 1, True, False), (11, 6, True, True),
                                                     (3, 1, True, False), (3, 6, True, True), (3, 33, False, False)])
def test_stream_steps_equal_one_clip_and_the_oracle(wsa, level, F, graph, host_in):
    from oracle import pyoracle
    from webspeechanalyzer_amd.synth import synth_clips
    fs, n = 16000, 12
    pcm = synth_clips(n, 5 * fs, fs=fs, seed=61, device="cuda")
    got, segs, used = _run_streams(wsa, pcm, fs, level, F, graph, host_in)
    fe = pyoracle.FrontEnd(pyoracle.fe_cfg(fs=fs))
    host = pcm[:, :used].cpu().numpy()
    nseg = 0
    for c in range(n):
        ref = pyoracle.run_backend(fe.run(host[c]), pyoracle.default_cfg(level=level))
        assert ref["segments_ci"] == segs[c], f"stream {c}"
        ok, why = callbacks_equal(level, ref["callbacks"], got[c], exact=False, tol=1e-4)
        assert ok, f"stream {c}: {why}"
        nseg += len(ref["segments_ci"])
    assert nseg > 20


def test_stream_overlapping_windows_and_48k(wsa):
    """window 30 ms / step 10 ms (two hops of history, warm-up frames skipped) at 48 kHz (3072-point FFT)."""
    from oracle import pyoracle
    from webspeechanalyzer_amd.synth import synth_clips
    fs, n = 48000, 5
    kw = dict(window_width=30.0, window_step=10.0)
    pcm = synth_clips(n, 4 * fs, fs=fs, seed=71, device="cuda")
    for F in (1, 5):
        got, segs, used = _run_streams(wsa, pcm, fs, 5, F, True, False, cfg_kw=kw)
        fe = pyoracle.FrontEnd(pyoracle.fe_cfg(fs=float(fs), **kw))
        host = pcm[:, :used].cpu().numpy()
        nseg = 0
        for c in range(n):
            ref = pyoracle.run_backend(fe.run(host[c]), pyoracle.default_cfg(level=5, window_step=10.0))
            assert ref["segments_ci"] == segs[c], f"F {F} stream {c}"
            ok, why = callbacks_equal(5, ref["callbacks"], got[c], exact=False, tol=1e-4)
            assert ok, f"F {F} stream {c}: {why}"
            nseg += len(ref["segments_ci"])
        assert nseg > 5


def test_stream_restart_and_idle_streams(wsa):
    """A stream that is idle for some steps, then STARTs again, behaves like a fresh launch."""
    from oracle import pyoracle
    from webspeechanalyzer_amd.synth import synth_clips
    fs, F = 16000, 8
    pcm = synth_clips(2, 4 * fs, fs=fs, seed=81, device="cuda")
    an = wsa.Analyzer(wsa.Config(output_level=5))
    st = an.streams(2, fs, frames_per_step=F)
    st.enable_graph(True)
    sps = st.samples_per_step
    nsteps = pcm.shape[1] // sps
    buf = torch.zeros((2, sps), device="cuda")
    runs = {0: [[], []], 1: [[], []]}
    for rep in range(2):
        for k in range(nsteps + 3):
            ctl = np.zeros(2, np.uint8)
            # stream 0 plays clip 0 in both repetitions; stream 1 idles during the first repetition
            live = [k < nsteps, rep == 1 and k < nsteps]
            for s in range(2):
                if live[s]:
                    ctl[s] = wsa.ACTIVE | (wsa.START if k == 0 else 0) | (wsa.STOP if k == nsteps - 1 else 0)
            buf.copy_(pcm[:, min(k, nsteps - 1) * sps:(min(k, nsteps - 1) + 1) * sps])
            st.step(buf.data_ptr(), buf.stride(0), ctl, _stream())
            r = st.collect(_stream())
            for m, f in zip(r["meta"], r["feat"]):
                runs[int(m[0])][rep].append((int(m[1]), int(m[2]), int(m[3]), f.copy()))
    st.close(); an.close()
    assert runs[1][0] == []
    fe = pyoracle.FrontEnd(pyoracle.fe_cfg(fs=fs))
    host = pcm[:, :nsteps * sps].cpu().numpy()
    for s, rep in ((0, 0), (0, 1), (1, 1)):
        ref = pyoracle.run_backend(fe.run(host[s]), pyoracle.default_cfg(level=5))["callbacks"]
        got = runs[s][rep]
        assert len(ref) == len(got) > 0
        for a, b in zip(ref, got):
            assert a[0] == b[0] and abs(a[2][0] - b[1] * 0.025) < 1e-12 and abs(a[2][1] - (b[2] + 1) * 0.025) < 1e-12
            assert np.allclose(a[3], b[3], rtol=1e-4, atol=1e-6)


def test_stream_span_longer_than_ring_is_cut_for_that_stream_only(wsa):
    """A source that does not pause for max_span_frames is cut there (segment_truncate semantics, counted in stream_cuts) — the
    other streams of the object keep matching the oracle, nothing raises; with room for the span the same signal matches the oracle."""
    from oracle import pyoracle
    from webspeechanalyzer_amd.synth import synth_clips
    fs = 16000
    base = synth_clips(1, 5 * fs, fs=fs, seed=61, device="cpu")[0].numpy()
    fe = pyoracle.FrontEnd(pyoracle.fe_cfg(fs=fs))
    segs = pyoracle.run_backend(fe.run(base), pyoracle.default_cfg(level=5))["segments_ci"]
    start, ln = max(segs, key=lambda s: s[1])
    assert ln >= 10
    mid = (start + ln // 2) * 400
    chunk = base[mid - 1600:mid + 1600]                      # 8 voiced frames, tiled into 20 s without a pause
    sig = np.tile(chunk, 100).astype(np.float32)
    ref = pyoracle.run_backend(fe.run(sig), pyoracle.default_cfg(level=5))["segments_ci"]
    assert max(s[1] for s in ref) > 200, ref                 # the oracle sees one very long segment
    other = synth_clips(1, len(sig), fs=fs, seed=62, device="cpu")[0].numpy()
    pcm = torch.from_numpy(np.stack([sig, other])).cuda().contiguous()
    an = wsa.Analyzer(wsa.Config(output_level=5))
    st = an.streams(2, fs, frames_per_step=16, max_span_frames=64)
    sps = st.samples_per_step
    nsteps = pcm.shape[1] // sps
    rows, seg0, cuts, nflag = [], [], None, 0
    for k in range(nsteps):
        buf = pcm[:, k * sps:(k + 1) * sps].contiguous()
        ctl = np.full(2, wsa.ACTIVE | (wsa.START if k == 0 else 0) | (wsa.STOP if k == nsteps - 1 else 0), np.uint8)
        st.step(buf.data_ptr(), buf.stride(0), ctl, _stream())
        r = st.collect(_stream())                            # no WsaError: the long span is cut, not fatal
        # WSA_FLAG_STREAM_CUT (8) is raised in exactly the steps in which a counter moved
        assert bool(r["flags"] & 8) == (cuts is not None and (r["cuts"] != cuts).any()) or (cuts is None and bool(r["flags"] & 8) == bool(r["cuts"].any()))
        nflag += 1 if r["flags"] & 8 else 0
        rows.append(r); cuts = r["cuts"]
        seg0 += [[int(g[1]), int(g[2])] for g in r["segments"] if g[0] == 0]
    st.close(); an.close()
    assert cuts[0] >= 2 and cuts[1] == 0 and nflag == cuts[0]
    assert len(seg0) >= 3 and max(l for _, l in seg0) <= 128 and sum(l for _, l in seg0) > 0.8 * max(s[1] for s in ref)
    got = _per_stream_callbacks(rows, 2, 5, 0.025)
    ref1 = pyoracle.run_backend(fe.run(other[:nsteps * sps]), pyoracle.default_cfg(level=5))
    ok, why = callbacks_equal(5, ref1["callbacks"], got[1], exact=False, tol=1e-4)
    assert ok and len(got[1]) > 3, why
    # with room for the span the same signal goes through and matches the oracle
    got, segs2, used = _run_streams(wsa, pcm[:1], fs, 5, 16, True, False, max_span=1024)
    ref2 = pyoracle.run_backend(fe.run(sig[:used]), pyoracle.default_cfg(level=5))
    assert ref2["segments_ci"] == segs2[0]
    ok, why = callbacks_equal(5, ref2["callbacks"], got[0], exact=False, tol=1e-4)
    assert ok, why


@pytest.mark.parametrize("seed", list(range(1, 13)))
def test_stream_random_settings_vs_oracle(wsa, seed):
    """Random rate / hop / window / frames per step / level / graph on or off: stream rows == oracle on the whole signal."""
    from oracle import pyoracle
    from webspeechanalyzer_amd.synth import synth_clips
    rng = np.random.default_rng(1000 + seed)
    fs = int(rng.choice([16000, 8000, 22050, 48000]))
    step = float(rng.choice([10.0, 15.0, 25.0, 40.0]))
    width = float(max(step, rng.choice([20.0, 25.0, 30.0, 60.0])))
    level = int(rng.choice([5, 13, 5, 13, 4, 10, 12, 11, 3]))
    F = int(rng.choice([1, 2, 3, 5, 8, 33]))
    kw = dict(window_step=step, window_width=width, pause_length=float(rng.choice([100.0, 200.0, 400.0])),
              min_seg_length=float(rng.choice([25.0, 50.0, 100.0])), auto_noise_gate=int(rng.random() < 0.7),
              voiced_max_dB=float(rng.choice([100.0, 140.0])), voiced_min_dB=float(rng.choice([10.0, 40.0])))
    n = 6
    pcm = synth_clips(n, int(fs * 4.5), fs=fs, seed=500 + seed, device="cuda")
    pcm = (pcm * torch.tensor(rng.uniform(0.1, 1.2, n), device="cuda", dtype=torch.float32)[:, None]).clamp(-1, 1).contiguous()
    got, segs, used = _run_streams(wsa, pcm, fs, level, F, bool(rng.random() < 0.7), bool(rng.random() < 0.5), cfg_kw=kw)
    fe = pyoracle.FrontEnd(pyoracle.fe_cfg(fs=float(fs), window_step=step, window_width=width))
    bkw = {k: v for k, v in kw.items() if k != "window_width"}
    host = pcm[:, :used].cpu().numpy()
    for c in range(n):
        ref = pyoracle.run_backend(fe.run(host[c]), pyoracle.default_cfg(level=level, **bkw))
        assert ref["segments_ci"] == segs[c], f"seed {seed} stream {c} fs {fs} F {F} {kw}"
        ok, why = callbacks_equal(level, ref["callbacks"], got[c], exact=False, tol=1e-4)
        assert ok, f"seed {seed} stream {c}: {why}"


def test_config5_512_streams_at_48k_equal_the_batch_run_and_the_oracle(wsa):
    """BASELINE config 5 at its full width: 512 concurrent 48 kHz streams, one 25 ms frame per hipGraph-replayed step, 80 steps
    (2 s of audio per stream).  The rows of all steps equal the rows of ONE batch run over the same 512 signals (indices and
    timestamps exactly, features to 1e-9: the two paths may finalize a segment with differently ordered wave sums), the segment
    tables agree, and 8 of the streams are checked against the oracle."""
    from oracle import pyoracle
    from webspeechanalyzer_amd.synth import synth_clips
    fs, n, nsteps = 48000, 512, 80
    an = wsa.Analyzer(wsa.Config(output_level=5))
    hop = an.geometry(fs)["hop"]
    pcm = synth_clips(n, nsteps * hop, fs=fs, seed=77, device="cuda")
    got, segs, used = _run_streams(wsa, pcm, fs, 5, 1, True, False)
    assert used == nsteps * hop
    b = an.batch([used] * n, fs)
    b.run(pcm.data_ptr(), pcm.stride(0), _stream())
    ref = b.callbacks(_stream())
    nrows = 0
    for s_ in range(n):
        assert segs[s_] == ref[s_]["segments_ci"], s_
        assert len(got[s_]) == len(ref[s_]["callbacks"]), s_
        for g, r in zip(got[s_], ref[s_]["callbacks"]):
            assert g[0] == r[0] and list(g[2]) == list(r[2])
            assert np.allclose(np.asarray(g[3]), np.asarray(r[3]), rtol=1e-9, atol=1e-12, equal_nan=True)
            nrows += 1
    assert nrows > n // 2
    fe = pyoracle.FrontEnd(pyoracle.fe_cfg(fs=fs))
    host = pcm[:8].cpu().numpy()
    for s_ in range(8):
        o = pyoracle.run_backend(fe.run(host[s_]), pyoracle.default_cfg(level=5))
        assert o["segments_ci"] == segs[s_]
        ok, why = callbacks_equal(5, o["callbacks"], got[s_], exact=False, tol=1e-4)
        assert ok, why
    b.close(); an.close()
