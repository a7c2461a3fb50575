"""The oracle front end FE-1 (oracle/frontend.c) — our own specification, parity unpinned against
the reference (its worklet source is absent) — checked against an fp64 textbook evaluation."""
import numpy as np

from oracle import pyoracle


def textbook_power4(fe, frame, cfgd):
    n = np.arange(fe.win)
    w = (0.5 - 0.5 * np.cos(2 * np.pi * n / fe.win)).astype(np.float32).astype(np.float64)
    x = np.zeros(fe.nfft)
    x[: fe.win] = frame[: fe.win].astype(np.float64) * w
    X = np.fft.rfft(x)
    return 4 * np.abs(X[: fe.kmax + 1]) ** 2


def test_geometry_config2():
    fe = pyoracle.FrontEnd(pyoracle.fe_cfg())
    assert (fe.nfft, fe.win, fe.hop, fe.bands, fe.kmax) == (1024, 400, 400, 128, 256)
    assert fe.n_frames(160000) == 400 and fe.n_frames(399) == 0 and fe.n_frames(400) == 1 and fe.n_frames(799) == 1


def test_power_spectrum_matches_fp64_dft():
    rng = np.random.default_rng(0)
    fe = pyoracle.FrontEnd(pyoracle.fe_cfg())
    for scale in (1e-3, 0.1, 0.9):
        x = (rng.standard_normal(400) * scale).clip(-1, 1).astype(np.float32)
        P = fe.power4(x).astype(np.float64)
        ref = textbook_power4(fe, x, None)
        # fp32 FFT: error relative to the spectrum's peak
        assert np.max(np.abs(P - ref)) <= 2e-5 * ref.max()


def test_full_scale_sine_lands_in_expected_band_and_range():
    fe = pyoracle.FrontEnd(pyoracle.fe_cfg())
    t = np.arange(1200) / 16000.0
    s = fe.run((0.5 * np.sin(2 * np.pi * 1000 * t)).astype(np.float32))
    assert s.shape == (3, 128)
    m = int(np.argmax(s[0]))
    assert abs(fe.bins_hz()[m] - 1000) < 30
    # dynamic range consistent with the reference's observed feature range: log10(ctx_max) <= 8.47
    # (dist/nnmodel/1/cats_emotion/model_meta.json, feature x3)
    assert 6.0 < np.log10(s[0].max()) < 8.47


def test_silence_and_saturation():
    fe = pyoracle.FrontEnd(pyoracle.fe_cfg())
    assert not fe.run(np.zeros(800, np.float32)).any()
    big = pyoracle.FrontEnd(pyoracle.fe_cfg(pre_norm_gain=1e9))
    s = big.run(np.ones(400, np.float32) * 0.99)
    assert s.max() == 0xFFFFFFFF


def test_other_rates_and_spec_types():
    for fs, nfft in ((16000, 1024), (48000, 3072), (44100, 3072), (8000, 512), (22050, 1536), (32000, 2048), (11025, 768), (96000, 6144), (6000, 384)):
        fe = pyoracle.FrontEnd(pyoracle.fe_cfg(fs=fs))
        assert fe.nfft == nfft
        x = (np.random.default_rng(1).standard_normal(fe.win + 3 * fe.hop) * 0.1).astype(np.float32)
        P = fe.power4(x).astype(np.float64)
        ref = textbook_power4(fe, x, None)
        assert np.max(np.abs(P - ref)) <= 5e-5 * ref.max()
        assert fe.run(x).shape == (4, 128)
    fe2 = pyoracle.FrontEnd(pyoracle.fe_cfg(spec_type=2))
    assert fe2.bands == 256


def test_resampler_rs1_fidelity_and_shape():
    """spec RS-1 (oracle/resample.c): output length trunc(n * fs_out / fs_in); a tone well below both Nyquist limits comes
    out as the same tone (up- and down-sampling); DC gain of the interpolated kernels within 1e-3 of one."""
    from oracle import pyoracle
    for fs_in, fs_out, tol in ((44100, 48000, 2e-4), (16000, 48000, 2e-4), (48000, 16000, 2e-4), (8000, 48000, 2e-4)):
        t = np.arange(fs_in) / fs_in
        x = (0.5 * np.sin(2 * np.pi * 440 * t)).astype(np.float32)
        y = pyoracle.resample(x, fs_in, fs_out)
        assert len(y) == fs_out
        ref = 0.5 * np.sin(2 * np.pi * 440 * np.arange(len(y)) / fs_out)
        assert np.abs(y[300:-300] - ref[300:-300]).max() < tol, (fs_in, fs_out)
        dc = pyoracle.resample(np.ones(4000, np.float32), fs_in, fs_out)
        assert np.abs(dc[200:-200] - 1).max() < 1.5e-3
    assert len(pyoracle.resample(np.zeros(0, np.float32), 44100, 48000)) == 0
    assert len(pyoracle.resample(np.zeros(1, np.float32), 44100, 48000)) == 1
