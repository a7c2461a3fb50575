"""Sample-rate conversion in front of the path (spec RS-1 in DESIGN.md): the HIP kernel against
oracle/resample.c bit for bit, and the whole path on the converted clips against the oracle chain
resample -> front end -> back end.  (The converter itself is not pinned by the reference: the browser does it there.)"""
import numpy as np
import pytest

from tests.util import callbacks_equal

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


def _stream():
    return torch.cuda.current_stream().cuda_stream


@pytest.fixture(scope="module")
def wsa():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    import webspeechanalyzer_amd as w
    return w


@pytest.mark.parametrize("fs_in,fs_out", [(44100, 48000), (16000, 48000), (8000, 48000), (22050, 48000), (32000, 48000),
                                          (48000, 16000), (44100, 16000), (96000, 48000), (11025, 44100), (48000, 48000), (128000, 8000), (3000, 48000)])
def test_resample_kernel_bit_exact_and_whole_path(wsa, fs_in, fs_out):
    from oracle import pyoracle
    from webspeechanalyzer_amd.synth import synth_clips
    rng = np.random.default_rng(fs_in + fs_out)
    n = 9
    lens = [int(fs_in * rng.uniform(0.3, 3.0)) for _ in range(n)]
    lens[0], lens[1], lens[2], lens[3] = 0, 1, 17, 300          # empty, shorter than the kernel, shorter than a window
    pcm = synth_clips(n, max(lens) + 8, fs=fs_in, seed=5, device="cuda")
    pcm = (pcm * torch.tensor(rng.uniform(0.2, 1.2, n), device="cuda", dtype=torch.float32)[:, None]).clamp(-1, 1).contiguous()
    an = wsa.Analyzer(wsa.Config(output_level=5))
    b = an.batch(lens, fs_in, resample_to=fs_out)
    b.run(pcm.data_ptr(), pcm.stride(0), _stream())
    got = b.callbacks(_stream())
    conv = b.converted_pcm(_stream())
    host = pcm.cpu().numpy()
    fe = pyoracle.FrontEnd(pyoracle.fe_cfg(fs=float(fs_out)))
    rows = 0
    for c in range(n):
        ref = pyoracle.resample(host[c, :lens[c]], fs_in, fs_out)
        assert len(ref) == int(b.n_samples[c]) and abs(len(ref) - lens[c] * fs_out / fs_in) <= 1
        assert np.array_equal(conv[c, :len(ref)].view(np.uint32), ref.view(np.uint32)), f"clip {c}: converted samples differ from the oracle"
        want = pyoracle.run_backend(fe.run(ref), pyoracle.default_cfg(level=5, bands=fe.bands))
        assert want["segments_ci"] == got[c]["segments_ci"], c
        ok, why = callbacks_equal(5, want["callbacks"], got[c]["callbacks"], exact=False, tol=1e-4)
        assert ok, why
        rows += len(want["callbacks"])
    assert rows > 0
    # the host-memory entry point takes the clips at the input rate as well
    b2 = an.batch(lens, fs_in, resample_to=fs_out)
    b2.run_host([host[c, :lens[c]] for c in range(n)], _stream())
    got2 = b2.callbacks(_stream())
    assert [g["segments_ci"] for g in got2] == [g["segments_ci"] for g in got]
    b.close(); b2.close(); an.close()


@pytest.mark.parametrize("pad", [0, 1, 2, 3])
def test_resample_staging_paths_by_clip_alignment(wsa, pad):
    """The kernel stages a block's inputs with 16-byte loads when the clip starts on a 16-byte boundary and word by word otherwise; a run
    without a whole 16-byte word inside the clip (clip edges, clips of < 4 samples) takes the word path as well.  Same samples bit for
    bit whatever the clips' stride and the buffer's offset are."""
    from oracle import pyoracle
    from webspeechanalyzer_amd.synth import synth_clips
    for fs_in in (44100, 16000):
        lens = [3, 5, 4481, 20000 + pad, 33333]
        stride = max(lens) + 4 + pad                              # stride mod 4 = (1 + pad) mod 4 ... every alignment occurs among the clips
        base = synth_clips(1, len(lens) * stride + 8, fs=fs_in, seed=11 + pad, device="cuda").reshape(-1)
        pcm = base[pad:pad + len(lens) * stride].reshape(len(lens), stride)
        an = wsa.Analyzer(wsa.Config(output_level=5))
        b = an.batch(lens, fs_in, resample_to=48000)
        b.run(pcm.data_ptr(), pcm.stride(0), _stream())
        b.device_result(_stream())
        conv = b.converted_pcm(_stream())
        host = pcm.cpu().numpy()
        for c, ln in enumerate(lens):
            ref = pyoracle.resample(host[c, :ln], fs_in, 48000)
            assert len(ref) == int(b.n_samples[c])
            assert np.array_equal(conv[c, :len(ref)].view(np.uint32), ref.view(np.uint32)), (fs_in, pad, c)
        b.close(); an.close()


def test_resample_fidelity_and_errors(wsa):
    """A 1 kHz tone comes out as a 1 kHz tone (44.1 -> 48 kHz), and bad rates are refused."""
    from oracle import pyoracle
    fs_in, fs_out = 44100, 48000
    t = np.arange(fs_in) / fs_in
    x = (0.5 * np.sin(2 * np.pi * 1000 * t)).astype(np.float32)
    y = pyoracle.resample(x, fs_in, fs_out)
    ref = 0.5 * np.sin(2 * np.pi * 1000 * np.arange(len(y)) / fs_out)
    assert len(y) == 48000 and np.abs(y[200:-200] - ref[200:-200]).max() < 2e-4
    an = wsa.Analyzer(wsa.Config())
    with pytest.raises(wsa.WsaError, match="sample rates"):
        an.batch([1000], 100, resample_to=48000)
    b = an.batch([1000], 16000)
    with pytest.raises(wsa.WsaError, match="converted PCM"):
        b.converted_pcm(_stream())
    b.close(); an.close()
