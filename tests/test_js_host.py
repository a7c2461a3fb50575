"""The JavaScript drop-in boundary (webspeechanalyzer_amd/js/formantanalyzer.js + N-API addon).
CPU part: loads, exports the reference's four functions, rejects like the reference, fails loudly
without a GPU.  GPU part (-m gpu): LaunchAudioNodes / LaunchBatch callbacks equal the oracle's."""
import json
import os
import shutil
import subprocess
import wave

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
JS = os.path.join(ROOT, "webspeechanalyzer_amd", "js", "formantanalyzer.js")
NODE = shutil.which("node")
pytestmark = pytest.mark.skipif(NODE is None, reason="node not installed")


def _build_addon():
    subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "webspeechanalyzer_amd", "csrc")], check=True)
    subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "webspeechanalyzer_amd", "napi")], check=True)


def _node(script):
    return subprocess.run([NODE, "-e", script], capture_output=True, text=True, timeout=120)


def test_module_shape_and_rejections():
    _build_addon()
    r = _node(f"""
const fa = require({json.dumps(JS)});
const out = {{keys: Object.keys(fa).filter(k => !k.startsWith('_')), rej: []}};
fa.configure({{spec_type:1, output_level:5, f_min:50, f_max:0, N_fft_bins:256, N_mel_bins:128, window_width:25, window_step:0,
              pause_length:200, min_seg_length:50, auto_noise_gate:false, voiced_max_dB:100, voiced_min_dB:0, pre_norm_gain:1000, high_f_emph:0}});
out.settings = fa._settings;
Promise.all([fa.LaunchAudioNodes(3).catch(e => out.rej.push(e)), fa.LaunchAudioNodes(2, {{}}).catch(e => out.rej.push(e)),
             fa.LaunchAudioNodes(1, null).catch(e => out.rej.push(e)), fa.LaunchAudioNodes(1, new ArrayBuffer(10)).catch(e => out.rej.push(e))])
  .then(() => console.log(JSON.stringify(out)));
""")
    assert r.returncode == 0, r.stderr
    out = json.loads(r.stdout)
    # ref dist/main.js:2 @B2750-2960: the module's four functions (+ our batch extension)
    assert out["keys"][:4] == ["configure", "LaunchAudioNodes", "StopAudioNodes", "set_predicted_label_for_segment"]
    # ref @B3292 truthy-merge: f_max:0 and window_step:0 ignored; the `null !==` keys honour 0 / false
    s = out["settings"]
    assert s["f_max"] == 4000 and s["window_step"] == 25 and s["auto_noise_gate"] is False and s["voiced_min_dB"] == 0
    assert out["rej"][:3] == ["Invalid audio source"] * 3 and out["rej"][3] == "Unable to decode audio data"


def test_fails_loudly_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    _build_addon()
    r = _node(f"""
const fa = require({json.dumps(JS)});
fa.configure({{spec_type:1, output_level:5, f_min:50, high_f_emph:0, auto_noise_gate:true, voiced_min_dB:10}});
fa.LaunchAudioNodes(1, new Float32Array(16000), () => {{}}, [], true, false).then(() => console.log('RESOLVED'), e => console.log('REJECT ' + e));
""")
    assert "REJECT" in r.stdout and "no CPU path" in r.stdout


def test_wav_decoder_formats(tmp_path):
    """PCM 8 / 16 / 24 / 32-bit, float32, stereo (channel 0), an odd-sized extra chunk and WAVE_FORMAT_EXTENSIBLE."""
    import struct
    rng = np.random.default_rng(0)
    x = rng.uniform(-0.9, 0.9, 257)
    y = rng.uniform(-0.9, 0.9, 257)

    def riff(fmt_body, data, extra=b""):
        chunks = b"fmt " + struct.pack("<I", len(fmt_body)) + fmt_body + extra + b"data" + struct.pack("<I", len(data)) + data
        return b"RIFF" + struct.pack("<I", 4 + len(chunks)) + b"WAVE" + chunks

    def fmt(tag, ch, rate, bits):
        return struct.pack("<HHIIHH", tag, ch, rate, rate * ch * bits // 8, ch * bits // 8, bits)

    cases = {}
    i16 = np.round(x * 32767).astype("<i2")
    cases["pcm16"] = (riff(fmt(1, 1, 16000, 16), i16.tobytes()), i16 / 32768.0, 16000)
    st = np.stack([i16, np.round(y * 32767).astype("<i2")], axis=1)
    cases["stereo16"] = (riff(fmt(1, 2, 44100, 16), st.tobytes(), extra=b"LIST" + struct.pack("<I", 3) + b"abc\0"), i16 / 32768.0, 44100)
    u8 = np.round(x * 127 + 128).astype(np.uint8)
    cases["pcm8"] = (riff(fmt(1, 1, 8000, 8), u8.tobytes()), (u8.astype(np.float64) - 128) / 128, 8000)
    i24 = np.round(x * 8388607).astype(np.int64)
    b24 = b"".join(int(v).to_bytes(3, "little", signed=True) for v in i24)
    cases["pcm24"] = (riff(fmt(1, 1, 48000, 24), b24), i24 / 8388608.0, 48000)
    i32 = np.round(x * 2147483647).astype("<i4")
    cases["pcm32"] = (riff(fmt(1, 1, 22050, 32), i32.tobytes()), i32 / 2147483648.0, 22050)
    f32 = x.astype("<f4")
    cases["float32"] = (riff(fmt(3, 1, 16000, 32), f32.tobytes()), f32.astype(np.float64), 16000)
    ext = fmt(0xFFFE, 1, 16000, 16) + struct.pack("<HHI", 22, 16, 4) + struct.pack("<H", 1) + bytes(14)
    cases["extensible"] = (riff(ext, i16.tobytes()), i16 / 32768.0, 16000)
    for name, (blob, expect, rate) in cases.items():
        f = tmp_path / (name + ".wav")
        f.write_bytes(blob)
        script = ("const fa=require(%r); const r=fa._decode_wav(require('fs').readFileSync(%r));"
                  "console.log(JSON.stringify({rate:r.sampleRate, pcm:Array.from(fa._clip_floats(r))}))") % (os.path.join(ROOT, "webspeechanalyzer_amd", "js", "formantanalyzer.js"), str(f))
        r = _node(script)
        assert r.returncode == 0, (name, r.stderr)
        got = json.loads(r.stdout)
        assert got["rate"] == rate, name
        assert np.array_equal(np.array(got["pcm"], dtype=np.float32), np.asarray(expect, dtype=np.float32)), name
    bad = tmp_path / "bad.wav"
    bad.write_bytes(b"RIFF....WAVEjunk")
    r = _node("const fa=require(%r); try{fa._decode_wav(require('fs').readFileSync(%r)); console.log('no')}catch(e){console.log('THREW '+e)}"
              % (os.path.join(ROOT, "webspeechanalyzer_amd", "js", "formantanalyzer.js"), str(bad)))
    assert "THREW Unable to decode audio data" in r.stdout


def _write_wav(path, pcm, fs):
    q = np.clip(np.round(pcm * 32768.0), -32768, 32767).astype(np.int16)
    with wave.open(path, "wb") as w:
        w.setnchannels(1); w.setsampwidth(2); w.setframerate(fs); w.writeframes(q.tobytes())
    return (q.astype(np.float64) / 32768.0).astype(np.float32)


@pytest.mark.gpu
@pytest.mark.parametrize("level", [5, 13])
def test_launch_audio_nodes_matches_oracle(tmp_path, level):
    import torch
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from oracle import pyoracle
    from tests.util import callbacks_equal
    from webspeechanalyzer_amd.synth import synth_clips
    _build_addon()
    fs = 16000
    pcm = synth_clips(3, 6 * fs, fs=fs, seed=31, device="cpu").numpy()
    clips, host = [], []
    pcm[0].tofile(tmp_path / "c0.f32"); clips.append(dict(file=str(tmp_path / "c0.f32"), kind="f32", fs=fs)); host.append(pcm[0])
    host.append(_write_wav(str(tmp_path / "c1.wav"), pcm[1], fs)); clips.append(dict(file=str(tmp_path / "c1.wav"), kind="wav"))
    job = tmp_path / "job.json"
    json.dump(dict(level=level, clips=clips, check_busy=True), open(job, "w"))
    r = subprocess.run([NODE, os.path.join(ROOT, "tests", "node_runner.js"), str(job)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    out = json.loads(r.stdout)
    fe = pyoracle.FrontEnd(pyoracle.fe_cfg(fs=fs))
    for o, x in zip(out, host):
        assert o["resolved"] is True
        assert o["busy"] == ["Error: Already playing"]                 # ref @B4554
        ref = pyoracle.run_backend(fe.run(x), pyoracle.default_cfg(level=level))
        got = [[c[0], c[1], c[2], c[3]] for c in o["calls"]]
        assert all(c[1] == ["lbl"] for c in got)
        ok, why = callbacks_equal(level, [[c[0], [], c[2], c[3]] for c in ref["callbacks"]], got, exact=False, tol=1e-4)
        assert ok, why
        assert len(got) > 0


@pytest.mark.gpu
def test_launch_batch_matches_oracle(tmp_path):
    import torch
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from oracle import pyoracle
    from tests.util import callbacks_equal
    from webspeechanalyzer_amd.synth import synth_clips
    _build_addon()
    fs, n = 16000, 6
    pcm = synth_clips(n, 5 * fs, fs=fs, seed=41, device="cpu").numpy()
    clips = []
    for i in range(n):
        pcm[i].tofile(tmp_path / f"c{i}.f32"); clips.append(dict(file=str(tmp_path / f"c{i}.f32"), kind="f32", fs=fs))
    job = tmp_path / "job.json"
    json.dump(dict(level=5, clips=clips, batch=True), open(job, "w"))
    r = subprocess.run([NODE, os.path.join(ROOT, "tests", "node_runner.js"), str(job)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    out = json.loads(r.stdout)
    fe = pyoracle.FrontEnd(pyoracle.fe_cfg(fs=fs))
    for i in range(n):
        ref = pyoracle.run_backend(fe.run(pcm[i]), pyoracle.default_cfg(level=5))
        assert all(c[1] == [f"clip{i}"] for c in out[i])
        ok, why = callbacks_equal(5, ref["callbacks"], [[c[0], [], c[2], c[3]] for c in out[i]], exact=False, tol=1e-4)
        assert ok, why


@pytest.mark.gpu
def test_clips_in_pinned_buffers_give_the_rows_of_clips_in_ordinary_memory(tmp_path):
    """allocPinned (wsa_host_alloc through the addon): clips held in page-locked ArrayBuffers — one buffer per clip, and views into one slab, which travel
    as a single copy — must give the callbacks of the same clips in ordinary memory."""
    import torch
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from webspeechanalyzer_amd.synth import synth_clips
    _build_addon()
    fs, n = 16000, 20
    pcm = synth_clips(n, 5 * fs, fs=fs, seed=43, device="cpu").numpy()
    clips = []
    for i in range(n):
        pcm[i].tofile(tmp_path / f"c{i}.f32"); clips.append(dict(file=str(tmp_path / f"c{i}.f32"), kind="f32", fs=fs))
    outs = []
    for mode in (None, "each", "slab"):
        job = tmp_path / f"job_{mode}.json"
        json.dump(dict(level=5, clips=clips, batch=True, pinned=mode), open(job, "w"))
        r = subprocess.run([NODE, os.path.join(ROOT, "tests", "node_runner.js"), str(job)], capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr
        outs.append(json.loads(r.stdout))
    assert sum(len(c) for c in outs[0]) > 20
    assert outs[0] == outs[1] == outs[2]


@pytest.mark.gpu
def test_free_pinned_releases_and_detaches():
    """freePinned(ab): the page-locked memory of an allocPinned buffer goes back when the caller says so (not in a finalizer on the event loop), the ArrayBuffer is
    detached (views have length 0), a second call and a call on an ordinary ArrayBuffer report false; a launch out of a buffer that is still allocated works before."""
    import torch
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    _build_addon()
    r = _node(f"""
const fa = require({json.dumps(JS)});
fa.configure({{spec_type:1, output_level:5, f_min:50, high_f_emph:0, auto_noise_gate:true, voiced_min_dB:10}});
(async () => {{
  const n = 16000 * 2, ab = fa.allocPinned(n * 4), x = new Float32Array(ab);
  for (let i = 0; i < n; i++) x[i] = 0.3 * Math.sin(2 * Math.PI * 220 * i / 16000) * (Math.sin(2 * Math.PI * 3 * i / 16000) > 0 ? 1 : 0);
  const info = await fa.LaunchBatch([{{pcm: x, sampleRate: 16000}}], () => {{}}, []);
  const out = {{rows: info.rows, before: ab.byteLength, first: fa.freePinned(ab), after: ab.byteLength, view: x.length, second: fa.freePinned(ab), plain: fa.freePinned(new ArrayBuffer(64))}};
  fa.shutdown();
  console.log(JSON.stringify(out));
}})().catch(e => {{ console.log('ERROR ' + e); process.exit(1); }});
""")
    assert r.returncode == 0, r.stdout + r.stderr
    out = json.loads(r.stdout.strip().splitlines()[-1])
    assert out["before"] == 16000 * 2 * 4 and out["first"] is True and out["after"] == 0 and out["view"] == 0 and out["second"] is False and out["plain"] is False


@pytest.mark.gpu
def test_pipelined_batches_give_the_callbacks_of_one_batch_at_a_time(tmp_path):
    """LaunchBatches (two contexts on one device, each with its own planned batch and HIP stream: batch k + 1 uploads while batch k computes and batch
    k - 1's callbacks run) must deliver, batch by batch and in order, exactly what LaunchBatch delivers for the same clips — ragged clips, five batches
    (so both contexts are reused, with a plan of another shape waiting in their boxes), clips in ordinary memory and in one page-locked slab."""
    import torch
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from webspeechanalyzer_amd.synth import synth_clips
    _build_addon()
    fs, n = 16000, 23
    pcm = synth_clips(n, 6 * fs, fs=fs, seed=47, device="cpu").numpy()
    clips = []
    for i in range(n):
        pcm[i][:6 * fs - 997 * (i % 5)].tofile(tmp_path / f"c{i}.f32"); clips.append(dict(file=str(tmp_path / f"c{i}.f32"), kind="f32", fs=fs))
    job = tmp_path / "one.json"
    json.dump(dict(level=5, clips=clips, batch=True), open(job, "w"))
    r = subprocess.run([NODE, os.path.join(ROOT, "tests", "node_runner.js"), str(job)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    want = json.loads(r.stdout)
    assert sum(len(c) for c in want) > 25
    for pinned in (None, "slab"):
        job = tmp_path / f"many_{pinned}.json"
        json.dump(dict(level=5, clips=clips, batches=5, pinned=pinned), open(job, "w"))
        r = subprocess.run([NODE, os.path.join(ROOT, "tests", "node_runner.js"), str(job)], capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr
        got = json.loads(r.stdout)
        assert got["info"]["batches"] == 5 and got["in_order"] is True
        assert got["info"]["rows"] == sum(len(c) for c in want)
        assert got["per"] == want, pinned


@pytest.mark.gpu
def test_wav_file_44k1_gpu_host_vs_js_cpu_path(tmp_path):
    """BASELINE config 1 shape: one 44.1 kHz WAV file, Segment Features — the Node host over the HIP path
    against the pure-JS CPU path (oracle/js) on the identical file."""
    import torch
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from tests.util import callbacks_equal, jsvec
    from webspeechanalyzer_amd.synth import synth_clips
    _build_addon()
    fs = 44100
    pcm = synth_clips(1, 8 * fs, fs=fs, seed=17, device="cpu").numpy()[0]
    wav = str(tmp_path / "clip.wav")
    _write_wav(wav, pcm, fs)
    job = tmp_path / "job.json"
    json.dump(dict(level=5, clips=[dict(file=wav, kind="wav")]), open(job, "w"))
    r = subprocess.run([NODE, os.path.join(ROOT, "tests", "node_runner.js"), str(job)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    got = json.loads(r.stdout)[0]
    assert got["resolved"] is True
    job2 = tmp_path / "job2.json"
    json.dump(dict(mode="e2e", wav=wav, settings=dict(output_level=5)), open(job2, "w"))
    r2 = subprocess.run([NODE, os.path.join(ROOT, "oracle", "js", "run.js"), str(job2)], capture_output=True, text=True, timeout=600)
    assert r2.returncode == 0, r2.stderr
    ref = json.loads(r2.stdout)
    assert ref["fs"] == fs and ref["nfft"] == 3072 and len(ref["callbacks"]) > 2
    refc = [[c[0], [], c[2], jsvec(c[3])] for c in ref["callbacks"]]
    gotc = [[c[0], [], np.array(c[2]), jsvec(c[3])] for c in got["calls"]]
    ok, why = callbacks_equal(5, refc, gotc, exact=False, tol=1e-4)
    assert ok, why


@pytest.mark.gpu
def test_stereo_16_bit_wav_goes_to_the_device_as_it_is(tmp_path):
    """A 16-bit stereo WAV is handed to the device undecoded (Int16Array view of the data chunk, interleaved; libwsa converts
    channel 0): the callbacks equal those of the mono file holding channel 0, bit for bit — also from an odd offset inside the
    file (an extra chunk of odd length in front of `data` plus its pad byte keeps the offset even; a hand-made odd offset takes
    the copy path), and in a batch that mixes the file with a Float32Array clip (sent as floats then)."""
    import struct
    import torch
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from webspeechanalyzer_amd.synth import synth_clips
    _build_addon()
    fs = 16000
    pcm = synth_clips(1, 6 * fs, fs=fs, seed=91, device="cpu").numpy()[0]
    mono = str(tmp_path / "mono.wav")
    q = _write_wav(mono, pcm, fs)
    i16 = np.round(q * 32768.0).astype("<i2")
    inter = np.empty((len(i16), 2), "<i2"); inter[:, 0] = i16; inter[:, 1] = np.random.default_rng(3).integers(-20000, 20000, len(i16))
    fmt = struct.pack("<HHIIHH", 1, 2, fs, fs * 4, 4, 16)
    def riff(chunks):
        body = b"WAVE" + b"".join(cid + struct.pack("<I", len(d)) + d + (b"\0" if len(d) & 1 else b"") for cid, d in chunks)
        return b"RIFF" + struct.pack("<I", len(body)) + body
    stereo = tmp_path / "stereo.wav"; stereo.write_bytes(riff([(b"fmt ", fmt), (b"data", inter.tobytes())]))
    odd = tmp_path / "odd.wav"; odd.write_bytes(riff([(b"fmt ", fmt), (b"LIST", b"abc"), (b"data", inter.tobytes())]))
    f32 = tmp_path / "c.f32"; q.astype(np.float32).tofile(f32)
    outs = {}
    for tag, clips, batch in (("mono", [dict(file=mono, kind="wav")], False), ("stereo", [dict(file=str(stereo), kind="wav")], False),
                              ("odd", [dict(file=str(odd), kind="wav")], False),
                              ("mixed", [dict(file=str(stereo), kind="wav"), dict(file=str(f32), kind="f32", fs=fs)], True)):
        job = tmp_path / f"job_{tag}.json"
        json.dump(dict(level=5, clips=clips, batch=batch), open(job, "w"))
        r = subprocess.run([NODE, os.path.join(ROOT, "tests", "node_runner.js"), str(job)], capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, (tag, r.stderr)
        outs[tag] = json.loads(r.stdout)
    ref = outs["mono"][0]["calls"]
    assert len(ref) > 2
    strip = lambda calls: [[c[0], c[2], c[3]] for c in calls]
    assert strip(outs["stereo"][0]["calls"]) == strip(ref) and strip(outs["odd"][0]["calls"]) == strip(ref)
    assert strip(outs["mixed"][0]) == strip(ref) and strip(outs["mixed"][1]) == strip(ref)


@pytest.mark.gpu
@pytest.mark.parametrize("level,fps", [(5, 1), (13, 3), (4, 2), (10, 1), (12, 2), (11, 1), (3, 2)])
def test_stream_open_matches_oracle(tmp_path, level, fps):
    """extension StreamOpen: concurrent streams pushed step by step from Node == the oracle on each whole signal (levels 4 / 10: the
    straightened formant frames of the segments / syllables that closed in a step)."""
    import torch
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from oracle import pyoracle
    from tests.util import callbacks_equal
    from webspeechanalyzer_amd.synth import synth_clips
    _build_addon()
    fs, n = 16000, 5
    pcm = synth_clips(n, 5 * fs, fs=fs, seed=51, device="cpu").numpy()
    clips = []
    for i in range(n):
        pcm[i].tofile(tmp_path / f"c{i}.f32"); clips.append(dict(file=str(tmp_path / f"c{i}.f32"), kind="f32", fs=fs))
    job = tmp_path / "job.json"
    json.dump(dict(level=level, clips=clips, stream=dict(frames_per_step=fps)), open(job, "w"))
    r = subprocess.run([NODE, os.path.join(ROOT, "tests", "node_runner.js"), str(job)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    out = json.loads(r.stdout)
    fe = pyoracle.FrontEnd(pyoracle.fe_cfg(fs=fs))
    total = 0
    for i in range(n):
        ref = pyoracle.run_backend(fe.run(pcm[i][:out["used"]]), pyoracle.default_cfg(level=level))
        got = out["per"][i]
        assert all(c[1] == [f"s{i}"] for c in got)
        if level in (4, 10):
            def arr(v):       # a Float32Array serialises as {"0":..,"1":..}
                return [[row[str(k)] for k in range(9)] if isinstance(row, dict) else row for row in v]
            mine = [[c[0], [], (np.array(c[2]) if level == 4 else c[2]), (arr(c[3]) if level == 4 else [arr(v) for v in c[3]])] for c in got]
            refc = [[c[0], [], c[2], (c[3].tolist() if level == 4 else [v.tolist() for v in c[3]])] for c in ref["callbacks"]]
            ok, why = callbacks_equal(level, refc, mine)
        elif level == 3:      # (segment index, label, ranked raw tracks): exact
            ok, why = callbacks_equal(level, ref["callbacks"], [[c[0], [], c[2]] for c in got])
        else:
            ok, why = callbacks_equal(level, ref["callbacks"], [[c[0], [], c[2], c[3]] for c in got], exact=False, tol=1e-4)
        assert ok, why
        total += len(got)
    assert total > 8


@pytest.mark.gpu
@pytest.mark.parametrize("level", [4, 10])
def test_launch_audio_nodes_formant_frames(tmp_path, level):
    """levels 4 / 10 through the Node host: Float32Array(9) frames per segment / per syllable == oracle, bit-exact."""
    import torch
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from oracle import pyoracle
    from tests.util import callbacks_equal
    from webspeechanalyzer_amd.synth import synth_clips
    _build_addon()
    fs = 16000
    pcm = synth_clips(2, 6 * fs, fs=fs, seed=33, device="cpu").numpy()
    clips = []
    for i in range(2):
        pcm[i].tofile(tmp_path / f"c{i}.f32"); clips.append(dict(file=str(tmp_path / f"c{i}.f32"), kind="f32", fs=fs))
    job = tmp_path / "job.json"
    json.dump(dict(level=level, clips=clips), open(job, "w"))
    r = subprocess.run([NODE, os.path.join(ROOT, "tests", "node_runner.js"), str(job)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    out = json.loads(r.stdout)
    fe = pyoracle.FrontEnd(pyoracle.fe_cfg(fs=fs))
    for o, x in zip(out, pcm):
        ref = pyoracle.run_backend(fe.run(x), pyoracle.default_cfg(level=level))
        def arr(v):       # a Float32Array serialises as {"0":..,"1":..}
            return [[row[str(k)] for k in range(9)] if isinstance(row, dict) else row for row in v]
        got = [[c[0], [], (np.array(c[2]) if level == 4 else c[2]), (arr(c[3]) if level == 4 else [arr(v) for v in c[3]])] for c in o["calls"]]
        refc = [[c[0], [], c[2], (c[3].tolist() if level == 4 else [v.tolist() for v in c[3]])] for c in ref["callbacks"]]
        ok, why = callbacks_equal(level, refc, got)
        assert ok, why
        assert len(got) > 0


@pytest.mark.gpu
def test_launch_audio_nodes_level_12(tmp_path):
    """level 12 through the Node host: per syllable the 23 polynomial coefficients, bit for bit == oracle."""
    import torch
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from oracle import pyoracle
    from tests.util import callbacks_equal
    from webspeechanalyzer_amd.synth import synth_clips
    _build_addon()
    fs = 16000
    pcm = synth_clips(1, 6 * fs, fs=fs, seed=36, device="cpu").numpy()
    pcm[0].tofile(tmp_path / "c0.f32")
    job = tmp_path / "job.json"
    json.dump(dict(level=12, clips=[dict(file=str(tmp_path / "c0.f32"), kind="f32", fs=fs)]), open(job, "w"))
    r = subprocess.run([NODE, os.path.join(ROOT, "tests", "node_runner.js"), str(job)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    o = json.loads(r.stdout)[0]
    fe = pyoracle.FrontEnd(pyoracle.fe_cfg(fs=fs))
    ref = pyoracle.run_backend(fe.run(pcm[0]), pyoracle.default_cfg(level=12))
    got = [[c[0], [], c[2], [np.array(v) for v in c[3]]] for c in o["calls"]]
    ok, why = callbacks_equal(12, ref["callbacks"], got)
    assert ok, why
    assert sum(len(c[3]) for c in got) > 5 and all(len(v) == 23 for c in got for v in c[3])


@pytest.mark.gpu
def test_launch_audio_nodes_level_11(tmp_path):
    """level 11 through the Node host: callback(0, label, [t0, dur], number[264]) after every result == oracle."""
    import torch
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from oracle import pyoracle
    from tests.util import callbacks_equal
    from webspeechanalyzer_amd.synth import synth_clips
    _build_addon()
    fs = 16000
    pcm = synth_clips(2, 8 * fs, fs=fs, seed=35, device="cpu").numpy()
    clips = []
    for i in range(2):
        pcm[i].tofile(tmp_path / f"c{i}.f32"); clips.append(dict(file=str(tmp_path / f"c{i}.f32"), kind="f32", fs=fs))
    job = tmp_path / "job.json"
    json.dump(dict(level=11, clips=clips), open(job, "w"))
    r = subprocess.run([NODE, os.path.join(ROOT, "tests", "node_runner.js"), str(job)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    out = json.loads(r.stdout)
    fe = pyoracle.FrontEnd(pyoracle.fe_cfg(fs=fs))
    n = 0
    for o, x in zip(out, pcm):
        ref = pyoracle.run_backend(fe.run(x), pyoracle.default_cfg(level=11))
        got = [[c[0], [], np.array(c[2]), np.array(c[3])] for c in o["calls"]]
        ok, why = callbacks_equal(11, ref["callbacks"], got)
        assert ok, why
        n += len(got)
    assert n > 3


@pytest.mark.gpu
@pytest.mark.parametrize("level", [5, 13])
def test_feature_db_files_from_the_gpu_host(tmp_path, level):
    """SURVEY.md 8f item 3 end to end: the GPU host's callbacks collected the way the app collects them
    (featuredb.js = src/localstore.js StoreFeatures) and exported as the app's JSON / CSV files; the files hold exactly
    the delivered callbacks (file, seg = si or si + ph/100, time, 53 features) and the JSON file loads back."""
    import torch
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from webspeechanalyzer_amd.synth import synth_clips
    _build_addon()
    fs = 16000
    pcm = synth_clips(2, 6 * fs, fs=fs, seed=77, device="cpu").numpy()
    clips = []
    for i in range(2):
        pcm[i].tofile(tmp_path / f"u{i}.f32"); clips.append(dict(file=str(tmp_path / f"u{i}.f32"), kind="f32", fs=fs))
    job = tmp_path / "job.json"
    json.dump(dict(level=level, clips=clips, featuredb=True), open(job, "w"))
    r = subprocess.run([NODE, os.path.join(ROOT, "tests", "node_runner.js"), str(job)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    out = json.loads(r.stdout)
    want = []
    for i, o in enumerate(out["clips"]):
        for si, _, t, f in o["calls"]:
            if level == 5:
                want.append((f"u{i}.f32", str(si), t, f))
            else:
                for ph in range(len(f)):
                    want.append((f"u{i}.f32", repr(si + ph / 100) if ph else str(si), t[ph], f[ph]))
    rows = json.loads(out["db_json"])
    assert len(rows) == len(want) > 0
    for row, (file, seg, t, f) in zip(rows, want):
        assert row["file"] == file and float(row["seg"]) == float(seg) and row["time"] == t and row["features"] == f
        assert row["origin"] is None and row["true"] is None and row["pred"] is None
    lines = out["db_csv"].split("\r\n")
    assert lines[0] == "file,seg,t0,td," + "".join(f"x{k}," for k in range(53)) and len(lines) == len(want) + 2
    # the exported JSON is what Load_JSON_Data takes
    chk = subprocess.run([NODE, "-e", "const {FeatureDB}=require(%r); const d=new FeatureDB(); const s=require('fs').readFileSync(0,'utf8');"
                          "const n=d.Load_JSON_Data(3,s); console.log(n, d.Download_DB(3,'JSON')===s)"
                          % os.path.join(ROOT, "webspeechanalyzer_amd", "js", "featuredb.js")], input=out["db_json"], capture_output=True, text=True, timeout=60)
    assert chk.stdout.split() == [str(len(want)), "true"], chk.stdout + chk.stderr


@pytest.mark.gpu
def test_launch_audio_nodes_level_3_raw_tracks(tmp_path):
    """level 3 through the Node host: the three-argument callback (si, label, ranked tracks) with every field of the
    18-field track records equal to the oracle's (pinned to the reference's own level-3 fixtures)."""
    import torch
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from oracle import pyoracle
    from tests.util import callbacks_equal
    from webspeechanalyzer_amd.synth import synth_clips
    _build_addon()
    fs = 16000
    pcm = synth_clips(2, 6 * fs, fs=fs, seed=35, device="cpu").numpy()
    clips = []
    for i in range(2):
        pcm[i].tofile(tmp_path / f"c{i}.f32"); clips.append(dict(file=str(tmp_path / f"c{i}.f32"), kind="f32", fs=fs))
    job = tmp_path / "job.json"
    json.dump(dict(level=3, clips=clips), open(job, "w"))
    r = subprocess.run([NODE, os.path.join(ROOT, "tests", "node_runner.js"), str(job)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    out = json.loads(r.stdout)
    fe = pyoracle.FrontEnd(pyoracle.fe_cfg(fs=fs))
    n = 0
    for o, x in zip(out, pcm):
        ref = pyoracle.run_backend(fe.run(x), pyoracle.default_cfg(level=3))
        assert all(len(c) == 3 for c in o["calls"])
        ok, why = callbacks_equal(3, ref["callbacks"], [[c[0], [], c[2]] for c in o["calls"]])
        assert ok, why
        n += sum(len(c[2]) for c in o["calls"])
    assert n > 10


@pytest.mark.gpu
def test_wav_file_converted_to_48k_like_the_reference_offline_path(tmp_path):
    """configure({resample_to: 48000}): a 44.1 kHz WAV file is converted to 48 kHz in front of the path (the reference's offline
    path always analyses at 48 kHz, ref @B18769: the browser converts; here spec RS-1 does) and the callbacks equal the oracle
    chain resample -> front end (48 kHz: 1200-sample hop, 3072-point FFT) -> back end."""
    import torch
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from oracle import pyoracle
    from tests.util import callbacks_equal
    from webspeechanalyzer_amd.synth import synth_clips
    _build_addon()
    fs = 44100
    pcm = synth_clips(1, 5 * fs, fs=fs, seed=91, device="cpu").numpy()[0]
    host = _write_wav(str(tmp_path / "a.wav"), pcm, fs)
    job = tmp_path / "job.json"
    json.dump(dict(level=5, clips=[dict(file=str(tmp_path / "a.wav"), kind="wav")], config=dict(resample_to=48000)), open(job, "w"))
    r = subprocess.run([NODE, os.path.join(ROOT, "tests", "node_runner.js"), str(job)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    out = json.loads(r.stdout)[0]
    fe = pyoracle.FrontEnd(pyoracle.fe_cfg(fs=48000.0))
    assert fe.nfft == 3072 and fe.hop == 1200
    ref = pyoracle.run_backend(fe.run(pyoracle.resample(host, fs, 48000)), pyoracle.default_cfg(level=5, bands=fe.bands))
    ok, why = callbacks_equal(5, [[c[0], [], c[2], c[3]] for c in ref["callbacks"]], [[c[0], [], c[2], c[3]] for c in out["calls"]], exact=False, tol=1e-4)
    assert ok, why
    assert len(out["calls"]) > 0


@pytest.mark.gpu
def test_stop_audio_nodes_batch_launch(tmp_path):
    """StopAudioNodes (ref @B5699 -> disconnect_nodes @B21559 -> teardown + segment_truncate + resolve, @B8851): called from inside a
    callback the remaining callbacks are not delivered and the launch still resolves true; called while the work is in flight nothing is
    dispatched; called when nothing is playing it has no effect on the next launch."""
    import torch
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from webspeechanalyzer_amd.synth import synth_clips
    _build_addon()
    fs = 16000
    pcm = synth_clips(1, 10 * fs, fs=fs, seed=7, device="cpu").numpy()[0]
    pcm.tofile(tmp_path / "a.f32")
    clip = dict(file=str(tmp_path / "a.f32"), kind="f32", fs=fs)

    def run(**kw):
        job = tmp_path / "job.json"
        json.dump(dict(level=5, clips=[clip, clip], **kw), open(job, "w"))
        r = subprocess.run([NODE, os.path.join(ROOT, "tests", "node_runner.js"), str(job)], capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr
        return json.loads(r.stdout)

    full = run(stop_before=True)
    assert full[0]["resolved"] is True and len(full[0]["calls"]) >= 3 and full[0]["calls"] == full[1]["calls"]
    cut = run(stop_after=2)
    assert [o["resolved"] for o in cut] == [True, True]
    assert cut[0]["calls"] == full[0]["calls"][:2] and cut[1]["calls"] == full[1]["calls"][:2]       # and the second launch starts afresh
    none = run(stop_in_flight=True)
    assert [o["resolved"] for o in none] == [True, True] and none[0]["calls"] == [] and none[1]["calls"] == []


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["stop_at", "close_at"])
def test_stop_audio_nodes_and_close_flush_open_streams(tmp_path, mode):
    """Streams (the reference's online path): StopAudioNodes lets the frame in flight through, truncates every source (segment_truncate,
    ref @B30757) so that the segment it is in the middle of is reported, and the object closes; close() alone flushes the same way.
    Either way the callbacks equal the oracle on the signal fed so far, and the pinned input buffer is detached afterwards."""
    import torch
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from oracle import pyoracle
    from tests.util import callbacks_equal
    from webspeechanalyzer_amd.synth import synth_clips
    _build_addon()
    fs, n = 16000, 3
    pcm = synth_clips(n, 6 * fs, fs=fs, seed=52, device="cpu").numpy()
    fe = pyoracle.FrontEnd(pyoracle.fe_cfg(fs=fs))
    # stop in the middle of a segment of stream 0
    segs = pyoracle.run_backend(fe.run(pcm[0]), pyoracle.default_cfg(level=5))["segments_ci"]
    st0, ln0 = max(segs, key=lambda s: s[1])
    k_stop = (st0 + ln0 // 2) // 2                     # 2 frames per step
    clips = []
    for i in range(n):
        pcm[i].tofile(tmp_path / f"c{i}.f32"); clips.append(dict(file=str(tmp_path / f"c{i}.f32"), kind="f32", fs=fs))
    job = tmp_path / "job.json"
    json.dump(dict(level=5, clips=clips, stream={"frames_per_step": 2, mode: k_stop}), open(job, "w"))
    r = subprocess.run([NODE, os.path.join(ROOT, "tests", "node_runner.js"), str(job)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    out = json.loads(r.stdout)
    assert out["used"] == (k_stop + (1 if mode == "stop_at" else 0)) * 800
    if mode == "stop_at":
        # the next step finds the object closed: its pinned input is detached (writing to it throws), push() would say "stream closed"
        assert out["closed_error"] == "stream closed" or "neutered" in out["closed_error"] or "detached" in out["closed_error"]
    assert out["input_length_after_close"] == 0                 # detached: no view on freed pinned memory
    total = 0
    for i in range(n):
        ref = pyoracle.run_backend(fe.run(pcm[i][:out["used"]]), pyoracle.default_cfg(level=5))
        ok, why = callbacks_equal(5, ref["callbacks"], [[c[0], [], c[2], c[3]] for c in out["per"][i]], exact=False, tol=1e-4)
        assert ok, f"stream {i}: {why}"
        total += len(ref["callbacks"])
    assert total >= 3


@pytest.mark.gpu
def test_launch_batch_sharded_over_two_contexts(tmp_path):
    """configure({devices: [0, 0]}): LaunchBatch shards the clips contiguously over two contexts (one worker thread each) and delivers the
    callbacks in (clip, si) order — the same calls as one context."""
    import torch
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from webspeechanalyzer_amd.synth import synth_clips
    _build_addon()
    fs, n = 16000, 5
    pcm = synth_clips(n, 4 * fs, fs=fs, seed=71, device="cpu").numpy()
    clips = []
    for i in range(n):
        pcm[i, :4 * fs - 997 * i].tofile(tmp_path / f"c{i}.f32"); clips.append(dict(file=str(tmp_path / f"c{i}.f32"), kind="f32", fs=fs))
    res = {}
    for tag, cfg in (("one", {}), ("two", {"devices": [0, 0]}), ("three", {"devices": [0, 0, 0]})):
        job = tmp_path / f"job_{tag}.json"
        json.dump(dict(level=13, clips=clips, batch=True, want_info=True, config=cfg), open(job, "w"))
        r = subprocess.run([NODE, os.path.join(ROOT, "tests", "node_runner.js"), str(job)], capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr
        res[tag] = json.loads(r.stdout)
    assert res["one"]["info"]["shards"] == 1 and res["two"]["info"]["shards"] == 2 and res["three"]["info"]["shards"] == 3
    assert sum(len(p) for p in res["one"]["per"]) >= 5
    assert res["two"]["per"] == res["one"]["per"] and res["three"]["per"] == res["one"]["per"]
    assert res["two"]["info"]["rows"] == res["one"]["info"]["rows"]


@pytest.mark.gpu
@pytest.mark.parametrize("level", [5, 13])
def test_launch_batch_rows_through_the_rccl_gather(tmp_path, level):
    """configure({gather: true}): the feature rows stay on the device after the batch, wsa_gather_rows collects them with one grouped RCCL
    send / receive (here one rank: the root's send to itself) and ONE copy brings them to the host — the callbacks are the same calls."""
    import torch
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from webspeechanalyzer_amd.synth import synth_clips
    _build_addon()
    fs, n = 16000, 5
    pcm = synth_clips(n, 4 * fs, fs=fs, seed=73, device="cpu").numpy()
    clips = []
    for i in range(n):
        pcm[i, :4 * fs - 911 * i].tofile(tmp_path / f"c{i}.f32"); clips.append(dict(file=str(tmp_path / f"c{i}.f32"), kind="f32", fs=fs))
    res = {}
    for tag, cfg in (("plain", {"gather": False}), ("gathered", {"gather": True})):
        job = tmp_path / f"job_{tag}.json"
        json.dump(dict(level=level, clips=clips, batch=True, want_info=True, config=cfg), open(job, "w"))
        r = subprocess.run([NODE, os.path.join(ROOT, "tests", "node_runner.js"), str(job)], capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr
        res[tag] = json.loads(r.stdout.strip().splitlines()[-1])       # (librccl prints a version banner on stdout when the communicator is made)
    assert sum(len(p) for p in res["plain"]["per"]) >= 5
    assert res["gathered"]["per"] == res["plain"]["per"]
    assert res["gathered"]["info"]["rows"] == res["plain"]["info"]["rows"]


@pytest.mark.gpu
def test_launch_batch_one_failing_shard_rejects_cleanly_and_the_module_stays_usable():
    """A multi-device LaunchBatch whose second shard fails while the first is still in flight: the launch rejects with that shard's
    error (not with "context still has batches in flight" from the clean-up), every context is destroyed once its work is through,
    and the module is not left "Already playing" — the next launch works."""
    import torch
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    _build_addon()
    r = _node(f"""
const fa = require({json.dumps(JS)});
const nat = require({json.dumps(os.path.join(ROOT, "webspeechanalyzer_amd", "lib", "wsa_napi.node"))});
fa.configure({{spec_type:1, output_level:5, f_min:50, high_f_emph:0, auto_noise_gate:true, voiced_min_dB:10, devices:[0, 0]}});
const clips = [];
for (let c = 0; c < 4; c++) {{ const x = new Float32Array(16000 * 20); for (let i = 0; i < x.length; i++) x[i] = 0.3 * Math.sin(i * 0.05 * (c + 1)) * Math.sin(i * 0.0004); clips.push({{pcm: x, sampleRate: 16000}}); }}
const real = nat.processBatch; let calls = 0;
nat.processBatch = function (...a) {{ calls++; return calls === 2 ? real.apply(nat, a).then(() => {{ throw 'shard two failed'; }}) : real.apply(nat, a); }};
const out = {{}};
fa.LaunchBatch(clips, () => {{}}).then(() => {{ out.first = 'resolved'; }}, (e) => {{ out.first = String(e); }})
  .then(() => {{ nat.processBatch = real; return fa.LaunchBatch(clips, () => {{}}); }})
  .then(() => {{ out.second = 'resolved'; }}, (e) => {{ out.second = String(e); }})
  .then(() => console.log(JSON.stringify(out)));
""")
    assert r.returncode == 0, r.stderr
    out = json.loads(r.stdout)
    assert out["first"] == "shard two failed" and out["second"] == "resolved", out


@pytest.mark.gpu
def test_addon_context_handle_is_safe_after_destroy():
    """napi/wsa_napi.c: the JS handle of a context is a box that outlives wsa_destroy — destroy() is refused while a stream created from the
    context is open, a second destroy() is a no-op and any use of the handle afterwards throws instead of touching freed memory."""
    import torch
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    prog = ("const nat=require(process.argv[1]);const d=nat.defaults();d.output_level=5;const c=nat.create(d,0);const st=nat.streamOpen(c,4,16000,1,256);const out={};"
            "try{nat.destroy(c);out.a='destroyed'}catch(e){out.a=String(e.message)}nat.streamClose(st);nat.destroy(c);nat.destroy(c);"
            "try{nat.geometry(c,16000);out.b='used'}catch(e){out.b=String(e.message)}process.stdout.write(JSON.stringify(out));")
    r = subprocess.run([NODE, "-e", prog, os.path.join(ROOT, "webspeechanalyzer_amd", "lib", "wsa_napi.node")], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    out = json.loads(r.stdout)
    assert "open streams" in out["a"] and out["b"] == "geometry(ctx, fs)"
