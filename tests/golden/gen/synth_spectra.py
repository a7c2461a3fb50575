"""Deterministic synthetic u32 spectrogram clips (formant-like ridges, syllabic envelopes, pauses,
drop-outs) used to drive both the reference (through ref_driver.js) and the oracle / HIP back end.
Pure numpy; the fixtures store the generated arrays, so this file only has to be reproducible
within one generator run."""
import numpy as np


def synth_clip(seed, frames=400, bands=128):
    rng = np.random.default_rng(seed)
    k = np.arange(bands)[None, :]
    spec = np.zeros((frames, bands))
    scale = 10.0 ** rng.uniform(2.0, 7.6)
    noise = scale * 10.0 ** rng.uniform(-4.0, -1.5)
    t = int(rng.integers(0, 20))
    while t < frames:
        dur = int(rng.integers(6, 90))
        end = min(frames, t + dur)
        n = end - t
        nform = int(rng.integers(3, 7))
        centers = np.sort(rng.uniform(9, bands * 0.8, nform))
        syl_rate = rng.uniform(0.03, 0.25)
        phase = rng.uniform(0, 6.28)
        env = 0.5 * (1 - np.cos(np.clip(np.arange(n) / max(n - 1, 1), 0, 1) * 2 * np.pi)) ** 0.3
        env = env * (0.55 + 0.45 * np.sin(phase + 2 * np.pi * syl_rate * np.arange(n)) ** 2)
        if rng.random() < 0.3 and n > 20:           # a hard dip inside the segment
            a = int(rng.integers(5, n - 8)); b = a + int(rng.integers(1, 6))
            env[a:b] *= rng.uniform(0.0, 0.05)
        for c in centers:
            drift = np.cumsum(rng.normal(0, rng.uniform(0.05, 0.9), n))
            width = rng.uniform(0.8, 3.5)
            amp = scale * 10.0 ** rng.uniform(-1.5, 0.0)
            ridge = np.exp(-0.5 * ((k - (c + drift)[:, None]) / width) ** 2)
            drop = rng.random(n) < rng.uniform(0, 0.08)
            e = env * amp
            e[drop] = 0
            spec[t:end] += e[:, None] * ridge
        t = end + int(rng.integers(2, 40))
    spec += rng.uniform(0, noise, spec.shape)
    if rng.random() < 0.25:                          # flat plateaus / ties exercise the flat counter
        spec = np.floor(spec / (scale / 64)) * (scale / 64)
    return np.minimum(spec, 4294967295.0).astype(np.uint32)


def _ridges(frames, bands, centers, widths, amps):
    k = np.arange(bands)[None, :]
    spec = np.zeros((frames, bands))
    for c, w, a in zip(centers, widths, amps):
        spec += a * np.exp(-0.5 * ((k - c) / w) ** 2)
    return spec


def designed_clip(name, frames, bands=128):
    """Hand-built clips aimed at single rules of the reference (G2):
    gate_sweep  five steady ridges whose level ramps over 7.5 decades and back, in bursts separated by pauses, so that the auto noise
                gate C(h) (ref @B28506) walks ctx_max through every branch of its piecewise floor (log10(y) = 1, 2, 4, 6, 7, ref @B28615)
    syl_edges   one long segment whose energy is on for u = 5 / 11 / 21 frames and off for 1 / 2 / 5 frames in every combination: the
                run / gap rule of sep_syllables (ref @B34864: (u>20 && c>0) || (u>10 && c>1) || (u>0 && c>4) || (last && u>4))"""
    rng = np.random.default_rng(12345)
    base = _ridges(1, bands, [14, 27, 43, 61, 80, 99], [1.6, 2.0, 2.2, 2.6, 3.0, 3.0], [1.0, 0.25, 0.2, 0.15, 0.1, 0.1])[0]   # one dominant ridge: the start test (ref @B26527) wants h (n - 1) / (d - h) > 4
    spec = np.zeros((frames, bands))
    if name == "gate_sweep":
        f = 0
        burst = 0
        while f < frames:
            n = min(frames - f, 34)
            x = (f + np.arange(n)) / frames
            level = 10.0 ** (0.7 + 7.6 * (1 - np.abs(2 * x - 1)))            # up to ~2e8 and back down
            wob = 1.0 + 0.3 * np.sin(0.9 * np.arange(n) + burst)
            spec[f:f + n] = (level * wob)[:, None] * base[None, :]
            f += n + 11                                                        # pause of 11 frames: the segment closes (breaker = 8)
            burst += 1
        spec += rng.uniform(0, 0.6, spec.shape)
    elif name == "syl_edges":
        f = 12
        for rep in range(2):
            for u in (5, 11, 21):
                for gap in (1, 2, 5):
                    if f + u + gap >= frames - 12:
                        break
                    spec[f:f + u] = 3.0e4 * (1.0 + 0.2 * np.sin(np.arange(u) + rep))[:, None] * base[None, :]
                    f += u + gap
        spec += rng.uniform(0, 1.5, spec.shape)
    else:
        raise KeyError(name)
    return np.minimum(spec, 4294967295.0).astype(np.uint32)
