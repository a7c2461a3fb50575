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
