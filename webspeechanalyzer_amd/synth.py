"""Synthetic speech-like PCM for the benchmark workload: harmonic source with slow
vibrato shaped by three moving formant resonances, a 3-6 Hz syllabic envelope, pauses >= 250 ms
every 0.7-2 s, a -45 dBFS noise floor, peak about 0.5.  Generated with torch on whatever device it
is asked for (the GPU for bench.py, the CPU in tests); float32 mono in [-1, 1)."""
import math

import torch


def synth_clips(n_clips, n_samples, fs=16000, seed=0, device="cpu", chunk=256):
    out = torch.empty((n_clips, n_samples), dtype=torch.float32, device=device)
    for c0 in range(0, n_clips, chunk):
        c1 = min(n_clips, c0 + chunk)
        out[c0:c1] = _synth(c1 - c0, n_samples, fs, seed * 1000003 + c0, device)
    return out


def _ctrl(gen, n, steps, lo, hi, device):
    """piecewise-linear random control track: `steps` knots per clip in [lo, hi]."""
    return lo + (hi - lo) * torch.rand((n, steps), generator=gen, device=device)


def _interp(knots, n_samples):
    return torch.nn.functional.interpolate(knots[:, None, :], size=n_samples, mode="linear", align_corners=True)[:, 0, :]


def _synth(n, ns, fs, seed, device):
    gen = torch.Generator(device=device)
    gen.manual_seed(seed)
    dur = ns / fs
    t = torch.arange(ns, device=device, dtype=torch.float32) / fs
    knots = max(2, int(dur / 0.14) + 1)                       # formant targets move every ~140 ms
    f0 = _interp(_ctrl(gen, n, max(2, int(dur / 0.5) + 1), 90.0, 250.0, device), ns)
    f0 = f0 * (1.0 + 0.02 * torch.sin(2 * math.pi * 5.0 * t)[None, :])
    phase = 2 * math.pi * torch.cumsum(f0, dim=1) / fs
    F = [_interp(_ctrl(gen, n, knots, lo, hi, device), ns) for lo, hi in ((300.0, 900.0), (900.0, 2400.0), (2400.0, 3500.0))]
    bw = (90.0, 130.0, 180.0)
    gains = (1.0, 0.6, 0.35)
    # voiced / pause gating: alternate voiced stretches (0.7-2 s) and pauses (0.25-0.6 s)
    gate = torch.zeros((n, ns), device=device)
    pos = torch.rand((n,), generator=gen, device=device) * 0.3
    idx = torch.arange(ns, device=device, dtype=torch.float32)[None, :] / fs
    for _ in range(int(dur / 0.95) + 2):
        von = 0.7 + 1.3 * torch.rand((n,), generator=gen, device=device)
        poff = 0.25 + 0.35 * torch.rand((n,), generator=gen, device=device)
        a, b = pos[:, None], (pos + von)[:, None]
        ramp = torch.clamp((idx - a) / 0.03, 0, 1) * torch.clamp((b - idx) / 0.05, 0, 1)
        gate = torch.maximum(gate, ramp)
        pos = pos + von + poff
    syl = 3.0 + 3.0 * torch.rand((n, 1), generator=gen, device=device)
    ph0 = 6.28 * torch.rand((n, 1), generator=gen, device=device)
    # syllabic envelope with short dead zones (2-3 frames) so that syllable splits occur
    env = gate * torch.clamp((torch.sin(math.pi * syl * t[None, :] + ph0) ** 2 - 0.15) / 0.85, 0, 1)
    x = torch.zeros((n, ns), device=device)
    nh = int(4000.0 / 90.0)
    for h in range(1, nh + 1):
        fh = h * f0
        amp = torch.zeros_like(fh)
        for Fk, bk, gk in zip(F, bw, gains):
            amp = amp + gk / (1.0 + ((fh - Fk) / bk) ** 2)
        amp = amp * (fh < 3900.0) / h ** 0.5
        x = x + amp * torch.sin(h * phase)
    x = x * env
    peak = x.abs().amax(dim=1, keepdim=True).clamp_min(1e-6)
    x = 0.5 * x / peak
    x = x + (10 ** (-45 / 20)) * torch.randn((n, ns), generator=gen, device=device)
    return x.clamp_(-0.999, 0.999).to(torch.float32)
