"""Seeded synthetic PCM used by tests and bench.py.

Per channel c:  A*sin(2*pi*f_c*t + 0.3c) + 0.3*A*sin(2*pi*3.1*f_c*t) + N(0, sigma),
f_c = 440*(1 + 0.26c) Hz, rounded to integers and clipped to the container width.
"""
import numpy as np

CONFIGS = {
    # name: (bits, channels, rate, seconds, A, sigma, seed)
    "cfg1": (16, 2, 44100, 60.0, 8000.0, 150.0, 1234),
    "cfg2": (24, 2, 96000, 3600.0, 2.0e6, 3000.0, 2),
    "cfg3_track": (24, 2, 48000, 10.0, 2.0e6, 3000.0, 0),
    "cfg4": (32, 8, 192000, 60.0, 5.0e8, 2.0e5, 4),
}


def synth_pcm(bits, channels, rate, seconds, amp, sigma, seed, t0=0.0):
    """Interleaved int64 samples (frames*channels,), numpy only."""
    frames = int(round(rate * seconds))
    rng = np.random.default_rng(seed)
    t = t0 + np.arange(frames, dtype=np.float64) / rate
    out = np.empty((frames, channels), dtype=np.int64)
    lim = (1 << (bits - 1)) - 1
    for c in range(channels):
        f = 440.0 * (1.0 + 0.26 * c)
        x = amp * np.sin(2 * np.pi * f * t + 0.3 * c) + 0.3 * amp * np.sin(2 * np.pi * 3.1 * f * t)
        x = x + rng.normal(0.0, sigma, frames)
        out[:, c] = np.clip(np.rint(x), -lim - 1, lim).astype(np.int64)
    return out.reshape(-1)


def pack_pcm(samples, bits):
    """int64 samples -> little-endian container bytes (uint8 array)."""
    s = np.asarray(samples, dtype=np.int64)
    if bits == 16:
        return s.astype("<i2").view(np.uint8).copy()
    if bits == 32:
        return s.astype("<i4").view(np.uint8).copy()
    if bits == 24:
        b = s.astype("<i4").view(np.uint8).reshape(-1, 4)[:, :3]
        return np.ascontiguousarray(b).reshape(-1)
    raise ValueError("bits must be 16, 24 or 32")


def unpack_pcm(raw, bits):
    r = np.asarray(raw, dtype=np.uint8)
    if bits == 16:
        return r.view("<i2").astype(np.int64)
    if bits == 32:
        return r.view("<i4").astype(np.int64)
    if bits == 24:
        b = r.reshape(-1, 3).astype(np.int32)
        v = b[:, 0] | (b[:, 1] << 8) | (b[:, 2] << 16)
        v = np.where(v & 0x800000, v - (1 << 24), v)
        return v.astype(np.int64)
    raise ValueError("bits must be 16, 24 or 32")


def synth_pcm_device(bits, channels, rate, frames, amp, sigma, seed, device, frame0=0):
    """Same signal shape generated on the GPU with torch (bench-size inputs; the RNG differs
    from numpy's, which is fine: parity is always checked on the same buffer).  Returns a uint8
    tensor of packed little-endian PCM, frames*channels*bits/8 bytes."""
    import torch
    g = torch.Generator(device=device)
    g.manual_seed(int(seed))
    W = bits // 8
    out = torch.empty(frames * channels * W, dtype=torch.uint8, device=device)
    ov = out.view(frames, channels, W)
    step = 1 << 22
    lim = (1 << (bits - 1)) - 1
    for f0 in range(0, frames, step):
        n = min(step, frames - f0)
        t = (torch.arange(n, device=device, dtype=torch.float64) + float(frame0 + f0)) / rate
        for c in range(channels):
            f = 440.0 * (1.0 + 0.26 * c)
            x = amp * torch.sin(2 * np.pi * f * t + 0.3 * c) + 0.3 * amp * torch.sin(2 * np.pi * 3.1 * f * t)
            x = x + torch.randn(n, device=device, dtype=torch.float32, generator=g).double() * sigma
            v = torch.clamp(torch.round(x), -lim - 1, lim).to(torch.int64)
            for b in range(W):
                ov[f0:f0 + n, c, b] = ((v >> (8 * b)) & 0xFF).to(torch.uint8)
    # A vlx_ctx runs on its own non-blocking stream, which is not ordered against torch's: hand the
    # buffer over only when it is complete.
    torch.cuda.synchronize(device)
    return out


def _splitmix64(x):
    """SplitMix64 finaliser on int64 tensors (wrapping arithmetic): a counter-based hash."""
    import torch
    def c(v):                                                            # uint64 constant as the int64 with the same bits
        return v - (1 << 64) if v >= (1 << 63) else v
    x = x + c(0x9E3779B97F4A7C15)
    x = (x ^ ((x >> 30) & 0x3FFFFFFFF)) * c(0xBF58476D1CE4E5B9)
    x = (x ^ ((x >> 27) & 0x1FFFFFFFFF)) * c(0x94D049BB133111EB)
    return x ^ ((x >> 31) & 0x1FFFFFFFF)


def synth_corpus_device(bits, channels, rate, frames, n_tracks, amp, sigma, device, stride=None, track0=0):
    """A corpus of `n_tracks` tracks of `frames` frames each (BASELINE configs[2]; SURVEY §8d: "seed = track
    index"): track t is the synth_pcm signal shape starting at time 0 with noise that is a pure function of
    (track0 + t, sample index) — a counter-based hash through Box-Muller — so any track can be regenerated from
    its index alone, whatever rank or batch it lands in.  Returns a uint8 tensor [n_tracks * stride (+16)] with
    track t at byte t * stride (stride defaults to the track's byte size rounded up to 16)."""
    import torch
    W = bits // 8
    nbytes = frames * channels * W
    stride = stride or (nbytes + 15) // 16 * 16
    out = torch.zeros(n_tracks * stride + 16, dtype=torch.uint8, device=device)
    lim = (1 << (bits - 1)) - 1
    t = torch.arange(frames, device=device, dtype=torch.float64) / rate
    tone = []
    for c in range(channels):
        f = 440.0 * (1.0 + 0.26 * c)
        tone.append(amp * torch.sin(2 * np.pi * f * t + 0.3 * c) + 0.3 * amp * torch.sin(2 * np.pi * 3.1 * f * t))
    tone = torch.stack(tone, 1)                                           # [frames, channels]
    idx = torch.arange(frames * channels, device=device, dtype=torch.int64).view(frames, channels)
    step = max(1, (1 << 25) // max(frames * channels, 1))                 # tracks per slab: ~32 M samples at a time
    for t0 in range(0, n_tracks, step):
        nt = min(step, n_tracks - t0)
        tr = (torch.arange(nt, device=device, dtype=torch.int64) + (track0 + t0)).view(nt, 1, 1)
        h1 = _splitmix64(idx.unsqueeze(0) * 2 + (tr << 40))
        h2 = _splitmix64(idx.unsqueeze(0) * 2 + 1 + (tr << 40))
        u1 = ((h1 >> 11) & ((1 << 53) - 1)).double() * (1.0 / (1 << 53)) + (0.5 / (1 << 53))
        u2 = ((h2 >> 11) & ((1 << 53) - 1)).double() * (1.0 / (1 << 53))
        noise = torch.sqrt(-2.0 * torch.log(u1)) * torch.cos(2 * np.pi * u2)
        v = torch.clamp(torch.round(tone.unsqueeze(0) + sigma * noise), -lim - 1, lim).to(torch.int64)
        dst = out[t0 * stride: (t0 + nt) * stride].view(nt, stride)[:, :nbytes].view(nt, frames, channels, W)
        for b in range(W):
            dst[..., b] = ((v >> (8 * b)) & 0xFF).to(torch.uint8)
    if torch.device(device).type == "cuda":
        torch.cuda.synchronize(device)                                    # a vlx_ctx stream is not ordered against torch's
    return out
