"""Seeded input catalogue shared by the golden-vector generator and the parity tests.

Each case is (name, bits, int64 interleaved samples).  Sizes are small (2-5 sub-blocks) so that
the CPU oracle finishes in milliseconds; the shapes follow SURVEY.md §8(c)'s fixture list and
the edge cases the reference's code paths distinguish (VeloxCore.h:73-81, :232-236, :242-253,
:284-293; VeloxFormat.h:196-212; VeloxEntropy.h:104-118).
"""
import numpy as np

from velox_amd.synth import synth_pcm


def _tone(bits, ch, rate, sec, amp, sigma, seed):
    return synth_pcm(bits, ch, rate, sec, amp, sigma, seed)


def pcm_cases():
    rng = np.random.default_rng(77)
    cases = []
    add = lambda n, b, s: cases.append((n, b, np.asarray(s, dtype=np.int64)))
    add("s16_stereo_44k", 16, _tone(16, 2, 44100, 0.5, 8000, 150, 1234))
    add("s24_stereo_96k_hires", 24, _tone(24, 2, 96000, 0.25, 2.0e6, 3000, 2))
    add("s24_quiet_lowres", 24, _tone(24, 2, 96000, 0.2, 30000, 200, 3))
    add("s32_8ch_192k_loud", 32, _tone(32, 8, 192000, 0.02, 5.0e8, 2.0e5, 4))       # R[0] >= 2^53
    add("s32_8ch_quiet", 32, _tone(32, 8, 192000, 0.02, 40000, 100, 5))
    add("s16_mono_even", 16, _tone(16, 1, 44100, 0.4, 9000, 80, 6))
    add("s16_3ch", 16, _tone(16, 3, 48000, 0.2, 7000, 120, 7)[: 3 * 9600 // 2 * 2])
    add("silence", 16, np.zeros(20000, dtype=np.int64))
    x = _tone(16, 2, 44100, 0.3, 8000, 100, 8).reshape(-1, 2).copy(); x[:, 1] = 0
    add("one_channel_silent", 16, x.reshape(-1))
    add("lsb_shifted_x16", 24, _tone(16, 2, 48000, 0.25, 6000, 90, 9) * 16)
    add("white_noise_fullscale_s16", 16, rng.integers(-32768, 32768, 30000))
    add("white_noise_fullscale_s32", 32, rng.integers(-2 ** 31, 2 ** 31, 20000))    # chunk stays < raw limit
    add("low_freq_pure_tone", 16, np.rint(30000 * np.sin(2 * np.pi * 55.0 * np.arange(40000) / 44100.0)).astype(np.int64).repeat(2))
    y = _tone(24, 2, 96000, 0.15, 3.0e6, 10, 10).reshape(-1, 2).copy(); y[:, 1] = y[:, 0] + rng.integers(0, 3, y.shape[0])
    add("near_identical_channels_hires", 24, y.reshape(-1))                          # lossy quirk, SURVEY §4
    add("short_tail_block", 16, _tone(16, 2, 44100, 0.2, 8000, 150, 11)[: 8192 * 2 + 346])
    add("tiny_2_samples", 16, np.array([5, -7]))
    add("tiny_20_samples", 16, rng.integers(-100, 100, 20))
    add("impulses_escape_codes", 32, np.where(rng.random(24000) < 0.002, rng.integers(-2 ** 30, 2 ** 30, 24000), rng.integers(-3, 4, 24000)))
    add("dc_offset_constant", 16, np.full(17000, 1234).astype(np.int64))
    add("alternating_max", 16, np.tile(np.array([32767, 32767, -32768, -32768]), 5000))
    add("square_wave_s24", 24, (np.sign(np.sin(2 * np.pi * 440 * np.arange(20000) / 48000.0)) * 4000000).astype(np.int64).repeat(2))
    add("ramp_up", 24, np.arange(-40000, 40000, 2).repeat(2))
    add("empty", 16, np.zeros(0, dtype=np.int64))
    return cases


def odd_cases():
    """Odd sample totals: encode-only parity (the reference cannot decode them)."""
    rng = np.random.default_rng(5)
    return [
        ("odd_small", 16, _tone(16, 1, 44100, 0.05, 9000, 80, 21)[:2001]),
        ("odd_over_4096", 16, _tone(16, 1, 44100, 0.3, 9000, 80, 22)[:12345]),
        ("odd_noise_raw", 32, rng.integers(-2 ** 31, 2 ** 31, 999).astype(np.int64)),
        ("odd_one", 16, np.array([3], dtype=np.int64)),
    ]


def wide_cases():
    """int64 samples beyond any PCM container (the boundary takes int64, VeloxArch.h:10)."""
    rng = np.random.default_rng(9)
    return [
        ("wide_2p40", 64, rng.integers(-2 ** 40, 2 ** 40, 16384).astype(np.int64)),
        ("wide_2p33_tone", 64, (np.rint(2.0 ** 33 * np.sin(np.arange(20000) * 0.01)).astype(np.int64)).repeat(2)[:30000]),
        ("wide_multiple_2p32", 64, (rng.integers(-1000, 1000, 8192).astype(np.int64)) << 32),
    ]


def float_cases():
    """(name, float32 array) for the float path (VeloxCore.h:214-230; VeloxFormat.h:14-128)."""
    rng = np.random.default_rng(3)
    t = np.arange(12000) / 48000.0
    f_true = (0.4 * np.sin(2 * np.pi * 440 * t) + 0.01 * rng.standard_normal(t.size)).astype(np.float32).repeat(2)
    i16 = np.rint(12000 * np.sin(2 * np.pi * 330 * t)).astype(np.int32).repeat(2)
    f_p16 = (i16.astype(np.float32) / np.float32(32768.0)).astype(np.float32)
    i24 = np.rint(3000000 * np.sin(2 * np.pi * 220 * t) + rng.normal(0, 50, t.size)).astype(np.int32).repeat(2)
    f_p24 = (i24.astype(np.float32) / np.float32(8388608.0)).astype(np.float32)
    sp = f_true.copy(); sp[5] = np.inf; sp[9] = np.nan; sp[11] = 0.0; sp[13] = np.float32(1e-42)
    return [("float_true", f_true), ("float_pseudo16", f_p16), ("float_pseudo24", f_p24), ("float_specials", sp)]
