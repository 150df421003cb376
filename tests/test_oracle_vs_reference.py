"""Pins the C oracle to the REAL reference (oracle/_ref, compiled from /root/reference) on seeded
random inputs beyond the golden set, stage by stage.  Skips when the reference build is absent."""
import numpy as np
import pytest

from cases import float_cases, odd_cases, pcm_cases, wide_cases


@pytest.mark.parametrize("case", pcm_cases() + odd_cases() + wide_cases(), ids=lambda c: c[0])
def test_whole_payload(oracle, reference, case):
    _, _, s = case
    a = reference.encode(s)
    assert np.array_equal(oracle.encode(s), a)
    if s.size % 2 == 0:
        assert np.array_equal(oracle.decode(a, s.size), reference.decode(a, s.size))


@pytest.mark.parametrize("seed", range(12))
def test_fuzz_streams(oracle, reference, seed):
    rng = np.random.default_rng(1000 + seed)
    n = int(rng.integers(1, 4097))
    scale = 2 ** int(rng.integers(1, 31))
    kind = seed % 4
    if kind == 0:
        x = rng.integers(-scale, scale, n)
    elif kind == 1:
        x = np.rint(scale * np.sin(np.arange(n) * rng.uniform(0.001, 1.0)) + rng.normal(0, scale / 300 + 1, n)).astype(np.int64)
    elif kind == 2:
        x = np.cumsum(rng.integers(-50, 51, n)) * int(rng.integers(1, 9))
    else:
        x = np.where(rng.random(n) < 0.01, rng.integers(-scale, scale, n), 0)
    x = x.astype(np.int64)
    for hr in (0, 1):
        a = reference.try_compress_channel(x, hr)
        b, P, nbits = oracle.encode_stream(x, hr)
        assert np.array_equal(a, b)
        assert np.array_equal(oracle.decode_stream(a, n, hr), reference.decode_channel(a, n, hr))
    if x.any():
        assert oracle.compute_lpc(x) == reference.compute_lpc(x)


def test_lpc_large_energy(oracle, reference):
    """R[0] >= 2^53: the sequential double accumulation order matters."""
    rng = np.random.default_rng(5)
    x = (rng.integers(-2 ** 30, 2 ** 30, 4096)).astype(np.int64)
    assert oracle.compute_lpc(x) == reference.compute_lpc(x)
    y = np.rint(2.0 ** 29 * np.sin(np.arange(4096) * 0.05)).astype(np.int64)
    assert oracle.compute_lpc(y) == reference.compute_lpc(y)
    z = np.rint(2.0 ** 29 * np.sin(np.arange(5000) * 0.05)).astype(np.int64)   # stride-4 path (> 4096)
    assert oracle.compute_lpc(z) == reference.compute_lpc(z)


@pytest.mark.parametrize("case", float_cases(), ids=lambda c: c[0])
def test_float_payload(oracle, reference, case):
    _, f = case
    raw = f.view(np.uint8)
    m, e = oracle.split_float32(raw)
    assert oracle.detect_pseudo_float(raw) == reference.detect_pseudo_float(raw)
    a = reference.encode(m, True, e, raw)
    assert np.array_equal(oracle.encode(m, True, e, raw), a)
    ra, rb = reference.decode(a, m.size, True), oracle.decode(a, m.size, True)
    assert np.array_equal(ra[0], rb[0]) and np.array_equal(ra[1], rb[1]) and ra[2:] == rb[2:]


def test_truncated_and_garbage_payloads(oracle, reference):
    """Reads past the end are zero bits (VeloxEntropy.h:61-62); a zero chunk size ends decoding."""
    from conftest import load_golden
    g = load_golden("s16_stereo_44k")
    p, n = g["payload"], g["samples"].size
    for cut in (p.size // 2, p.size - 3, 7, 1):
        assert np.array_equal(oracle.decode(p[:cut], n), reference.decode(p[:cut], n))
    rng = np.random.default_rng(0)
    junk = p.copy()
    junk[200:260] = rng.integers(0, 256, 60)
    assert np.array_equal(oracle.decode(junk, n), reference.decode(junk, n))


def test_int64_min_samples_are_not_loud(oracle, reference):
    """std::abs(INT64_MIN) stays negative in the reference's build, so a track of small values and INT64_MIN
    (what DemoteFloatToInt makes of inf / NaN / 1e30f, VeloxFormat.h:68) has high_res_mode 0 (VeloxCore.h:234)."""
    rng = np.random.default_rng(5)
    s = rng.integers(-3000, 3000, 2 * 9000).astype(np.int64)
    s[5::97] = np.iinfo(np.int64).min
    a, b = oracle.encode(s), reference.encode(s)
    assert np.array_equal(a, b) and (int(a[0]) >> 1) & 1 == 0
