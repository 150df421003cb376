"""The C oracle against the committed golden vectors (generated from the real reference by
tests/golden/make_golden.py).  CPU only."""
import numpy as np
import pytest

from conftest import golden_names, load_golden

PCM = [n for n in golden_names() if not n.startswith("float_")]
FLT = [n for n in golden_names() if n.startswith("float_")]


@pytest.mark.parametrize("name", PCM)
def test_encode_matches_golden(oracle, name):
    g = load_golden(name)
    got = oracle.encode(g["samples"].astype(np.int64))
    assert got.size == g["payload"].size
    assert np.array_equal(got, g["payload"])


@pytest.mark.parametrize("name", [n for n in PCM if not n.startswith("odd_")])
def test_decode_matches_golden(oracle, name):
    g = load_golden(name)
    s = g["samples"].astype(np.int64)
    want = g["decoded"].astype(np.int64) if "decoded" in g else s
    got = oracle.decode(g["payload"], s.size)
    assert np.array_equal(got, want)
    if s.size % 8192 == 0 or True:
        assert np.array_equal(oracle.decode_mt(g["payload"], s.size, 3), want)


@pytest.mark.parametrize("name", FLT)
def test_float_path_matches_golden(oracle, name):
    g = load_golden(name)
    m, e = oracle.split_float32(g["raw"])
    assert np.array_equal(m, g["mant"].astype(np.int64)) and np.array_equal(e, g["exps"])
    got = oracle.encode(m, True, e, g["raw"])
    assert np.array_equal(got, g["payload"])
    dv, de, isf, fm = oracle.decode(g["payload"], m.size, True)
    assert np.array_equal(dv, g["dec_vals"].astype(np.int64)) and np.array_equal(de, g["dec_exps"])
    assert isf == int(g["is_float"]) and fm == int(g["float_mode"])


def test_known_answer_invariants(oracle):
    """payload starts with bits is_float, high_res, then u32 chunk size at bit 2."""
    g = load_golden("s24_stereo_96k_hires")
    p = g["payload"]
    assert p[0] & 1 == 0 and (p[0] >> 1) & 1 == 1            # integer PCM, high-res on
    size0 = int.from_bytes(bytes(p[:5]), "little") >> 2 & 0xFFFFFFFF
    n1 = int.from_bytes(bytes(p[4 + size0: 4 + size0 + 5]), "little") >> 2 & 0xFFFFFFFF
    assert 0 < size0 < 8192 * 5 and 0 < n1 < 8192 * 5
    assert load_golden("empty")["payload"].tolist() == [0]
    assert load_golden("silence")["payload"].size == 16    # 3 chunks x (4 + 1) bytes + 1


def test_mt_encode_is_deterministic(oracle):
    g = load_golden("s16_stereo_44k")
    s = g["samples"].astype(np.int64)
    assert np.array_equal(oracle.encode(s, threads=4), g["payload"])


@pytest.mark.parametrize("name", [n for n in golden_names() if n.startswith("float_")])
def test_oracle_float_epilogue_restores_the_golden_input(oracle, name):
    """MergeFloat32 / PromoteIntToFloat (VeloxFormat.h:72-82, :107-128) over the golden decode output give
    back the float32 bytes the reference encoded: the codec is lossless on these fixtures (no -0.0 in them)."""
    g = load_golden(name)
    f32, fm = oracle.decode_f32(g["payload"], g["mant"].size)
    assert fm == int(g["float_mode"])
    assert np.array_equal(f32, np.ascontiguousarray(g["raw"], dtype=np.uint8))
