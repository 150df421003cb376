"""velox_amd/container.py against the reference's own CLI (oracle/_ref/velox_ref_cli, compiled from the
reference's main.cpp where the reference tree is present; the binary travels to the GPU box).

* CPU: everything around the codec — input parsing, tag import, header, metadata block, blobs — by
  assembling the file around the ORACLE's payload and comparing the whole .vlx with the reference CLI's.
* GPU: encode_audio_file / decode_vlx_file end to end: byte-identical .vlx, byte-identical WAV back."""
import os
import struct
import subprocess

import numpy as np
import pytest

from conftest import ROOT

REF_CLI = os.path.join(ROOT, "oracle", "_ref", "velox_ref_cli")
needs_ref = pytest.mark.skipif(not os.path.exists(REF_CLI), reason="reference CLI not built (needs /root/reference at build time)")


def _chunk(cid, body):
    return cid + struct.pack("<I", len(body)) + body + (b"\0" if len(body) % 2 else b"")


def _wav(data, rate, ch, bits, fmt=1, before=b"", after=b""):
    fmt_c = _chunk(b"fmt ", struct.pack("<HHIIHH", fmt, ch, rate, rate * ch * bits // 8, ch * bits // 8, bits))
    body = b"WAVE" + fmt_c + before + _chunk(b"data", data) + after
    return b"RIFF" + struct.pack("<I", len(body)) + body


def _aiff(data_be, rate, ch, bits, frames):
    # 80-bit extended sample rate: exponent 16383 + floor(log2(rate)), mantissa with the explicit leading one
    e = rate.bit_length() - 1
    ext = struct.pack(">HQ", 16383 + e, rate << (63 - e))
    comm = b"COMM" + struct.pack(">I", 18) + struct.pack(">HIH", ch, frames, bits) + ext
    ssnd = b"SSND" + struct.pack(">I", 8 + len(data_be)) + struct.pack(">II", 0, 0) + data_be + (b"\0" if len(data_be) % 2 else b"")
    body = b"AIFF" + comm + ssnd
    return b"FORM" + struct.pack(">I", len(body)) + body


def cases():
    """(file name, file bytes, [artist, title] or [])"""
    from velox_amd.synth import pack_pcm, synth_pcm
    out = []
    s = synth_pcm(16, 2, 44100, 0.7, 8000.0, 150.0, 1234)
    out.append(("plain16.wav", _wav(pack_pcm(s, 16).tobytes(), 44100, 2, 16), ["Some Artist", "A Title"]))
    info = b"INFO" + _chunk(b"INAM", b"Imported Title\0") + _chunk(b"IART", b"Imported Artist") + _chunk(b"ICMT", b"ignored")
    s = synth_pcm(24, 2, 96000, 0.3, 2.0e6, 3000.0, 5)
    out.append(("tagged24.wav", _wav(pack_pcm(s, 24).tobytes(), 96000, 2, 24, before=_chunk(b"LIST", info), after=_chunk(b"cue ", b"\1\2\3\4\5\6")), []))
    out.append(("untagged24.wav", _wav(pack_pcm(s[:20000], 24).tobytes(), 48000, 2, 24), []))
    out.append(("artist_only.wav", _wav(pack_pcm(s[:9000], 24).tobytes(), 48000, 2, 24), ["Only Artist"]))
    s8 = np.random.default_rng(8).integers(-2000, 2000, 2 * 5001).astype(np.int64)
    out.append(("odd_bytes.wav", _wav(pack_pcm(s8, 16).tobytes() + b"\x07", 22050, 2, 16), ["a", "b"]))
    t = np.arange(20000) / 48000.0
    f = (0.4 * np.sin(2 * np.pi * 440 * t) + 0.01 * np.random.default_rng(3).standard_normal(t.size)).astype("<f4").repeat(2)
    out.append(("float.wav", _wav(f.tobytes(), 48000, 2, 32, fmt=3), []))
    pf = (np.random.default_rng(4).integers(-32768, 32768, 30000) / 32768.0).astype("<f4")
    out.append(("pseudo_float16.wav", _wav(pf.tobytes(), 44100, 2, 32, fmt=3), []))
    s = synth_pcm(16, 2, 44100, 0.4, 9000.0, 100.0, 9)
    out.append(("big_endian.aif", _aiff(pack_pcm(s, 16).reshape(-1, 2)[:, ::-1].tobytes(), 44100, 2, 16, s.size // 2), ["x", "y"]))
    s = synth_pcm(24, 2, 48000, 0.2, 1.0e6, 500.0, 10)
    out.append(("big_endian24.aif", _aiff(pack_pcm(s, 24).reshape(-1, 3)[:, ::-1].tobytes(), 48000, 2, 24, s.size // 2), []))
    return out


def _ref(tmp, name, blob, tags):
    src, vlx, back = (os.path.join(tmp, name), os.path.join(tmp, name + ".vlx"), os.path.join(tmp, name + ".out.wav"))
    with open(src, "wb") as fh:
        fh.write(blob)
    for args in (["-c", src, vlx] + tags, ["-d", vlx, back]):
        r = subprocess.run([REF_CLI] + args, capture_output=True, text=True, timeout=120)
        assert r.returncode == 0, r.stdout + r.stderr
    return open(vlx, "rb").read(), open(back, "rb").read()


@needs_ref
def test_container_pieces_match_the_reference_cli(tmp_path, oracle):
    from velox_amd import container as C
    for name, blob, tags in cases():
        ref_vlx, _ = _ref(str(tmp_path), name, blob, tags)
        info, raw, t, hb, fb = C.prepare_audio_file(blob, name, *tags)
        total = C.sample_count(info, raw)
        if info.format_code == 3:
            m, e = C.split_float32(raw[: 4 * total])
            om, oe = oracle.split_float32(np.frombuffer(raw[: 4 * total], np.uint8))
            assert np.array_equal(m, om) and np.array_equal(e, oe)
            payload = oracle.encode(m, True, e, np.frombuffer(raw, np.uint8))
        else:
            payload = oracle.encode(oracle.bytes_to_samples(np.frombuffer(raw, np.uint8)[: total * (info.bits // 8)], info.bits))
        mine = C.assemble_vlx(info, len(raw), total, t, hb, fb, payload.tobytes())
        assert mine == ref_vlx, name
        v = C.parse_vlx(ref_vlx)
        assert v["total_samples"] == total and v["tags"][b"ENCODER"] == b"Velox v1.1" and v["header_blob"] == hb and v["footer_blob"] == fb
        assert bytes(v["payload"]) == payload.tobytes()


def test_metadata_block_padding_rule():
    """4 + block is a multiple of 4096; an exact fit gets a whole extra 4096 (VeloxMetadata.h:115-130)."""
    from velox_amd import container as C
    b = C.metadata_block({b"ARTIST": b"a", b"TITLE": b"t"})
    assert len(b) == 4096 and struct.unpack_from("<I", b)[0] == 4092
    base = len(C.metadata_block({b"K": b""})) and 4 + 4 + len(C.VENDOR) + 4 + 4 + len(b"K=") + 1
    exact = C.metadata_block({b"K": b"x" * (4096 - base)})
    assert len(exact) == 8192
    tags, end = C.read_metadata_block(exact, 0)
    assert end == 8192 and tags == {b"K": b"x" * (4096 - base)}


@pytest.mark.gpu
@needs_ref
def test_whole_files_match_the_reference_cli(tmp_path, ctx):
    from velox_amd import container as C
    for name, blob, tags in cases():
        ref_vlx, ref_wav = _ref(str(tmp_path), name, blob, tags)
        mine = C.encode_audio_file(blob, name, *tags, ctx=ctx)
        assert mine == ref_vlx, name
        assert C.decode_vlx_file(mine, ctx=ctx) == ref_wav, name


@pytest.mark.gpu
@needs_ref
def test_whole_files_over_a_pool(tmp_path, monkeypatch):
    """The same files with VELOX_AMD_DEVICES naming several devices (three shards on the one GPU of the box): the
    container module then encodes and decodes whole tracks through velox_amd.codec.Pool (vlx_pool_*) — same bytes."""
    from velox_amd import container as C
    monkeypatch.setenv("VELOX_AMD_DEVICES", "0,0,0")
    for name, blob, tags in cases():
        ref_vlx, ref_wav = _ref(str(tmp_path), name, blob, tags)
        mine = C.encode_audio_file(blob, name, *tags)
        assert mine == ref_vlx, name
        assert C.decode_vlx_file(mine) == ref_wav, name
