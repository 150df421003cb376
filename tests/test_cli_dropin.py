"""BASELINE config 1 through the reference's own CLI.

* CPU: the reference CLI (oracle/_ref/velox_ref_cli, compiled from the reference's main.cpp) round-trips
  a WAV, and the payload inside its .vlx equals the oracle's — the oracle pinned through the CLI path.
* GPU: the SAME main.cpp compiled against velox_amd/cpp/VeloxCodecAMD.h (oracle/_ref/velox_amd_cli) writes
  byte-identical .vlx files and decodes them to byte-identical WAVs: the drop-in, end to end, including
  header / footer blobs, the pad byte and float input.
Both binaries exist only where the reference tree was present at build time; tests skip otherwise."""
import os
import struct
import subprocess

import numpy as np
import pytest

from conftest import ROOT

REF_CLI = os.path.join(ROOT, "oracle", "_ref", "velox_ref_cli")
AMD_CLI = os.path.join(ROOT, "oracle", "_ref", "velox_amd_cli")


def _wav(path, data_bytes, rate, ch, bits, fmt=1, extra_before=b"", extra_after=b""):
    pad = b"\x00" if len(data_bytes) % 2 else b""
    fmt_chunk = struct.pack("<4sIHHIIHH", b"fmt ", 16, fmt, ch, rate, rate * ch * bits // 8, ch * bits // 8, bits)
    body = b"WAVE" + fmt_chunk + extra_before + b"data" + struct.pack("<I", len(data_bytes)) + data_bytes + pad + extra_after
    with open(path, "wb") as fh:
        fh.write(b"RIFF" + struct.pack("<I", len(body)) + body)


def _cases(tmp):
    """(name, int64 samples or None) — WAV files written into tmp."""
    from velox_amd.synth import pack_pcm, synth_pcm
    out = []
    s = synth_pcm(16, 2, 44100, 60.0, 8000.0, 150.0, 1234)               # BASELINE configs[0]
    _wav(os.path.join(tmp, "cfg1.wav"), pack_pcm(s, 16).tobytes(), 44100, 2, 16)
    out.append(("cfg1", s))
    s = synth_pcm(24, 2, 96000, 0.5, 2.0e6, 3000.0, 5)
    lst = b"LIST" + struct.pack("<I", 12) + b"INFOISFT" + struct.pack("<I", 0)   # a chunk after the data -> footer blob
    _wav(os.path.join(tmp, "s24_extra.wav"), pack_pcm(s, 24).tobytes(), 96000, 2, 24, extra_after=lst)
    out.append(("s24_extra", s))
    s3 = synth_pcm(24, 3, 48000, 0.1, 1.0e6, 500.0, 7)                  # 3 channels, 14400 samples
    _wav(os.path.join(tmp, "s24_3ch.wav"), pack_pcm(s3, 24).tobytes(), 48000, 3, 24)
    out.append(("s24_3ch", s3))
    t = np.arange(20000) / 48000.0
    f = (0.4 * np.sin(2 * np.pi * 440 * t) + 0.01 * np.random.default_rng(3).standard_normal(t.size)).astype("<f4").repeat(2)
    _wav(os.path.join(tmp, "float.wav"), f.tobytes(), 48000, 2, 32, fmt=3)
    out.append(("float", None))
    s8 = np.random.default_rng(8).integers(-2000, 2000, 2 * 5001).astype(np.int64)
    _wav(os.path.join(tmp, "odd_bytes.wav"), pack_pcm(s8, 16).tobytes() + b"\x07", 22050, 2, 16)   # odd data size: pad byte, bits_flag 0x8000
    out.append(("odd_bytes", None))
    return out


def _payload_of(vlx):
    magic, ver, rate, ch, bits, fmt, total, hb, fb = struct.unpack_from("<IHIHHHQII", vlx, 0)
    assert magic == 0x584C4556 and ver == 0x0800
    pos = 40
    (meta_len,) = struct.unpack_from("<I", vlx, pos)                      # metadata block: u32 size + payload, 4 KiB padded
    pos += 4 + meta_len
    return vlx[pos + hb + fb:], total


def _run(cli, *args, env=None):
    r = subprocess.run([cli, *args], capture_output=True, text=True, timeout=300, env=dict(os.environ, **env) if env else None)
    assert r.returncode == 0, r.stdout + r.stderr


@pytest.mark.skipif(not os.path.exists(REF_CLI), reason="reference CLI not built (needs /root/reference)")
def test_reference_cli_roundtrip_and_oracle_payload(tmp_path, oracle):
    tmp = str(tmp_path)
    for name, s in _cases(tmp)[:2]:
        wav, vlx, back = (os.path.join(tmp, name + e) for e in (".wav", ".vlx", ".out.wav"))
        _run(REF_CLI, "-c", wav, vlx, "Artist", "Title")
        _run(REF_CLI, "-d", vlx, back)
        assert open(back, "rb").read() == open(wav, "rb").read()
        payload, total = _payload_of(open(vlx, "rb").read())
        assert total == s.size
        assert np.array_equal(np.frombuffer(payload, np.uint8), oracle.encode(s)), name


@pytest.mark.gpu
@pytest.mark.skipif(not (os.path.exists(REF_CLI) and os.path.exists(AMD_CLI)), reason="drop-in CLI not built (needs /root/reference)")
def test_dropin_cli_files_are_byte_identical(tmp_path):
    tmp = str(tmp_path)
    for name, _ in _cases(tmp):
        wav = os.path.join(tmp, name + ".wav")
        v_ref, v_amd = os.path.join(tmp, name + ".ref.vlx"), os.path.join(tmp, name + ".amd.vlx")
        _run(REF_CLI, "-c", wav, v_ref, "Artist", "Title")
        _run(AMD_CLI, "-c", wav, v_amd, "Artist", "Title")
        assert open(v_ref, "rb").read() == open(v_amd, "rb").read(), f"{name}: .vlx differs"
        w_ref, w_amd = os.path.join(tmp, name + ".ref.wav"), os.path.join(tmp, name + ".amd.wav")
        _run(REF_CLI, "-d", v_ref, w_ref)
        _run(AMD_CLI, "-d", v_ref, w_amd)
        assert open(w_ref, "rb").read() == open(w_amd, "rb").read(), f"{name}: decoded WAV differs"
        if name != "odd_bytes":
            assert open(w_amd, "rb").read() == open(wav, "rb").read(), f"{name}: did not round-trip"


@pytest.mark.gpu
@pytest.mark.skipif(not (os.path.exists(REF_CLI) and os.path.exists(AMD_CLI)), reason="drop-in CLI not built (needs /root/reference)")
@pytest.mark.parametrize("devices", ["0,0,0", "0,0,0,0,0", "all"])
def test_dropin_cli_over_a_pool_of_contexts(tmp_path, devices):
    """The reference's unmodified main.cpp on several GPUs: VELOX_AMD_DEVICES puts a vlx_pool behind ProcessBlock (one
    worker thread + context per entry; here K virtual shards on the one GPU of the box, "all" = every visible GPU).
    Config 1 and the other CLI cases stay byte-identical to the reference CLI's files."""
    tmp = str(tmp_path)
    env = {"VELOX_AMD_DEVICES": devices}
    for name, _ in _cases(tmp):
        wav = os.path.join(tmp, name + ".wav")
        v_ref, v_amd = os.path.join(tmp, name + ".ref.vlx"), os.path.join(tmp, name + ".amd.vlx")
        _run(REF_CLI, "-c", wav, v_ref, "Artist", "Title")
        _run(AMD_CLI, "-c", wav, v_amd, "Artist", "Title", env=env)
        assert open(v_ref, "rb").read() == open(v_amd, "rb").read(), f"{name}: .vlx differs"
        w_amd = os.path.join(tmp, name + ".amd.wav")
        _run(AMD_CLI, "-d", v_ref, w_amd, env=env)
        if name != "odd_bytes":
            assert open(w_amd, "rb").read() == open(wav, "rb").read(), f"{name}: did not round-trip"
