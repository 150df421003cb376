"""GPU parity tests proper: everything goes through the C-ABI (libvelox_amd.so) and is compared
bit-exactly with the golden vectors / the CPU oracle on the same inputs.  Integer work: bit-exact,
no tolerance anywhere."""
import numpy as np
import pytest

from conftest import golden_names, load_golden

pytestmark = pytest.mark.gpu

@pytest.fixture(params=["auto", "stream", "chunk", "pipe16", "pipe32", "pipe64"])
def wave_shape(request):
    """The fast decode path's shapes (vlx_set_decode_shape): "auto" sends small batches to the pipelined
    small-batch kernel and mixes stream and chunk waves for large ones; every decode test also runs with each
    shape forced, so every kernel shape sees every case."""
    from velox_amd.codec import set_decode_shape
    set_decode_shape(request.param)
    yield request.param
    set_decode_shape("auto")


@pytest.fixture(params=["auto", "stream", "pipe16", "pipe32", "pipe64"])
def enc_shape(request):
    """The sample-serial encode kernel's shapes (vlx_set_encode_shape): the throughput shape and the two
    pipelined small-batch shapes; "auto" picks by batch size."""
    from velox_amd.codec import set_encode_shape
    set_encode_shape(request.param)
    yield request.param
    set_encode_shape("auto")


PCM = [n for n in golden_names() if not n.startswith(("float_", "odd_"))]
FLT = [n for n in golden_names() if n.startswith("float_")]


@pytest.mark.parametrize("name", PCM)
def test_encode_block_matches_golden(ctx, name, enc_shape):
    g = load_golden(name)
    got = ctx.encode_block(g["samples"].astype(np.int64))
    assert got.size == g["payload"].size and np.array_equal(got, g["payload"])


ODD = [n for n in golden_names() if n.startswith("odd_")]


@pytest.mark.parametrize("name", ODD)
def test_odd_total_encode_matches_golden(ctx, name):
    """Odd sample totals (VeloxCore.h:242-253): one stream, no use_MS bit, stride-4 LPC above 4096
    samples, raw fallback at > 5 B/sample.  Encode parity only — the reference cannot decode these."""
    from velox_amd.synth import pack_pcm
    g = load_golden(name)
    s = g["samples"].astype(np.int64)
    assert np.array_equal(ctx.encode_block(s), g["payload"])
    bits = int(g["bits"])
    assert np.array_equal(ctx.encode_pcm(pack_pcm(s, bits), bits), g["payload"])


def test_odd_total_hires_vs_oracle(ctx, oracle):
    from velox_amd.synth import synth_pcm
    s = synth_pcm(24, 1, 96000, 0.3, 2.0e6, 3000.0, 77)[:20001]
    assert np.array_equal(ctx.encode_block(s), oracle.encode(s))
    z = np.zeros(4099, dtype=np.int64)
    assert np.array_equal(ctx.encode_block(z), oracle.encode(z))


@pytest.mark.parametrize("name", [n for n in PCM if not n.startswith("wide_")])
def test_encode_pcm_matches_golden(ctx, name):
    from velox_amd.synth import pack_pcm
    g = load_golden(name)
    bits = int(g["bits"])
    got = ctx.encode_pcm(pack_pcm(g["samples"].astype(np.int64), bits), bits)
    assert np.array_equal(got, g["payload"])


@pytest.mark.parametrize("name", PCM)
@pytest.mark.parametrize("lookahead", [1, 5])
def test_streaming_decoder_matches_golden(ctx, name, lookahead, wave_shape):
    from velox_amd.codec import StreamingDecoder
    g = load_golden(name)
    s = g["samples"].astype(np.int64)
    want = g["decoded"].astype(np.int64) if "decoded" in g else s
    d = StreamingDecoder(g["payload"], s.size, ctx=ctx, lookahead=lookahead)
    vals, exps = d.read(s.size + 7)
    assert np.array_equal(vals, want)
    assert not exps.any() and not d.IsFloat()
    ok, _, _ = d.DecodeNext()
    assert not ok


@pytest.mark.parametrize("name", FLT)
def test_float_path_matches_golden(ctx, name, wave_shape, enc_shape):
    from velox_amd.codec import StreamingDecoder
    g = load_golden(name)
    got = ctx.encode_block(g["mant"].astype(np.int64), True, g["exps"], g["raw"])
    assert np.array_equal(got, g["payload"])
    d = StreamingDecoder(g["payload"], g["mant"].size, ctx=ctx, lookahead=2)
    vals, exps = d.read(g["mant"].size)
    assert np.array_equal(vals, g["dec_vals"].astype(np.int64)) and np.array_equal(exps, g["dec_exps"])
    assert d.IsFloat() == bool(g["is_float"]) and d.GetFloatMode() == int(g["float_mode"])


def _f32_device_round_trip(ctx, oracle, raw_u8, want_payload=None):
    """raw float32 bytes -> vlx_encode_f32_device -> payload == the oracle's; -> vlx_decode_f32_device ->
    float32 bytes == Merge / Promote of the oracle's decode (main.cpp:253-270).  Bit patterns, no tolerance."""
    import torch
    n = raw_u8.size // 4
    if want_payload is None:
        m, e = oracle.split_float32(raw_u8)
        want_payload = oracle.encode(m, True, e, raw_u8)
    d_raw = torch.from_numpy(raw_u8.copy()).cuda()
    cap = ctx.encode_bound([n]) + 2 * n + 256
    d_pay = torch.zeros(cap + 16, dtype=torch.uint8, device="cuda")
    plen, fm = ctx.encode_f32_device(d_raw.data_ptr(), n, d_pay.data_ptr(), cap)
    got = d_pay[:plen].cpu().numpy()
    assert plen == want_payload.size and np.array_equal(got, want_payload)
    sizes = ctx.last_chunk_sizes()
    want_f32, want_fm = oracle.decode_f32(want_payload, n)
    assert fm == want_fm
    d_out = torch.full((4 * n + 16,), 0xAB, dtype=torch.uint8, device="cuda")
    fm2, n_chunks = ctx.decode_f32_device(d_pay.data_ptr(), plen, n, sizes, d_out.data_ptr())
    assert fm2 == want_fm and n_chunks == sizes.size
    assert np.array_equal(d_out[:4 * n].cpu().numpy(), want_f32)
    assert bool((d_out[4 * n:] == 0xAB).all())
    return fm


@pytest.mark.parametrize("name", FLT)
def test_f32_device_path_matches_golden(ctx, oracle, name, wave_shape, enc_shape):
    """The float front-end with the samples staying in HBM: detect / demote / split / exponent RLE on the
    device, payload byte-identical to the reference's golden one, and the float32 output of the decode side."""
    g = load_golden(name)
    fm = _f32_device_round_trip(ctx, oracle, np.ascontiguousarray(g["raw"], dtype=np.uint8), g["payload"])
    assert fm == int(g["float_mode"])


def _f32_cases():
    rng = np.random.default_rng(11)
    n = 2 * 150_000                                                       # > 100,000: DetectPseudoFloat looks at every 4th sample
    t = np.arange(n // 2) / 48000.0
    true = (0.3 * np.sin(2 * np.pi * 200 * t) * (1 + 0.5 * np.sin(2 * np.pi * 0.7 * t))).astype(np.float32).repeat(2)
    true[40_000:140_000] = 0.0                                            # exponent runs far beyond 255, cut in many places
    true[140_000:140_600] = np.float32(1e-41)                             # denormals: exponent 0, no hidden bit
    true[7] = np.inf; true[8] = -np.inf; true[9] = np.nan; true[11] = -0.0
    yield "true_long_runs", true
    noisy = rng.standard_normal(n).astype(np.float32) * np.float32(10.0) ** rng.integers(-20, 20, n).astype(np.float32)
    yield "true_exponent_noise", noisy                                    # nearly every exponent differs: ~2 header bytes per sample
    i16 = rng.integers(-32768, 32768, n).astype(np.float32) / np.float32(32768.0)
    yield "pseudo16_random", i16
    i24 = rng.integers(-(1 << 23), 1 << 23, n).astype(np.float32) / np.float32(8388608.0)
    yield "pseudo24_random", i24
    off = i16.copy()                                                      # only the off-stride samples break the fit: still mode 1,
    off[1::4] = np.float32(0.3); off[2::4] = np.float32(1e30); off[3::8] = np.nan; off[7::8] = -np.inf   # demoted as x86 converts them
    yield "pseudo16_off_stride_garbage", off
    big = i24.copy(); big[::4][:100] = np.float32(3e9)                    # a probed value beyond int32: cvttss2si gives INT_MIN, no fit
    yield "probed_value_beyond_int32", big
    small = i16[:100_000].copy(); small[99_999] = np.float32(0.1)         # <= 100,000: every sample is probed
    yield "every_sample_probed", small
    k255 = np.full(2 * 255 * 40, 1.5, dtype=np.float32); k255[255 * 7:] = 3.0     # runs that end exactly on the 255 cap
    yield "runs_of_exactly_255", k255


@pytest.mark.parametrize("case", [c[0] for c in _f32_cases()])
def test_f32_device_path_vs_oracle(ctx, oracle, case):
    f = dict(_f32_cases())[case]
    _f32_device_round_trip(ctx, oracle, f.view(np.uint8))


@pytest.mark.parametrize("name", ["float_true", "float_pseudo24"])
def test_f32_device_decode_of_a_stream_that_ends_early(ctx, oracle, name):
    """The callers decode into zero-initialised vectors and stop at the first zero chunk size (main.cpp:254-261,
    VeloxCore.h:341): the samples the stream does not reach come out as +0.0f — also in true-float mode, where the
    exponent runs of the header cover them (their exponent bytes are never handed out)."""
    import torch
    from velox_amd.codec import walk_chunks
    g = load_golden(name)
    n = int(g["mant"].size)
    pay = np.ascontiguousarray(g["payload"]).copy()
    want_full, fm = oracle.decode_f32(pay, n)
    vals, _, _, _ = oracle.decode(pay, n, want_meta=True)
    assert vals.size == n
    # chunk sizes by walking the prefixes behind the header (its length: 3 bits + 16 per RLE pair + high_res)
    d_p = torch.from_numpy(np.concatenate([pay, np.zeros(32, np.uint8)])).cuda()
    d_raw = torch.from_numpy(np.ascontiguousarray(g["raw"], dtype=np.uint8).copy()).cuda()
    cap = ctx.encode_bound([n]) + 2 * n + 256
    d_enc = torch.zeros(cap + 16, dtype=torch.uint8, device="cuda")
    plen, _ = ctx.encode_f32_device(d_raw.data_ptr(), n, d_enc.data_ptr(), cap)
    assert plen == pay.size
    sizes = ctx.last_chunk_sizes().copy()
    assert sizes.size >= 2
    keep = sizes.size - 1                                                 # the table ends one chunk early
    d_out = torch.full((4 * n + 16,), 0xCD, dtype=torch.uint8, device="cuda")
    fm2, n_chunks = ctx.decode_f32_device(d_p.data_ptr(), pay.size, n, sizes[:keep], d_out.data_ptr())
    assert fm2 == fm and n_chunks == keep
    got = d_out[:4 * n].cpu().numpy()
    reached = 8192 * keep
    assert np.array_equal(got[:4 * reached], want_full[:4 * reached])
    assert not got[4 * reached:].any()


def test_f32_device_argument_checks(ctx):
    import torch
    g = load_golden("s16_stereo_44k")
    d = torch.from_numpy(np.ascontiguousarray(g["payload"])).cuda()
    pad = torch.cat([d, torch.zeros(32, dtype=torch.uint8, device="cuda")])
    out = torch.zeros(4 * g["samples"].size, dtype=torch.uint8, device="cuda")
    with pytest.raises(RuntimeError):                                     # not a float payload
        ctx.decode_f32_device(pad.data_ptr(), d.numel(), g["samples"].size, np.zeros(1, np.uint32), out.data_ptr())


def test_corpus_high_res_scan_walks_each_track(ctx, oracle):
    """A corpus of at least 256 short tracks gets one workgroup per track for the file-global high_res scan
    (k_hires_scan_tracks: gone at a track's first loud chunk, VeloxCore.h:232-236).  Tracks that never get loud, that get
    loud only in their LAST sub-block (one sample), that are loud from the start, ragged tails and an empty track, in one
    batch: every payload equals the oracle's."""
    import torch
    from velox_amd.synth import pack_pcm, synth_pcm
    rng = np.random.default_rng(77)
    base = synth_pcm(24, 2, 48000, 0.6, 30000.0, 200.0, 12)               # |s| stays below 65536
    assert np.abs(base).max() < 65536
    tracks = []
    for i in range(300):
        n = 3 * 8192 + 2 * int(rng.integers(0, 700)) if i % 7 else 2 * 8192
        t = np.roll(base, 97 * i)[:n].copy()
        if i % 3 == 1:
            t[n - 1 - 2 * int(rng.integers(0, 50))] = 70000 + i           # one loud sample near the very end
        elif i % 3 == 2:
            t = t * 40                                                      # loud throughout
        if i == 123:
            t = t[:0]
        tracks.append(t)
    offs, blob, pos = [], [], 0
    for t in tracks:
        offs.append(pos)
        b = pack_pcm(t, 24)
        blob.append(np.concatenate([b, np.zeros((-b.size) % 16 + 16, np.uint8)]))
        pos += blob[-1].size
    pcm = torch.from_numpy(np.concatenate(blob)).cuda()
    ns = [t.size for t in tracks]
    cap = ctx.encode_bound(ns)
    out = torch.zeros(cap + 16, dtype=torch.uint8, device="cuda"); torch.cuda.synchronize()
    poff, plen = ctx.encode_device(pcm.data_ptr(), 24, offs, ns, out.data_ptr(), cap)
    host = out.cpu().numpy()
    kinds = set()
    for k, t in enumerate(tracks):
        want = oracle.encode(t)
        got = host[int(poff[k]): int(poff[k]) + int(plen[k])]
        assert got.size == want.size and np.array_equal(got, want), (k, t.size)
        kinds.add((k % 3, int(want[0] >> 1) & 1))
    assert {(0, 0), (1, 1), (2, 1)} <= kinds                               # quiet, late-loud and loud tracks all occurred


def test_odd_totals_on_the_device_entry_points(ctx, oracle):
    """Odd totals are routed to the odd path (VeloxCore.h:242-253: one stream over the whole input) by the
    device-resident entry points too — vlx_encode_device (alone and mixed with even tracks in one batch) and
    vlx_encode_f32_device (pseudo float and true float) — instead of being refused.  Encode parity only: the
    reference's own decoder cannot read these payloads."""
    import torch
    from velox_amd.synth import pack_pcm, synth_pcm
    tracks, bits = [], 24
    for k, n in enumerate((20001, 6 * 8192, 4099, 2 * 8192 + 10, 1)):
        tracks.append(synth_pcm(bits, 1, 48000, 1.5, 2.0e6 if k != 2 else 900.0, 300.0, 50 + k)[:n])
    W = bits // 8
    offs, blob, pos = [], [], 0
    for t in tracks:
        offs.append(pos)
        b = pack_pcm(t, bits)
        blob.append(np.concatenate([b, np.zeros((-b.size) % 16 + 16, np.uint8)]))
        pos += blob[-1].size
    pcm = torch.from_numpy(np.concatenate(blob)).cuda()
    ns = [t.size for t in tracks]
    cap = ctx.encode_bound(ns)
    out = torch.zeros(cap + 16, dtype=torch.uint8, device="cuda"); torch.cuda.synchronize()
    poff, plen = ctx.encode_device(pcm.data_ptr(), bits, offs, ns, out.data_ptr(), cap)
    host = out.cpu().numpy()
    for k, t in enumerate(tracks):
        want = oracle.encode(t)
        got = host[int(poff[k]): int(poff[k]) + int(plen[k])]
        assert int(poff[k]) % 256 == 0 and got.size == want.size and np.array_equal(got, want), (k, t.size)
    ends = sorted((int(poff[k]), int(poff[k]) + int(plen[k])) for k in range(len(tracks)))
    assert all(a[1] <= b[0] for a, b in zip(ends, ends[1:])), "payloads overlap"
    # the one-lane path is bounded (VLX_ODD_MAX_SAMPLES = 2^26): beyond it a clear refusal, before anything is read or launched
    from velox_amd.codec import VeloxError
    with pytest.raises(VeloxError) as e:
        ctx.encode_device(pcm.data_ptr(), bits, [0], [(1 << 26) + 1], out.data_ptr(), cap)
    assert e.value.code == -5 and "VLX_ODD_MAX_SAMPLES" in str(e.value)
    # float32 tracks with an odd total: pseudo float (demoted integers) and true float (mantissas + exponent runs)
    rng = np.random.default_rng(5)
    pseudo = (rng.integers(-30000, 30000, 9001).astype(np.float32) / np.float32(32768.0)).astype("<f4")
    true_f = (0.3 * np.sin(np.arange(12345) * 0.01) + 1e-3 * rng.standard_normal(12345)).astype("<f4")
    for f in (pseudo, true_f):
        raw = f.view(np.uint8)
        m, e = oracle.split_float32(raw)
        want = oracle.encode(m, True, e, raw)
        d_raw = torch.from_numpy(raw.copy()).cuda()
        capf = ctx.encode_bound([f.size]) + 2 * f.size + 512
        d_pay = torch.zeros(capf + 16, dtype=torch.uint8, device="cuda"); torch.cuda.synchronize()
        ln, fm = ctx.encode_f32_device(d_raw.data_ptr(), f.size, d_pay.data_ptr(), capf)
        assert ln == want.size and np.array_equal(d_pay[:ln].cpu().numpy(), want)
        assert fm == ((int(want[0]) >> 1) & 3)


def test_decode_next_pull_semantics_and_clone(ctx, oracle):
    """DecodeNext one sample at a time; copy-assign mid-stream (VeloxStreamClient.cpp:396)."""
    from velox_amd.codec import StreamingDecoder
    g = load_golden("short_tail_block")
    s = g["samples"].astype(np.int64)
    d = StreamingDecoder(g["payload"], s.size, ctx=ctx)
    got = []
    for _ in range(9000):
        ok, v, e = d.DecodeNext()
        assert ok and e == 0
        got.append(v)
    d2 = d.clone()
    a, _ = d.read(s.size)
    b, _ = d2.read(s.size)
    assert np.array_equal(np.array(got + a.tolist()), s) and np.array_equal(a, b)


def test_odd_total_tail_decode(ctx, oracle, wave_shape):
    """total odd on the DECODE side (the decoder has no odd path; remaining==1 -> frames=1)."""
    from velox_amd.codec import StreamingDecoder
    g = load_golden("s16_mono_even")
    n = g["samples"].size - 1
    want = oracle.decode(g["payload"], n)
    got, _ = StreamingDecoder(g["payload"], n, ctx=ctx, lookahead=2).read(n + 3)
    assert np.array_equal(got, want)


def test_truncated_and_garbage_payloads(ctx, oracle, wave_shape):
    from velox_amd.codec import StreamingDecoder
    g = load_golden("s16_stereo_44k")
    p, n = g["payload"], g["samples"].size
    for cut in (p.size // 2, p.size - 3, 7, 1):
        want = oracle.decode(p[:cut], n)
        got, _ = StreamingDecoder(p[:cut].copy(), n, ctx=ctx, lookahead=3).read(n)
        assert np.array_equal(got, want), cut
    rng = np.random.default_rng(0)
    junk = p.copy()
    junk[200:260] = rng.integers(0, 256, 60)
    want = oracle.decode(junk, n)
    got, _ = StreamingDecoder(junk, n, ctx=ctx, lookahead=2).read(n)
    assert np.array_equal(got, want)


@pytest.mark.parametrize("seed", range(6))
def test_fuzz_against_oracle(ctx, oracle, seed, wave_shape, enc_shape):
    from velox_amd.codec import StreamingDecoder
    rng = np.random.default_rng(4000 + seed)
    n = 2 * int(rng.integers(1, 30000))
    bits = [16, 24, 32][seed % 3]
    lim = 2 ** (bits - 1)
    t = np.arange(n // 2)
    l = lim * rng.uniform(0.01, 0.9) * np.sin(t * rng.uniform(0.001, 0.5)) + rng.normal(0, lim * 10 ** rng.uniform(-4, -1), n // 2)
    r = 0.7 * l + rng.normal(0, lim * 10 ** rng.uniform(-4, -1), n // 2)
    s = np.clip(np.rint(np.stack([l, r], 1).reshape(-1)), -lim, lim - 1).astype(np.int64)
    want = oracle.encode(s)
    assert np.array_equal(ctx.encode_block(s), want)
    got, _ = StreamingDecoder(want, n, ctx=ctx, lookahead=4).read(n)
    assert np.array_equal(got, oracle.decode(want, n))


def _dev(t):
    return t.data_ptr()


@pytest.mark.parametrize("bits", [16, 24, 32])
def test_device_batch_multi_track(ctx, oracle, bits, wave_shape, enc_shape):
    """vlx_encode_device / vlx_decode_device over several tracks in one HBM buffer, incl. an empty
    track, a tail block and all three packed output formats."""
    import torch
    from velox_amd.codec import walk_chunks
    from velox_amd.synth import pack_pcm, synth_pcm
    amp = {16: 9000, 24: 2.5e6, 32: 4e8}[bits]
    tracks = [synth_pcm(bits, 2, 48000, 0.21, amp, amp / 500, 50 + i)[: ln] for i, ln in enumerate([16384, 0, 20000, 8192 + 2, 2])]
    offs, blobs, pos = [], [], 0
    for s in tracks:
        offs.append(pos)
        b = pack_pcm(s, bits)
        pad = (-b.size) % 16
        blobs.append(np.concatenate([b, np.zeros(pad, np.uint8)]))
        pos += b.size + pad
    pcm = torch.from_numpy(np.concatenate(blobs + [np.zeros(16, np.uint8)])).cuda()
    ns = [s.size for s in tracks]
    cap = ctx.encode_bound(ns)
    out = torch.zeros(cap, dtype=torch.uint8, device="cuda"); torch.cuda.synchronize()
    poff, plen = ctx.encode_device(_dev(pcm), bits, offs, ns, _dev(out), cap)
    host = out.cpu().numpy()
    chunks, total_out = [], 0
    for i, s in enumerate(tracks):
        got = host[int(poff[i]): int(poff[i]) + int(plen[i])]
        want = oracle.encode(s)
        assert np.array_equal(got, want), f"track {i}"
        ch = walk_chunks(got, s.size, base_bit_off=8 * int(poff[i]), out_base=total_out)
        chunks.append(ch)
        total_out += s.size
    ch = np.concatenate(chunks)
    allpcm = np.concatenate([pack_pcm(s, bits) for s in tracks])
    for fmt in (bits, 64):
        w = 8 if fmt == 64 else fmt // 8
        dst = torch.zeros(total_out * w + 16, dtype=torch.uint8, device="cuda"); torch.cuda.synchronize()
        ctx.decode_device(_dev(out), cap, ch, _dev(dst), fmt)
        res = dst.cpu().numpy()[: total_out * w]
        if fmt == 64:
            assert np.array_equal(res.view("<i8"), np.concatenate(tracks))
        else:
            assert np.array_equal(res, allpcm)


def test_one_wave_mixes_streams_that_shift_and_streams_that_stay(ctx, oracle, wave_shape, enc_shape):
    """The predictor's history is a ring of registers that every lane of a wave turns in lock-step; a stream whose error is
    zero must not shift (VeloxCore.h:30) and repairs its ring instead (vlx_device.h:VlxLmsT).  One batch of one-chunk tracks
    whose noise floor runs from 0.2 LSB (zero errors at most samples) over 1-3 LSB (zero and non-zero errors alternate at
    every ring phase) to thousands of LSB (never zero), so that the 64 lanes of a wave hold all of them side by side; plus
    constants, ramps and a track that falls silent half-way.  Encoder bytes per track against the oracle, decode of the batch
    back to the samples, in every encode and decode shape."""
    import torch
    from velox_amd.codec import walk_chunks
    from velox_amd.synth import pack_pcm
    rng = np.random.default_rng(77)
    tracks = []
    for i in range(80):
        sigma = [0.2, 0.5, 1.0, 2.0, 3.0, 10.0, 100.0, 3000.0][i % 8]
        n = 8192 if i % 5 else 8192 - 2 * int(rng.integers(1, 300))
        l = np.rint(rng.normal(0, sigma, n // 2) + (200.0 * np.sin(np.arange(n // 2) * 0.01) if i % 3 == 0 else 0.0))
        r = np.rint(rng.normal(0, sigma, n // 2)) if i % 4 else l.copy()
        if i % 16 == 7: l[:] = 5; r = np.rint(np.linspace(-40, 40, n // 2))
        if i % 16 == 9: l[n // 4:] = 0; r[n // 4:] = 0
        tracks.append(np.clip(np.stack([l, r], 1).reshape(-1), -32768, 32767).astype(np.int64))
    bits = 16
    offs, blobs, pos = [], [], 0
    for t in tracks:
        offs.append(pos)
        b = pack_pcm(t, bits)
        pad = (-b.size) % 16
        blobs.append(np.concatenate([b, np.zeros(pad, np.uint8)]))
        pos += b.size + pad
    pcm = torch.from_numpy(np.concatenate(blobs + [np.zeros(16, np.uint8)])).cuda()
    ns = [t.size for t in tracks]
    cap = ctx.encode_bound(ns)
    out = torch.zeros(cap, dtype=torch.uint8, device="cuda"); torch.cuda.synchronize()
    poff, plen = ctx.encode_device(_dev(pcm), bits, offs, ns, _dev(out), cap)
    host = out.cpu().numpy()
    chunks, total_out = [], 0
    for i, t in enumerate(tracks):
        got = host[int(poff[i]): int(poff[i]) + int(plen[i])]
        assert np.array_equal(got, oracle.encode(t)), f"track {i}"
        chunks.append(walk_chunks(got, t.size, base_bit_off=8 * int(poff[i]), out_base=total_out))
        total_out += t.size
    dst = torch.zeros(total_out * 8 + 16, dtype=torch.uint8, device="cuda"); torch.cuda.synchronize()
    ctx.decode_device(_dev(out), cap, np.concatenate(chunks), _dev(dst), 64)
    assert np.array_equal(dst.cpu().numpy()[: total_out * 8].view("<i8"), np.concatenate(tracks))


@pytest.mark.parametrize("bits", [16, 24])
@pytest.mark.parametrize("shape", ["stream", "auto"])
@pytest.mark.parametrize("misalign", [0, 1])
def test_stream_waves_finish_lowres_chunks_themselves(ctx, oracle, bits, shape, misalign):
    """Chunks without high-res low bytes are finished by the stream waves of k_decode_streams (neighbour-lane swap,
    LSB restore, inverse M/S, packed PCM) and skipped by k_decode_finalize; everything else in the same batch — a
    ragged tail chunk, two silent streams, raw chunks, high-res tracks, an output that is not 4-byte aligned — still
    goes the long way.  The bytes must not depend on which way a chunk took."""
    import torch
    from velox_amd.codec import set_decode_shape, walk_chunks
    from velox_amd.synth import pack_pcm, synth_pcm
    rng = np.random.default_rng(17 + bits)
    lim = 30000 if bits == 16 else 60000
    corr = synth_pcm(16, 2, 44100, 0.6, 8000.0, 150.0, 5)[: 3 * 8192 + 20]                  # M/S material, ragged tail (10 frames)
    indep = rng.integers(-lim, lim, 2 * 8192).astype(np.int64)                             # L/R material
    one_silent = indep.copy(); one_silent[1::2] = 0                                       # stream R silent
    all_silent = np.zeros(8192, dtype=np.int64)                                           # both silent: finalize writes the zeros
    shifted = (rng.integers(-2000, 2000, 8192) * 16).astype(np.int64)                     # LSB shift 4 on both streams
    shifted[1::2] = (shifted[1::2] // 32) * 32                                            # ... 5 on stream R
    fullscale = rng.integers(-32768, 32768, 8192).astype(np.int64) if bits == 16 else rng.integers(-(1 << 23), 1 << 23, 8192).astype(np.int64)
    tracks = [corr, indep, one_silent, all_silent, shifted, fullscale]
    if bits == 24:
        tracks.append(synth_pcm(24, 2, 96000, 0.1, 2.0e6, 3000.0, 9)[: 2 * 8192])          # a high-res track between the others
        tracks.append(indep[::-1].copy())
    payloads = [oracle.encode(t) for t in tracks]
    offs, pos = [], 0
    for pl in payloads:
        offs.append(pos)
        pos += (pl.size + 15) // 16 * 16
    buf = np.zeros(pos + 32, dtype=np.uint8)
    chunks, base = [], 0
    for t, pl, o in zip(tracks, payloads, offs):
        buf[o: o + pl.size] = pl
        chunks.append(walk_chunks(pl, t.size, base_bit_off=8 * o, out_base=base))
        base += t.size
    ch = np.concatenate(chunks)
    d_p = torch.from_numpy(buf).cuda()
    w = bits // 8
    want = np.concatenate([pack_pcm(t, bits) for t in tracks])
    dst = torch.full((base * w + 32,), 0x5A, dtype=torch.uint8, device="cuda"); torch.cuda.synchronize()
    set_decode_shape(shape)
    try:
        ctx.decode_device(d_p.data_ptr(), buf.size, ch, dst.data_ptr() + misalign, bits)
    finally:
        set_decode_shape("auto")
    res = dst.cpu().numpy()
    assert np.array_equal(res[misalign: misalign + base * w], want)
    assert (res[:misalign] == 0x5A).all() and (res[misalign + base * w:] == 0x5A).all()


def test_decode_tracks_device_matches_the_two_call_form(ctx, oracle):
    """vlx_decode_tracks_device (chunk table built in HBM from the byte sizes) against
    vlx_chunks_from_sizes + vlx_decode_device on the same batch: several tracks, an empty one, tails,
    totals smaller than the sizes cover (even and odd: the one-sample last chunk), a zero size that ends
    a track early, and a size table longer than the track."""
    import torch
    from velox_amd.codec import TRACK_SIZES_DTYPE, chunks_from_sizes
    from velox_amd.synth import pack_pcm, synth_pcm
    lens = [16384, 0, 20000, 8192 + 2, 2, 3 * 8192, 4 * 8192]
    tracks = [synth_pcm(24, 2, 48000, 0.4, 2.5e6, 4000, 70 + i)[: ln] for i, ln in enumerate(lens)]
    offs, blobs, pos = [], [], 0
    for s in tracks:
        offs.append(pos)
        b = pack_pcm(s, 24)
        pad = (-b.size) % 16
        blobs.append(np.concatenate([b, np.zeros(pad, np.uint8)]))
        pos += b.size + pad
    pcm = torch.from_numpy(np.concatenate(blobs + [np.zeros(16, np.uint8)])).cuda()
    cap = ctx.encode_bound(lens)
    out = torch.zeros(cap, dtype=torch.uint8, device="cuda"); torch.cuda.synchronize()
    poff, plen = ctx.encode_device(_dev(pcm), 24, offs, lens, _dev(out), cap)
    sizes = ctx.last_chunk_sizes().copy()
    host = out.cpu().numpy()
    per_track = [(ln + 8191) // 8192 for ln in lens]
    first = np.concatenate([[0], np.cumsum(per_track)]).astype(int)
    totals = list(lens)
    totals[5] = 2 * 8192 + 4097          # odd total inside the third chunk: 4096 samples of it, then a chunk that is not there
    totals[6] = 8192 + 100               # even total: the last two sizes are never walked
    sizes[first[2] + 1] = 0              # the walk of track 2 ends at its second chunk (VeloxCore.h:341)
    tr = np.zeros(len(lens), dtype=TRACK_SIZES_DTYPE)
    table, chunks, out_base = [], [], 0
    for i, ln in enumerate(lens):
        sz = sizes[first[i]: first[i + 1]]
        if i == 4:
            sz = np.concatenate([sz, [77, 88]]).astype(np.uint32)     # more sizes than the total needs
        hires = int((host[int(poff[i])] >> 1) & 1)
        tr[i] = (8 * int(poff[i]), out_base, totals[i], sz.size, hires, 2, 0)
        table.append(sz)
        chunks.append(chunks_from_sizes(sz, totals[i], 8 * int(poff[i]), out_base, hires))
        out_base += ln
    ch = np.concatenate(chunks)
    for fmt in (24, 64):
        w = 8 if fmt == 64 else 3
        a = torch.zeros(out_base * w + 16, dtype=torch.uint8, device="cuda")
        b = torch.zeros(out_base * w + 16, dtype=torch.uint8, device="cuda"); torch.cuda.synchronize()
        ctx.decode_device(_dev(out), cap, ch, _dev(a), fmt)
        n = ctx.decode_tracks_device(_dev(out), cap, tr, np.concatenate(table), _dev(b), fmt)
        assert n == ch.size
        assert torch.equal(a, b)
        tr2 = tr.copy(); tr2["high_res"] = 0xFFFFFFFF                  # VLX_HIGH_RES_FROM_PAYLOAD: the device reads the header bit
        b.zero_(); torch.cuda.synchronize()
        assert ctx.decode_tracks_device(_dev(out), cap, tr2, np.concatenate(table), _dev(b), fmt) == ch.size
        assert torch.equal(a, b)
    # the untouched tracks decode to their input
    res = b.cpu().numpy()[: out_base * 8].view("<i8")
    assert np.array_equal(res[: lens[0]], tracks[0])
    o3 = sum(lens[:3])
    assert np.array_equal(res[o3: o3 + lens[3]], tracks[3])
    # nothing to decode at all
    assert ctx.decode_tracks_device(_dev(out), cap, tr[1:2], np.zeros(0, np.uint32), _dev(b), 24) == 0
    # a size table that runs past the payload is refused before anything is launched
    bad = tr[:1].copy(); big = np.full(2, 2 ** 31, dtype=np.uint32)
    with pytest.raises(Exception):
        ctx.decode_tracks_device(_dev(out), cap, bad, big, _dev(b), 24)


@pytest.mark.parametrize("name", [n for n in PCM if not n.startswith("wide_")])
def test_decode_pcm_matches_golden(ctx, oracle, name):
    """vlx_decode_pcm (host payload -> packed host PCM, the mirror of vlx_encode_pcm) on every integer fixture,
    in the fixture's own width and truncated to 16 bits (SamplesToBytes, VeloxFormat.h:160-189)."""
    g = load_golden(name)
    s = g["samples"].astype(np.int64)
    for bits in sorted({int(g["bits"]), 16}):
        got, n = ctx.decode_pcm(g["payload"], s.size, bits)
        assert n == s.size
        assert np.array_equal(got, oracle.samples_to_bytes(s, bits)), bits


def test_decode_pcm_short_streams_and_float(ctx, oracle):
    """A truncated payload decodes like the reference's loop over a zero-initialised vector (main.cpp:254-261):
    what the stream reaches, then zeros; a float payload is refused."""
    g = load_golden("s16_stereo_44k")
    s = g["samples"].astype(np.int64)
    p = g["payload"]
    cut = p[: p.size // 2]
    want = oracle.decode(cut, s.size)                                  # int64 samples the reference decoder yields
    got, n = ctx.decode_pcm(cut, s.size, 16)
    full = np.zeros(s.size, np.int64); full[: want.size] = want
    assert n == want.size and n < s.size
    assert np.array_equal(got, oracle.samples_to_bytes(full, 16))
    for cut_at in (1, 3, 5, 6, 7, 9, p.size // 3 + 1):                     # incl. cuts inside a u32 size prefix
        cut = p[:cut_at]
        want = oracle.decode(cut, s.size)
        full = np.zeros(s.size, np.int64); full[: want.size] = want
        got, n = ctx.decode_pcm(cut, s.size, 16)
        assert n == want.size, cut_at
        assert np.array_equal(got, oracle.samples_to_bytes(full, 16)), cut_at
    # the same cut through vlx_walk_chunks + vlx_decode_device: the chunk named by the cut prefix is accepted
    import torch
    from velox_amd.codec import walk_chunks
    cut = p[:6]
    ch = walk_chunks(cut, s.size)
    want = oracle.decode(cut, s.size)
    assert int(ch["n_out"].sum()) == want.size and int(ch["bit_off"].max()) <= 8 * cut.size
    d_p = torch.from_numpy(np.concatenate([cut, np.zeros(10, np.uint8)])).cuda()
    dst = torch.zeros(s.size * 8 + 16, dtype=torch.uint8, device="cuda"); torch.cuda.synchronize()
    ctx.decode_device(d_p.data_ptr(), cut.size, ch, dst.data_ptr(), 64)
    assert np.array_equal(dst.cpu().numpy()[: want.size * 8].view("<i8"), want)
    got, n = ctx.decode_pcm(p, 0, 16)
    assert got.size == 0 and n == 0
    f = load_golden(FLT[0])
    with pytest.raises(Exception):
        ctx.decode_pcm(f["payload"], int(f["dec_vals"].size), 16)


def test_decode_truncating_pack(ctx, oracle, wave_shape):
    """SamplesToBytes truncates (VeloxFormat.h:160-189): decode a 24-bit track into 16-bit output."""
    import torch
    from velox_amd.codec import walk_chunks
    g = load_golden("s24_stereo_96k_hires")
    s = g["samples"].astype(np.int64)
    p = np.concatenate([g["payload"], np.zeros(8, np.uint8)])
    d_p = torch.from_numpy(p).cuda()
    ch = walk_chunks(g["payload"], s.size)
    dst = torch.zeros(s.size * 2 + 16, dtype=torch.uint8, device="cuda"); torch.cuda.synchronize()
    ctx.decode_device(_dev(d_p), p.size, ch, _dev(dst), 16)
    assert np.array_equal(dst.cpu().numpy()[: s.size * 2], oracle.samples_to_bytes(s, 16))


def test_full_size_properties(ctx, oracle):
    """BASELINE config-2 shape at reduced duration (60 s of 24-bit/96 kHz stereo = 1407 chunks):
    size-independent properties — encode -> decode round-trips to the original PCM bytes, chunk sizes
    chain to the payload length, and the leading chunks match the oracle byte for byte."""
    import torch
    from velox_amd.codec import chunks_from_sizes
    from velox_amd.synth import synth_pcm_device, unpack_pcm
    frames = 96000 * 60
    pcm = synth_pcm_device(24, 2, 96000, frames, 2.0e6, 3000.0, 7, "cuda")
    n = frames * 2
    cap = ctx.encode_bound([n])
    out = torch.zeros(cap, dtype=torch.uint8, device="cuda"); torch.cuda.synchronize()
    poff, plen = ctx.encode_device(_dev(pcm), 24, [0], [n], _dev(out), cap)
    sizes = ctx.last_chunk_sizes()
    assert sizes.size == (n + 8191) // 8192
    assert int(plen[0]) == int(sizes.astype(np.int64).sum()) + 4 * sizes.size + 1
    head = out[: int(plen[0])].cpu().numpy()
    hires = (head[0] >> 1) & 1
    assert hires == 1
    ch = chunks_from_sizes(sizes, n, 8 * int(poff[0]), 0, hires)
    dst = torch.zeros(n * 3 + 16, dtype=torch.uint8, device="cuda"); torch.cuda.synchronize()
    ctx.decode_device(_dev(out), cap, ch, _dev(dst), 24)
    assert torch.equal(dst[: n * 3], pcm[: n * 3])
    # oracle spot check: the first 3 chunks as their own (equally loud) track give the same leading bytes
    host = pcm[: 3 * 8192 * 3].cpu().numpy()
    want = oracle.encode(unpack_pcm(host, 24))
    assert np.array_equal(head[: want.size - 1], want[:-1])    # last byte also carries the next chunk's bits


def _host_threads():
    import os
    try:
        return max(1, min(16, len(os.sched_getaffinity(0))))
    except AttributeError:
        return max(1, min(16, os.cpu_count() or 1))


def test_full_hour_every_chunk_matches_the_oracle(ctx, oracle):
    """BASELINE configs[1] at its real size: 24-bit stereo 96 kHz, one hour = 2.07 GB of PCM, 84,375 chunks,
    encoded by vlx_encode_device in one batch.  EVERY chunk of the device payload is compared with the oracle's
    encoding of the same 8192 samples (vlxo_verify_payload: the hour re-encoded on the host cores, about half a
    minute on 16 threads) — an encoder deviation that still decodes (another M/S choice, LPC coefficient or
    raw-fallback decision) cannot hide.  Before that, 1,024 sampled chunk positions go through the Python-side
    sampler bench.py and tools/bench_configs.py gate their timed payloads with, so that the sampler itself is pinned
    (a flipped bit in a sampled chunk is seen).  Size-independent properties: the sizes chain to the payload
    length; decode returns the input PCM bytes."""
    import torch
    from _sampling import check_chunks, pick
    from velox_amd.codec import HIGH_RES_FROM_PAYLOAD, TRACK_SIZES_DTYPE
    from velox_amd.synth import synth_pcm_device, unpack_pcm
    frames = 96000 * 3600
    n, W = frames * 2, 3
    pcm = synth_pcm_device(24, 2, 96000, frames, 2.0e6, 3000.0, 2, "cuda")
    cap = ctx.encode_bound([n])
    out = torch.empty(cap, dtype=torch.uint8, device="cuda"); torch.cuda.synchronize()
    poff, plen = ctx.encode_device(_dev(pcm), 24, [0], [n], _dev(out), cap)
    sizes = ctx.last_chunk_sizes().copy()
    po, pl = int(poff[0]), int(plen[0])
    assert sizes.size == 84375
    assert pl == int(sizes.astype(np.int64).sum()) + 4 * sizes.size + 1
    assert (int(out[po].item()) & 3) == 2                                 # is_float = 0, high_res = 1
    picks = pick(sizes.size, 1024, seed=11)
    assert 0 in picks and sizes.size - 1 in picks and len(picks) >= 1000
    bad = check_chunks(oracle, lambda a, b: pcm[a * W: b * W].cpu().numpy(), lambda a, b: out[po + a: po + b].cpu().numpy(),
                       sizes, n, 24, 1, picks, unpack_pcm)
    assert bad == []
    n_cmp, n_bad, first_bad = oracle.verify_payload(pcm[: n * W].cpu().numpy(), 24, n, out[po: po + pl].cpu().numpy(), _host_threads())
    assert (n_cmp, n_bad, first_bad) == (84375, 0, -1), "the device payload of the hour differs from the oracle's"
    tr = np.zeros(1, dtype=TRACK_SIZES_DTYPE)
    tr[0] = (8 * po, 0, n, sizes.size, HIGH_RES_FROM_PAYLOAD, 2, 0)
    dst = torch.empty(n * W + 16, dtype=torch.uint8, device="cuda"); torch.cuda.synchronize()
    assert ctx.decode_tracks_device(_dev(out), cap, tr, sizes, _dev(dst), 24) == sizes.size
    assert torch.equal(dst[: n * W], pcm[: n * W])
    # a flipped payload bit is seen by the sampling when it falls into a sampled chunk (the check can fail)
    c = picks[len(picks) // 2]
    at = po + int(np.concatenate(([0], np.cumsum(sizes.astype(np.int64) + 4)))[c]) + 4 + int(sizes[c]) // 2
    out[at] ^= 0x10; torch.cuda.synchronize()
    assert check_chunks(oracle, lambda a, b: pcm[a * W: b * W].cpu().numpy(), lambda a, b: out[po + a: po + b].cpu().numpy(),
                        sizes, n, 24, 1, picks, unpack_pcm) == [c]
    del pcm, out, dst
    torch.cuda.empty_cache()
    ctx.trim()


def test_cfg4_full_size_every_chunk_matches_the_oracle(ctx, oracle):
    """BASELINE configs[3] at its real size: 32-bit, 8 channels, 192 kHz, 60 s = 368 MB of PCM, 11,250 chunks whose
    streams all carry an energy >= 2^53 (the sequential fp64 autocorrelation) — the batch size at which both serial
    kernels run in their pipelined shapes.  64 sampled chunks against the oracle, round trip of the whole track."""
    import torch
    from _sampling import check_chunks, pick
    from velox_amd.codec import HIGH_RES_FROM_PAYLOAD, TRACK_SIZES_DTYPE
    from velox_amd.synth import synth_pcm_device, unpack_pcm
    frames, chn, W = 192000 * 60, 8, 4
    n = frames * chn
    pcm = synth_pcm_device(32, chn, 192000, frames, 5.0e8, 2.0e5, 4, "cuda")
    cap = ctx.encode_bound([n])
    out = torch.empty(cap, dtype=torch.uint8, device="cuda"); torch.cuda.synchronize()
    poff, plen = ctx.encode_device(_dev(pcm), 32, [0], [n], _dev(out), cap)
    sizes = ctx.last_chunk_sizes().copy()
    po, pl = int(poff[0]), int(plen[0])
    assert sizes.size == 11250 and pl == int(sizes.astype(np.int64).sum()) + 4 * sizes.size + 1
    hires = (int(out[po].item()) >> 1) & 1
    assert hires == 1
    picks = pick(sizes.size, 64, seed=12)
    bad = check_chunks(oracle, lambda a, b: pcm[a * W: b * W].cpu().numpy(), lambda a, b: out[po + a: po + b].cpu().numpy(),
                       sizes, n, 32, hires, picks, unpack_pcm)
    assert bad == []
    # ... and every one of the 11,250 chunks (the whole track re-encoded on the host cores)
    assert oracle.verify_payload(pcm[: n * W].cpu().numpy(), 32, n, out[po: po + pl].cpu().numpy(), _host_threads()) == (11250, 0, -1)
    tr = np.zeros(1, dtype=TRACK_SIZES_DTYPE)
    tr[0] = (8 * po, 0, n, sizes.size, HIGH_RES_FROM_PAYLOAD, 2, 0)
    dst = torch.empty(n * W + 16, dtype=torch.uint8, device="cuda"); torch.cuda.synchronize()
    assert ctx.decode_tracks_device(_dev(out), cap, tr, sizes, _dev(dst), 32) == sizes.size
    assert torch.equal(dst[: n * W], pcm[: n * W])
    assert ctx.last_redo_chunks() == 0            # none of it is beyond the fast kernels (see the test below)
    del pcm, out, dst


@pytest.mark.gpu
@pytest.mark.parametrize("shape", ["stream", "pipe16", "pipe32", "pipe64", "auto"])
@pytest.mark.parametrize("bits,chn,amp,sigma", [(32, 8, 5.0e8, 2.0e5), (32, 2, 5.0e8, 2.0e5), (24, 2, 2.0e6, 3000.0), (16, 6, 8000.0, 150.0)])
def test_ordinary_material_never_needs_the_careful_kernel(ctx, bits, chn, amp, sigma, shape):
    """A fast decode kernel that flags a chunk it could have decoded costs nothing in parity — the careful kernel decodes
    it again, bit-exactly — and everything in time (cfg4 went from 1.5 to 11 ms when the pipelined parser's bit ring
    started too thin for a stream's first, longest codes).  So: ordinary material, every shape, zero chunks redone."""
    import torch
    from velox_amd.codec import chunks_from_sizes, set_decode_shape
    from velox_amd.synth import synth_pcm_device
    W = bits // 8
    n = 600 * 8192 + 2 * chn * 37                                       # a ragged tail as well
    pcm = synth_pcm_device(bits, chn, 96000, n // chn, amp, sigma, 9, "cuda")
    n = pcm.numel() // W
    cap = ctx.encode_bound([n])
    payload = torch.empty(cap, dtype=torch.uint8, device="cuda"); torch.cuda.synchronize()
    poff, plen = ctx.encode_device(_dev(pcm), bits, [0], [n], _dev(payload), cap)
    hires = (int(payload[int(poff[0])].item()) >> 1) & 1
    chunks = chunks_from_sizes(ctx.last_chunk_sizes(), n, 8 * int(poff[0]), 0, hires)
    out = torch.empty(n * W + 16, dtype=torch.uint8, device="cuda"); torch.cuda.synchronize()
    set_decode_shape(shape)
    try:
        ctx.decode_device(_dev(payload), cap, chunks, _dev(out), bits)
    finally:
        set_decode_shape("auto")
    assert torch.equal(out[: n * W], pcm[: n * W])
    assert ctx.last_redo_chunks() == 0
    torch.cuda.empty_cache()


def test_cfg3_corpus_share_every_chunk_matches_the_oracle(ctx, oracle):
    """BASELINE configs[2] in the shape one GPU of eight gets: 512 tracks of 24-bit / 48 kHz stereo, 10 s each
    (seed = track index), one vlx_encode_device batch of 60,416 chunks.  Three whole tracks (first, last, one in
    between) byte for byte against the oracle, 48 sampled chunks of other tracks, round trip of the whole batch."""
    import torch
    from _sampling import check_chunks, pick
    from velox_amd.codec import HIGH_RES_FROM_PAYLOAD, TRACK_SIZES_DTYPE
    from velox_amd.synth import synth_corpus_device, unpack_pcm
    T, frames, W = 512, 480000, 3
    n = frames * 2
    stride = (n * W + 15) // 16 * 16
    pcm = synth_corpus_device(24, 2, 48000, frames, T, 2.0e6, 3000.0, "cuda", stride=stride)
    offs, ns = [t * stride for t in range(T)], [n] * T
    cap = ctx.encode_bound(ns)
    out = torch.empty(cap, dtype=torch.uint8, device="cuda"); torch.cuda.synchronize()
    poff, plen = ctx.encode_device(_dev(pcm), 24, offs, ns, _dev(out), cap)
    sizes = ctx.last_chunk_sizes().copy()
    cpt = (n + 8191) // 8192
    assert sizes.size == T * cpt
    for t in (0, 257, T - 1):                                             # whole tracks
        s = unpack_pcm(pcm[t * stride: t * stride + n * W].cpu().numpy(), 24)
        got = out[int(poff[t]): int(poff[t]) + int(plen[t])].cpu().numpy()
        assert np.array_equal(got, oracle.encode(s)), t
    rng = np.random.default_rng(5)
    for t in rng.choice(T, 12, replace=False):                            # sampled chunks of other tracks
        t = int(t)
        sz = sizes[t * cpt: (t + 1) * cpt]
        po = int(poff[t])
        hires = (int(out[po].item()) >> 1) & 1
        bad = check_chunks(oracle, lambda a, b: pcm[t * stride + a * W: t * stride + b * W].cpu().numpy(),
                           lambda a, b: out[po + a: po + b].cpu().numpy(), sz, n, 24, hires, pick(cpt, 4, seed=t), unpack_pcm)
        assert bad == [], t
    # ... and every chunk of every track of the batch (60,416 chunks re-encoded on the host cores)
    host_pcm, host_out, n_cmp = pcm.cpu().numpy(), out.cpu().numpy(), 0
    for t in range(T):
        got = oracle.verify_payload(host_pcm[t * stride: t * stride + n * W], 24, n, host_out[int(poff[t]): int(poff[t]) + int(plen[t])], _host_threads())
        assert got == (cpt, 0, -1), (t, got)
        n_cmp += got[0]
    assert n_cmp == 60416
    del host_pcm, host_out
    tr = np.zeros(T, dtype=TRACK_SIZES_DTYPE)
    tr["base_bit_off"] = 8 * poff.astype(np.uint64); tr["out_base"] = np.arange(T, dtype=np.uint64) * n
    tr["total_samples"] = n; tr["n_sizes"] = cpt; tr["high_res"] = HIGH_RES_FROM_PAYLOAD; tr["hdr_bits"] = 2
    dst = torch.empty(T * n * W + 16, dtype=torch.uint8, device="cuda"); torch.cuda.synchronize()
    assert ctx.decode_tracks_device(_dev(out), cap, tr, sizes, _dev(dst), 24) == sizes.size
    for t in range(T):
        assert torch.equal(dst[t * n * W: (t + 1) * n * W], pcm[t * stride: t * stride + n * W]), t
    del pcm, out, dst
    torch.cuda.empty_cache()
    ctx.trim()


def test_kernel_time_slots_name_the_kernel_that_ran(ctx):
    """vlx_last_kernel_times: the slot of the sample-serial stage carries the name of the kernel the batch size (or a
    forced shape) selected, so that a roofline line never quotes k_encode_serial / k_decode_streams for a batch the
    pipelined shapes took."""
    import torch
    from velox_amd.codec import chunks_from_sizes, set_decode_shape, set_encode_shape
    from velox_amd.synth import synth_pcm_device
    frames = 4096 * 40
    n = 2 * frames
    pcm = synth_pcm_device(16, 2, 44100, frames, 8000.0, 150.0, 9, "cuda")
    cap = ctx.encode_bound([n])
    out = torch.empty(cap, dtype=torch.uint8, device="cuda")
    dst = torch.empty(n * 2 + 16, dtype=torch.uint8, device="cuda"); torch.cuda.synchronize()
    ctx.set_profiling(True)
    try:
        for shape, enc_name, dec_name in (("auto", "k_encode_pipe", "k_decode_pipe"), ("stream", "k_encode_serial", "k_decode_streams"),
                                          ("pipe32", "k_encode_pipe", "k_decode_pipe")):
            set_encode_shape(shape); set_decode_shape(shape)
            poff, plen = ctx.encode_device(_dev(pcm), 16, [0], [n], _dev(out), cap)
            names = [k for k, _ in ctx.last_kernel_times()]
            assert enc_name in names and not ({"k_encode_pipe", "k_encode_serial"} - {enc_name}) & set(names), (shape, names)
            sizes = ctx.last_chunk_sizes()
            chunks = chunks_from_sizes(sizes, n, 8 * int(poff[0]), 0, (int(out[int(poff[0])].item()) >> 1) & 1)
            ctx.decode_device(_dev(out), cap, chunks, _dev(dst), 16)
            names = [k for k, _ in ctx.last_kernel_times()]
            assert dec_name in names and not ({"k_decode_pipe", "k_decode_streams"} - {dec_name}) & set(names), (shape, names)
            assert torch.equal(dst[: n * 2], pcm[: n * 2])
    finally:
        ctx.set_profiling(False)
        set_encode_shape("auto"); set_decode_shape("auto")


def test_high_res_scan_walks_long_tracks(ctx, oracle):
    """k_hires_scan on a batch of long tracks (more than 512 chunks each, so the "already found" poll and the jump to
    the next track's chunks are in use) next to short ones: loud only in the last chunk, never loud, loud from the first
    chunk, loud only in the middle.  The high-res bit of every track's header against |x| > 65536 over the track, two
    tracks whole against the oracle, round trip of the batch."""
    import torch
    from velox_amd.codec import HIGH_RES_FROM_PAYLOAD, TRACK_SIZES_DTYPE
    from velox_amd.synth import synth_corpus_device, unpack_pcm
    W = 3
    chunks = [700, 40, 1500, 513, 900, 3, 520]
    loud_at = [699, None, 0, None, 450, 2, None]                          # chunk that holds the one loud frame
    ns = [8192 * k - (0 if i % 2 else 6) for i, k in enumerate(chunks)]  # some tracks end in a short chunk
    T = len(chunks)
    stride = (max(ns) * W + 15) // 16 * 16
    pcm = synth_corpus_device(24, 2, 48000, max(ns) // 2, T, 2.0e4, 300.0, "cuda", stride=stride)   # |x| stays far below 65536
    for t in range(T):
        b = pcm[t * stride: t * stride + ns[t] * W].view(-1, 3).to(torch.int32)
        v = b[:, 0] | (b[:, 1] << 8) | (b[:, 2] << 16)
        v = torch.where(v >= (1 << 23), v - (1 << 24), v)
        assert int(v.abs().max().item()) <= 65536, t
        del b, v
        if loud_at[t] is not None:
            at = (loud_at[t] * 8192 + 4000) * W
            pcm[t * stride + at: t * stride + at + 3] = torch.tensor([0x10, 0x00, 0x02], dtype=torch.uint8, device="cuda")   # 0x020010 = 131088
    torch.cuda.synchronize()
    offs = [t * stride for t in range(T)]
    cap = ctx.encode_bound(ns)
    out = torch.empty(cap, dtype=torch.uint8, device="cuda"); torch.cuda.synchronize()
    poff, plen = ctx.encode_device(_dev(pcm), 24, offs, ns, _dev(out), cap)
    sizes = ctx.last_chunk_sizes().copy()
    assert sizes.size == sum(chunks)
    for t in range(T):
        assert (int(out[int(poff[t])].item()) >> 1) & 1 == (1 if loud_at[t] is not None else 0), t
    for t in (1, 5, 3):                                                   # whole tracks: never loud (short), loud in its last chunk, never loud (long)
        s = unpack_pcm(pcm[t * stride: t * stride + ns[t] * W].cpu().numpy(), 24)
        got = out[int(poff[t]): int(poff[t]) + int(plen[t])].cpu().numpy()
        assert np.array_equal(got, oracle.encode(s)), t
    tr = np.zeros(T, dtype=TRACK_SIZES_DTYPE)
    tr["base_bit_off"] = 8 * poff.astype(np.uint64); tr["out_base"] = np.concatenate([[0], np.cumsum(ns)[:-1]]).astype(np.uint64)
    tr["total_samples"] = ns; tr["n_sizes"] = chunks; tr["high_res"] = HIGH_RES_FROM_PAYLOAD; tr["hdr_bits"] = 2
    dst = torch.empty(sum(ns) * W + 16, dtype=torch.uint8, device="cuda"); torch.cuda.synchronize()
    assert ctx.decode_tracks_device(_dev(out), cap, tr, sizes, _dev(dst), 24) == sizes.size
    o = 0
    for t in range(T):
        assert torch.equal(dst[o * W: (o + ns[t]) * W], pcm[t * stride: t * stride + ns[t] * W]), t
        o += ns[t]
    del pcm, out, dst
    torch.cuda.empty_cache()
    ctx.trim()


def test_batches_beyond_one_scan_tile(ctx, oracle):
    """More than 4096 chunks and more than 4096 tracks in one batch: the layout scans (chunk offsets,
    track offsets, the decoder's chunk positions) then run tiled over several workgroups."""
    import torch
    from velox_amd.codec import HIGH_RES_FROM_PAYLOAD, TRACK_SIZES_DTYPE
    from velox_amd.synth import pack_pcm, synth_pcm, synth_pcm_device
    # one long track: 5000 full chunks and a tail
    frames = 4096 * 5000 + 100
    pcm = synth_pcm_device(16, 2, 44100, frames, 8000.0, 150.0, 5, "cuda")
    n = 2 * frames
    cap = ctx.encode_bound([n])
    out = torch.zeros(cap, dtype=torch.uint8, device="cuda"); torch.cuda.synchronize()
    poff, plen = ctx.encode_device(_dev(pcm), 16, [0], [n], _dev(out), cap)
    sizes = ctx.last_chunk_sizes()
    assert sizes.size == 5001
    assert int(plen[0]) == int(sizes.astype(np.int64).sum()) + 4 * sizes.size + 1
    tr = np.zeros(1, dtype=TRACK_SIZES_DTYPE)
    tr[0] = (8 * int(poff[0]), 0, n, sizes.size, HIGH_RES_FROM_PAYLOAD, 2, 0)
    dst = torch.zeros(n * 2 + 16, dtype=torch.uint8, device="cuda"); torch.cuda.synchronize()
    assert ctx.decode_tracks_device(_dev(out), cap, tr, sizes, _dev(dst), 16) == 5001
    assert torch.equal(dst[: n * 2], pcm[: n * 2])
    # 4200 short tracks of different lengths
    rng = np.random.default_rng(3)
    lens = (2 * rng.integers(1, 120, 4200)).astype(np.int64)
    base = synth_pcm(16, 2, 44100, 0.2, 9000, 120, 77)
    starts = rng.integers(0, base.size // 2 - 130, 4200) * 2
    stride = 512                                            # bytes per track slot (>= 240 samples * 2 B, multiple of 16)
    host = np.zeros(4200 * stride + 16, np.uint8)
    for i in range(4200):
        b = pack_pcm(base[starts[i]: starts[i] + lens[i]], 16)
        host[i * stride: i * stride + b.size] = b
    d_pcm = torch.from_numpy(host).cuda()
    offs = [i * stride for i in range(4200)]
    cap = ctx.encode_bound(list(lens))
    out = torch.zeros(cap, dtype=torch.uint8, device="cuda"); torch.cuda.synchronize()
    poff, plen = ctx.encode_device(_dev(d_pcm), 16, offs, list(lens), _dev(out), cap)
    sizes = ctx.last_chunk_sizes()
    assert sizes.size == 4200
    got = out.cpu().numpy()
    for i in (0, 1, 2047, 4095, 4096, 4097, 4199):
        want = oracle.encode(base[starts[i]: starts[i] + lens[i]])
        assert np.array_equal(got[int(poff[i]): int(poff[i]) + int(plen[i])], want), f"track {i}"
    tr = np.zeros(4200, dtype=TRACK_SIZES_DTYPE)
    tr["base_bit_off"] = 8 * poff.astype(np.uint64); tr["out_base"] = np.concatenate([[0], np.cumsum(lens)[:-1]]).astype(np.uint64)
    tr["total_samples"] = lens; tr["n_sizes"] = 1; tr["high_res"] = HIGH_RES_FROM_PAYLOAD; tr["hdr_bits"] = 2
    total = int(lens.sum())
    dst = torch.zeros(total * 8 + 16, dtype=torch.uint8, device="cuda"); torch.cuda.synchronize()
    assert ctx.decode_tracks_device(_dev(out), cap, tr, sizes, _dev(dst), 64) == 4200
    res = dst.cpu().numpy()[: total * 8].view("<i8")
    want = np.concatenate([base[starts[i]: starts[i] + lens[i]] for i in range(4200)])
    assert np.array_equal(res, want)


def test_cpp_shim_roundtrip():
    """The reference-shaped C++ classes (velox_amd/cpp/VeloxCodecAMD.h) called the way main.cpp and the
    players call the reference; the binary checks itself against the oracle (built by build())."""
    import os
    import subprocess
    from conftest import ROOT
    exe = os.path.join(ROOT, "tests", "cpp", "shim_roundtrip")
    assert os.path.exists(exe), "run __graft_entry__.build() first"
    r = subprocess.run([exe], capture_output=True, text=True, timeout=100)
    assert r.returncode == 0, r.stdout + r.stderr


@pytest.mark.parametrize("parts", [2, 3])
def test_shard_encode_merges_to_whole_payload(ctx, oracle, parts):
    """The multi-GPU building blocks on one GPU: cut a track into runs of whole sub-blocks, encode each
    with vlx_encode_shard_device (file-global high_res from vlx_scan_high_res_device, MAX-combined), place
    the shard payloads with the offset table -> byte-identical to the whole-track oracle payload; every
    shard also decodes on its own from its chunk sizes."""
    import torch
    from velox_amd import sharded
    from velox_amd.codec import chunks_from_sizes
    from velox_amd.synth import pack_pcm, synth_pcm
    s = synth_pcm(24, 2, 96000, 0.3, 2.0e6, 3000.0, 91)
    ranges = sharded.shard_ranges(s.size, parts)
    s[: ranges[0][1]] //= 64                              # the first shard alone is quiet: high_res must come from the others
    want = oracle.encode(s)
    shards = []
    for a, n in ranges:
        pcm = torch.from_numpy(np.concatenate([pack_pcm(s[a: a + n], 24), np.zeros(16, np.uint8)])).cuda()
        shards.append(sharded.GpuShard(ctx, pcm, 24, n))
    flags = [sh.scan() for sh in shards]
    assert flags[0] == 0 and max(flags) == 1
    segs, sizes, outs = [], [], []
    for i, sh in enumerate(shards):
        seg, sz = sh.encode(max(flags), i == 0)
        segs.append(seg.cpu().numpy().copy()); sizes.append(sz.copy())
        ch = chunks_from_sizes(sz, sh.n, 0, 0, max(flags))
        dst = torch.zeros(sh.n * 3 + 16, dtype=torch.uint8, device="cuda"); torch.cuda.synchronize()
        ctx.decode_device(sh.payload.data_ptr(), sh.cap, ch, dst.data_ptr(), 24)
        outs.append(dst[: sh.n * 3].cpu().numpy())
    assert np.array_equal(sharded.merge_shards_host(segs, sizes), want)
    assert np.array_equal(np.concatenate(outs), pack_pcm(s, 24))


@pytest.mark.parametrize("parts", [2, 3])
def test_shard_decode_concatenates_to_whole_track(ctx, oracle, parts):
    """SURVEY §8e decode side on one GPU: each "rank" uploads only its byte range of the payload and decodes
    its chunk run (velox_amd.sharded.decode_track_sharded); the slices concatenate to the reference decode."""
    from velox_amd import sharded
    from velox_amd.codec import walk_chunks
    g = load_golden("s24_stereo_96k_hires")
    s = g["samples"].astype(np.int64)
    ch = walk_chunks(g["payload"], s.size)
    got = np.zeros(s.size, np.int64)
    for r in range(parts):
        lo, t = sharded.decode_track_sharded(ctx, g["payload"], ch, r, parts, 64, "cuda")
        v = t.cpu().numpy().view("<i8")
        got[lo: lo + v.size] = v
    assert np.array_equal(got, oracle.decode(g["payload"], s.size))


def test_many_streams_one_batch_and_seek(ctx, oracle, wave_shape):
    """BASELINE config 5 through the decoder API: S decoders over different payloads are refilled by ONE
    GPU batch per tick (vlx_decoder_refill_many) and each hands out 4096 samples per tick
    (VeloxStreamClient.cpp:450); seek lands where restart-and-fast-forward would."""
    from velox_amd.codec import StreamingDecoder, refill_many
    names = ["s16_stereo_44k", "s24_stereo_96k_hires", "short_tail_block", "s16_3ch", "silence", "float_true",
             "white_noise_fullscale_s16", "lsb_shifted_x16", "tiny_20_samples", "impulses_escape_codes"]
    decs, wants, gots = [], [], []
    for n in names * 3:
        g = load_golden(n)
        total = g["mant"].size if "mant" in g else g["samples"].size
        want = g["dec_vals"].astype(np.int64) if "dec_vals" in g else (g["decoded"].astype(np.int64) if "decoded" in g else g["samples"].astype(np.int64))
        decs.append(StreamingDecoder(g["payload"], total, ctx=ctx, lookahead=1))
        wants.append(want); gots.append([])
    live = list(range(len(decs)))
    while live:
        refill_many([decs[i] for i in live])
        nxt = []
        for i in live:
            v, _ = decs[i].read(4096)
            gots[i].append(v)
            if v.size == 4096:
                nxt.append(i)
        live = nxt
    for i in range(len(decs)):
        assert np.array_equal(np.concatenate(gots[i]), wants[i]), names[i % len(names)]
    # seek
    g = load_golden("s24_stereo_96k_hires")
    s = g["samples"].astype(np.int64)
    d = StreamingDecoder(g["payload"], s.size, ctx=ctx, lookahead=2)
    for target in (0, 1, 8191, 8192, 8193, 20000, s.size - 1, s.size, s.size + 5):
        d.seek(target)
        v, _ = d.read(3000)
        assert np.array_equal(v, s[target: target + 3000]), target
    f = load_golden("float_true")
    d = StreamingDecoder(f["payload"], f["mant"].size, ctx=ctx)
    d.seek(9000)
    v, e = d.read(500)
    assert np.array_equal(v, f["dec_vals"][9000:9500].astype(np.int64)) and np.array_equal(e, f["dec_exps"][9000:9500])


def test_decoder_over_a_buffer_that_is_still_downloading(ctx, oracle):
    """VeloxStreamClient.cpp:379-459: the decoder is handed a RAM buffer the downloader is still filling
    (256 KiB GETs) and decodes 4096 samples per batch once `downloaded >= estimated position + 16384`.
    Same pacing here with vlx_decoder_extend; the samples must equal a decode of the complete payload."""
    from velox_amd.codec import StreamingDecoder
    g = load_golden("s16_stereo_44k")
    full, total = g["payload"], g["samples"].size
    ram = np.zeros(full.size, dtype=np.uint8)                       # globalFileRAM: zero-filled, filled front to back
    downloaded = 0
    step = 4000                                                       # a slow link: 4000 bytes arrive between decoder polls
    downloaded = min(full.size, 8192)
    ram[:downloaded] = full[:downloaded]
    d = StreamingDecoder(ram, total, ctx=ctx, lookahead=1, size=downloaded)
    got, decoded = [], 0
    while decoded < total:
        est = decoded * full.size // total
        while downloaded < est + 16384 and downloaded < full.size:
            n = min(step, full.size - downloaded)
            ram[downloaded: downloaded + n] = full[downloaded: downloaded + n]
            downloaded += n
            d.extend(downloaded)
        # the reference client trusts its linear estimate; vlx_decoder_bytes_needed says exactly how far the
        # buffer must reach before the next 4096 samples can be decoded
        while downloaded < min(full.size, d.bytes_needed(4096)):
            n = min(step, full.size - downloaded)
            ram[downloaded: downloaded + n] = full[downloaded: downloaded + n]
            downloaded += n
            d.extend(downloaded)
        v, _ = d.read(4096)
        if v.size == 0:
            break
        got.append(v); decoded += v.size
    assert np.array_equal(np.concatenate(got), oracle.decode(full, total))


def test_side_term_beyond_int32_into_a_32_bit_container(ctx, oracle, wave_shape):
    """32-bit stereo whose channels agree almost everywhere (mid/side wins, VeloxCore.h:262-271) except for
    a few full-scale opposite-sign frames: there L-R needs 33 bits, the chunk leaves the 32-bit fast path,
    and the inverse transform (VeloxCore.h:357-364) needs the high word of the side value even though the
    output container is 32-bit.  Also many such chunks at once (more than the hi-word pool starts with)."""
    import torch
    from velox_amd.codec import walk_chunks
    rng = np.random.default_rng(12)
    n_frames = 8192 * 12                                                 # 24 chunks
    l = rng.integers(-2 ** 29, 2 ** 29, n_frames)
    r = l + rng.integers(-50, 50, n_frames)
    for c in range(24):                                                  # one 33-bit side term per chunk
        f = c * 4096 + int(rng.integers(0, 4096))
        l[f], r[f] = 2 ** 31 - 1, -2 ** 31
    s = np.stack([l, r], 1).reshape(-1).astype(np.int64)
    p = oracle.encode(s)
    assert np.array_equal(ctx.encode_block(s), p)
    want = oracle.decode(p, s.size)
    assert np.array_equal(want, s)
    ch = walk_chunks(p, s.size)
    d_p = torch.from_numpy(np.concatenate([p, np.zeros(16, np.uint8)])).cuda()
    for fmt, w in ((32, 4), (64, 8)):
        dst = torch.zeros(s.size * w + 16, dtype=torch.uint8, device="cuda"); torch.cuda.synchronize()
        ctx.decode_device(d_p.data_ptr(), p.size + 16, ch, dst.data_ptr(), fmt)
        got = dst.cpu().numpy()[: s.size * w]
        if fmt == 64:
            assert np.array_equal(got.view("<i8"), want)
        else:
            assert np.array_equal(got, oracle.samples_to_bytes(want, 32))


def test_workspace_reserve_trim_and_buffer_contract(oracle):
    """vlx_ctx_reserve sizes the workspace up front (a batch of that size then allocates nothing more),
    vlx_ctx_trim gives it back and the next call grows it again; a payload pointer that is not 4-byte
    aligned is refused (include/velox_amd.h: buffer contract of the device decode entry points)."""
    import torch
    from velox_amd.codec import Context, VeloxError, walk_chunks
    c = Context(0)
    assert c.workspace_bytes() == 0
    g = load_golden("s24_stereo_96k_hires")
    s = g["samples"].astype(np.int64)
    n_chunks = (s.size + 8191) // 8192
    c.reserve(encode_chunks=n_chunks, encode_tracks=1, decode_chunks=n_chunks)
    held = c.workspace_bytes()
    assert held > n_chunks * 2 * 4096 * 4
    from velox_amd.synth import pack_pcm
    pcm = torch.from_numpy(np.concatenate([pack_pcm(s, 24), np.zeros(16, np.uint8)])).cuda()
    cap = c.encode_bound([s.size])
    out = torch.zeros(cap, dtype=torch.uint8, device="cuda"); torch.cuda.synchronize()
    poff, plen = c.encode_device(pcm.data_ptr(), 24, [0], [s.size], out.data_ptr(), cap)
    assert np.array_equal(out[: int(plen[0])].cpu().numpy(), g["payload"])
    ch = walk_chunks(g["payload"], s.size)
    dst = torch.zeros(s.size * 8 + 16, dtype=torch.uint8, device="cuda"); torch.cuda.synchronize()
    c.decode_device(out.data_ptr(), cap, ch, dst.data_ptr(), 64)
    assert np.array_equal(dst.cpu().numpy()[: s.size * 8].view("<i8"), s)
    assert c.workspace_bytes() == held                                # nothing grew
    # 4-byte and wider formats take the slots of the sequential autocorrelation too (loud 32-bit material parks
    # every chunk there) and several tracks more pinned staging: reserved for, then nothing grows either
    from velox_amd.synth import synth_pcm_device
    for fmt, W in ((32, 4), (64, 8)):
        n32 = 12 * 8192 + 64
        src = synth_pcm_device(32, 8, 192000, n32 // 8, 5.0e8, 2.0e5, 3, "cuda")
        if fmt == 64:
            src = src.view(torch.int32).to(torch.int64).view(torch.uint8)
        three = torch.cat([src, torch.zeros((-src.numel()) % 16, dtype=torch.uint8, device="cuda")] * 3)
        stride = three.numel() // 3
        ns = [n32] * 3
        cap3 = c.encode_bound(ns)
        out3 = torch.zeros(cap3, dtype=torch.uint8, device="cuda"); torch.cuda.synchronize()
        c.reserve(encode_chunks=3 * 13, encode_tracks=3, fmt=fmt)
        held = c.workspace_bytes()
        po3, pl3 = c.encode_device(three.data_ptr(), fmt, [0, stride, 2 * stride], ns, out3.data_ptr(), cap3)
        assert c.workspace_bytes() == held, fmt
        want = oracle.encode(src[: n32 * W].cpu().numpy().view("<i4" if fmt == 32 else "<i8").astype(np.int64))
        for t in range(3):
            assert np.array_equal(out3[int(po3[t]): int(po3[t]) + int(pl3[t])].cpu().numpy(), want), (fmt, t)
    c.trim()
    assert c.workspace_bytes() == 0
    c.decode_device(out.data_ptr(), cap, ch, dst.data_ptr(), 64)       # grows again on demand
    assert c.workspace_bytes() > 0
    # misaligned payload pointer: same bytes one byte further on
    shifted = torch.zeros(cap + 16, dtype=torch.uint8, device="cuda")
    shifted[1: 1 + cap] = out; torch.cuda.synchronize()
    with pytest.raises(VeloxError) as e:
        c.decode_device(shifted.data_ptr() + 1, cap, ch, dst.data_ptr(), 64)
    assert e.value.code == -1 and "aligned" in str(e.value)
    with pytest.raises(VeloxError):
        c.decode_device_noalloc(shifted.data_ptr() + 1, cap, ch, dst.data_ptr(), 64)
    # 4-byte (not 16-byte) aligned is fine: the fast kernel works on the enclosing 16-B groups
    shifted.zero_(); shifted[4: 4 + cap] = out; dst.zero_(); torch.cuda.synchronize()
    c.decode_device(shifted.data_ptr() + 4, cap, ch, dst.data_ptr(), 64)
    assert np.array_equal(dst.cpu().numpy()[: s.size * 8].view("<i8"), s)
    c.close()


def test_decoder_handed_to_another_context():
    """A decoder opened on one context keeps working on another (vlx_decoder_set_ctx) after the first
    context is gone — what VeloxCodecAMD.h does when a StreamingDecoder moves to another thread."""
    import ctypes as C
    from velox_amd.codec import Context, StreamingDecoder
    g = load_golden("s16_stereo_44k")
    s = g["samples"].astype(np.int64)
    a, b = Context(0), Context(0)
    d = StreamingDecoder(g["payload"], s.size, ctx=a, lookahead=2)
    first, _ = d.read(10000)
    assert a.L.vlx_decoder_set_ctx(d.h, b.h) == 0
    d.ctx = b
    a.close()
    rest, _ = d.read(s.size)
    assert np.array_equal(np.concatenate([first, rest]), s)


def test_native_node_layer_one_rank(ctx, oracle):
    """The native multi-GPU layer (vlx_node_*: RCCL called from the library) with a world of ONE rank — all a
    one-GPU box can run: communicator set-up, the high_res all-reduce on the device flag, the size-row
    all-gather, the offset table, the gather into a root buffer, the corpus length table.  The shard of the
    only rank is the whole payload, so it must equal the oracle's bytes."""
    import torch
    from velox_amd import sharded
    from velox_amd.synth import pack_pcm, synth_pcm
    uid = sharded.NativeNode.unique_id()
    node = sharded.NativeNode(ctx, uid, 0, 1)
    try:
        for bits, amp in ((24, 2.0e6), (16, 8000.0), (24, 2000.0)):       # high-res, 16-bit (no scan), quiet 24-bit (scan says no)
            s = synth_pcm(bits, 2, 48000, 0.5, amp, amp / 600, 31)
            a, nloc = sharded.native_shard_range(s.size, 0, 1)
            assert (a, nloc) == (0, s.size)
            pcm = torch.from_numpy(np.concatenate([pack_pcm(s, bits), np.zeros(16, np.uint8)])).cuda()
            cap = ctx.encode_bound([s.size])
            out = torch.zeros(cap, dtype=torch.uint8, device="cuda"); torch.cuda.synchronize()
            ln, off, total, hr = node.encode_track(pcm.data_ptr(), bits, s.size, out.data_ptr(), cap)
            want = oracle.encode(s)
            assert (ln, off, total) == (want.size, 0, want.size) and hr == ((int(want[0]) >> 1) & 1)
            assert np.array_equal(out[:ln].cpu().numpy(), want)
            sizes = node.chunk_sizes()
            assert sizes.size == (s.size + 8191) // 8192 and int(sizes.astype(np.int64).sum()) + 4 * sizes.size + 1 == total
            full = torch.full((total + 16,), 0xEE, dtype=torch.uint8, device="cuda"); torch.cuda.synchronize()
            node.gather_payload(0, out.data_ptr(), ln, full.data_ptr(), total + 16)
            assert np.array_equal(full[:total].cpu().numpy(), want) and int(full[total].item()) == 0xEE
        tab = node.all_gather_lengths([5, 7, 11], 4)
        assert len(tab) == 1 and tab[0].tolist() == [5, 7, 11]
    finally:
        node.close()


@pytest.mark.gpu
@pytest.mark.parametrize("slots", [0, 3, -1])
def test_loud_32_bit_chunks_in_slots_and_beyond_them(ctx, oracle, slots):
    """Chunks whose autocorrelation is the reference's sequential fp64 sum (energy >= 2^53) are parked in slots and their
    chains run in a kernel of their own; a batch has a bounded number of slots, and chunks beyond them run their chains
    inside k_analyze.  The same bytes on every route, and the oracle's: no slots, three slots for 20 chunks, the default."""
    import ctypes
    import torch
    from velox_amd import _lib
    from velox_amd.synth import synth_pcm_device, unpack_pcm
    L = _lib.load()
    L.vlx_debug_set_seq_slots.argtypes, L.vlx_debug_set_seq_slots.restype = [ctypes.c_long], None
    chn, W = 8, 4
    n = 20 * 8192 + 2 * chn * 11
    pcm = synth_pcm_device(32, chn, 192000, n // chn, 5.0e8, 2.0e5, 21, "cuda")
    n = pcm.numel() // W
    cap = ctx.encode_bound([n])
    out = torch.empty(cap, dtype=torch.uint8, device="cuda"); torch.cuda.synchronize()
    L.vlx_debug_set_seq_slots(slots)
    try:
        poff, plen = ctx.encode_device(_dev(pcm), 32, [0], [n], _dev(out), cap)
    finally:
        L.vlx_debug_set_seq_slots(-1)
    got = out[int(poff[0]): int(poff[0]) + int(plen[0])].cpu().numpy()
    want = oracle.encode(unpack_pcm(pcm.cpu().numpy(), 32))
    assert got.tobytes() == want.tobytes()
