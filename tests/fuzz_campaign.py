#!/usr/bin/env python3
"""Randomised parity campaign: HIP path vs the CPU oracle, bit-exact, many signal families.

Not collected by pytest (a long-running hunt, run by hand on a GPU box):

    python tests/fuzz_campaign.py --seconds 120 --seed 1

Per case: encode_block (int64 boundary) and, where the values fit, encode_pcm (packed container)
against oracle.encode; StreamingDecoder over the oracle's payload against oracle.decode under every
decode shape; the same for a corrupted and a truncated copy of the payload (the reference
defines reads past the end as zero bits, VeloxEntropy.h:61-62, so every byte string has one answer).
Prints one line per failure with the seed that reproduces it, and a summary.
"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def gen(rng):
    """One random interleaved int64 signal + the container width it fits (or 0 for int64 only)."""
    fam = int(rng.integers(0, 14))
    bits = int(rng.choice([16, 24, 32]))
    lim = 1 << (bits - 1)
    n = 2 * int(rng.integers(1, 21000))
    if rng.random() < 0.15:
        n = 2 * int(rng.integers(1, 40))
    t = np.arange(n // 2)
    amp = lim * 10 ** rng.uniform(-4, 0)
    if fam == 0:                                           # tone + noise, correlated channels
        l = amp * np.sin(t * rng.uniform(1e-3, 1.5)) + rng.normal(0, amp * 10 ** rng.uniform(-4, -0.5), t.size)
        r = rng.uniform(-1, 1) * l + rng.normal(0, amp * 10 ** rng.uniform(-4, -0.5), t.size)
    elif fam == 1:                                         # white noise
        l = rng.normal(0, amp, t.size); r = rng.normal(0, amp, t.size)
    elif fam == 2:                                         # uniform full scale
        l = rng.integers(-lim, lim, t.size).astype(float); r = rng.integers(-lim, lim, t.size).astype(float)
    elif fam == 3:                                         # silence with islands
        l = np.zeros(t.size); r = np.zeros(t.size)
        for _ in range(int(rng.integers(0, 4))):
            a = int(rng.integers(0, t.size)); b = min(t.size, a + int(rng.integers(1, 3000)))
            l[a:b] = rng.normal(0, amp, b - a)
            if rng.random() < 0.5:
                r[a:b] = rng.normal(0, amp, b - a)
    elif fam == 4:                                         # LSB-shifted
        sh = int(rng.integers(1, bits - 2))
        l = (rng.normal(0, max(amp / (1 << sh), 1.0), t.size)).round() * (1 << sh); r = (rng.normal(0, max(amp / (1 << sh), 1.0), t.size)).round() * (1 << sh)
    elif fam == 5:                                         # identical / near-identical channels
        l = amp * np.sin(t * rng.uniform(1e-3, 0.5)) + rng.normal(0, amp * 0.01, t.size); r = l + rng.integers(0, 3, t.size)
    elif fam == 6:                                         # impulses on a quiet floor (escape codes)
        l = rng.integers(-3, 4, t.size).astype(float); r = rng.integers(-3, 4, t.size).astype(float)
        m = rng.random(t.size) < rng.uniform(0, 0.01)
        l[m] = rng.integers(-lim, lim, int(m.sum()))
    elif fam == 7:                                         # DC / constant / ramps
        l = np.full(t.size, float(rng.integers(-lim, lim))); r = np.linspace(-amp, amp, t.size)
    elif fam == 8:                                         # square wave / clipping
        l = np.clip(3 * amp * np.sin(t * rng.uniform(1e-3, 0.3)), -amp, amp); r = np.sign(np.sin(t * rng.uniform(1e-3, 0.3))) * amp
    elif fam == 9:                                         # random walk
        l = np.cumsum(rng.normal(0, amp * 0.01, t.size)); r = np.cumsum(rng.normal(0, amp * 0.01, t.size))
    elif fam == 10:                                        # alternating extremes
        l = np.where(t % 2 == 0, lim - 1, -lim).astype(float); r = -l
    elif fam == 11:                                        # one channel silent
        l = amp * np.sin(t * rng.uniform(1e-3, 0.5)) + rng.normal(0, amp * 0.01, t.size); r = np.zeros(t.size)
    elif fam == 12:                                        # int64-boundary values beyond any container
        big = 2.0 ** rng.uniform(31, 44)
        l = rng.normal(0, big, t.size); r = rng.normal(0, big * 10 ** rng.uniform(-6, 0), t.size)
        bits = 0
    else:                                                  # mostly small with rare huge int64 outliers
        l = rng.normal(0, 50, t.size); r = rng.normal(0, 50, t.size)
        m = rng.random(t.size) < 0.001
        l[m] = rng.normal(0, 2.0 ** 45, int(m.sum()))
        bits = 0
    s = np.rint(np.stack([l, r], 1).reshape(-1))
    if bits:
        s = np.clip(s, -lim, lim - 1)
    return s.astype(np.int64), bits, fam


def float_case(rng, O, ctx, StreamingDecoder):
    """True float32, pseudo-float (16/24-bit integers stored as floats) and mixtures through the float
    prologue (VeloxCore.h:214-230) and back; returns the list of failing checks."""
    n = 2 * int(rng.integers(1, 12000))
    kind = int(rng.integers(0, 4))
    t = np.arange(n)
    if kind == 0:
        f = (rng.uniform(0.001, 1.0) * np.sin(t * rng.uniform(1e-3, 0.5)) + rng.normal(0, 10 ** rng.uniform(-5, -1), n)).astype("<f4")
    elif kind == 1:
        f = (rng.integers(-32768, 32768, n) / 32768.0).astype("<f4")
    elif kind == 2:
        f = (rng.integers(-(1 << 23), 1 << 23, n) / float(1 << 23)).astype("<f4")
    else:
        f = rng.normal(0, 10 ** rng.uniform(-3, 3), n).astype("<f4")
        f[rng.random(n) < 0.01] = 0.0
    raw = f.view(np.uint8)
    m, e = O.split_float32(raw)
    bad = []
    want = O.encode(m, True, e, raw)
    if not np.array_equal(ctx.encode_block(m, True, e, raw), want):
        bad.append(f"float{kind}:encode_block")
    dref = O.decode(want, m.size, True)
    d = StreamingDecoder(want, m.size, ctx=ctx, lookahead=int(rng.integers(1, 4)))
    dv, de = d.read(m.size)
    if not (np.array_equal(dv, dref[0]) and np.array_equal(de, dref[1]) and d.IsFloat() == bool(dref[2]) and d.GetFloatMode() == dref[3]):
        bad.append(f"float{kind}:decode")
    # the same track with the samples staying in HBM: vlx_encode_f32_device / vlx_decode_f32_device
    import torch
    d_raw = torch.from_numpy(raw.copy()).cuda()
    cap = ctx.encode_bound([n]) + 2 * n + 256
    d_pay = torch.zeros(cap + 16, dtype=torch.uint8, device="cuda")
    plen, fm = ctx.encode_f32_device(d_raw.data_ptr(), n, d_pay.data_ptr(), cap)
    if plen != want.size or not np.array_equal(d_pay[:plen].cpu().numpy(), want):
        bad.append(f"float{kind}:encode_f32_device")
    else:
        sizes = ctx.last_chunk_sizes()
        if rng.random() < 0.3 and sizes.size > 1:
            sizes = sizes[: int(rng.integers(1, sizes.size))]               # a chunk list that ends early
        vals, exps = dref[0], dref[1]
        reach = min(n, 8192 * sizes.size)
        v = np.zeros(n, np.int64); v[:reach] = vals[:reach]
        e2 = np.zeros(n, np.uint8); e2[:reach] = exps[:reach]
        ref = O.merge_float32(v, e2) if dref[3] == 0 else O.promote_int_to_float(v, 16 if dref[3] == 1 else 24)
        d_out = torch.full((4 * n + 16,), 0x77, dtype=torch.uint8, device="cuda")
        fm2, _ = ctx.decode_f32_device(d_pay.data_ptr(), plen, n, sizes, d_out.data_ptr())
        if fm2 != dref[3] or fm != dref[3] or not np.array_equal(d_out[: 4 * n].cpu().numpy(), ref) or not bool((d_out[4 * n:] == 0x77).all()):
            bad.append(f"float{kind}:decode_f32_device")
    return bad


def device_case(rng, O, ctx):
    """vlx_encode_device / vlx_decode_device: a random corpus of tracks of one container width in one HBM
    buffer; every track's payload against the oracle, then a random subset of the chunks, shuffled,
    decoded in one batch (both wave shapes, packed and int64 output)."""
    import torch
    from velox_amd.codec import set_decode_shape, walk_chunks
    from velox_amd.synth import pack_pcm
    bits = int(rng.choice([16, 24, 32]))
    lim = 1 << (bits - 1)
    tracks = []
    many = rng.random() < 0.15                             # a corpus of >= 256 short tracks: the per-track high-res scan
    with_odd = rng.random() < 0.2                          # some odd totals: routed to the one-stream path, placed behind the batch
    for _ in range(int(rng.integers(256, 420)) if many else int(rng.integers(1, 24))):
        n = 2 * int(rng.integers(0, 9000 if many else 14000)) if rng.random() < 0.9 else 0
        if with_odd and rng.random() < 0.3:
            n = 2 * int(rng.integers(0, 3000)) + 1
        amp = lim * 10 ** rng.uniform(-4, 0)
        t = np.arange((n + 1) // 2)
        l = amp * np.sin(t * rng.uniform(1e-3, 1.0)) + rng.normal(0, amp * 10 ** rng.uniform(-3, -0.5), t.size)
        r = rng.uniform(-1, 1) * l + rng.normal(0, amp * 10 ** rng.uniform(-3, -0.5), t.size)
        x = np.clip(np.rint(np.stack([l, r], 1).reshape(-1)), -lim, lim - 1).astype(np.int64)[:n]
        if many and x.size and rng.random() < 0.3:         # loud in one late sample only
            x[int(rng.integers(x.size * 3 // 4, x.size))] = lim - 1
        tracks.append(x)
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
    poff, plen = ctx.encode_device(pcm.data_ptr(), bits, offs, ns, out.data_ptr(), cap)
    host = out.cpu().numpy()
    bad, chunks, total_out = [], [], 0
    for i, s in enumerate(tracks):
        got = host[int(poff[i]): int(poff[i]) + int(plen[i])]
        if not np.array_equal(got, O.encode(s)):
            bad.append(f"device:encode track {i}/{len(tracks)}")
        chunks.append(walk_chunks(got, s.size, base_bit_off=8 * int(poff[i]), out_base=total_out))
        total_out += s.size
    ch = np.concatenate(chunks)
    if ch.size == 0 or bad or any(x.size & 1 for x in tracks):            # (the reference's decoder cannot read an odd total's payload)
        return bad
    keep = rng.permutation(ch.size)[: max(1, int(ch.size * rng.uniform(0.3, 1.0)))]
    sub = ch[keep]
    want = np.concatenate(tracks)
    mask = np.zeros(total_out, bool)
    for c in sub:
        mask[int(c["out_off"]): int(c["out_off"]) + int(c["n_out"])] = True
    for shape in ("stream", "chunk", "pipe16", "pipe64"):
        set_decode_shape(shape)
        for fmt in (bits, 64):
            w = 8 if fmt == 64 else fmt // 8
            dst = torch.zeros(total_out * w + 16, dtype=torch.uint8, device="cuda"); torch.cuda.synchronize()
            ctx.decode_device(out.data_ptr(), cap, sub, dst.data_ptr(), fmt)
            res = dst.cpu().numpy()[: total_out * w]
            if fmt == 64:
                ok = np.array_equal(res.view("<i8")[mask], want[mask]) and not res.view("<i8")[~mask].any()
            else:
                ref = pack_pcm(want, bits).reshape(-1, w)
                ok = np.array_equal(res.reshape(-1, w)[mask], ref[mask]) and not res.reshape(-1, w)[~mask].any()
            if not ok:
                bad.append(f"device:decode[{shape},fmt{fmt}] {sub.size}/{ch.size} chunks")
    set_decode_shape("auto")
    # the whole corpus through vlx_decode_tracks_device (chunk table built on the device from the byte sizes)
    from velox_amd.codec import HIGH_RES_FROM_PAYLOAD, TRACK_SIZES_DTYPE
    sizes = ctx.last_chunk_sizes()
    tr = np.zeros(len(tracks), dtype=TRACK_SIZES_DTYPE)
    ob = 0
    for i, s in enumerate(tracks):
        hires = HIGH_RES_FROM_PAYLOAD if rng.random() < 0.5 else int((host[int(poff[i])] >> 1) & 1)
        tr[i] = (8 * int(poff[i]), ob, s.size, (s.size + 8191) // 8192, hires, 2, 0)
        ob += s.size
    fmt = bits if rng.random() < 0.5 else 64
    w = 8 if fmt == 64 else fmt // 8
    dst = torch.zeros(total_out * w + 16, dtype=torch.uint8, device="cuda"); torch.cuda.synchronize()
    n = ctx.decode_tracks_device(out.data_ptr(), cap, tr, sizes, dst.data_ptr(), fmt)
    res = dst.cpu().numpy()[: total_out * w]
    ok = n == ch.size and (np.array_equal(res.view("<i8"), want) if fmt == 64 else np.array_equal(res, pack_pcm(want, bits)))
    if not ok:
        bad.append(f"device:decode_tracks[fmt{fmt}] {n}/{ch.size} chunks")
    return bad


_POOLS = {}


def pool_case(rng, O, s, bits):
    """vlx_pool_*: the host-pointer drop-ins over K contexts on one GPU (K shards of the track in flight) — encoder bytes
    and packed-PCM decodes of a clean, a corrupted and a truncated payload against the oracle."""
    from velox_amd.codec import Pool
    from velox_amd.synth import pack_pcm
    K = int(rng.choice([2, 3, 5]))
    if K not in _POOLS:
        _POOLS[K] = Pool([0] * K)
    P = _POOLS[K]
    bad = []
    want = O.encode(s)
    if not np.array_equal(P.encode_block(s), want):
        bad.append(f"pool{K}:encode_block")
    if bits and not np.array_equal(P.encode_pcm(pack_pcm(s, bits), bits), want):
        bad.append(f"pool{K}:encode_pcm")
    if bits and s.size and not (s.size & 1):
        variants = [("clean", want)]
        if want.size > 12:
            j = want.copy()
            p = int(rng.integers(1, j.size)); q = min(j.size, p + int(rng.integers(1, 40)))
            j[p:q] = rng.integers(0, 256, q - p)
            variants += [("corrupt", j), ("truncated", want[: int(rng.integers(1, want.size))].copy())]
        for vname, payload in variants:
            dref = O.decode(payload, s.size)
            full = np.zeros(s.size, np.int64); full[: dref.size] = dref
            try:
                got, n = P.decode_pcm(payload, s.size, bits)
                ok = n == dref.size and np.array_equal(got, O.samples_to_bytes(full, bits))
            except Exception as e:                         # noqa: BLE001
                ok = bool(payload[0] & 1) and getattr(e, "code", 0) == -5
            if not ok:
                bad.append(f"pool{K}:decode_pcm[{vname}]")
    return bad


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=60.0)
    ap.add_argument("--seed", type=int, default=1)
    a = ap.parse_args()
    from _oracle import Oracle
    from velox_amd.codec import Context, StreamingDecoder, set_decode_shape, set_encode_shape
    from velox_amd.synth import pack_pcm
    O, ctx = Oracle(), Context(0)
    t_end = time.time() + a.seconds
    case = fails = 0
    fam_count = {}
    while time.time() < t_end:
        seed = a.seed * 1000003 + case
        rng = np.random.default_rng(seed)
        if case % 8 == 7:
            case += 1
            fam_count["float"] = fam_count.get("float", 0) + 1
            bad = float_case(rng, O, ctx, StreamingDecoder)
            if bad:
                fails += 1
                print(f"FAIL seed={seed} float: {', '.join(bad)}", flush=True)
            continue
        if case % 8 == 3:
            case += 1
            fam_count["device"] = fam_count.get("device", 0) + 1
            bad = device_case(rng, O, ctx)
            if bad:
                fails += 1
                print(f"FAIL seed={seed} device: {', '.join(bad)}", flush=True)
            continue
        s, bits, fam = gen(rng)
        fam_count[fam] = fam_count.get(fam, 0) + 1
        case += 1
        if case % 8 == 6:                                  # (case was already advanced: every eighth signal goes through a pool)
            if s.size and rng.random() < 0.4:              # long enough for every shard to hold several sub-blocks
                s = np.resize(s, int(rng.integers(4, 16)) * 8192 - 2 * int(rng.integers(0, 500)))
            fam_count["pool"] = fam_count.get("pool", 0) + 1
            bad = pool_case(rng, O, s, bits)
            if bad:
                fails += 1
                print(f"FAIL seed={seed} pool fam={fam} bits={bits} n={s.size}: {', '.join(bad)}", flush=True)
            continue
        bad = []
        want = O.encode(s)
        eshape = str(rng.choice(["auto", "stream", "pipe16", "pipe64"]))
        set_encode_shape(eshape)
        if not np.array_equal(ctx.encode_block(s), want):
            bad.append(f"encode_block[{eshape}]")
        if bits and not np.array_equal(ctx.encode_pcm(pack_pcm(s, bits), bits), want):
            bad.append(f"encode_pcm[{eshape}]")
        set_encode_shape("auto")
        variants = [("clean", want)]
        if want.size > 12:
            j = want.copy()
            for _ in range(int(rng.integers(1, 4))):
                p = int(rng.integers(1, j.size)); q = min(j.size, p + int(rng.integers(1, 40)))
                j[p:q] = rng.integers(0, 256, q - p)
            variants.append(("corrupt", j))
            variants.append(("truncated", want[: int(rng.integers(1, want.size))].copy()))
        for vname, payload in variants:
            dref = O.decode(payload, s.size)
            for shape in ("stream", "chunk", "pipe16", "pipe64"):
                set_decode_shape(shape)
                try:
                    got, _ = StreamingDecoder(payload, s.size, ctx=ctx, lookahead=int(rng.integers(1, 6))).read(s.size + 3)
                    ok = np.array_equal(got, dref)
                except Exception as e:                     # noqa: BLE001 - a campaign reports, it does not stop
                    ok = False
                    vname += f" EXC {e}"
                if not ok:
                    bad.append(f"decode[{vname},{shape}]")
        set_decode_shape("auto")
        if bits and s.size:                                # vlx_decode_pcm: what the stream reaches, then zeros, packed
            for vname, payload in variants:
                dref = O.decode(payload, s.size)
                full = np.zeros(s.size, np.int64); full[: dref.size] = dref
                for shape in ("auto", "stream"):            # "stream": packed 16 / 24-bit output lets the stream waves finish chunks themselves
                    set_decode_shape(shape)
                    try:
                        got, n = ctx.decode_pcm(payload, s.size, bits)
                        ok = n == dref.size and np.array_equal(got, O.samples_to_bytes(full, bits))
                    except Exception as e:                     # noqa: BLE001
                        ok = bool(payload[0] & 1) and getattr(e, "code", 0) == -5   # corruption may have set is_float: refused, not decoded
                    if not ok:
                        bad.append(f"decode_pcm[{vname},{shape}]")
                set_decode_shape("auto")
        if bad:
            fails += 1
            print(f"FAIL seed={seed} fam={fam} bits={bits} n={s.size}: {', '.join(bad)}", flush=True)
        if case % 25 == 0:
            print(f"... {case} cases, {fails} failing", flush=True)
    print(f"cases={case} failing={fails} families={dict(sorted(fam_count.items(), key=str))}")
    sys.exit(1 if fails else 0)


if __name__ == "__main__":
    main()
