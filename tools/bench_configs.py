#!/usr/bin/env python3
"""Secondary measurements for BASELINE.json configs[2..4] on ONE MI355X (bench.py keeps the
headline configs[1]).  Every run is parity-gated twice: encode -> decode must return the input bytes, and
chunks sampled across the batch (first, last, tail blocks, random ones) must equal the CPU oracle's encoding
of the same samples byte for byte (tests/_sampling.py); decode batches are compared with the oracle's decode
of sampled streams.  Every line carries `roofline` (algorithmic bytes of the direction / its dominant
kernel's time, against 8 TB/s) and `cpu_baseline` (the reference, or its C port, on the host cores over a
bounded sample of the same data).

    python tools/bench_configs.py cfg1                     16-bit stereo 44.1 kHz 60 s through the HOST-pointer entry points
    python tools/bench_configs.py cfg3 [--tracks 4096]     corpus of 24-bit/48 kHz stereo tracks, one batch
    python tools/bench_configs.py cfg4                     32-bit 8-channel 192 kHz, 60 s
    python tools/bench_configs.py cfg5                     streaming decode batches: S streams x next chunk

Prints one JSON line per measurement.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


HBM_PEAK_GBS = 8000.0   # /opt/skills/guides/MI355X_MICROARCH.md
QUICK = False           # --quick: GPU work only (the profiling passes of tools/prof_pmc.sh): no oracle checks, no CPU baseline


def _cores():
    from bench import host_cpu_share
    return host_cpu_share()


_ORACLES = []


def _oracles():
    if not _ORACLES:                                      # one Reference per process: its thread pool is a process-wide static
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        from _oracle import Oracle, Reference
        _ORACLES.append((Oracle(), Reference(threads=_cores()[0]) if Reference.available() else None))
    return _ORACLES[0]


def _profile(tag):
    """profiles/<tag>_pmc.json (tools/prof_pmc.sh), quoted only while it was taken from the kernel sources this run uses:
    the same hash rule as bench.py's.  -> (table or None, file name, note)."""
    from bench import csrc_hash
    f = os.path.join(ROOT, "profiles", tag + "_pmc.json")
    if not os.path.exists(f):
        return None, None, "no committed profile of this workload"
    try:
        with open(f) as fh:
            prof = json.load(fh)
    except (OSError, ValueError):
        return None, os.path.basename(f), "unreadable profile"
    if (prof.get("_meta") or {}).get("csrc_sha256") != csrc_hash():
        return None, os.path.basename(f), "profile taken from other kernel sources (hash of velox_amd/csrc differs): re-run tools/prof_pmc.sh"
    return prof, os.path.basename(f), None


def _roofline(alg_bytes, kernel_ms, names, tag=None):
    """Dominant kernel among `names` and the whole direction's algorithmic bytes over its time; HBM traffic and VALUBusy
    of that kernel from the workload's committed counter profile (null, with the reason, when the sources moved on)."""
    from bench import pmc_traffic, pmc_valu_busy
    ks = {k: kernel_ms[k] for k in names if k in kernel_ms}
    dom = max(ks, key=ks.get)
    ach = alg_bytes / (ks[dom] * 1e-3) / 1e9
    prof, pfile, note = _profile(tag) if tag else (None, None, "no counter profile is kept for this line")
    return {"bound": "hbm", "kernel": dom, "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 4),
            "traffic": pmc_traffic(prof, dom), "valu_busy_profiled": pmc_valu_busy(prof, dom), "profile": pfile, "profile_note": note,
            "algorithmic_bytes_per_launch": int(alg_bytes)}


def _cpu_baseline(samples, bits, what):
    """The reference (or the C port) on the host cores over `samples` (int64, a bounded sample of the workload)."""
    O, R = _oracles()
    cores, host_cores, cores_why = _cores()
    t0 = time.perf_counter()
    p = R.encode(samples) if R else O.encode(samples, threads=cores)
    t1 = time.perf_counter()
    d = R.decode(p, samples.size) if R else O.decode(p, samples.size)
    t2 = time.perf_counter()
    mb = samples.size * (bits // 8) / 1e6
    return {"value": round(mb / (t2 - t0), 2), "unit": "MB/s", "cores": R.pool_threads() if R else cores, "host_cores": host_cores, "cores_limited_by": cores_why,
            "kind": "reference" if R else "port",
            "sample": what, "encode_mb_s": round(mb / (t1 - t0), 2), "decode_mb_s": round(mb / (t2 - t1), 2),
            "round_trip": bool(np.array_equal(d, samples))}


def _time(fn, reps):
    import torch
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


def corpus(ctx, bits, ch, rate, seconds, n_tracks, amp, sigma, reps, name, tag=None):
    import torch
    from velox_amd.codec import chunks_from_sizes
    from velox_amd.synth import synth_corpus_device, synth_pcm_device, unpack_pcm
    W = bits // 8
    frames = int(rate * seconds)
    n = frames * ch                                       # samples per track
    stride = (n * W + 15) // 16 * 16
    if n_tracks > 1:                                      # every track its own signal, seed = track index
        pcm = synth_corpus_device(bits, ch, rate, frames, n_tracks, amp, sigma, "cuda", stride=stride)
    else:
        pcm = torch.zeros(stride + 16, dtype=torch.uint8, device="cuda"); torch.cuda.synchronize()
        pcm[: n * W] = synth_pcm_device(bits, ch, rate, frames, amp, sigma, 4, "cuda")
    torch.cuda.synchronize()                              # the ctx stream is not ordered against torch's
    offs = [t * stride for t in range(n_tracks)]
    ns = [n] * n_tracks
    cap = ctx.encode_bound(ns)
    payload = torch.empty(cap, dtype=torch.uint8, device="cuda")
    out = torch.empty(n * n_tracks * W + 16, dtype=torch.uint8, device="cuda")
    res = {}

    def enc():
        res["poff"], res["plen"] = ctx.encode_device(pcm.data_ptr(), bits, offs, ns, payload.data_ptr(), cap)
    te = _time(enc, reps)
    ctx.set_profiling(True); enc(); kt = dict(ctx.last_kernel_times()); ctx.set_profiling(False)
    sizes = ctx.last_chunk_sizes()
    cpt = (n + 8191) // 8192
    first = payload[torch.from_numpy(res["poff"].astype(np.int64)).cuda()].cpu().numpy()
    chunks = np.concatenate([chunks_from_sizes(sizes[t * cpt:(t + 1) * cpt], n, 8 * int(res["poff"][t]), t * n, int(first[t] >> 1) & 1)
                             for t in range(n_tracks)])

    from velox_amd.codec import HIGH_RES_FROM_PAYLOAD, TRACK_SIZES_DTYPE
    tr = np.zeros(n_tracks, dtype=TRACK_SIZES_DTYPE)
    tr["base_bit_off"] = 8 * res["poff"].astype(np.uint64); tr["out_base"] = np.arange(n_tracks, dtype=np.uint64) * n
    tr["total_samples"] = n; tr["n_sizes"] = cpt; tr["high_res"] = HIGH_RES_FROM_PAYLOAD; tr["hdr_bits"] = 2

    def dec():                                            # chunk table built on the device from the byte sizes
        ctx.decode_tracks_device(payload.data_ptr(), cap, tr, sizes, out.data_ptr(), bits)
    td = _time(dec, reps)
    chk = torch.empty_like(out)                           # the two-call form gives the same bytes
    ctx.decode_device(payload.data_ptr(), cap, chunks, chk.data_ptr(), bits)
    same = bool(torch.equal(chk[: n * n_tracks * W], out[: n * n_tracks * W]))
    del chk
    ctx.set_profiling(True); dec(); kt.update(dict(ctx.last_kernel_times())); ctx.set_profiling(False)
    ok = bool(torch.equal(out[: n * n_tracks * W].view(n_tracks, n * W), pcm[: stride * n_tracks].view(n_tracks, stride)[:, : n * W]))
    total = n * n_tracks * W
    # sampled chunks of sampled tracks against the oracle, on the payload the timed encode left in HBM
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from _sampling import check_chunks, pick
    O, _ = _oracles()
    tracks_checked = [] if QUICK else pick(n_tracks, 8, seed=5)
    per_track = 64 if n_tracks == 1 else 6
    bad, checked = [], 0
    for t in tracks_checked:
        po, hr = int(res["poff"][t]), int(first[t] >> 1) & 1
        picks = pick(cpt, per_track, seed=t)
        checked += len(picks)
        b = check_chunks(O, lambda a, bb: pcm[t * stride + a * W: t * stride + bb * W].cpu().numpy(),
                         lambda a, bb: payload[po + a: po + bb].cpu().numpy(), sizes[t * cpt:(t + 1) * cpt], n, bits, hr, picks, unpack_pcm)
        bad += [(int(t), c) for c in b]
    # ... and every chunk of every track, re-encoded by the oracle on the host threads
    tv = time.perf_counter()
    host_pcm, host_pay, every = (None, None, "skipped (--quick)") if QUICK else (pcm.cpu().numpy(), payload.cpu().numpy(), {"compared": 0, "mismatching": 0, "broken_tracks": []})
    for t in range(0 if QUICK else n_tracks):
        nc, nb, _ = O.verify_payload(host_pcm[t * stride: t * stride + n * W], bits, n, host_pay[int(res["poff"][t]): int(res["poff"][t]) + int(res["plen"][t])], _cores()[0])
        if nb < 0:
            every["broken_tracks"].append(t)
        else:
            every["compared"] += nc; every["mismatching"] += nb
    if not QUICK:
        every["of"] = int(chunks.size); every["seconds"] = round(time.perf_counter() - tv, 1)
    del host_pcm, host_pay
    comp = float(res["plen"].sum())
    sample_n = min(n, int(rate * 10) * ch // 8192 * 8192) if n_tracks == 1 else n
    sample_tracks = 1 if n_tracks == 1 else min(n_tracks, 8)
    cpu_samples = np.concatenate([unpack_pcm(pcm[t * stride: t * stride + sample_n * W].cpu().numpy(), bits) for t in range(sample_tracks)])
    print(json.dumps({"workload": name, "tracks": n_tracks, "pcm_bytes": total, "chunks": int(chunks.size),
                      "sampled_chunks_vs_oracle": {"checked": checked, "mismatching": bad}, "every_chunk_vs_oracle": every,
                      "roofline_encode": _roofline(total + comp, kt, ["k_hires_scan", "k_analyze", "k_encode_serial", "k_encode_pipe", "k_assemble"], tag),
                      "roofline_decode": _roofline(total + comp, kt, ["k_decode_streams", "k_decode_pipe", "k_decode_finalize"], tag),
                      "cpu_baseline": None if QUICK else _cpu_baseline(cpu_samples, bits, f"{sample_tracks} track(s) x {sample_n // ch / rate:.1f} s of the same data (tracks encoded back to back as one input)"),
                      "encode_mb_s": round(total / te / 1e6, 1), "decode_mb_s": round(total / td / 1e6, 1),
                      "encode_ms": round(te * 1e3, 3), "decode_ms": round(td * 1e3, 3),
                      "ratio": round(float(res["plen"].sum()) / total, 4), "round_trip_bit_exact": ok, "two_call_form_same": same,
                      "kernel_ms": {k: round(v, 3) for k, v in kt.items()}}), flush=True)
    return payload, chunks, out, pcm


def streaming(ctx, reps):
    """configs[4]: S concurrent streams each ask for their next 4096 (stream client,
    VeloxStreamClient.cpp:450) or 16384 (Qt player, VeloxQtPlayerEngine.cpp:620) samples = 1 or 2
    chunks per stream and batch; all requests of a batch go to the GPU as ONE vlx_decode_device call."""
    import torch
    from velox_amd.codec import chunks_from_sizes
    from velox_amd.synth import synth_pcm_device
    bits, ch, rate = 16, 2, 44100
    for S in (64, 1024, 16384):
        for per in (1, 2):
            n = 8192 * per
            pcm = synth_pcm_device(bits, ch, rate, (n // ch) * S, 8000.0, 150.0, 5, "cuda")
            offs = [t * n * 2 for t in range(S)]
            ns = [n] * S
            cap = ctx.encode_bound(ns)
            payload = torch.empty(cap, dtype=torch.uint8, device="cuda")
            poff, plen = ctx.encode_device(pcm.data_ptr(), bits, offs, ns, payload.data_ptr(), cap)
            sizes = ctx.last_chunk_sizes()
            chunks = np.concatenate([chunks_from_sizes(sizes[t * per:(t + 1) * per], n, 8 * int(poff[t]), t * n, 0) for t in range(S)])
            out = torch.empty(S * n * 2 + 16, dtype=torch.uint8, device="cuda")

            def dec():
                ctx.decode_device(payload.data_ptr(), cap, chunks, out.data_ptr(), bits)
            td = _time(dec, reps)
            ok = bool(torch.equal(out[: S * n * 2], pcm[: S * n * 2]))
            ctx.set_profiling(True); dec(); kt = dict(ctx.last_kernel_times()); ctx.set_profiling(False)
            # sampled streams against the oracle's decode of the same payload bytes
            O, _ = _oracles()
            from velox_amd.synth import unpack_pcm
            bad = []
            for t in sorted(set([0, S - 1] + np.random.default_rng(S + per).choice(S, size=min(S, 6), replace=False).tolist())):
                pb = payload[int(poff[t]): int(poff[t]) + int(plen[t])].cpu().numpy()
                want = O.decode(pb, n)
                got = unpack_pcm(out[t * n * 2: (t + 1) * n * 2].cpu().numpy(), bits)
                if not np.array_equal(want, got):
                    bad.append(int(t))
            comp = float(plen.sum())
            cpu = _cpu_baseline(np.concatenate([unpack_pcm(pcm[t * n * 2: (t + 1) * n * 2].cpu().numpy(), bits) for t in range(min(S, 64))]), bits,
                                f"{min(S, 64)} of the batch's streams, {n} samples each, back to back as one input")
            print(json.dumps({"workload": "cfg5 streaming decode batch", "streams": S, "samples_per_request": 4096 if per == 1 else 16384,
                              "sampled_streams_vs_oracle_decode": {"mismatching": bad},
                              "kernel_ms": {k: round(v, 3) for k, v in kt.items()},
                              "roofline": _roofline(S * n * 2 + comp, kt, ["k_decode_streams", "k_decode_pipe", "k_decode_finalize"], "cfg5"),
                              "cpu_baseline": cpu,
                              "chunks_per_batch": int(chunks.size), "batch_latency_ms": round(td * 1e3, 3),
                              "aggregate_mb_s": round(S * n * 2 / td / 1e6, 1),
                              "x_realtime_per_stream": round((n // ch) / rate / td, 1), "round_trip_bit_exact": ok}), flush=True)


def host_path(ctx, reps, lookahead=64):
    """configs[0], the reference's own CPU-runnable case, the way the CLI drives it: packed PCM in host
    memory -> vlx_encode_pcm -> payload in host memory -> StreamingDecoder pull (64 chunks per refill).
    Every call pays the PCIe copies and a synchronisation, so this is a latency figure, not the headline."""
    from velox_amd.codec import StreamingDecoder
    from velox_amd.synth import pack_pcm, synth_pcm
    s = synth_pcm(16, 2, 44100, 60.0, 8000.0, 150.0, 1234)
    pcm = pack_pcm(s, 16)
    payload = ctx.encode_pcm(pcm, 16)
    t0 = time.perf_counter()
    for _ in range(reps):
        payload = ctx.encode_pcm(pcm, 16)
    te = (time.perf_counter() - t0) / reps
    t0 = time.perf_counter()
    for _ in range(reps):
        vals, _ = StreamingDecoder(payload, s.size, ctx=ctx, lookahead=lookahead).read(s.size)
    td = (time.perf_counter() - t0) / reps
    got, n = ctx.decode_pcm(payload, s.size, 16)          # whole file -> packed PCM in one call (what the container path uses)
    t0 = time.perf_counter()
    for _ in range(reps):
        got, n = ctx.decode_pcm(payload, s.size, 16)
    tp = (time.perf_counter() - t0) / reps
    print(json.dumps({"workload": "cfg1 16-bit stereo 44.1 kHz 60 s, host-pointer entry points (PCIe + sync inclusive)", "decode_lookahead_chunks": lookahead,
                      "decode_pcm_ms": round(tp * 1e3, 2), "decode_pcm_mb_s": round(pcm.size / tp / 1e6, 1),
                      "decode_pcm_bit_exact": bool(n == s.size and np.array_equal(got, pcm)),
                      "pcm_bytes": int(pcm.size), "encode_ms": round(te * 1e3, 2), "decode_ms": round(td * 1e3, 2),
                      "encode_mb_s": round(pcm.size / te / 1e6, 1), "decode_mb_s": round(pcm.size / td / 1e6, 1),
                      "ratio": round(payload.size / pcm.size, 4), "round_trip_bit_exact": bool(np.array_equal(vals, s))}), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("what", choices=["cfg1", "cfg3", "cfg4", "cfg5"])
    ap.add_argument("--tracks", type=int, default=4096)
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--quick", action="store_true", help="GPU work only: no oracle checks, no CPU baseline (profiling passes)")
    a = ap.parse_args()
    global QUICK
    QUICK = a.quick
    from velox_amd.codec import Context
    ctx = Context(0)
    if a.what == "cfg1":
        host_path(ctx, a.reps, 64)                          # the drop-in CLI's setting
        host_path(ctx, a.reps, 1024)                        # the whole file in one refill
    elif a.what == "cfg3":
        corpus(ctx, 24, 2, 48000, 10.0, a.tracks, 2.0e6, 3000.0, a.reps, "cfg3 corpus 24-bit/48 kHz stereo 10 s tracks (seed = track index), one GPU",
               "cfg3" if a.tracks == 4096 else ("cfg3share" if a.tracks == 512 else None))
    elif a.what == "cfg4":
        corpus(ctx, 32, 8, 192000, 60.0, 1, 5.0e8, 2.0e5, a.reps, "cfg4 32-bit 8-channel 192 kHz 60 s", "cfg4")
    else:
        streaming(ctx, a.reps)


if __name__ == "__main__":
    main()
