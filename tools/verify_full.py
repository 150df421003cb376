#!/usr/bin/env python3
"""One exhaustive full-size comparison per round: EVERY chunk of the device payload of a
BASELINE config, at its real size, against the CPU oracle's encoding of the same sub-block — not a sample.

    python tools/verify_full.py cfg2      24-bit stereo 96 kHz, one hour: 84,375 chunks
    python tools/verify_full.py cfg4      32-bit 8-channel 192 kHz, 60 s: 11,250 chunks
    python tools/verify_full.py cfg3      the share of configs[2] one GPU of eight gets: 512 tracks, 60,416 chunks
    python tools/verify_full.py all [--out profiles/round3_verify_full.json]

The GPU encodes the batch with vlx_encode_device (the timed path of bench.py); PCM and payload come down to the host;
oracle/velox_oracle.c:vlxo_verify_payload walks every track's size-prefixed chunk list and re-encodes every sub-block
on the host cores (test infrastructure: the oracle is only ever the checker).  A decode of the same payload back to
the input bytes rides along.  Prints one JSON line per config: chunks compared, mismatches, seconds, and the hash of
the kernel sources the payload came from (tools/pmc_summary.py's stamp), and appends them to --out."""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def host_threads():
    """Threads for the checker: the cores this process may run on, at most 16 on a shared GPU box (its CPU share)."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    return max(1, min(n, int(os.environ.get("VELOX_VERIFY_THREADS", "16"))))


def verify(name, ctx, oracle, pcm, bits, offs, ns, stride_bytes):
    """pcm: device uint8 tensor with the tracks at byte offsets `offs`, `ns` samples each."""
    import torch
    from velox_amd.codec import HIGH_RES_FROM_PAYLOAD, TRACK_SIZES_DTYPE
    W = bits // 8
    cap = ctx.encode_bound(ns)
    out = torch.empty(cap, dtype=torch.uint8, device=pcm.device)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    poff, plen = ctx.encode_device(pcm.data_ptr(), bits, offs, ns, out.data_ptr(), cap)
    enc_ms = (time.perf_counter() - t0) * 1e3
    sizes = ctx.last_chunk_sizes().copy()
    # decode everything back first (device-side compare)
    tr = np.zeros(len(ns), dtype=TRACK_SIZES_DTYPE)
    cpt = [(n + 8191) // 8192 for n in ns]
    tr["base_bit_off"] = 8 * poff.astype(np.uint64)
    tr["out_base"] = np.concatenate(([0], np.cumsum(ns)[:-1])).astype(np.uint64)
    tr["total_samples"] = ns; tr["n_sizes"] = cpt; tr["high_res"] = HIGH_RES_FROM_PAYLOAD; tr["hdr_bits"] = 2
    dst = torch.empty(sum(ns) * W + 16, dtype=torch.uint8, device=pcm.device)
    torch.cuda.synchronize()
    ctx.decode_tracks_device(out.data_ptr(), cap, tr, sizes, dst.data_ptr(), bits)
    rt = True
    pos = 0
    for t, n in enumerate(ns):
        rt = rt and bool(torch.equal(dst[pos: pos + n * W], pcm[offs[t]: offs[t] + n * W]))
        pos += n * W
    del dst
    host_pcm = pcm.cpu().numpy()
    host_out = out.cpu().numpy()
    threads = host_threads()
    t0 = time.perf_counter()
    chunks = bad = 0
    broken, first = [], None
    for t, n in enumerate(ns):
        nc, b, fb = oracle.verify_payload(host_pcm[offs[t]: offs[t] + n * W], bits, n, host_out[int(poff[t]): int(poff[t]) + int(plen[t])], threads)
        if b < 0:
            broken.append(t)
            continue
        chunks += nc
        bad += b
        if b and first is None:
            first = (t, fb)
    secs = time.perf_counter() - t0
    from bench import csrc_hash
    res = {"config": name, "chunks_compared": chunks, "chunks_expected": int(sum(cpt)), "mismatching_chunks": bad,
           "tracks": len(ns), "tracks_with_a_broken_chunk_list": broken, "first_mismatch_track_chunk": first,
           "decode_round_trip_bit_exact": rt, "checker": "oracle/velox_oracle.c:vlxo_verify_payload (every chunk re-encoded on the host)",
           "checker_threads": threads, "checker_seconds": round(secs, 1), "pcm_bytes": int(sum(ns)) * W, "payload_bytes": int(plen.sum()),
           "gpu_encode_call_ms": round(enc_ms, 2), "csrc_sha256": csrc_hash()}
    res["verdict"] = (f"{chunks} of {int(sum(cpt))} chunks equal" if not bad and not broken and chunks == sum(cpt) else "MISMATCH") + \
                     (", decode returns the input" if rt else ", DECODE DIFFERS")
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("what", choices=["cfg2", "cfg3", "cfg4", "all"])
    ap.add_argument("--out", default=None, help="append the JSON lines to this file (e.g. profiles/round3_verify_full.json)")
    ap.add_argument("--seconds", type=float, default=None, help="cfg2 / cfg4: shorter track (rehearsal)")
    args = ap.parse_args()
    import torch
    from _oracle import Oracle
    from velox_amd.codec import Context
    from velox_amd.synth import synth_corpus_device, synth_pcm_device
    ctx, oracle = Context(0), Oracle()
    lines = []
    for what in (["cfg2", "cfg4", "cfg3"] if args.what == "all" else [args.what]):
        if what == "cfg2":
            frames = int(96000 * (args.seconds or 3600.0))
            pcm = synth_pcm_device(24, 2, 96000, frames, 2.0e6, 3000.0, 2, "cuda")
            res = verify("configs[1]: 24-bit stereo 96 kHz, %g s" % (frames / 96000), ctx, oracle, pcm, 24, [0], [frames * 2], 0)
        elif what == "cfg4":
            frames = int(192000 * (args.seconds or 60.0))
            pcm = synth_pcm_device(32, 8, 192000, frames, 5.0e8, 2.0e5, 4, "cuda")
            res = verify("configs[3]: 32-bit 8-channel 192 kHz, %g s" % (frames / 192000), ctx, oracle, pcm, 32, [0], [frames * 8], 0)
        else:
            T, frames = 512, 480000
            n = frames * 2
            stride = (n * 3 + 15) // 16 * 16
            pcm = synth_corpus_device(24, 2, 48000, frames, T, 2.0e6, 3000.0, "cuda", stride=stride)
            res = verify("configs[2], one GPU's share: 512 tracks of 24-bit/48 kHz stereo, 10 s each", ctx, oracle, pcm, 24,
                         [t * stride for t in range(T)], [n] * T, stride)
        del pcm
        torch.cuda.empty_cache()
        ctx.trim()
        print(json.dumps(res), flush=True)
        lines.append(res)
    if args.out:
        with open(os.path.join(ROOT, args.out) if not os.path.isabs(args.out) else args.out, "a") as fh:
            for r in lines:
                fh.write(json.dumps(r) + "\n")
    return 0 if all(r["verdict"].startswith(str(r["chunks_expected"])) and r["decode_round_trip_bit_exact"] for r in lines) else 1


if __name__ == "__main__":
    sys.exit(main())
