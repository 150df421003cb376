#!/usr/bin/env python3
"""k_encode_serial's store path: direct 4-byte stores (G = 0) against the per-lane LDS ring flushed 16 / 32
bytes at a time (G = 4 / 8), on cfg2-shaped material.  Prints one JSON line: kernel time per variant and whether
the payload bytes are the same.  With --store G only that variant runs (for a rocprofv3 --pmc pass:
PROF_PROG="python3 tools/bench_encode_store.py --store 8 --reps 0" tools/prof_pmc.sh store8).

    python tools/bench_encode_store.py [--bits 24] [--store G] [chunks]
"""
import argparse
import ctypes
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("chunks", nargs="?", type=int, default=84375)
    ap.add_argument("--bits", type=int, default=24)
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--store", type=int, default=None)
    ap.add_argument("--order", default="0,4,8", help="variants to time, in this order")
    a = ap.parse_args()
    import torch
    from velox_amd import _lib
    from velox_amd.codec import Context, set_encode_shape
    from velox_amd.synth import synth_pcm_device
    L = _lib.load()
    L.vlx_debug_set_encode_store.argtypes = [ctypes.c_int]
    L.vlx_debug_set_encode_store.restype = None
    ctx = Context(0)
    bits = a.bits
    amp, sigma = {24: (2.0e6, 3000.0), 16: (8000.0, 150.0), 32: (5.0e8, 2.0e5)}[bits]
    n = a.chunks * 8192
    pcm = synth_pcm_device(bits, 2, 96000, n // 2, amp, sigma, 3, "cuda")
    cap = ctx.encode_bound([n])
    payload = torch.empty(cap, dtype=torch.uint8, device="cuda")
    set_encode_shape("stream")
    row = {"chunks": a.chunks, "bits": bits}
    ctx.set_profiling(True)
    ref = None
    for g in ([a.store] if a.store is not None else [int(x) for x in a.order.split(",")]):
        L.vlx_debug_set_encode_store(g)
        ts = []
        for _ in range(a.reps + 1):
            poff, plen = ctx.encode_device(pcm.data_ptr(), bits, [0], [n], payload.data_ptr(), cap)
            kt = dict(ctx.last_kernel_times()); ts.append(kt.get("k_encode_pipe", kt.get("k_encode_serial")))
        got = payload[: int(plen[0])].clone()
        if ref is None:
            ref = got
        row[f"g{g}_ms"] = round(float(np.mean(ts[1:] or ts)), 4)
        row[f"g{g}_same_bytes"] = bool(got.numel() == ref.numel() and torch.equal(got, ref))
        row["payload_bytes"] = int(plen[0])
    print(json.dumps(row), flush=True)


if __name__ == "__main__":
    main()
