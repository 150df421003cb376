#!/usr/bin/env python3
"""k_encode_serial time against batch size for the throughput shape and the two pipelined small-batch
shapes (vlx_set_encode_shape), and what the launcher picks.  One JSON line per batch size.

    python tools/bench_encode_shapes.py [--bits 24] [chunks ...]
"""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("chunks", nargs="*", type=int, default=[64, 1024, 4096, 8192, 11250, 16384, 32768, 84375])
    ap.add_argument("--bits", type=int, default=24)
    ap.add_argument("--reps", type=int, default=3)
    a = ap.parse_args()
    import torch
    from velox_amd.codec import Context, set_encode_shape
    from velox_amd.synth import synth_pcm_device
    ctx = Context(0)
    bits = a.bits
    amp, sigma = {24: (2.0e6, 3000.0), 16: (8000.0, 150.0), 32: (5.0e8, 2.0e5)}[bits]
    for nc in a.chunks:
        n = nc * 8192
        pcm = synth_pcm_device(bits, 2, 96000, n // 2, amp, sigma, 3, "cuda")
        cap = ctx.encode_bound([n])
        payload = torch.empty(cap, dtype=torch.uint8, device="cuda")
        row = {"chunks": nc, "streams_per_simd": round(2 * nc / 1024, 1)}
        ctx.set_profiling(True)
        ref = None
        for shape in ("stream", "pipe16", "pipe32", "pipe64", "auto"):
            set_encode_shape(shape)
            ts = []
            for _ in range(a.reps + 1):
                poff, plen = ctx.encode_device(pcm.data_ptr(), bits, [0], [n], payload.data_ptr(), cap)
                kt = dict(ctx.last_kernel_times()); ts.append(kt.get("k_encode_pipe", kt.get("k_encode_serial")))
            got = payload[: int(plen[0])].clone()
            if ref is None:
                ref = got
            row[shape + "_ms"] = round(float(np.mean(ts[1:])), 4)
            row[shape + "_same_bytes"] = bool(got.numel() == ref.numel() and torch.equal(got, ref))
        ctx.set_profiling(False)
        set_encode_shape("auto")
        print(json.dumps(row), flush=True)
        del pcm, payload


if __name__ == "__main__":
    main()
