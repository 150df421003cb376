#!/usr/bin/env python3
"""k_encode_serial (the throughput shape) against the number of waves per SIMD: is a SIMD with two resident waves as
fast per wave-instruction as one with three?  One JSON line per batch size (streams = 64 x waves).

    python tools/encode_fill.py [--bits 24] [waves_per_simd_x100 ...]
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
    ap.add_argument("fill", nargs="*", type=int, default=[100, 150, 200, 250, 257, 300, 350, 400, 600])
    ap.add_argument("--bits", type=int, default=24)
    ap.add_argument("--reps", type=int, default=4)
    ap.add_argument("--shape", default="stream")
    a = ap.parse_args()
    import torch
    from velox_amd.codec import Context, set_encode_shape
    from velox_amd.synth import synth_pcm_device
    ctx = Context(0)
    bits = a.bits
    amp, sigma = {24: (2.0e6, 3000.0), 16: (8000.0, 150.0), 32: (5.0e8, 2.0e5)}[bits]
    set_encode_shape(a.shape)
    ctx.set_profiling(True)
    for f in a.fill:
        waves = 1024 * f // 100
        nc = waves * 32                                   # 64 streams per wave = 32 chunks
        n = nc * 8192
        pcm = synth_pcm_device(bits, 2, 96000, n // 2, amp, sigma, 3, "cuda")
        cap = ctx.encode_bound([n])
        payload = torch.empty(cap, dtype=torch.uint8, device="cuda")
        ts = []
        for _ in range(a.reps + 1):
            ctx.encode_device(pcm.data_ptr(), bits, [0], [n], payload.data_ptr(), cap)
            kt = dict(ctx.last_kernel_times())
            ts.append(kt.get("k_encode_serial", kt.get("k_encode_pipe")))
        ms = float(np.mean(ts[1:]))
        print(json.dumps({"waves_per_simd": f / 100, "waves": waves, "chunks": nc, "kernel_ms": round(ms, 4),
                          "ms_per_wave_per_simd": round(ms / (f / 100), 4)}), flush=True)
        del pcm, payload
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
