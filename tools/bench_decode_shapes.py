#!/usr/bin/env python3
"""k_decode_streams time against batch size for the two wave shapes and the launcher's own mix
(calibration of vlx_chunk_wave_count's cost model).  One JSON line per batch size.

    python tools/bench_decode_shapes.py [--bits 24] [chunks ...]
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
    ap.add_argument("chunks", nargs="*", type=int, default=[8192, 16384, 32768, 40000, 49152, 65536, 84375, 98304, 131072])
    ap.add_argument("--bits", type=int, default=24)
    ap.add_argument("--reps", type=int, default=3)
    a = ap.parse_args()
    import torch
    from velox_amd.codec import Context, chunks_from_sizes, set_decode_shape
    from velox_amd.synth import synth_pcm_device
    ctx = Context(0)
    bits, W = a.bits, a.bits // 8
    amp, sigma = (2.0e6, 3000.0) if bits == 24 else (8000.0, 150.0)
    for nc in a.chunks:
        n = nc * 8192
        pcm = synth_pcm_device(bits, 2, 96000, n // 2, amp, sigma, 3, "cuda")
        cap = ctx.encode_bound([n])
        payload = torch.empty(cap, dtype=torch.uint8, device="cuda")
        out = torch.empty(n * W + 16, dtype=torch.uint8, device="cuda")
        poff, plen = ctx.encode_device(pcm.data_ptr(), bits, [0], [n], payload.data_ptr(), cap)
        hires = int(payload[int(poff[0])].item() >> 1) & 1
        chunks = chunks_from_sizes(ctx.last_chunk_sizes(), n, 8 * int(poff[0]), 0, hires)
        row = {"chunks": nc, "chunks_per_simd": round(nc / 1024, 1)}
        ctx.set_profiling(True)
        for shape in ("stream", "chunk", "pipe16", "pipe32", "pipe64", "auto"):
            set_decode_shape(shape)
            ts, fs = [], []
            for _ in range(a.reps + 1):
                ctx.decode_device(payload.data_ptr(), cap, chunks, out.data_ptr(), bits)
                kt = dict(ctx.last_kernel_times())
                ts.append(kt.get("k_decode_pipe", kt.get("k_decode_streams")))
                fs.append(kt["k_decode_finalize"])
            row[shape + "_ms"] = round(float(np.mean(ts[1:])), 4)
            row[shape + "_finalize_ms"] = round(float(np.mean(fs[1:])), 4)       # ~0 when the stream waves stored the PCM themselves
            row[shape + "_ok"] = bool(torch.equal(out[: n * W], pcm[: n * W]))
        ctx.set_profiling(False)
        set_decode_shape("auto")
        print(json.dumps(row), flush=True)
        del pcm, payload, out


if __name__ == "__main__":
    main()
