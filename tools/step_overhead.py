#!/usr/bin/env python3
"""What the per-kernel HIP events of vlx_set_profiling cost a cfg2 step: wall time of encode + decode over a few
steps with the events on and off.  One JSON line."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    from velox_amd.codec import HIGH_RES_FROM_PAYLOAD, TRACK_SIZES_DTYPE, Context
    from velox_amd.synth import synth_pcm_device
    ctx = Context(0)
    bits, n = 24, 84375 * 8192
    pcm = synth_pcm_device(bits, 2, 96000, n // 2, 2.0e6, 3000.0, 3, "cuda")
    cap = ctx.encode_bound([n])
    payload = torch.empty(cap, dtype=torch.uint8, device="cuda")
    out = torch.empty(n * 3 + 16, dtype=torch.uint8, device="cuda")
    track = np.zeros(1, dtype=TRACK_SIZES_DTYPE)

    def step():
        poff, plen = ctx.encode_device(pcm.data_ptr(), bits, [0], [n], payload.data_ptr(), cap)
        sizes = ctx.last_chunk_sizes()
        track[0] = (8 * int(poff[0]), 0, n, sizes.size, HIGH_RES_FROM_PAYLOAD, 2, 0)
        ctx.decode_tracks_device(payload.data_ptr(), cap, track, sizes, out.data_ptr(), bits)

    row = {}
    for rep in range(2):
        for mode in (True, False):
            ctx.set_profiling(mode)
            step(); step()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(8):
                step()
            torch.cuda.synchronize()
            row.setdefault("events_on_ms" if mode else "events_off_ms", []).append(round((time.perf_counter() - t0) / 8 * 1e3, 3))
    print(json.dumps(row))


if __name__ == "__main__":
    main()
