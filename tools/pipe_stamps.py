#!/usr/bin/env python3
"""Where the pipelined small-batch decode kernel (vlx_pipe.hip) spends its time: s_memtime stamps of the
five waves of workgroup 0 (diagnostic hook vlx_debug_set_stamp_buffer; 100 MHz ticks -> microseconds).

    python tools/pipe_stamps.py [--chunks 64] [--bits 24]
"""
import argparse
import ctypes as C
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--chunks", type=int, default=64)
    ap.add_argument("--bits", type=int, default=24)
    ap.add_argument("--shape", default="pipe64", choices=["pipe16", "pipe32", "pipe64"])
    ap.add_argument("--channels", type=int, default=2)
    ap.add_argument("--waits", action="store_true", help="also print the wait accumulators of a -DDP_PROFILE build")
    a = ap.parse_args()
    import torch
    from velox_amd import _lib
    from velox_amd.codec import Context, chunks_from_sizes, set_decode_shape
    from velox_amd.synth import synth_pcm_device
    ctx = Context(0)
    L = _lib.load()
    bits, W = a.bits, a.bits // 8
    amp, sigma = {24: (2.0e6, 3000.0), 16: (8000.0, 150.0), 32: (5.0e8, 2.0e5)}[bits]
    n = a.chunks * 8192
    pcm = synth_pcm_device(bits, a.channels, 96000, n // a.channels, amp, sigma, 3, "cuda")
    cap = ctx.encode_bound([n])
    payload = torch.empty(cap, dtype=torch.uint8, device="cuda")
    out = torch.empty(n * W + 16, dtype=torch.uint8, device="cuda")
    poff, plen = ctx.encode_device(pcm.data_ptr(), bits, [0], [n], payload.data_ptr(), cap)
    hires = int(payload[int(poff[0])].item() >> 1) & 1
    chunks = chunks_from_sizes(ctx.last_chunk_sizes(), n, 8 * int(poff[0]), 0, hires)
    stamps = torch.zeros(32, dtype=torch.int64, device="cuda"); torch.cuda.synchronize()
    L.vlx_debug_set_stamp_buffer.argtypes = [C.c_void_p]
    L.vlx_debug_set_stamp_buffer.restype = None
    set_decode_shape(a.shape)
    ctx.set_profiling(True)
    for _ in range(3):
        ctx.decode_device(payload.data_ptr(), cap, chunks, out.data_ptr(), bits)
    L.vlx_debug_set_stamp_buffer(C.c_void_p(stamps.data_ptr()))
    ctx.decode_device(payload.data_ptr(), cap, chunks, out.data_ptr(), bits)
    L.vlx_debug_set_stamp_buffer(None)
    kt = dict(ctx.last_kernel_times())
    s = stamps.cpu().numpy().reshape(8, 4)
    t0 = int(s[:5, 0].min())
    names = ["XL", "P", "ML", "MR", "XR"]
    row = {"chunks": a.chunks, "bits": bits, "shape": a.shape, "kernel_ms": round(kt.get("k_decode_pipe", kt.get("k_decode_streams")), 4),
           "ok": bool(torch.equal(out[: n * W], pcm[: n * W]))}
    for r, nm in enumerate(names):
        row[nm] = {"start_us": (int(s[r, 0]) - t0) / 100.0, "end_us": (int(s[r, 1]) - t0) / 100.0}
    row["P"]["end_of_L_us"] = (int(s[1, 2]) - t0) / 100.0
    if a.waits:                                            # a -DDP_PROFILE build (VELOX_AMD_LIB): cycles / 100 spent waiting
        flat = s.reshape(-1)
        for r, nm in enumerate(names):
            row[nm].update({"wait_producer": int(s[r, 3]) / 100.0, "publish_drain": int(flat[20 + r]) / 100.0, "load_wait": int(flat[25 + r]) / 100.0})
    print(json.dumps(row), flush=True)


if __name__ == "__main__":
    main()
