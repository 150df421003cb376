import sys, os, json
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from velox_amd.codec import Context, set_decode_shape, chunks_from_sizes
from velox_amd.synth import synth_pcm_device
ctx = Context(0)
def run(ch, rate, seconds, amp, sigma, seed, shape):
    bits, W = 32, 4
    frames = int(rate * seconds); n = frames * ch
    pcm = torch.zeros(n * W + 32, dtype=torch.uint8, device="cuda"); torch.cuda.synchronize()
    pcm[: n * W] = synth_pcm_device(bits, ch, rate, frames, amp, sigma, seed, "cuda"); torch.cuda.synchronize()
    cap = ctx.encode_bound([n]); payload = torch.empty(cap, dtype=torch.uint8, device="cuda")
    out = torch.empty(n * W + 16, dtype=torch.uint8, device="cuda")
    poff, plen = ctx.encode_device(pcm.data_ptr(), bits, [0], [n], payload.data_ptr(), cap)
    sizes = ctx.last_chunk_sizes()
    hires = int(payload[int(poff[0])].item() >> 1) & 1
    chunks = chunks_from_sizes(sizes, n, 8 * int(poff[0]), 0, hires)
    set_decode_shape(shape); ctx.set_profiling(True)
    for _ in range(2): ctx.decode_device(payload.data_ptr(), cap, chunks, out.data_ptr(), bits)
    kt = dict(ctx.last_kernel_times()); ctx.set_profiling(False)
    ok = bool(torch.equal(out[: n * W], pcm[: n * W]))
    print(dict(ch=ch, rate=rate, amp=amp, sigma=sigma, seed=seed, shape=shape, chunks=len(chunks), ratio=round(float(plen[0]) / (n * W), 3), hires=hires),
          {k: round(v, 3) for k, v in kt.items() if "decode" in k}, ok, flush=True)
for args in [(2, 96000, 480.0, 5.0e8, 2.0e5, 3), (8, 192000, 60.0, 5.0e8, 2.0e5, 4), (2, 192000, 240.0, 5.0e8, 2.0e5, 4), (8, 192000, 60.0, 5.0e8, 2.0e5, 3), (8, 96000, 120.0, 5.0e8, 2.0e5, 3)]:
    run(*args, "pipe64")
