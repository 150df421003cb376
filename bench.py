#!/usr/bin/env python3
"""Headline benchmark: PCM MB/s encode + decode of the Velox frame-codec hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W

One "step" = one pass of the hot path over one batch of synthetic PCM that is already resident in
HBM: encode the whole track with vlx_encode_device, then decode the produced payload back to packed
PCM with vlx_decode_device.  The workload at N=1 is BASELINE.json configs[1]: 24-bit stereo 96 kHz,
1 h (2.0736 GB of PCM, 84,375 sub-blocks).  With N>1 the track is N hours long and rank r holds
hour r (weak scaling: contiguous runs of whole sub-blocks per GPU, velox_amd/sharded.py): the timed
step then contains the path's real exchange — the 1-int MAX all-reduce of high_res and the
all-gather of per-chunk byte counts that gives every rank its place in the global offset table —
and each rank decodes its own shard.  The compressed bytes stay sharded in HBM; gathering them onto
one GPU is timed once, separately (`gather_to_root_ms`), because nothing on the path consumes it.

Prints ONE JSON line on rank 0.  `value` = PCM bytes through encode AND decode per second of the
timed region's wall clock (K steps between two barrier + synchronize brackets, max over ranks),
whole job — so it includes the host glue between the two calls (chunk sizes read back, track table
filled in, `host_glue_ms_per_step`); the rates of the encode and decode calls alone ride along.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def csrc_hash():
    """sha256 over the kernel sources, as tools/pmc_summary.py stamps it into a profile."""
    import hashlib
    h = hashlib.sha256()
    d = os.path.join(ROOT, "velox_amd", "csrc")
    for name in sorted(os.listdir(d)):
        if name.endswith((".hip", ".h", ".cpp")):
            h.update(name.encode())
            with open(os.path.join(d, name), "rb") as fh:
                h.update(fh.read())
    return h.hexdigest()


def load_profile():
    """The newest committed PMC profile of this workload (profiles/round*_pmc.json, made by tools/prof_pmc.sh),
    but only if it was taken from the kernel sources this run uses: the file carries a hash of
    velox_amd/csrc, and counters from other sources are not quoted (traffic / valu_busy come out null).
    Returns (table or None, file name, reason when None)."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "round*_pmc.json")))
    if not files:
        return None, None, "no committed profile"
    f = files[-1]
    try:
        with open(f) as fh:
            prof = json.load(fh)
    except (OSError, ValueError):
        return None, os.path.basename(f), "unreadable profile"
    meta = prof.get("_meta") or {}
    if meta.get("csrc_sha256") != csrc_hash():
        return None, os.path.basename(f), "profile taken from other kernel sources (hash of velox_amd/csrc differs): re-run tools/prof_pmc.sh"
    return prof, os.path.basename(f), None


def pmc_traffic(prof, kernel):
    """HBM bytes per launch of `kernel`: 2 x FETCH_SIZE (gfx950 tallies 128-B read requests at 64 B,
    MI355X_MICROARCH.md §HBM; checked here on k_decode_finalize / k_analyze, whose byte counts are
    known) + WRITE_SIZE, both in KiB.  None when the profile has no such kernel."""
    for name, c in (prof or {}).items():
        if name.split("<")[0] == kernel and "FETCH_SIZE" in c and "WRITE_SIZE" in c:
            return int((2.0 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024)
    return None


def pmc_valu_busy(prof, kernel):
    """VALUBusy of `kernel` from the same profile: SQ_ACTIVE_INST_VALU x 4 cycles / (1024 SIMDs x
    GRBM_GUI_ACTIVE per XCD) — the fraction of SIMD issue time spent on vector ALU instructions."""
    for name, c in (prof or {}).items():
        if name.split("<")[0] == kernel and c.get("GRBM_GUI_ACTIVE") and "SQ_ACTIVE_INST_VALU" in c:
            return round(c["SQ_ACTIVE_INST_VALU"] * 4.0 / 1024.0 / (c["GRBM_GUI_ACTIVE"] / 8.0), 3)
    return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--seconds", type=float, default=3600.0, help="track duration (default: the 1 h of configs[1])")
    ap.add_argument("--bits", type=int, default=24)
    ap.add_argument("--rate", type=int, default=96000)
    ap.add_argument("--cpu-sample-seconds", type=float, default=60.0)
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--decode-shape", choices=["auto", "stream", "chunk"], default="auto", help="tuning aid: force one wave shape of the fast decode kernel")
    ap.add_argument("--workload", choices=["track", "corpus"], default="track",
                    help="track = configs[1] (one track, N hours on N GPUs, sub-block runs per rank); corpus = configs[2] (whole 24-bit/48 kHz tracks per rank)")
    ap.add_argument("--tracks-per-gpu", type=int, default=512, help="corpus: tracks per rank (512 x 8 GPUs = the 4096 tracks of configs[2])")
    ap.add_argument("--track-seconds", type=float, default=10.0, help="corpus: seconds per track")
    ap.add_argument("--gather", action="store_true", help="track, N > 1: also gather the whole payload onto rank 0 inside every timed step")
    ap.add_argument("--exchange", choices=["native", "torch"], default="native",
                    help="N > 1: who runs the exchange steps: the library itself over RCCL (vlx_node_*) or torch.distributed (velox_amd/sharded.py)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    # VELOX_DIST_BACKEND=gloo is a rehearsal aid (several ranks on ONE GPU, small collectives on the CPU);
    # the driver's multi-GPU run uses nccl = RCCL, one rank per GPU.
    backend = os.environ.get("VELOX_DIST_BACKEND", "nccl")
    local = local % max(torch.cuda.device_count(), 1)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    coll_dev = dev if backend == "nccl" else torch.device("cpu")

    from velox_amd import sharded
    from velox_amd.codec import Context
    from velox_amd.synth import synth_pcm_device

    if args.workload == "corpus":
        return corpus_main(args, rank, world, local, dev, coll_dev, backend)
    bits, W, ch = args.bits, args.bits // 8, 2
    frames = int(round(args.rate * args.seconds))
    n = frames * ch
    amp, sigma = (2.0e6, 3000.0) if bits == 24 else ((8000.0, 150.0) if bits == 16 else (5.0e8, 2.0e5))
    pcm = synth_pcm_device(bits, ch, args.rate, frames, amp, sigma, 2 + rank, dev)
    ctx = Context(local)
    if args.decode_shape != "auto":
        from velox_amd.codec import set_decode_shape
        set_decode_shape(args.decode_shape)
    cap = ctx.encode_bound([n])
    payload = torch.empty(cap, dtype=torch.uint8, device=dev)
    out = torch.empty(n * W + 16, dtype=torch.uint8, device=dev)
    pcm_bytes = n * W

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    ktimes = {}
    from velox_amd.codec import HIGH_RES_FROM_PAYLOAD, TRACK_SIZES_DTYPE
    track = np.zeros(1, dtype=TRACK_SIZES_DTYPE)                         # the chunk table itself is built on the device

    # VELOX_FORCE_NODE=1: rehearse the N > 1 code path (native exchange, gather) with a world of one rank on a one-GPU box
    force_node = world == 1 and bool(os.environ.get("VELOX_FORCE_NODE"))
    shard = sharded.GpuShard(ctx, pcm, bits, n) if (world > 1 or force_node) else None
    last_sp = [None]
    last_enc = [None]                                                    # (payload offset, length, chunk sizes) of the last timed encode
    node, exchange = None, "none"
    if world > 1:
        node, exchange = make_node(args, ctx, rank, world, backend, coll_dev)
    elif force_node:
        node, exchange = sharded.NativeNode(ctx, sharded.NativeNode.unique_id(), 0, 1), "native (forced, world of 1)"
        if not dist.is_initialized():                                     # the self-check below compares with the torch.distributed exchange
            dist.init_process_group("nccl", init_method="tcp://127.0.0.1:29537", rank=0, world_size=1, device_id=torch.device("cuda", local))
    max_chunks = (n + 8191) // 8192
    full = None
    last_native = [None]
    self_check = None
    if node is not None and (world > 1 or force_node):
        # the native exchange has only ever run with one rank before this job: check it, untimed, against the
        # torch.distributed exchange of the same shards, and fall back to that one if anything differs
        why = native_self_check(node, shard, pcm, bits, n, max_chunks, rank, world, coll_dev, dev, backend == "nccl")
        self_check = "failed: " + why if why else "passed"
        if why:
            print(f"[bench] rank {rank}: native exchange failed its self-check ({why}); using torch.distributed", file=sys.stderr, flush=True)
            node, exchange = None, "torch.distributed (nccl); the native layer failed its self-check: " + why
        else:
            exchange += "; checked against the torch.distributed exchange before timing (shard bytes, offsets, size table, gathered payload)"

    def step_native(record):
        nonlocal full
        t0 = time.perf_counter()
        ln, off, total, hr = node.encode_track(pcm.data_ptr(), bits, n, shard.payload.data_ptr(), shard.cap, max_chunks)
        if args.gather:
            if rank == 0 and (full is None or full.numel() < total):
                full = torch.empty(total + 16, dtype=torch.uint8, device=dev); torch.cuda.synchronize()
            node.gather_payload(0, shard.payload.data_ptr(), ln, full.data_ptr() if rank == 0 else 0, total + 16 if rank == 0 else 0)
        t1 = time.perf_counter()
        if record:
            for k, ms in ctx.last_kernel_times():
                ktimes.setdefault(k, []).append(ms)
        sizes = node.chunk_sizes()
        track[0] = (0, 0, n, sizes.size, hr, 2, 0)
        t2 = time.perf_counter()
        ctx.decode_tracks_device(shard.payload.data_ptr(), shard.cap, track, sizes, out.data_ptr(), bits)
        t3 = time.perf_counter()
        if record:
            for k, ms in ctx.last_kernel_times():
                ktimes.setdefault(k, []).append(ms)
        last_native[0] = (ln, off, total, hr)
        return (t1 - t0), (t3 - t2), ln

    def step_sharded(record):
        t0 = time.perf_counter()
        sp = sharded.encode_track_sharded(shard.scan, shard.encode, coll_dev)
        t1 = time.perf_counter()
        if record:
            for k, ms in ctx.last_kernel_times():
                ktimes.setdefault(k, []).append(ms)
        track[0] = (0, 0, n, sp.local_sizes.size, sp.high_res, 2, 0)
        t2 = time.perf_counter()
        ctx.decode_tracks_device(shard.payload.data_ptr(), shard.cap, track, sp.local_sizes, out.data_ptr(), bits)
        t3 = time.perf_counter()
        if record:
            for k, ms in ctx.last_kernel_times():
                ktimes.setdefault(k, []).append(ms)
        last_sp[0] = sp
        return (t1 - t0), (t3 - t2), int(sp.seg.numel())

    def step(record):
        if world > 1 or force_node:
            return step_native(record) if node is not None else step_sharded(record)
        t0 = time.perf_counter()
        poff, plen = ctx.encode_device(pcm.data_ptr(), bits, [0], [n], payload.data_ptr(), cap)
        t1 = time.perf_counter()
        if record:
            for k, ms in ctx.last_kernel_times():
                ktimes.setdefault(k, []).append(ms)
        sizes = ctx.last_chunk_sizes()
        last_enc[0] = (int(poff[0]), int(plen[0]), sizes)
        track[0] = (8 * int(poff[0]), 0, n, sizes.size, HIGH_RES_FROM_PAYLOAD, 2, 0)
        t2 = time.perf_counter()
        ctx.decode_tracks_device(payload.data_ptr(), cap, track, sizes, out.data_ptr(), bits)
        t3 = time.perf_counter()
        if record:
            for k, ms in ctx.last_kernel_times():
                ktimes.setdefault(k, []).append(ms)
        return (t1 - t0), (t3 - t2), int(plen[0])

    ctx.set_profiling(True)
    for _ in range(args.warmup):
        step(False)
    barrier()
    t_start = time.perf_counter()
    enc_t = dec_t = 0.0
    plen = 0
    for _ in range(args.steps):
        e, d, plen = step(True)
        enc_t += e
        dec_t += d
    barrier()
    wall = time.perf_counter() - t_start
    ok = bool(torch.equal(out[:pcm_bytes], pcm[:pcm_bytes]))
    gather_ms, total_len = None, None
    if node is not None:
        total_len = last_native[0][2]
        if not args.gather:                             # untimed extra: the whole payload onto rank 0, once
            ln, _, total, _ = last_native[0]
            if rank == 0:
                full = torch.empty(total + 16, dtype=torch.uint8, device=dev); torch.cuda.synchronize()
            barrier()
            g0 = time.perf_counter()
            node.gather_payload(0, shard.payload.data_ptr(), ln, full.data_ptr() if rank == 0 else 0, total + 16 if rank == 0 else 0)
            barrier()
            gather_ms = (time.perf_counter() - g0) * 1e3
        if rank == 0:                                   # the gathered payload starts with rank 0's own shard, and its last shard ends it
            ln0 = last_native[0][0]
            ok = ok and bool(torch.equal(full[1: ln0 - 1], shard.payload[1: ln0 - 1]))
    elif world > 1 and backend == "nccl":               # torch.distributed exchange: untimed gather
        barrier()
        g0 = time.perf_counter()
        full = sharded.gather_payload(last_sp[0], root=0)
        barrier()
        gather_ms = (time.perf_counter() - g0) * 1e3
        total_len = last_sp[0].total_len
        if rank == 0:
            ok = ok and int(full.numel()) == last_sp[0].total_len

    # max over ranks
    stats = torch.tensor([enc_t, dec_t, wall, wall - enc_t - dec_t], dtype=torch.float64, device=coll_dev)
    if world > 1:
        dist.all_reduce(stats, op=dist.ReduceOp.MAX)
        okt = torch.tensor([1 if ok else 0], device=coll_dev)
        dist.all_reduce(okt, op=dist.ReduceOp.MIN)
        ok = bool(okt.item())
    enc_t, dec_t, wall, glue_t = [float(x) for x in stats.tolist()]     # each the slowest rank's

    if rank == 0:
        K = args.steps
        total_pcm = pcm_bytes * world
        mbs = lambda t: total_pcm * K / t / 1e6
        c_per_sample = plen / n
        kavg = {k: float(np.mean(v)) for k, v in ktimes.items()}
        dom = max(kavg, key=kavg.get)
        alg_bytes = (W + c_per_sample) * n          # W read + c written (encode) / c read + W written (decode)
        achieved = alg_bytes / (kavg[dom] * 1e-3) / 1e9
        standard = world == 1 and args.seconds == 3600.0 and bits == 24 and args.rate == 96000
        prof, prof_file, prof_note = load_profile() if standard else (None, None, "counters are profiled on the default workload only")
        gbs = lambda t: alg_bytes * world * K / t / 1e9              # one direction's algorithmic bytes per second of that direction
        res = {
            "metric": "pcm_mb_per_s_encode_plus_decode",
            "value": round(mbs(wall), 1),
            "unit": "MB/s",
            "n_gpus": world, "steps": K, "warmup": args.warmup,
            "ms_per_step": round(wall / K * 1e3, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "int64 (semantics of velox_sample_t; int32 fast path with exact overflow detection and int64 redo)", "data": "synthetic",
            "config": {"workload": f"configs[1]: {bits}-bit stereo {args.rate} Hz {args.seconds:g} s synthetic PCM per GPU, "
                                   f"encode + decode, default LPC order 8", "pcm_bytes_per_gpu": pcm_bytes,
                       "sub_blocks_per_gpu": (n + 8191) // 8192,
                       "parallelism": "1 GPU" if world == 1 else f"one track of {world} x {args.seconds:g} s, contiguous sub-block runs x{world}, "
                                      "high_res all-reduce + chunk-size all-gather in the step" + (", payload gather to rank 0 in the step" if args.gather else ""),
                       "exchange": exchange},
            "encode_mb_s": round(mbs(enc_t), 1), "decode_mb_s": round(mbs(dec_t), 1),
            "encode_plus_decode_calls_mb_s": round(mbs(enc_t + dec_t), 1),
            "host_glue_ms_per_step": round(glue_t / K * 1e3, 3),
            "compressed_ratio": round(plen / pcm_bytes, 4),
            "round_trip_bit_exact": ok,
            "kernel_ms": {k: round(v, 4) for k, v in sorted(kavg.items(), key=lambda kv: -kv[1])},
            "roofline": {"bound": "hbm", "kernel": dom, "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 4),
                         "traffic": pmc_traffic(prof, dom),
                         "algorithmic_bytes_per_launch": int(alg_bytes),
                         "valu_busy_profiled": pmc_valu_busy(prof, dom),
                         "profile": prof_file, "profile_note": prof_note,
                         # the same algorithmic bytes over a whole direction (all its kernels, launches, the host's
                         # length read-back) and over the whole step, per GPU: what a caller gets, not one kernel
                         "whole_path": {"encode_gbs": round(gbs(enc_t) / world, 1), "encode_frac": round(gbs(enc_t) / world / HBM_PEAK_GBS, 4),
                                        "decode_gbs": round(gbs(dec_t) / world, 1), "decode_frac": round(gbs(dec_t) / world / HBM_PEAK_GBS, 4),
                                        "step_gbs": round(2 * gbs(wall) / world, 1), "step_frac": round(2 * gbs(wall) / world / HBM_PEAK_GBS, 4)},
                         "note": "sample-serial recurrence: the kernel runs at the VALU issue limit, not at the HBM limit (DESIGN.md §3)",
                         "algorithmic_bytes_per_sample": round(W + c_per_sample, 4)},
        }
        if self_check is not None:                     # never silent: which exchange was timed, and whether the native one passed its check
            res["native_exchange_self_check"] = self_check
        if gather_ms is not None:
            res["gather_to_root_ms"] = round(gather_ms, 2)
        if total_len is not None:
            res["payload_bytes_total"] = int(total_len)
        if world == 1 and not args.no_cpu and not force_node:
            res["cpu_baseline"] = cpu_baseline(args, pcm, ctx, bits, ch, payload, last_enc[0], n)
        if not ok:
            res["value"] = None
            res["error"] = "round trip mismatch: no throughput is reported for a failing run"
        print(json.dumps(res), flush=True)
    if node is not None:
        node.close()
    if world > 1 or dist.is_initialized():
        dist.destroy_process_group()


def make_node(args, ctx, rank, world, backend, coll_dev):
    """The native exchange layer (vlx_node_*: the library calls RCCL itself) when every rank can bring it up,
    else None and the torch.distributed path of velox_amd/sharded.py is used.  Returns (node, description)."""
    import torch
    import torch.distributed as dist
    from velox_amd import sharded
    if args.exchange != "native":
        return None, "torch.distributed (" + backend + ")"
    if backend != "nccl":
        # rehearsal of the N > 1 code path where RCCL cannot run (several ranks on ONE GPU): the same vlx_node_* calls, the
        # six collectives carried by torch.distributed on the CPU (velox_amd/sharded.py:TorchTransport)
        tr = sharded.TorchTransport()
        node = sharded.NativeNode(ctx, None, rank, world, collectives=tr.table)
        node.transport = tr                                               # keeps the callbacks alive
        return node, f"native node layer (vlx_node_*) over a torch.distributed ({backend}) transport: a rehearsal on shared GPUs, not RCCL"
    uid = torch.zeros(128, dtype=torch.uint8, device=coll_dev)
    okf = torch.ones(1, dtype=torch.int32, device=coll_dev)
    if rank == 0:
        try:
            uid.copy_(torch.from_numpy(sharded.NativeNode.unique_id()).to(coll_dev))
        except Exception:                                                 # no RCCL to bind: every rank learns it from the flag
            okf.zero_()
    dist.broadcast(okf, 0)
    if int(okf.item()) == 0:
        return None, "torch.distributed (nccl); native layer unavailable"
    dist.broadcast(uid, 0)
    node = sharded.NativeNode(ctx, uid.cpu().numpy(), rank, world)       # collective: ncclCommInitRank
    return node, "native: libvelox_amd.so calls RCCL (ncclAllReduce of the device flag, ncclAllGather of size rows, ncclSend/Recv gather)"


def _agree(bad, coll_dev):
    """True when ANY rank says bad (collective)."""
    import torch
    import torch.distributed as dist
    f = torch.tensor([1 if bad else 0], dtype=torch.int32, device=coll_dev)
    dist.all_reduce(f, op=dist.ReduceOp.MAX)
    return bool(int(f.item()))


def native_self_check(node, shard, pcm, bits, n, max_chunks, rank, world, coll_dev, dev, torch_can_gather=True):
    """One untimed encode of this rank's shard through vlx_node_encode_track + vlx_node_gather_payload, compared with
    the torch.distributed exchange (velox_amd/sharded.py, covered by the gloo tests) of the same shards: shard bytes,
    offset, total length, high_res, the size table of every rank and the gathered payload.  Returns None when they
    agree on every rank, else a short reason (collective: every rank gets the same answer)."""
    import torch
    from velox_amd import sharded
    why, got = None, None
    try:
        ln, off, total, hr = node.encode_track(pcm.data_ptr(), bits, n, shard.payload.data_ptr(), shard.cap, max_chunks)
        mine = shard.payload[:ln].clone()
        tables = [node.chunk_sizes(r).copy() for r in range(world)]
        full = torch.zeros(total + 16, dtype=torch.uint8, device=dev) if rank == 0 else None
        torch.cuda.synchronize()
        node.gather_payload(0, shard.payload.data_ptr(), ln, full.data_ptr() if rank == 0 else 0, total + 16 if rank == 0 else 0)
        got = (ln, off, total, hr, mine, tables, full)
    except Exception as e:                                                # noqa: BLE001 - any failure means: do not use it
        why = f"{type(e).__name__}: {e}"
    if _agree(why is not None, coll_dev):
        return why or "an error on another rank"
    ln, off, total, hr, mine, tables, full = got
    sp = sharded.encode_track_sharded(shard.scan, shard.encode, coll_dev)
    if ln != int(sp.seg.numel()) or off != int(sp.off[rank]) or total != int(sp.total_len) or hr != sp.high_res:
        why = f"shard length/offset/total/high_res {(ln, off, total, hr)} != {(int(sp.seg.numel()), int(sp.off[rank]), int(sp.total_len), sp.high_res)}"
    elif not torch.equal(mine, sp.seg[:ln]):
        why = "shard bytes differ"
    elif any(not np.array_equal(tables[r], sp.sizes_per_rank[r]) for r in range(world)):
        why = "gathered size table differs"
    if torch_can_gather:                                                  # (device tensors: not over a CPU backend)
        ref = sharded.gather_payload(sp, root=0)
        if why is None and rank == 0 and not torch.equal(full[:total], ref[:total]):
            why = "gathered payload differs"
    elif why is None and rank == 0 and not torch.equal(full[off: off + ln - 1], mine[: ln - 1]):
        why = "gathered payload does not start with this rank's own shard"
    if _agree(why is not None, coll_dev):
        return why or "a mismatch on another rank"
    return None


def corpus_main(args, rank, world, local, dev, coll_dev, backend):
    """BASELINE configs[2]: a corpus of 24-bit/48 kHz stereo tracks, whole tracks per rank (weak scaling:
    --tracks-per-gpu each; 512 x 8 GPUs = the 4096 tracks), every track its own signal (seed = global track
    index).  One step = encode this rank's tracks in one batch, exchange the per-track payload lengths (the
    archive's offset table; native RCCL all-gather), decode them back to PCM."""
    import torch
    import torch.distributed as dist
    from velox_amd.codec import HIGH_RES_FROM_PAYLOAD, TRACK_SIZES_DTYPE, Context
    from velox_amd.synth import synth_corpus_device
    bits, W, ch, rate = 24, 3, 2, 48000
    T = args.tracks_per_gpu
    frames = int(rate * args.track_seconds)
    n = frames * ch
    stride = (n * W + 15) // 16 * 16
    pcm = synth_corpus_device(bits, ch, rate, frames, T, 2.0e6, 3000.0, dev, stride=stride, track0=rank * T)
    ctx = Context(local)
    node, exchange = (None, "none") if world == 1 else make_node(args, ctx, rank, world, backend, coll_dev)
    offs = [t * stride for t in range(T)]
    ns = [n] * T
    cap = ctx.encode_bound(ns)
    payload = torch.empty(cap, dtype=torch.uint8, device=dev)
    out = torch.empty(T * n * W + 16, dtype=torch.uint8, device=dev)
    cpt = (n + 8191) // 8192
    tr = np.zeros(T, dtype=TRACK_SIZES_DTYPE)
    tr["out_base"] = np.arange(T, dtype=np.uint64) * n
    tr["total_samples"] = n; tr["n_sizes"] = cpt; tr["high_res"] = HIGH_RES_FROM_PAYLOAD; tr["hdr_bits"] = 2
    ktimes, table = {}, [None]

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def step(record):
        t0 = time.perf_counter()
        poff, plen = ctx.encode_device(pcm.data_ptr(), bits, offs, ns, payload.data_ptr(), cap)
        if node is not None:
            table[0] = node.all_gather_lengths(plen, T)
        elif world > 1:
            from velox_amd import sharded
            table[0] = sharded.all_gather_lengths(plen, coll_dev)
        else:
            table[0] = [plen]
        t1 = time.perf_counter()
        if record:
            for k, ms in ctx.last_kernel_times():
                ktimes.setdefault(k, []).append(ms)
        sizes = ctx.last_chunk_sizes()
        tr["base_bit_off"] = 8 * poff.astype(np.uint64)
        t2 = time.perf_counter()
        ctx.decode_tracks_device(payload.data_ptr(), cap, tr, sizes, out.data_ptr(), bits)
        t3 = time.perf_counter()
        if record:
            for k, ms in ctx.last_kernel_times():
                ktimes.setdefault(k, []).append(ms)
        return (t1 - t0), (t3 - t2), int(plen.sum())

    if node is not None:                                                  # untimed: the native length table against torch's
        from velox_amd import sharded
        bad = None
        try:
            _, plen0 = ctx.encode_device(pcm.data_ptr(), bits, offs, ns, payload.data_ptr(), cap)
            a = node.all_gather_lengths(plen0, T)
        except Exception as e:                                            # noqa: BLE001
            bad = f"{type(e).__name__}: {e}"
        if _agree(bad is not None, coll_dev):
            node, exchange = None, "torch.distributed (nccl); the native layer failed its self-check: " + (bad or "an error on another rank")
        else:
            b = sharded.all_gather_lengths(plen0, coll_dev)
            bad = None if len(a) == len(b) and all(np.array_equal(np.asarray(x, dtype=np.uint64), np.asarray(y, dtype=np.uint64)) for x, y in zip(a, b)) else "length table differs"
            if _agree(bad is not None, coll_dev):
                node, exchange = None, "torch.distributed (nccl); the native layer failed its self-check: length table differs"
            else:
                exchange += "; checked against the torch.distributed table before timing"
    ctx.set_profiling(True)
    for _ in range(args.warmup):
        step(False)
    barrier()
    t_start = time.perf_counter()
    enc_t = dec_t = 0.0
    comp = 0
    for _ in range(args.steps):
        e, d, comp = step(True)
        enc_t += e; dec_t += d
    barrier()
    wall = time.perf_counter() - t_start
    ok = bool(torch.equal(out[: T * n * W].view(T, n * W), pcm[: T * stride].view(T, stride)[:, : n * W]))
    ok = ok and len(table[0]) == world and all(len(x) == T for x in table[0])
    stats = torch.tensor([enc_t, dec_t, wall], dtype=torch.float64, device=coll_dev)
    if world > 1:
        dist.all_reduce(stats, op=dist.ReduceOp.MAX)
        okt = torch.tensor([1 if ok else 0], device=coll_dev)
        dist.all_reduce(okt, op=dist.ReduceOp.MIN)
        ok = bool(okt.item())
    enc_t, dec_t, wall = [float(x) for x in stats.tolist()]
    if rank == 0:
        K = args.steps
        pcm_bytes = T * n * W
        total_pcm = pcm_bytes * world
        mbs = lambda t: total_pcm * K / t / 1e6
        kavg = {k: float(np.mean(v)) for k, v in ktimes.items()}
        dom = max(kavg, key=kavg.get)
        alg = pcm_bytes + comp
        ach = alg / (kavg[dom] * 1e-3) / 1e9
        res = {"metric": "pcm_mb_per_s_encode_plus_decode", "value": round(mbs(wall), 1) if ok else None, "unit": "MB/s",
               "n_gpus": world, "steps": K, "warmup": args.warmup, "ms_per_step": round(wall / K * 1e3, 3),
               "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
               "dtype": "int64 (semantics of velox_sample_t; int32 fast path with exact overflow detection and int64 redo)", "data": "synthetic",
               "config": {"workload": f"configs[2]: corpus of {T * world} synthetic 24-bit/48 kHz stereo tracks of {args.track_seconds:g} s, "
                                      f"{T} whole tracks per GPU (seed = global track index), encode + decode",
                          "pcm_bytes_per_gpu": pcm_bytes, "sub_blocks_per_gpu": T * cpt,
                          "parallelism": "1 GPU" if world == 1 else f"whole tracks x{world}, all-gather of per-track payload lengths in the step",
                          "exchange": exchange},
               "encode_mb_s": round(mbs(enc_t), 1), "decode_mb_s": round(mbs(dec_t), 1),
               "compressed_ratio": round(comp / pcm_bytes, 4), "round_trip_bit_exact": ok,
               "archive_bytes_total": int(sum(int(x.sum()) for x in table[0])),
               "kernel_ms": {k: round(v, 4) for k, v in sorted(kavg.items(), key=lambda kv: -kv[1])},
               "roofline": {"bound": "hbm", "kernel": dom, "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                            "frac": round(ach / HBM_PEAK_GBS, 4), "traffic": None, "algorithmic_bytes_per_launch": int(alg)}}
        if not ok:
            res["error"] = "round trip or length-table mismatch: no throughput is reported for a failing run"
        print(json.dumps(res), flush=True)
    if node is not None:
        node.close()
    if world > 1:
        dist.destroy_process_group()


def host_cpu_share():
    """Threads the CPU baseline may use, and why: every core this process is allowed to run on — the affinity mask, cut
    down to a cgroup CPU quota when there is one, and to the CPU share the box announces for one GPU's job through
    OMP_NUM_THREADS (the GPU pool's rule: worker pools are sized to that share, not to the machine).  Returns
    (threads, cores the machine has, what set the limit)."""
    host = os.cpu_count() or 1
    try:
        n, why = len(os.sched_getaffinity(0)), "affinity mask"
    except AttributeError:
        n, why = host, "cpu_count"
    if n == host:
        why = "none: every core of the host"
    try:
        with open("/sys/fs/cgroup/cpu.max") as fh:
            q, per = fh.read().split()[:2]
        if q != "max" and int(q) > 0 and -(-int(q) // int(per)) < n:
            n, why = -(-int(q) // int(per)), "cgroup cpu.max quota"
    except (OSError, ValueError):
        pass
    share = os.environ.get("VELOX_CPU_THREADS") or os.environ.get("OMP_NUM_THREADS")
    if share and share.isdigit() and 0 < int(share) < n:
        n, why = int(share), ("VELOX_CPU_THREADS" if os.environ.get("VELOX_CPU_THREADS") else "OMP_NUM_THREADS = this job's CPU share of the box")
    return max(1, n), host, why


def cpu_baseline(args, pcm, ctx, bits, ch, payload, last_enc, n):
    """The reference itself (oracle/_ref, compiled from the reference's sources) on the host cores,
    over a bounded sample of the same PCM; falls back to the C port when that build is absent.  Also
    the parity gate, on the payload the TIMED region produced (still in HBM): its first 60 s must equal
    the CPU encoder's payload of those 60 s byte for byte, and 64 chunks sampled across the whole track
    (first, last, the tail block, random ones) must each equal the oracle's encoding of the same 8192
    samples, size prefix included."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from _oracle import Oracle, Reference
    from _sampling import check_chunks, pick
    from velox_amd.synth import unpack_pcm
    W = bits // 8
    ns = int(min(args.cpu_sample_seconds, args.seconds) * args.rate) * ch
    ns -= ns % 8192
    raw = pcm[: ns * W].cpu().numpy()
    s = unpack_pcm(raw, bits)
    cores, host_cores, cores_why = host_cpu_share()
    if Reference.available():
        R = Reference(threads=cores)
        kind, threads = "reference", R.pool_threads()
        t0 = time.perf_counter(); p = R.encode(s); t1 = time.perf_counter()
        d = R.decode(p, s.size); t2 = time.perf_counter()
    else:
        O = Oracle()
        kind, threads = "port", cores
        t0 = time.perf_counter(); p = O.encode(s, threads=cores); t1 = time.perf_counter()
        d = O.decode(p, s.size); t2 = time.perf_counter()
    poff, plen, sizes = last_enc
    # the timed payload's prefix: every byte but the CPU payload's last one, which holds the final chunk's last
    # 2 bits plus padding where the long payload goes on with the next size prefix
    head = payload[poff: poff + p.size].cpu().numpy()
    prefix_ok = bool(p.size >= 2 and np.array_equal(head[:-1], p[:-1]) and (int(head[-1]) & 3) == (int(p[-1]) & 3))
    O2 = Oracle()
    picks = pick(sizes.size, 64, seed=1)
    high_res = int(head[0] >> 1) & 1
    bad = check_chunks(O2, lambda a, b: pcm[a * W: b * W].cpu().numpy(), lambda a, b: payload[poff + a: poff + b].cpu().numpy(),
                       sizes, n, bits, high_res, picks, unpack_pcm)
    # ... and EVERY chunk of the timed payload, re-encoded by the oracle on the host threads (a few seconds for the hour)
    tv = time.perf_counter()
    n_cmp, n_bad, first_bad = O2.verify_payload(pcm[: n * W].cpu().numpy(), bits, n, payload[poff: poff + plen].cpu().numpy(), cores)
    tv = time.perf_counter() - tv
    parity = prefix_ok and not bad and bool(np.array_equal(d, s)) and n_bad == 0 and n_cmp == sizes.size
    mb = ns * W / 1e6
    # what the reference does not do: chunk-parallel decode on the same cores (the C port, for scale)
    t3 = time.perf_counter(); d2 = O2.decode_mt(p, s.size, cores); t4 = time.perf_counter()
    parity = parity and bool(np.array_equal(d2, s))
    extra = {"decode_chunk_parallel_port_mb_s": round(mb / (t4 - t3), 2)}
    extra["timed_payload_every_chunk"] = {"compared": n_cmp, "of": int(sizes.size), "mismatching": n_bad, "first_bad": first_bad,
                                          "checker": "oracle (vlxo_verify_payload)", "threads": cores, "seconds": round(tv, 1)}
    return {**extra, "value": round(mb / (t2 - t0), 2), "unit": "MB/s", "cores": threads, "host_cores": host_cores, "cores_limited_by": cores_why, "kind": kind,
            "sample": f"first {ns * W / 1e6:.1f} MB ({ns // ch / args.rate:.1f} s) of the same track; "
                      f"encode on {threads} pool threads + decode on 1 thread (the reference decoder is single-threaded)",
            "encode_mb_s": round(mb / (t1 - t0), 2), "decode_mb_s": round(mb / (t2 - t1), 2),
            "timed_payload_prefix_equals_cpu_payload": prefix_ok,
            "timed_payload_sampled_chunks": {"checked": len(picks), "mismatching": bad},
            "gpu_payload_equals_cpu_payload": parity}


if __name__ == "__main__":
    main()
