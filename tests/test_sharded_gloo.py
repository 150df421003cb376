"""The N>1 path on CPU: world_size 2 and 3 over gloo.  The exchange logic of velox_amd/sharded.py
(high_res MAX all-reduce, chunk-size all-gather -> offset table, point-to-point gather with the
seam-byte OR) runs for real; the per-shard codec work, which is GPU-only in the product, is stood in
for by shards cut out of the oracle's whole-track payload (the oracle is the checker here)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import load_golden


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _cut_shard(full, sizes, lo, hi, first):
    """Bits [2 + 8*off_lo, 2 + 8*off_hi) of the whole payload as a shard payload (2 leading bits:
    the stream header on the first shard, zeros elsewhere)."""
    body = np.concatenate(([0], np.cumsum(sizes.astype(np.int64) + 4)))
    a, b = int(body[lo]), int(body[hi])
    seg = full[a: b + 1].copy()
    if b == a:                                   # no chunks: just the 2 header bits (or nothing)
        seg[0] = (full[a] & 0x03) if first else 0
        return seg
    if not first:
        seg[0] &= 0xFC                           # bits 0-1 belong to the predecessor's last chunk
    seg[-1] &= 0x03                              # bits 2-7 belong to the successor's first chunk
    return seg


def _worker(rank, world, port, name, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from velox_amd import sharded
        from velox_amd.codec import walk_chunks
        g = load_golden(name)
        s, full = g["samples"].astype(np.int64), g["payload"]
        ch = walk_chunks(full, s.size)
        sizes = (ch["n_bits"] // 8).astype(np.uint32)
        ranges = sharded.shard_ranges(s.size, world)
        c_lo = ranges[rank][0] // 8192
        c_hi = c_lo + (ranges[rank][1] + 8191) // 8192
        mine = s[ranges[rank][0]: ranges[rank][0] + ranges[rank][1]]

        def scan():
            return int(np.any(np.abs(mine) > 65536))

        def enc(high_res, first):
            assert high_res == int((full[0] >> 1) & 1), "file-global high_res must survive the all-reduce"
            seg = _cut_shard(full, sizes, c_lo, c_hi, first)
            return torch.from_numpy(seg), sizes[c_lo:c_hi]

        sp = sharded.encode_track_sharded(scan, enc, torch.device("cpu"))
        assert sp.total_len == full.size
        assert [len(x) for x in sp.sizes_per_rank] == [(r[1] + 8191) // 8192 for r in ranges]
        final = sharded.gather_payload(sp, root=0)
        if rank == 0:
            ok = final is not None and np.array_equal(final.numpy(), full)
            q.put(("ok" if ok else "payload mismatch", rank))
        else:
            assert final is None
            q.put(("ok", rank))
        lens = sharded.all_gather_lengths([100 + rank, 7], torch.device("cpu"))
        assert [list(x) for x in lens] == [[100 + r, 7] for r in range(world)]
    except Exception as e:  # noqa: BLE001
        q.put((f"rank {rank}: {type(e).__name__}: {e}", rank))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,name", [(2, "s24_stereo_96k_hires"), (3, "s16_stereo_44k"), (2, "tiny_20_samples"),
                                        (3, "near_identical_channels_hires")])
def test_sharded_encode_exchange_over_gloo(world, name):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, name, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
    assert all(r[0] == "ok" for r in res), res


def test_shard_ranges_and_offsets():
    from velox_amd import sharded
    r = sharded.shard_ranges(8192 * 5 + 346, 4)
    assert r == [(0, 16384), (16384, 16384), (32768, 8192), (40960, 346)]
    assert sharded.shard_ranges(20, 3) == [(0, 20), (20, 0), (20, 0)]
    off, seg_len, total = sharded.offset_table([np.array([10, 20], np.uint32), np.array([], np.uint32), np.array([5], np.uint32)])
    assert list(off) == [0, 38, 38] and list(seg_len) == [39, 1, 10] and total == 48


def test_merge_shards_host_matches_whole_payload():
    from velox_amd import sharded
    from velox_amd.codec import walk_chunks
    g = load_golden("s16_stereo_44k")
    full = g["payload"]
    sizes = (walk_chunks(full, g["samples"].size)["n_bits"] // 8).astype(np.uint32)
    cuts = [0, 2, 2, len(sizes)]
    segs = [_cut_shard(full, sizes, cuts[i], cuts[i + 1], i == 0) for i in range(3)]
    merged = sharded.merge_shards_host(segs, [sizes[cuts[i]: cuts[i + 1]] for i in range(3)])
    assert np.array_equal(merged, full)


def _decode_worker(rank, world, port, name, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from _oracle import Oracle
        from velox_amd import sharded
        from velox_amd.codec import walk_chunks
        O = Oracle()
        g = load_golden(name)
        s, full = g["samples"].astype(np.int64), g["payload"]
        ch = walk_chunks(full, s.size)                      # the one host walk; every rank can do it, nothing is exchanged
        rng = sharded.decode_ranges(ch, world)[rank]
        seg = sharded.shard_as_payload(full, ch, rng)       # the bytes this rank would upload
        mine = O.decode(seg, rng[5]) if rng[5] else np.zeros(0, np.int64)   # stand-in for the GPU decode of the shard
        assert mine.size == rng[5]
        # the re-based chunk table the GPU path would get stays inside the shard's bytes
        sub = sharded.shard_chunks(ch, rng)
        if len(sub):
            assert int(sub["bit_off"][0]) == 32 + 2 and int(sub["out_off"][0]) == 0
            assert int((sub["bit_off"] + sub["n_bits"]).max()) <= 8 * seg.size
        # PCM left sharded, or gathered: static offsets, so a plain all-gather of equal-size buffers does it
        buf = torch.zeros(s.size, dtype=torch.int64)
        buf[rng[4]: rng[4] + rng[5]] = torch.from_numpy(mine)
        dist.all_reduce(buf)                                # disjoint slices: the sum is the concatenation
        want = O.decode(full, s.size)
        ok = np.array_equal(buf.numpy()[: want.size], want)
        q.put(("ok" if ok else "decode mismatch", rank))
    except Exception as e:  # noqa: BLE001
        q.put((f"rank {rank}: {type(e).__name__}: {e}", rank))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,name", [(2, "s24_stereo_96k_hires"), (3, "s16_stereo_44k"), (3, "tiny_20_samples")])
def test_sharded_decode_over_gloo(world, name):
    """SURVEY §8e decode partitioning: chunk ranges per rank from one host walk, no exchange before the decode."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_decode_worker, args=(r, world, port, name, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
    assert all(r[0] == "ok" for r in res), res


def test_decode_ranges_cover_the_track():
    from velox_amd import sharded
    from velox_amd.codec import walk_chunks
    g = load_golden("short_tail_block")
    ch = walk_chunks(g["payload"], g["samples"].size)
    for world in (1, 2, 3, 7):
        r = sharded.decode_ranges(ch, world)
        assert [x[0] for x in r][0] == 0 and r[-1][1] == len(ch)
        assert all(r[i][1] == r[i + 1][0] for i in range(world - 1))
        assert sum(x[5] for x in r) == g["samples"].size
        nz = [x for x in r if x[5]]
        assert all(nz[i][4] + nz[i][5] == nz[i + 1][4] for i in range(len(nz) - 1))


@pytest.mark.parametrize("cut", [1, 3, 5, 6, 7, 9, 40, 1000])
def test_decode_ranges_on_a_cut_payload(cut, oracle):
    """Truncated payloads (incl. cuts inside a u32 size prefix, where vlx_walk_chunks places the chunk at the
    end of the data with no bits): the per-rank byte ranges stay inside the payload, never negative, the
    re-based chunk tables stay inside their range, and decoding every rank's range as a payload of its own
    gives the reference's decode of the cut payload."""
    from velox_amd import sharded
    from velox_amd.codec import walk_chunks
    g = load_golden("s16_stereo_44k")
    p, n = g["payload"][:cut], g["samples"].size
    ch = walk_chunks(p, n)
    want = oracle.decode(p, n)
    assert int(ch["n_out"].sum()) == want.size
    for world in (1, 2, 3):
        got = np.zeros(want.size, np.int64)
        for rng in sharded.decode_ranges(ch, world):
            c_lo, c_hi, byte_lo, byte_hi, out_lo, out_n = rng
            assert 0 <= byte_lo <= byte_hi <= p.size
            sub = sharded.shard_chunks(ch, rng)
            if len(sub):
                assert int((sub["bit_off"] + sub["n_bits"]).max()) <= 8 * (byte_hi - byte_lo)
            if out_n and byte_hi > byte_lo and int(ch["bit_off"][c_lo]) >= 34:
                seg = sharded.shard_as_payload(p, ch, rng)
                mine = oracle.decode(seg, out_n)
                got[out_lo: out_lo + mine.size] = mine
            # (a chunk named only by a cut prefix decodes to zeros from no bits: `got` is already zero there)
        assert np.array_equal(got, want), world


def test_native_offset_table_and_shard_range_match_the_python_ones():
    """vlx_node_offset_table / vlx_node_shard_range (the C side of the native RCCL layer, no GPU needed)
    against sharded.offset_table / shard_ranges on ragged shards, an empty rank and one-rank worlds."""
    from velox_amd import sharded
    rng = np.random.default_rng(9)
    for world in (1, 2, 3, 8):
        shards = [rng.integers(1, 40000, size=int(rng.integers(0, 7))).astype(np.uint32) for _ in range(world)]
        shards[world // 2] = np.zeros(0, np.uint32)                       # a rank without chunks
        a = sharded.offset_table(shards)
        b = sharded.native_offset_table(shards)
        assert list(a[0]) == list(b[0]) and list(a[1]) == list(b[1]) and a[2] == b[2]
        for total in (0, 2, 8192, 8194, 5 * 8192, 5 * 8192 + 100, 1234567 * 2):
            want = sharded.shard_ranges(total, world)
            got = [sharded.native_shard_range(total, r, world) for r in range(world)]
            assert got == [(int(x), int(y)) for x, y in want], (world, total)
    assert sharded.native_offset_table([np.array([10, 20], np.uint32), np.zeros(0, np.uint32), np.array([5], np.uint32)])[2] == 48
