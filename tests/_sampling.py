"""Sampled-chunk parity for payloads too large to re-encode on the CPU: pick chunk positions across a
device-resident payload, pull only those chunks' PCM and payload bytes to the host and compare each with
the oracle's encoding of the same 8192 samples (oracle.encode_chunk = one task of the reference's
sub-block loop, VeloxCore.h:279-293).  Test infrastructure: imported by tests/, by bench.py's
cpu_baseline leg and by tools/bench_configs.py's parity gate only."""
import numpy as np

SUB_BLOCK = 8192


def chunk_positions(sizes):
    """Byte offset (in the track's payload) of every chunk's u32 size prefix — it sits at bit 2 of that
    byte (2 header bits, then whole bytes: VeloxCore.h:226-236, :301-302)."""
    sizes = np.asarray(sizes, dtype=np.int64)
    return np.concatenate(([0], np.cumsum(sizes + 4)[:-1]))


def pick(n_chunks, k, seed=0):
    """First, second, last two (the tail block) and random chunks in between."""
    fixed = [c for c in (0, 1, n_chunks - 2, n_chunks - 1) if 0 <= c < n_chunks]
    rng = np.random.default_rng(seed)
    rest = rng.choice(n_chunks, size=min(max(k - len(fixed), 0), n_chunks), replace=False).tolist() if n_chunks else []
    return sorted(set(fixed + rest))


def unshift2(b):
    """Bytes of a bit string that starts at bit 2 of b[0]: len(b) - 1 whole bytes."""
    b = np.asarray(b, dtype=np.uint8).astype(np.uint16)
    return (((b[:-1] >> 2) | (b[1:] << 6)) & 0xFF).astype(np.uint8)


def check_chunks(oracle, fetch_pcm, fetch_payload, sizes, n_samples, bits, high_res, picks, unpack):
    """fetch_pcm(sample_lo, sample_hi) -> packed PCM bytes (numpy uint8) of that sample range of the track;
    fetch_payload(byte_lo, byte_hi) -> payload bytes of the track.  Returns the list of mismatching chunk
    indices (empty = every sampled chunk equals the oracle's bytes, size prefix included)."""
    pos = chunk_positions(sizes)
    bad = []
    for c in picks:
        lo, hi = c * SUB_BLOCK, min((c + 1) * SUB_BLOCK, n_samples)
        s = unpack(fetch_pcm(lo, hi), bits)
        want, _ = oracle.encode_chunk(s, high_res)
        raw = fetch_payload(int(pos[c]), int(pos[c]) + 4 + int(sizes[c]) + 1)
        got = unshift2(raw)
        prefix = int(got[0]) | (int(got[1]) << 8) | (int(got[2]) << 16) | (int(got[3]) << 24)
        if prefix != int(sizes[c]) or want.size != int(sizes[c]) or not np.array_equal(got[4:], want):
            bad.append(int(c))
    return bad
