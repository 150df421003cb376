"""vlx_decode_device_noalloc — the entry point that allocates nothing (include/velox_amd.h): its workspace is
the library's file-scope device array.  Checked through the C-ABI against vlx_decode_device on the same
batch, bit for bit, on two calls in a row, the second with a smaller input (the array keeps the first
call's contents), plus the documented limit.  Collected last (the file name sorts behind every other test
file) so that a failure here cannot hide another test under `pytest -x`."""
import numpy as np
import pytest

from conftest import load_golden

pytestmark = pytest.mark.gpu


def _batch(names, oracle):
    """Several golden payloads back to back in one buffer (16-B aligned starts) + their chunk tables."""
    from velox_amd.codec import walk_chunks
    blobs, chunks, wants, pos, out = [], [], [], 0, 0
    for n in names:
        g = load_golden(n)
        p, total = g["payload"], g["samples"].size
        chunks.append(walk_chunks(p, total, base_bit_off=8 * pos, out_base=out))
        wants.append(oracle.decode(p, total))
        pad = (-p.size) % 16
        blobs.append(np.concatenate([p, np.zeros(pad, np.uint8)]))
        pos += p.size + pad
        out += total
    return np.concatenate(blobs + [np.zeros(32, np.uint8)]), np.concatenate(chunks), np.concatenate(wants)


def test_noalloc_decode_equals_decode_device_twice_second_smaller(ctx, oracle):
    import torch
    first = ["s24_stereo_96k_hires", "s16_stereo_44k", "white_noise_fullscale_s32", "near_identical_channels_hires",
             "impulses_escape_codes", "silence", "short_tail_block", "wide_2p40"]
    second = ["s16_3ch", "tiny_20_samples", "lsb_shifted_x16"]
    for names, fmts in ((first, (64, 24)), (second, (64, 16))):
        host, ch, want = _batch(names, oracle)
        d_p = torch.from_numpy(host).cuda()
        for fmt in fmts:
            w = 8 if fmt == 64 else fmt // 8
            a = torch.zeros(want.size * w + 16, dtype=torch.uint8, device="cuda")
            b = torch.full((want.size * w + 16,), 0, dtype=torch.uint8, device="cuda"); torch.cuda.synchronize()
            ctx.decode_device(d_p.data_ptr(), host.size, ch, a.data_ptr(), fmt)
            ctx.decode_device_noalloc(d_p.data_ptr(), host.size, ch, b.data_ptr(), fmt)
            assert torch.equal(a, b), (names, fmt)
            if fmt == 64:                                              # and against the oracle, not only each other
                assert np.array_equal(b.cpu().numpy()[: want.size * 8].view("<i8"), want)


def test_noalloc_limit_is_a_clear_error(ctx, oracle):
    import torch
    from velox_amd.codec import VeloxError
    host, ch, want = _batch(["tiny_20_samples"], oracle)
    d_p = torch.from_numpy(host).cuda()
    dst = torch.zeros(8 * want.size + 16, dtype=torch.uint8, device="cuda"); torch.cuda.synchronize()
    too_many = np.repeat(ch[:1], 1025)
    with pytest.raises(VeloxError) as e:
        ctx.decode_device_noalloc(d_p.data_ptr(), host.size, too_many, dst.data_ptr(), 64)
    assert e.value.code == -6 and "1024" in str(e.value)
    # exactly the maximum is accepted (1024 copies of one small chunk, all writing the same output)
    ctx.decode_device_noalloc(d_p.data_ptr(), host.size, np.repeat(ch[:1], 1024), dst.data_ptr(), 64)
    assert np.array_equal(dst.cpu().numpy()[: want.size * 8].view("<i8"), want)


def test_noalloc_decode_leaves_the_workspace_empty(oracle):
    """The "allocates nothing" claim itself: on a FRESH context that has never grown a workspace buffer,
    vlx_decode_device_noalloc decodes a batch and vlx_ctx_workspace_bytes (device buffers + pinned staging) still says 0."""
    import torch
    from velox_amd.codec import Context
    host, ch, want = _batch(["s24_stereo_96k_hires", "s16_3ch", "wide_2p40"], oracle)
    d_p = torch.from_numpy(host).cuda()
    dst = torch.zeros(8 * want.size + 16, dtype=torch.uint8, device="cuda"); torch.cuda.synchronize()
    c = Context(0)
    try:
        assert c.workspace_bytes() == 0
        c.decode_device_noalloc(d_p.data_ptr(), host.size, ch, dst.data_ptr(), 64)
        assert c.workspace_bytes() == 0
        assert np.array_equal(dst.cpu().numpy()[: want.size * 8].view("<i8"), want)
    finally:
        c.close()
