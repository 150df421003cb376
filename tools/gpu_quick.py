"""Quick GPU parity sweep over the case catalogue (debug helper; the real tests are in tests/)."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from _oracle import Oracle
from cases import pcm_cases, wide_cases, float_cases, odd_cases
from velox_amd.codec import Context, StreamingDecoder
from velox_amd.synth import pack_pcm

O = Oracle(); ctx = Context(0)
bad = 0
def first_diff(a, b):
    n = min(a.size, b.size)
    d = np.nonzero(a[:n] != b[:n])[0]
    return (int(d[0]) if d.size else n), a.size, b.size
for name, bits, s in pcm_cases() + wide_cases():
    ref = O.encode(s)
    t = time.time()
    try:
        got = ctx.encode_block(s)
    except Exception as e:
        print(f"{name:32s} ENCODE EXC {e}"); bad += 1; continue
    ok = np.array_equal(ref, got)
    msg = "" if ok else f" first_diff={first_diff(ref, got)}"
    ok2 = True
    if bits in (16, 24, 32) and s.size:
        got2 = ctx.encode_pcm(pack_pcm(s, bits), bits)
        ok2 = np.array_equal(ref, got2)
        if not ok2: msg += f" pcm_first_diff={first_diff(ref, got2)}"
    # decode
    dref = O.decode(ref, s.size)
    try:
        d = StreamingDecoder(ref, s.size, ctx=ctx, lookahead=3)
        dv, de = d.read(s.size + 5)
        okd = np.array_equal(dv, dref)
        if not okd: msg += f" dec_first_diff={first_diff(dref, dv)}"
    except Exception as e:
        okd = False; msg += f" DEC EXC {e}"
    print(f"{name:32s} n={s.size:7d} enc={ok} pcm={ok2} dec={okd} {time.time()-t:.2f}s{msg}", flush=True)
    bad += (not ok) + (not ok2) + (not okd)
for name, f in float_cases():
    raw = f.view(np.uint8); m, e = O.split_float32(raw)
    ref = O.encode(m, True, e, raw)
    try:
        got = ctx.encode_block(m, True, e, raw)
        ok = np.array_equal(ref, got)
        msg = "" if ok else f" first_diff={first_diff(ref, got)}"
        dref = O.decode(ref, m.size, True)
        d = StreamingDecoder(ref, m.size, ctx=ctx, lookahead=2)
        dv, de = d.read(m.size)
        okd = np.array_equal(dv, dref[0]) and np.array_equal(de, dref[1]) and d.IsFloat() == bool(dref[2]) and d.GetFloatMode() == dref[3]
    except Exception as ex:
        ok = okd = False; msg = f" EXC {ex}"
    print(f"{name:32s} n={m.size:7d} enc={ok} dec={okd}{msg}", flush=True)
    bad += (not ok) + (not okd)
print("BAD", bad)
sys.exit(1 if bad else 0)
