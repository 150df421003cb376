"""Generate the golden vectors under tests/golden/ from the REAL reference (oracle/_ref,
compiled from /root/reference by oracle/Makefile).  Run in the build container only:

    make -C oracle && python tests/golden/make_golden.py

Each fixture is data only: the input samples (or float32 bytes) and the payload / decoded samples
the reference produced for them.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from _oracle import Reference  # noqa: E402
from cases import float_cases, odd_cases, pcm_cases, wide_cases  # noqa: E402


def small(a):
    a = np.asarray(a, dtype=np.int64)
    for dt in (np.int16, np.int32):
        if a.size == 0 or (a.min() >= np.iinfo(dt).min and a.max() <= np.iinfo(dt).max):
            return a.astype(dt)
    return a


def main():
    R = Reference(threads=8)
    out = {}
    names = []
    for name, bits, s in pcm_cases() + odd_cases() + wide_cases():
        payload = R.encode(s)
        rec = {"samples": small(s), "payload": payload, "bits": np.int32(bits)}
        if s.size % 2 == 0:
            dec = R.decode(payload, s.size)
            if not np.array_equal(dec, s):
                rec["decoded"] = small(dec)      # the reference itself is lossy here
        np.savez_compressed(os.path.join(HERE, name + ".npz"), **rec)
        names.append(name)
    # LPC known answers: coefficients for the first stream of a few cases
    for name, f in float_cases():
        raw = f.view(np.uint8)
        m = np.zeros(f.size, dtype=np.int64); e = np.zeros(f.size, dtype=np.uint8)
        R.L.ref_split_float32(raw.ctypes.data_as(R.L.ref_split_float32.argtypes[0]), f.size,
                              m.ctypes.data_as(R.L.ref_split_float32.argtypes[2]),
                              e.ctypes.data_as(R.L.ref_split_float32.argtypes[3]))
        payload = R.encode(m, True, e, raw)
        dv, de, isf, fm = R.decode(payload, m.size, True)
        np.savez_compressed(os.path.join(HERE, name + ".npz"), raw=raw, mant=small(m), exps=e, payload=payload,
                            dec_vals=small(dv), dec_exps=de, is_float=np.int32(isf), float_mode=np.int32(fm))
        names.append(name)
    with open(os.path.join(HERE, "INDEX.txt"), "w") as fh:
        fh.write("\n".join(names) + "\n")
    tot = sum(os.path.getsize(os.path.join(HERE, n + ".npz")) for n in names)
    print(f"wrote {len(names)} fixtures, {tot/1e6:.2f} MB")


if __name__ == "__main__":
    main()
