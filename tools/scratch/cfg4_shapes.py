import sys, os, json, io, contextlib
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tools"))
import bench_configs as B
from velox_amd.codec import Context, set_decode_shape
ctx = Context(0)
B._cpu_baseline = lambda *a, **k: None
for shape in ("stream", "pipe64", "pipe32", "auto"):
    set_decode_shape(shape)
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        B.corpus(ctx, 32, 8, 192000, 60.0, 1, 5.0e8, 2.0e5, 3, "cfg4 " + shape)
    d = json.loads(buf.getvalue().strip().splitlines()[-1])
    print(shape, d["decode_ms"], {k: v for k, v in d["kernel_ms"].items() if "decode" in k}, d["round_trip_bit_exact"], flush=True)
