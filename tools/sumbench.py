import json,sys
for l in open(sys.argv[1]):
    if l.startswith("{"):
        d=json.loads(l); print(d["value"], d["ms_per_step"], d["encode_mb_s"], d["decode_mb_s"], d["kernel_ms"], d["round_trip_bit_exact"])
