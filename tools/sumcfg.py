import json, sys
for f in sys.argv[1:]:
    for l in open(f):
        if l.startswith("{"):
            x = json.loads(l)
            print(f, x.get("encode_ms"), x.get("decode_ms"), x.get("encode_mb_s"), x.get("decode_mb_s"), x.get("kernel_ms"), x.get("every_chunk_vs_oracle"), x.get("round_trip_bit_exact"))
