"""One-line digest of bench.py JSON lines (files given as arguments)."""
import json, sys
for f in sys.argv[1:]:
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1]); r = d.get("roofline") or {}
        print(f, d["value"], d["ms_per_step"], "loop", r.get("decode_loop_ms"), "step", r.get("decode_step_ms"), "beside-enc", r.get("decode_step_beside_encoder_ms"),
              "kv", r.get("in_situ_launch_ms"), "chain", r.get("in_situ_chain_between_launches_ms"), r.get("in_situ_launches_in_flight"))
    except Exception as e:
        print(f, "failed", e)
