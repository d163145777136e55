import json,sys
for f in sys.argv[1:]:
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); r=d["roofline"]
        print(f, d["value"], d["ms_per_step"], r.get("decode_loop_ms"), r["decode_step_ms"], r["in_situ_launch_ms"], r["in_situ_chain_between_launches_ms"], r["in_situ_launches_in_flight"])
    except Exception as e: print(f, "failed", e)
