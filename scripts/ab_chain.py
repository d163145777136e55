"""A/B of the one-row decode chain (csrc/gemv_chain.hip) on ONE box: bench.py at small batches with wm_set_decode_chain(0 / 1 / 2), interleaved.
    python scripts/ab_chain.py [batch=1] [rounds=2]"""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
batch = sys.argv[1] if len(sys.argv) > 1 else "1"
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 2
code = r"""
import sys, os
sys.path[:0] = [%r, %r]
import native
native.load_library().wm_set_decode_chain(int(sys.argv[1]))
sys.argv = ["bench.py", "--batch", sys.argv[2], "--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--length-dist", "forced", "--no-measure-traffic"]
import bench
bench.main()
""" % (ROOT, os.path.join(ROOT, "eddie-wang-hackathon2023_amd"))
for r in range(rounds):
    for on in (0, 1, 2):
        out = subprocess.run([sys.executable, "-c", code, str(on), batch], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True).stdout
        line = [l for l in out.splitlines() if l.startswith("{")][-1]
        res = json.loads(line)
        name = ("off", "a launch per layer", "one launch per step")[on]
        print(f"batch {batch} chain {name} round {r}: {res['roofline']['decode_step_ms']} ms per token, {res['value']} tokens/s whole job", flush=True)
