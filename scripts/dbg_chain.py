import sys, os
ROOT="/root/repo"
sys.path[:0] = [ROOT, os.path.join(ROOT, "eddie-wang-hackathon2023_amd")]
import native
if len(sys.argv) > 2:
    native.load_library().wm_set_decode_chain(int(sys.argv[2]))
b = sys.argv[1]
sys.argv = ["bench.py", "--batch", b, "--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--length-dist", "forced", "--no-measure-traffic"]
import bench
bench.main()
print("STATUS", native.chain_status())
