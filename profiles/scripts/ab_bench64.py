"""One bench.py line at batch N (default 64) reduced to the numbers an A/B needs (for profiles/scripts/ab_lib.sh)."""
import json
import subprocess
import sys

n = sys.argv[1] if len(sys.argv) > 1 else "64"
out = subprocess.run([sys.executable, "bench.py", "--batch", n, "--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--no-batch64"] + sys.argv[2:],
                     capture_output=True, text=True).stdout.strip().splitlines()[-1]
j = json.loads(out)
r = j["roofline"]
print("batch", n, "clips/s %.2f" % j["value"], j["stage_ms"], "attn frac", r["frac"], "attn launch us", r.get("avg_launch_us"),
      "step ms", r.get("decode_step", {}).get("ms"), "linear launch us", r.get("linear_layers", {}).get("avg_launch_us"))
