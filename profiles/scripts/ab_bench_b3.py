"""bench.py --batch N --no-extras, decode ms per call only (for profiles/scripts/ab_lib.sh). usage: ab_bench_b3.py [batch 3]"""
import json, os, subprocess, sys
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
B = sys.argv[1] if len(sys.argv) > 1 else "3"
out = subprocess.run([sys.executable, os.path.join(R, "bench.py"), "--batch", B, "--steps", "5", "--warmup", "2", "--no-extras"], capture_output=True, text=True).stdout
d = json.loads([l for l in out.splitlines() if l.startswith("{")][-1])
print("batch", B, "clips/s", d["value"], "decode ms", d["stage_ms"]["decode_ms"])
