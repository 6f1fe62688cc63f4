#!/usr/bin/env python3
"""One short line from a bench.py JSON line on stdin (A/B runs): label, clips/s, stage ms, roofline frac, step ms."""
import json
import sys

j = json.loads(sys.stdin.read())
r = j["roofline"]
print(sys.argv[1] if len(sys.argv) > 1 else "", j["value"], j["stage_ms"], "frac", r["frac"], r.get("decode_step", {}).get("ms"))
