#!/usr/bin/env python3
"""print a few fields of bench.py's JSON line read from stdin:  python bench.py ... | python tools/bench_fields.py [label]"""
import json, sys
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
r = d.get("roofline") or {}
print(" ".join(sys.argv[1:]), "value", d["value"], "ms", d["ms_per_step"], "single", d.get("single_stream_ms_per_step"), "frac", r.get("frac"),
      "in_flight", r.get("frac_in_flight"))
