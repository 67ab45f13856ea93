#!/usr/bin/env python3
"""Prints value + stage times from a bench.py JSON line on stdin (helper for experiments)."""
import json
import sys
tag = sys.argv[1] if len(sys.argv) > 1 else ""
for line in sys.stdin:
    line = line.strip()
    if line.startswith("{"):
        d = json.loads(line)
        print(tag, d["value"], d["ms_per_step"], d.get("stage_ms_per_launch", d.get("stage_ms_per_step")))
