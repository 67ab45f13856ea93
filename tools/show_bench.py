#!/usr/bin/env python3
"""Prints value + stage times from a bench.py JSON line (file argument or stdin; helper for experiments)."""
import json
import sys
import os
args = sys.argv[1:]
src = sys.stdin
if args and os.path.exists(args[0]):  # a file path instead of stdin
    src = open(args.pop(0))
tag = args[0] if args else ""
for line in src:
    line = line.strip()
    if line.startswith("{"):
        d = json.loads(line)
        print(tag, d["value"], d["ms_per_step"], d.get("stage_ms_per_launch", d.get("stage_ms_per_step")))
        if "gated_match" in d:
            print(tag, "gated:", d["gated_match"])
