#!/usr/bin/env python3
"""Regenerates include/amos_orb_pattern.h (numeric rBRIEF pattern table) from a file that holds
the 1024 integers of ORB's learned `bit_pattern_31` table, and checks the sha256 from SURVEY.md."""
import hashlib
import re
import struct
import sys

src = open(sys.argv[1], encoding="utf-8", errors="replace").read()
i = src.index("bit_pattern_31_[256*4]")
body = src[i:src.index("};", i)]
body = re.sub(r"/\*.*?\*/", "", body[body.index("{") + 1:], flags=re.S)
body = re.sub(r"//.*", "", body)
nums = [int(x) for x in re.findall(r"-?\d+", body)]
assert len(nums) == 1024
assert hashlib.sha256(struct.pack("<1024i", *nums)).hexdigest() == "7e645581387b82784797e8adddb9b6f0c12611859fda09ca8a9bec96d767a05f"
for k in range(0, 1024, 16):
    print("    " + ",".join("%d" % v for v in nums[k:k + 16]) + ",")
