#!/bin/bash
# GPU box: HBM-side traffic of every extractor kernel (FETCH_SIZE x 2 + WRITE_SIZE, KB units, gfx950 correction) for one lane of 128 frames.
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/$1
rm -rf $O; mkdir -p $O
for set in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 200 rocprofv3 --pmc $set --output-format csv -d $O/$set -- python3 $R/bench.py --config c2 --streams 1 --batch 128 --steps 3 --warmup 1 --cpu-frames 0 > $O/$set.log 2>&1 || { echo "$set failed"; tail -3 $O/$set.log; }
done
python3 - $O <<'PY'
import csv, glob, sys, collections
o = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    for f in glob.glob(f"{o}/{c}/*/*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            if "amos::" in r["Kernel_Name"]:
                acc[r["Kernel_Name"].split("(")[0].replace("void ", "")][c].append(float(r["Counter_Value"]))
tot = 0
for k, v in sorted(acc.items()):
    n = len(v["FETCH_SIZE"]) // 3  # launches per pass (3 timed steps + 1 warm-up are in the trace; per-pass = count / 4)
    fe, wr = sum(v["FETCH_SIZE"]) / len(v["FETCH_SIZE"]), sum(v["WRITE_SIZE"]) / max(len(v["WRITE_SIZE"]), 1)
    per_launch = (2 * fe + wr) * 1024 / 1e6
    launches_per_pass = len(v["FETCH_SIZE"]) / 4.0
    tot += per_launch * launches_per_pass
    print(f"{k:45s} launches/pass {launches_per_pass:4.1f}  fetch x2 {2*fe*1024/1e6:8.1f} MB  write {wr*1024/1e6:8.1f} MB  per pass {per_launch*launches_per_pass:8.1f} MB")
print(f"total per pass of 128 frames: {tot:.1f} MB = {tot/128:.2f} MB per frame")
PY
