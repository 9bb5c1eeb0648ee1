#!/bin/bash
# C2 (2-D P1 Laplacian, ~1 M triangles) under rocprofv3: kernel trace of the default path (the whole CG as one persistent launch) and
# of the multi-launch path (PERSIST=0), plus the FETCH / WRITE counters of both.  usage: tools/profile_c2.sh <tag>
set -u
TAG=${1:-r2}
REPO=$(pwd)
OUT=$REPO/gpurun_out/prof_c2_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
for P in 1 0; do
  export PERSIST=$P
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_p$P -- python3 $REPO/tools/run_c2.py > $OUT/trace_p$P.log 2>&1
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/fetch_p$P -- python3 $REPO/tools/run_c2.py > $OUT/fetch_p$P.log 2>&1
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/write_p$P -- python3 $REPO/tools/run_c2.py > $OUT/write_p$P.log 2>&1
done
cd $REPO
python3 - $OUT > $OUT/summary.txt 2>&1 <<'PY'
import csv, glob, os, sys
from collections import defaultdict
out = sys.argv[1]
def find(d, pat):
    r = glob.glob(os.path.join(d, "**", pat), recursive=True)
    return r[0] if r else None
def short(n):
    return n.split("(")[0].replace("fdapde_hip::", "").replace("void ", "").strip()
for P in ("1", "0"):
    print(f"==== C2, persist={P} ({'one persistent launch per solve' if P == '1' else 'multi-launch fused-update CG'}) ====")
    print(open(os.path.join(out, f"trace_p{P}.log")).read().strip().splitlines()[-1])
    kt = find(os.path.join(out, f"trace_p{P}"), "*kernel_trace.csv")
    dur = defaultdict(list)
    for r in csv.DictReader(open(kt)):
        dur[short(r["Kernel_Name"])].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    tot = sum(sum(v) for v in dur.values())
    print(f"{'kernel':60s} {'calls':>7s} {'total_us':>12s} {'avg_us':>10s} {'%':>6s}")
    for k, v in sorted(dur.items(), key=lambda kv: -sum(kv[1]))[:8]:
        print(f"{k[:60]:60s} {len(v):7d} {sum(v):12.1f} {sum(v)/len(v):10.2f} {100*sum(v)/tot:6.2f}")
    for sub, ctr in ((f"fetch_p{P}", "FETCH_SIZE"), (f"write_p{P}", "WRITE_SIZE")):
        f = find(os.path.join(out, sub), "*counter_collection.csv")
        if not f:
            continue
        acc = defaultdict(list)
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == ctr and any(s in r["Kernel_Name"] for s in ("k_cg_persist", "k_spmv_team2", "k_cgf_update")):
                acc[short(r["Kernel_Name"])].append(float(r["Counter_Value"]))
        for k, v in acc.items():
            print(f"  {ctr} {k[:50]:50s} avg {sum(v)/len(v):12.1f} KB per dispatch (n={len(v)})")
PY
cat $OUT/summary.txt
find $OUT -name "*.csv" -size +4M -delete
