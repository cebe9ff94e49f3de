#!/bin/bash
# usage: tools/pmc.sh <tag> "<counters pass 1>" ["<counters pass 2>" ...]   (each pass: space-separated counters)
cd /tmp && export TMPDIR=/tmp
tag=$1; shift
i=0
for set in "$@"; do
  i=$((i+1))
  out=$GRAFT_REPO_ROOT/gpurun_out/pmc_${tag}_$i
  rm -rf $out
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $out -o p -- python3 $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline $BENCH_ARGS > $out.log 2>&1 || { echo "pass $i failed"; tail -5 $out.log; exit 1; }
done
python3 - "$tag" <<'PY'
import csv, glob, sys, os, json, collections
tag = sys.argv[1]
root = os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out"
agg = collections.defaultdict(list)
for f in glob.glob(root + "/pmc_%s_*/**/*counter_collection.csv" % tag, recursive=True):
    for row in csv.DictReader(open(f)):
        if os.environ.get("PMC_KERNEL", "cbc_encode_blocks_kernel") in row["Kernel_Name"]:
            agg[row["Counter_Name"]].append(float(row["Counter_Value"]))
res = {k: sum(v) / len(v) for k, v in agg.items()}
print(json.dumps(res, indent=1))
json.dump(res, open(root + "/pmc_%s.json" % tag, "w"), indent=1)
PY
