#!/bin/bash
# usage (on the GPU box, through gpurun): tools/profile_round.sh <tag> [encode|decode|long_encode|long_decode|all]
#   -> gpurun_out/<tag>_<leg>_{bench.json,bench_under_rocprof.json,kernel_stats.csv,pmc.json}
# One leg = the bench line, rocprofv3 --kernel-trace --stats of the same command, and the --pmc passes (each in its own
# run, with --kernel-trace only: MI355X_MICROARCH.md, HBM / rocprofv3 section).  The pmc JSON is stamped with the sha of the
# kernel sources (bench.py kernel_source_sha): bench.py reports a pass only for the build it was taken from.
# Copy what should be judged into profiles/ and name it in profiles/CURRENT.json.
set -e
tag=$1
legs=${2:-all}
[ "$legs" = all ] && legs="encode decode"
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for leg in $legs; do
  case $leg in
    encode)      args="";                               kern=cbc_encode_blocks_kernel ;;
    decode)      args="--mode decode";                  kern=cbc_decode_blocks_kernel ;;
    long_encode) args="--workload cfg5";                kern=cbc_long_encode_kernel ;;
    long_decode) args="--workload cfg5 --mode decode";  kern=cbc_long_decode_kernel ;;
    *) echo "unknown leg $leg"; exit 2 ;;
  esac
  P=$O/${tag}_${leg}
  python3 $R/bench.py $args --no-e2e > ${P}_bench.log 2>&1; tail -1 ${P}_bench.log > ${P}_bench.json
  rm -rf $O/prof_${tag}_${leg}
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_${tag}_${leg} -o p -- python3 $R/bench.py $args --steps 5 --warmup 2 --no-cpu-baseline --no-e2e > ${P}_rp.log 2>&1
  tail -1 ${P}_rp.log > ${P}_bench_under_rocprof.json
  cp $(find $O/prof_${tag}_${leg} -name '*kernel_stats.csv' | head -1) ${P}_kernel_stats.csv
  for c in FETCH_SIZE WRITE_SIZE "SQ_INSTS_SALU SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAVES" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VALU" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAIT_INST_LDS"; do
    d=$O/pmc_${tag}_${leg}_$(echo $c | tr ' ' '_' | cut -c1-20)
    rm -rf $d
    rocprofv3 --pmc $c --kernel-trace --output-format csv -d $d -o p -- python3 $R/bench.py $args --steps 2 --warmup 1 --no-cpu-baseline --no-e2e > $d.log 2>&1 || echo "counter pass '$c' failed (see $d.log)"
  done
  python3 - "$tag" "$leg" "$kern" <<'PY'
import csv, glob, sys, os, json, collections
tag, leg, kern = sys.argv[1:4]
R = os.environ["GRAFT_REPO_ROOT"]; O = R + "/gpurun_out"
sys.path.insert(0, R)
import bench
agg = collections.defaultdict(list); info = {}
for f in glob.glob(O + "/pmc_%s_%s_*/**/*counter_collection.csv" % (tag, leg), recursive=True):
    for row in csv.DictReader(open(f)):
        name = row["Kernel_Name"]
        if name.startswith(kern):                            # both register budgets of the encode kernel (.._w6) count
            agg[row["Counter_Name"]].append(float(row["Counter_Value"]))
            info = {k: row[k] for k in ("Grid_Size", "Workgroup_Size", "LDS_Block_Size", "Scratch_Size", "VGPR_Count", "SGPR_Count") if k in row}
res = {"kernel": kern, "kernel_source_sha": bench.kernel_source_sha(leg), "FETCH_SIZE": agg.pop("FETCH_SIZE", []), "WRITE_SIZE": agg.pop("WRITE_SIZE", []),
       "dispatch_info": info, "SQ_per_launch": {k: sum(v) / len(v) for k, v in sorted(agg.items())}}
try:
    ks = list(csv.DictReader(open(O + "/%s_%s_kernel_stats.csv" % (tag, leg))))
    res["kernel_ms"] = [float(r["AverageNs"]) / 1e6 for r in ks if r["Name"].startswith(kern)][0]
except Exception as e:
    res["kernel_ms_error"] = str(e)
if res["FETCH_SIZE"] and res["WRITE_SIZE"]:
    f = sum(res["FETCH_SIZE"]) / len(res["FETCH_SIZE"]); w = sum(res["WRITE_SIZE"]) / len(res["WRITE_SIZE"])
    res["hbm_bytes_per_launch"] = int(f * 1024 * 2 + w * 1024)
    res["note"] = "hbm_bytes = FETCH_SIZE*1024*2 + WRITE_SIZE*1024 (gfx950 FETCH_SIZE reports half of a coalesced streaming read; see profiles/README.md)"
json.dump(res, open(O + "/%s_%s_pmc.json" % (tag, leg), "w"), indent=1)
print(json.dumps(res, indent=1)[:1800])
PY
  cut -c1-400 ${P}_bench.json; head -3 ${P}_kernel_stats.csv
done
