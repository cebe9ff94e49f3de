#!/bin/bash
# usage: tools/profile_round.sh <tag>   -> gpurun_out/<tag>_{bench.json,bench_under_rocprof.json,kernel_stats.csv,pmc_hbm.json,...}
set -e
tag=$1
leg=${2:-all}          # encode | decode | all  (one gpurun call holds 1200 s: run the two legs in separate calls)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
if [ "$leg" != decode ]; then
python3 $R/bench.py > $O/${tag}_bench.log 2>&1; tail -1 $O/${tag}_bench.log > $O/${tag}_bench.json
rm -rf $O/prof_$tag
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$tag -o p -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $O/${tag}_rp.log 2>&1
tail -1 $O/${tag}_rp.log > $O/${tag}_bench_under_rocprof.json
cp $(find $O/prof_$tag -name '*kernel_stats.csv' | head -1) $O/${tag}_kernel_stats.csv
for c in FETCH_SIZE WRITE_SIZE "SQ_INSTS_SALU SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAVES" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VALU" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"; do
  d=$O/pmc_${tag}_$(echo $c | tr ' ' '_' | cut -c1-20)
  rm -rf $d
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $d -o p -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $d.log 2>&1 || echo "counter pass '$c' failed (see $d.log)"
done
python3 - "$tag" <<'PY'
import csv, glob, sys, os, json, collections
tag = sys.argv[1]; O = os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out"
agg = collections.defaultdict(list); info = {}
for f in glob.glob(O + "/pmc_%s_*/**/*counter_collection.csv" % tag, recursive=True):
    for row in csv.DictReader(open(f)):
        if "cbc_encode_blocks_kernel" in row["Kernel_Name"]:
            agg[row["Counter_Name"]].append(float(row["Counter_Value"]))
            info = {k: row[k] for k in ("Grid_Size", "Workgroup_Size", "LDS_Block_Size", "Scratch_Size", "VGPR_Count", "SGPR_Count") if k in row}
res = {"FETCH_SIZE": agg.pop("FETCH_SIZE", []), "WRITE_SIZE": agg.pop("WRITE_SIZE", []), "dispatch_info": info,
       "SQ_per_launch": {k: sum(v) / len(v) for k, v in sorted(agg.items())}}
try:
    ks = list(csv.DictReader(open(O + "/%s_kernel_stats.csv" % tag)))
    res["kernel_ms"] = [float(r["AverageNs"]) / 1e6 for r in ks if "cbc_encode_blocks_kernel" in r["Name"]][0]
except Exception:
    pass
if res["FETCH_SIZE"] and res["WRITE_SIZE"]:
    f = sum(res["FETCH_SIZE"]) / len(res["FETCH_SIZE"]); w = sum(res["WRITE_SIZE"]) / len(res["WRITE_SIZE"])
    res["hbm_bytes_per_launch"] = int(f * 1024 * 2 + w * 1024)
    res["note"] = "hbm_bytes = FETCH_SIZE*1024*2 + WRITE_SIZE*1024 (gfx950 FETCH_SIZE reports half of a coalesced streaming read; see profiles/README.md)"
json.dump(res, open(O + "/%s_pmc_hbm.json" % tag, "w"), indent=1)
print(json.dumps(res, indent=1)[:1500])
PY
cat $O/${tag}_bench.json | cut -c1-400; head -3 $O/${tag}_kernel_stats.csv
fi
[ "$leg" = encode ] && exit 0
# decode leg
cd /tmp
python3 $R/bench.py --mode decode --no-cpu-baseline > $O/${tag}_bench_decode.log 2>&1; tail -1 $O/${tag}_bench_decode.log > $O/${tag}_bench_decode.json
rm -rf $O/prof_${tag}_dec
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_${tag}_dec -o p -- python3 $R/bench.py --mode decode --steps 5 --warmup 2 --no-cpu-baseline > $O/${tag}_rp_dec.log 2>&1
cp $(find $O/prof_${tag}_dec -name '*kernel_stats.csv' | head -1) $O/${tag}_decode_kernel_stats.csv
cut -c1-300 $O/${tag}_bench_decode.json; head -3 $O/${tag}_decode_kernel_stats.csv

# decode leg: the same counter passes for cbc_decode_blocks_kernel (bench.py --mode decode reads *_decode_pmc.json)
for c in FETCH_SIZE WRITE_SIZE "SQ_INSTS_SALU SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAVES" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VALU"; do
  d=$O/pmcdec_${tag}_$(echo $c | tr ' ' '_' | cut -c1-20)
  rm -rf $d
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $d -o p -- python3 $R/bench.py --mode decode --steps 2 --warmup 1 --no-cpu-baseline > $d.log 2>&1
done
python3 - "$tag" <<'PY'
import csv, glob, sys, os, json, collections
tag = sys.argv[1]; O = os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out"
agg = collections.defaultdict(list); info = {}
for f in glob.glob(O + "/pmcdec_%s_*/**/*counter_collection.csv" % tag, recursive=True):
    for row in csv.DictReader(open(f)):
        if "cbc_decode_blocks_kernel" in row["Kernel_Name"]:
            agg[row["Counter_Name"]].append(float(row["Counter_Value"]))
            info = {k: row[k] for k in ("Grid_Size", "Workgroup_Size", "LDS_Block_Size", "Scratch_Size", "VGPR_Count", "SGPR_Count") if k in row}
res = {"FETCH_SIZE": agg.pop("FETCH_SIZE", []), "WRITE_SIZE": agg.pop("WRITE_SIZE", []), "dispatch_info": info,
       "SQ_per_launch": {k: sum(v) / len(v) for k, v in sorted(agg.items())}}
try:
    ks = list(csv.DictReader(open(O + "/%s_decode_kernel_stats.csv" % tag)))
    res["kernel_ms"] = [float(r["AverageNs"]) / 1e6 for r in ks if "cbc_decode_blocks_kernel" in r["Name"]][0]
except Exception:
    pass
if res["FETCH_SIZE"] and res["WRITE_SIZE"]:
    f = sum(res["FETCH_SIZE"]) / len(res["FETCH_SIZE"]); w = sum(res["WRITE_SIZE"]) / len(res["WRITE_SIZE"])
    res["hbm_bytes_per_launch"] = int(f * 1024 * 2 + w * 1024)
    res["note"] = "hbm_bytes = FETCH_SIZE*1024*2 + WRITE_SIZE*1024 (gfx950 FETCH_SIZE reports half of a coalesced streaming read; see profiles/README.md)"
json.dump(res, open(O + "/%s_decode_pmc.json" % tag, "w"), indent=1)
print(json.dumps(res["SQ_per_launch"], indent=1)[:600])
PY
