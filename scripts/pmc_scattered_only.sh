#!/bin/bash
# Run ON THE GPU BOX: only the two PMC passes of the unordered paint (quick check between kernel variants).
# usage: scripts/pmc_scattered_only.sh <tag>
set -e
tag=${1:-x}
R=$PWD
out=$R/gpurun_out/pmc_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
lean="--cpu-sample 0 --kappa 0 --bispec 0 --legs 0"
rocprofv3 --pmc FETCH_SIZE -d $out/pmc_sfetch -o f --output-format csv -- python3 $R/bench.py $lean --order shuffled --steps 2 --warmup 1 > /dev/null 2> $out/pmc_sfetch.stderr
rocprofv3 --pmc WRITE_SIZE -d $out/pmc_swrite -o w --output-format csv -- python3 $R/bench.py $lean --order shuffled --steps 2 --warmup 1 > /dev/null 2> $out/pmc_swrite.stderr
cd $R
python3 scripts/pmc_traffic_json.py $out/pmc_sfetch $out/pmc_swrite $out/pmc_scattered_$tag.json --order shuffled
rm -rf $out/pmc_sfetch $out/pmc_swrite
python3 - <<PY
import json
d=json.load(open("$out/pmc_scattered_$tag.json"))
print({k:v for k,v in d.items() if k not in("kernels","_doc")})
for k,v in d["kernels"].items():
    if v["corrected_GB"]>0.5: print("  ",k[:60], round(2*v["FETCH_SIZE_KiB"]*1024/1e9,2),"R", round(v["WRITE_SIZE_KiB"]*1024/1e9,2),"W", v["launches"])
PY
