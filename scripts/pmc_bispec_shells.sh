#!/bin/bash
# Run ON THE GPU BOX: per-dispatch traffic of the bispectrum's masked inverse passes.  usage: scripts/pmc_bispec_shells.sh <tag>
set -e
tag=${1:-x}
R=$PWD
out=$R/gpurun_out/pmc_bshell_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc FETCH_SIZE -d $out/f -o f --output-format csv -- python3 $R/scripts/bispec_shells_once.py > $out/f.stdout 2> $out/f.stderr
rocprofv3 --pmc WRITE_SIZE -d $out/w -o w --output-format csv -- python3 $R/scripts/bispec_shells_once.py > $out/w.stdout 2> $out/w.stderr
cd $R
python3 scripts/pmc_per_launch.py $out/f $out/w | tee $out/per_launch_$tag.txt
rm -rf $out/f $out/w
