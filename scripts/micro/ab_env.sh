#!/bin/bash
# usage (ON THE GPU BOX): ab_env.sh VAR v1 v2 ...   - placement_step.py (5 placements) once per value, interleaved twice
R=$PWD
var=$1; shift
out=$R/gpurun_out/ab_$var.txt
for rep in 1 2; do
for v in "$@"; do
env $var=$v timeout -k 10 200 python3 scripts/micro/placement_step.py 5 2> /dev/null | python3 -c "
import sys, re
rows = [l for l in sys.stdin if l.startswith('trial')]
st = [float(re.search(r'step ([0-9.]+)', l).group(1)) for l in rows]
fi = [float(re.search(r\"fill': ([0-9.]+)\", l).group(1)) for l in rows]
de = [float(re.search(r\"deposit': ([0-9.]+)\", l).group(1)) for l in rows]
ff = [float(re.search(r'fft ([0-9.]+)', l).group(1)) for l in rows]
print('$var=$v: step min %.3f mean %.3f max %.3f | fill %.3f | deposit %.3f | fft %.3f' % (min(st), sum(st) / len(st), max(st), sum(fi) / len(fi), sum(de) / len(de), sum(ff) / len(ff)))" >> $out || exit 1
done
done
cat $out
