#!/bin/bash
# A/B of REC_LINE_SKEW (pitch of the paint's halo-record lines) ON THE GPU BOX: rebuilds the library twice.
set -e
R=$PWD
F="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -munsafe-fp-atomics -Wall -Wno-unused-function"
out=$R/gpurun_out/recskew.txt
summ() {
python3 -c "
import sys, re
rows = [l for l in sys.stdin if l.startswith('trial')]
st = [float(re.search(r'step ([0-9.]+)', l).group(1)) for l in rows]
fi = [float(re.search(r\"fill': ([0-9.]+)\", l).group(1)) for l in rows]
de = [float(re.search(r\"deposit': ([0-9.]+)\", l).group(1)) for l in rows]
ff = [float(re.search(r'fft ([0-9.]+)', l).group(1)) for l in rows]
print('$1: step min %.3f mean %.3f max %.3f | fill %.3f | deposit %.3f | fft %.3f' % (min(st), sum(st) / len(st), max(st), sum(fi) / len(fi), sum(de) / len(de), sum(ff) / len(ff)))"
}
for skew in ${SKEWS:-32 0 32 0}; do
    make -C astrild_amd/csrc clean > /dev/null
    make -j12 -C astrild_amd/csrc CXXFLAGS="$F -DREC_LINE_SKEW=$skew" > $R/gpurun_out/make_$skew.log 2>&1
    timeout -k 10 200 python3 scripts/micro/placement_step.py 5 2> /dev/null | summ "REC_LINE_SKEW=$skew" >> $out
done
cat $out
