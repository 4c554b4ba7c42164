#!/bin/bash
# Run ON THE GPU BOX (through gpurun): bench line, rocprofv3 kernel stats and the two PMC passes of the same
# command; everything lands in gpurun_out/profiles_<tag>/ for copying into profiles/.
# usage: scripts/refresh_profiles.sh <tag> [a|b|all]   (two halves: one gpurun call holds ~20 minutes)
set -e
tag=${1:-r05}
part=${2:-all}
R=$PWD
out=$R/gpurun_out/profiles_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
lean="--cpu-sample 0 --kappa 0 --bispec 0 --legs 0"
if [ "$part" != "b" ]; then
rocprofv3 --kernel-trace --stats -d $out/stats -o s --output-format csv -- python3 $R/bench.py $lean > $out/${tag}_bench_under_rocprof.json 2> $out/rocprof_stats.stderr
rocprofv3 --pmc FETCH_SIZE -d $out/pmc_fetch -o f --output-format csv -- python3 $R/bench.py $lean --steps 2 --warmup 1 > /dev/null 2> $out/pmc_fetch.stderr
rocprofv3 --pmc WRITE_SIZE -d $out/pmc_write -o w --output-format csv -- python3 $R/bench.py $lean --steps 2 --warmup 1 > /dev/null 2> $out/pmc_write.stderr
# the secondary legs (kappa pipeline with its rocFFT kernels, bispectrum, shuffled / TSC paints): kernel stats only
rocprofv3 --kernel-trace --stats -d $out/stats_legs -o s --output-format csv -- python3 $R/bench.py --cpu-sample 0 --steps 3 --warmup 1 > $out/${tag}_legs_under_rocprof.json 2> $out/rocprof_legs.stderr
fi
if [ "$part" = "a" ]; then ls -la $out; exit 0; fi
# the bispectrum leg's real traffic (its roofline fraction in the bench line is computed from this, not from the
# unpruned algorithmic bytes) and the SQ / LDS counters of the paint kernels (what bounds them)
rocprofv3 --pmc FETCH_SIZE -d $out/pmc_bfetch -o f --output-format csv -- python3 $R/bench.py --cpu-sample 0 --kappa 0 --legs 0 --steps 1 --warmup 0 > /dev/null 2> $out/pmc_bfetch.stderr
rocprofv3 --pmc WRITE_SIZE -d $out/pmc_bwrite -o w --output-format csv -- python3 $R/bench.py --cpu-sample 0 --kappa 0 --legs 0 --steps 1 --warmup 0 > /dev/null 2> $out/pmc_bwrite.stderr
# the unordered (scattered) paint: level A, level B and the walk over stray copies (verdict r3 item 4: its PMC table)
rocprofv3 --pmc FETCH_SIZE -d $out/pmc_sfetch -o f --output-format csv -- python3 $R/bench.py $lean --order shuffled --steps 2 --warmup 1 > /dev/null 2> $out/pmc_sfetch.stderr
rocprofv3 --pmc WRITE_SIZE -d $out/pmc_swrite -o w --output-format csv -- python3 $R/bench.py $lean --order shuffled --steps 2 --warmup 1 > /dev/null 2> $out/pmc_swrite.stderr
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d $out/pmc_sq -o q --output-format csv -- python3 $R/scripts/perf_paint.py 1024 cic,tsc natural 1 > $out/pmc_sq.stdout 2> $out/pmc_sq.stderr
# per-dispatch traffic of the bispectrum's masked inverse passes against the pruning model
rocprofv3 --pmc FETCH_SIZE -d $out/pmc_shf -o f --output-format csv -- python3 $R/scripts/bispec_shells_once.py > /dev/null 2> $out/pmc_shf.stderr
rocprofv3 --pmc WRITE_SIZE -d $out/pmc_shw -o w --output-format csv -- python3 $R/scripts/bispec_shells_once.py > /dev/null 2> $out/pmc_shw.stderr
cd $R
python3 scripts/pmc_per_launch.py $out/pmc_shf $out/pmc_shw > $out/${tag}_bispec_pruning.txt
python3 scripts/pmc_summary.py $out/pmc_sq > $out/${tag}_paint_sq_counters.txt
python3 scripts/pmc_bispec_json.py $out/pmc_bfetch $out/pmc_bwrite $out/${tag}_pmc_bispectrum.json
mkdir -p profiles && cp $out/${tag}_pmc_bispectrum.json profiles/${tag}_pmc_bispectrum.json
python3 scripts/pmc_traffic_json.py $out/pmc_sfetch $out/pmc_swrite $out/${tag}_pmc_scattered.json --order shuffled
cp $out/${tag}_pmc_scattered.json profiles/${tag}_pmc_scattered.json
python3 scripts/pmc_traffic_json.py $out/pmc_fetch $out/pmc_write $out/${tag}_pmc_traffic.json
# the plain bench line last: it quotes the PMC traffic just measured (same paint source, checked by hash)
mkdir -p profiles && cp $out/${tag}_pmc_traffic.json profiles/${tag}_pmc_traffic.json
python3 bench.py > $out/${tag}_bench_1gpu.json 2> $out/bench.stderr
[ -d $out/stats ] && find $out/stats -name "*kernel_stats.csv" -exec cp {} $out/${tag}_bench_kernel_stats.csv \;
[ -d $out/stats_legs ] && find $out/stats_legs -name "*kernel_stats.csv" -exec cp {} $out/${tag}_legs_kernel_stats.csv \;
# the raw traces are large: keep only the summaries
rm -rf $out/stats $out/stats_legs $out/pmc_fetch $out/pmc_write $out/pmc_bfetch $out/pmc_bwrite $out/pmc_sq $out/pmc_sfetch $out/pmc_swrite $out/pmc_shf $out/pmc_shw
ls -la $out
