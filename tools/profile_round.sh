#!/bin/bash
# One round's profile set for bench.py (run on the GPU box from the repo root):
#   tools/profile_round.sh r02 [cfg2]
# 1. rocprofv3 --kernel-trace --stats      -> profiles/<tag>_kernel_stats.csv      (per-kernel durations, one stream)
# 2. rocprofv3 --pmc FETCH_SIZE, WRITE_SIZE (separate passes, no trace domains) -> profiles/<tag>_pmc_traffic.txt + pmc_traffic.json
# 3. the default bench line (eager, two streams, with cpu_baseline)                      -> profiles/<tag>_bench_default.json
# Steps are joined with &&: a failed or killed GPU step starts no further one.
set -o pipefail
tag=$1; wl=${2:-cfg2}
root=$(pwd)
out=$root/gpurun_out/prof_$tag
mkdir -p $out $root/profiles
export TMPDIR=/tmp
flags="--workload $wl --steps 3 --warmup 2 --eager --no-overlap --no-cpu-baseline --no-kernel-timing --no-render-forward"
cd /tmp &&
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -o s -- python3 $root/bench.py $flags > $out/stats.log 2>&1 &&
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/pmc_fetch -o p -- python3 $root/bench.py $flags > $out/fetch.log 2>&1 &&
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/pmc_write -o p -- python3 $root/bench.py $flags > $out/write.log 2>&1 &&
cd $root &&
cp "$(ls $out/stats/*kernel_stats.csv $out/stats/*/*kernel_stats.csv 2>/dev/null | head -1)" profiles/${tag}_kernel_stats.csv &&
python3 tools/pmc_summary.py $out/pmc_fetch $out/pmc_write $wl $tag 5 > $out/pmc_summary.log 2>&1 &&
cp profiles/${tag}_pmc_traffic.txt profiles/${tag}_kernel_stats.csv profiles/pmc_traffic.json gpurun_out/ &&
python3 bench.py --workload $wl > gpurun_out/${tag}_bench_default.json 2> $out/bench.err &&
cp gpurun_out/${tag}_bench_default.json profiles/
echo "profile_round rc=$?"
