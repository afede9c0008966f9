#!/bin/bash
# Evidence run of a round on the GPU box (every step under its own time limit; stops at the first failure):
#   gpurun --timeout 1200 -- 'bash tools/profile_round.sh r3'
# then, back here:  python tools/refresh_profiles.py r3
set -e -o pipefail
tag=${1:-r3}
out=gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
B="bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline --no-other-configs"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof -o $tag -- python3 $B > $out/prof.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_serial -o ${tag}s -- python3 $B --serial > $out/prof_serial.log 2>&1
timeout -s KILL 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $out/pmc_fetch -o $tag -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-roofline --no-other-configs --serial > $out/pmc_fetch.log 2>&1
timeout -s KILL 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $out/pmc_write -o $tag -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-roofline --no-other-configs --serial > $out/pmc_write.log 2>&1
# the trace files are large and not needed back (only the stats and the counter tables)
rm -f $out/prof/*_kernel_trace.csv $out/prof_serial/*_kernel_trace.csv
# the bench lines read the dominant kernel's HBM traffic from profiles/pmc_traffic.json: rebuild it from the passes above first
python3 tools/pmc_traffic.py $out/pmc_fetch/${tag}_counter_collection.csv $out/pmc_write/${tag}_counter_collection.csv > profiles/pmc_traffic.json
timeout -k 10 300 python3 bench.py > $out/bench.json 2> $out/bench.err
for cfg in train_tsm_mingru infer_fwd infer_long_video; do
    timeout -k 10 300 python3 bench.py --config $cfg --no-cpu-baseline > $out/bench_$cfg.json 2> $out/bench_$cfg.err
done
timeout -k 10 120 python3 tools/bench_augment.py > $out/augment.txt 2>&1
# the evaluation forward (configuration infer_fwd) by kernel
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_infer -o ${tag}i -- python3 bench.py --config infer_fwd --steps 5 --warmup 2 --no-cpu-baseline --no-roofline > $out/prof_infer.log 2>&1
rm -f $out/prof_infer/*_kernel_trace.csv
echo profiles done
