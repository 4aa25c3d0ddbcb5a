cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r05f
mkdir -p $O
python bench.py > $O/bench_n1.json.log 2> $O/bench_n1.stderr
A="--steps 8 --warmup 2 --no-also-configs --no-cpu-baseline --no-f32-mode"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_run -- python3 bench.py $A > $O/bench_n1_under_rocprof.json.log 2>/dev/null
MULAN_SIDE_STREAM=0 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_serial -- python3 bench.py $A > $O/bench_n1_serial_under_rocprof.json.log 2>/dev/null
find $O -name "*kernel_stats.csv" | head
tail -c 400 $O/bench_n1.json.log
