cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r06m
mkdir -p $O
A="--steps 8 --warmup 2 --no-also-configs --no-cpu-baseline --no-f32-mode"
MULAN_SIDE_STREAM=0 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_serial -- python3 bench.py $A > $O/bench_n1_serial_under_rocprof.json.log 2>/dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_run -- python3 bench.py $A > $O/bench_n1_under_rocprof.json.log 2>/dev/null
for d in prof_serial prof_run; do f=$(find $O/$d -name "*kernel_stats.csv" | head -1); cp $f $O/${d}_kernel_stats.csv; done
rm -rf $O/prof_serial $O/prof_run
head -30 $O/prof_serial_kernel_stats.csv | cut -c1-200
