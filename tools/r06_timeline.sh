cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r06t
mkdir -p $O
rocprofv3 --kernel-trace --output-format csv -d $O/kt -o kt -- python3 bench.py --steps 6 --warmup 3 --no-also-configs --no-cpu-baseline --no-f32-mode > $O/bench.log 2>/dev/null
f=$(find $O/kt -name "*kernel_trace.csv" | head -1)
python tools/kt_shorten.py $f $O/kt_tail.csv 8000
python tools/step_timeline.py $O/kt_tail.csv > $O/step_timeline.log 2>&1
rm -rf $O/kt
cat $O/step_timeline.log
