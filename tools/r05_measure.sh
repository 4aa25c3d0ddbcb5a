cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r05
mkdir -p $O
python bench.py > $O/bench_n1.json.log 2> $O/bench_n1.stderr
A="--steps 8 --warmup 2 --no-also-configs --no-cpu-baseline --no-f32-mode"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_run -- python3 bench.py $A > $O/bench_n1_under_rocprof.json.log 2>/dev/null
MULAN_SIDE_STREAM=0 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_serial -- python3 bench.py $A > $O/bench_n1_serial_under_rocprof.json.log 2>/dev/null
for c in FETCH_SIZE WRITE_SIZE SQ_VALU_MFMA_BUSY_CYCLES; do
  rocprofv3 --kernel-trace --pmc $c -d $O/pmc_conv_$c -o pmc --output-format csv -- python3 tools/pmc_conv.py > /dev/null 2>&1
  rocprofv3 --kernel-trace --pmc $c -d $O/pmc_more_$c -o $c --output-format csv -- python3 tools/pmc_more.py > /dev/null 2>&1
done
find $O -name "*kernel_stats.csv" | head; find $O -name "*counter_collection.csv" | head -8
tail -c 600 $O/bench_n1.json.log
