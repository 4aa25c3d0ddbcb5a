# round 6: the record set of the final binary (one gpurun call); copies go to profiles/r06_*
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r06f
rm -rf $O; mkdir -p $O
python bench.py > $O/bench_n1.json.log 2> $O/bench_n1.stderr
A="--steps 8 --warmup 2 --no-also-configs --no-cpu-baseline --no-f32-mode"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_run -- python3 bench.py $A > $O/bench_n1_under_rocprof.json.log 2>/dev/null
MULAN_SIDE_STREAM=0 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_serial -- python3 bench.py $A > $O/bench_n1_serial_under_rocprof.json.log 2>/dev/null
for d in prof_run prof_serial; do f=$(find $O/$d -name "*kernel_stats.csv" | head -1); cp $f $O/${d}_kernel_stats.csv; done
rm -rf $O/prof_run $O/prof_serial
# PMC passes: one counter group per pass (never combined with a trace domain other than --kernel-trace)
for c in FETCH_SIZE WRITE_SIZE "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
  n=$(echo $c | cut -d' ' -f1)
  rocprofv3 --kernel-trace --pmc $c -d $O/pmc/conv_$n -o pmc --output-format csv -- python3 tools/pmc_conv.py > /dev/null 2>&1
  rocprofv3 --kernel-trace --pmc $c -d $O/pmc/more_$n -o pmc --output-format csv -- python3 tools/pmc_more.py > /dev/null 2>&1
done
mkdir -p $O/pmc_conv $O/pmc_more
for n in FETCH_SIZE WRITE_SIZE SQ_VALU_MFMA_BUSY_CYCLES; do
  for k in conv more; do f=$(find $O/pmc/${k}_$n -name "*counter_collection.csv" | head -1); [ -n "$f" ] && cp $f $O/pmc_$k/${n}_counter_collection.csv; done
done
rm -rf $O/pmc
python3 tools/pmc_conv.py --parse $O/pmc_conv > $O/pmc_conv3x3_f16x3.json
python3 tools/pmc_more.py --parse $O/pmc_more > $O/pmc_wgrad_groupnorm.json
python tools/launch_census.py > $O/launch_census.log 2>&1
python tools/wgrad_w8_ab.py --shapes 128x128 256x128 256x256 --tunes 29=1 29=2 "" 29=2 "" 29=1 > $O/wgrad_ab.log 2>&1
tail -c 300 $O/bench_n1.json.log; ls -la $O
