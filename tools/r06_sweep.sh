O=gpurun_out/r06s; mkdir -p $O; : > $O/sweep.log
run() { MULAN_TUNE=$1 MULAN_SIDE_DEPTH=$2 python bench.py --no-cpu-baseline --no-f32-mode --no-also-configs --steps 20 --warmup 5 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('tune=$1 depth=$2', d['ms_per_step'], d['value'], d['last_train_bpd'])" >> $O/sweep.log; }
for rep in 1 2; do
  for t in "" 1=84 1=96 1=108 1=132; do run "$t" 6; done
  for d in 3 4 8 12 24; do run "" $d; done
done
cat $O/sweep.log
