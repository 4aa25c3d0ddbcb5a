cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r05pmc
rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/conv_FETCH_SIZE -o pmc --output-format csv -- python3 tools/pmc_conv.py > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/conv_WRITE_SIZE -o pmc --output-format csv -- python3 tools/pmc_conv.py > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -d $O/conv_MFMA -o pmc --output-format csv -- python3 tools/pmc_conv.py > /dev/null 2>&1
python3 tools/pmc_conv.py --parse $O > $O/pmc_conv3x3_f16x3.json
find $O -name "*counter_collection.csv" | head
python3 - <<'PY'
import json
d=json.load(open("gpurun_out/r05pmc/pmc_conv3x3_f16x3.json"))
for k,v in d["shapes"].items():
    print(k, round(v.get("avg_duration_us",0),1), round(v.get("mfma_util",0),4), round(v.get("traffic_over_algorithmic",0),3))
PY
