timeout -k 10 600 python -m pytest tests/test_bh_gpu.py -q -m gpu 2>&1 | tail -2
for spec in "8192 100" "65536 50" "262144 20" "1048576 10"; do set -- $spec; python3 tools/bh_ticks.py $1 $2 step; done
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/bhstats_1m -o bh -- python3 $GRAFT_REPO_ROOT/tools/bh_ticks.py 1048576 10 step > /dev/null 2>&1
f=$(find $GRAFT_REPO_ROOT/gpurun_out/bhstats_1m -name "*kernel_stats.csv" | head -1); head -12 $f | cut -c1-70,150-260
