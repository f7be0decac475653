# kernel trace of the bench at b = $1 (default 32) -> gpurun_out/trace_b$1/
R=${GRAFT_REPO_ROOT:-/root/repo}
B=${1:-32}
O=$R/gpurun_out/trace_b$B
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d $O --output-format csv -- python3 $R/bench.py --steps 6 --warmup 3 --global-batch $B --no-cpu-baseline --no-kernel-table > $O/bench.json 2> $O/err.log || { tail -5 $O/err.log; exit 1; }
cat $O/bench.json | cut -c1-200
F=$(find $O -name "*kernel_trace.csv" | head -1)
python3 $R/tools/timeline.py $F 4
