cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/trcols; rm -rf $O; mkdir -p $O
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/prof -o tr -- python3 bench.py --steps 1 --warmup 1 --cells 256 --no-cpu-baseline --skip-fine-bench --skip-csr > $O/bench.json 2> $O/bench.err || { tail -5 $O/bench.err; exit 1; }
f=$(find $O/prof -name '*kernel_trace.csv' | head -1)
head -1 $f > $O/cols.txt
tail -400 $f > $O/tail.csv
rm -rf $O/prof
