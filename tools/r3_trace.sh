# kernel trace of two default 256^3 steps (per-level timing analysis: tools/r3_trace_levels.py)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/${TRACE_OUT:-trace3}; rm -rf $O; mkdir -p $O
timeout -k 5 200 rocprofv3 --kernel-trace --output-format csv -d $O/kt -o kt -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-api-wall --skip-fine-bench --skip-csr $EXTRA > $O/line.json 2> $O/kt.err || { tail -5 $O/kt.err; exit 1; }
cp $(find $O/kt -name '*kernel_trace.csv' | head -1) $O/kernel_trace.csv
rm -rf $O/kt
ls -la $O; cat $O/line.json | cut -c1-300
