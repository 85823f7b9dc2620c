# round-3 baseline: GPU parity tests, default bench line, kernel stats of the 256^3 step
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r3base; rm -rf $O; mkdir -p $O
timeout -k 10 600 python3 -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log; tail -3 $O/pytest.log
timeout -k 10 300 python3 bench.py > $O/bench.json 2> $O/bench.err && tail -c 3000 $O/bench.json
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/ks -o ks -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --skip-fine-bench --skip-csr > $O/under_rocprof.json 2> $O/ks.err || { tail -5 $O/ks.err; exit 1; }
cp $(find $O/ks -name '*kernel_stats.csv' | head -1) $O/kernel_stats.csv
rm -rf $O/ks
