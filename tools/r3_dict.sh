cd $GRAFT_REPO_ROOT
O=gpurun_out/r3dict; mkdir -p $O
timeout -k 5 300 python -m pytest tests/test_gpu_parity.py -x -q -k "row_dictionary" > $O/test.log 2>&1; echo "pytest exit $?" >> $O/test.log; tail -15 $O/test.log
grep -q "pytest exit 0" $O/test.log || exit 1
timeout -k 5 200 python tools/r3_dict_probe.py 256 30 > $O/probe.txt 2>&1; tail -20 $O/probe.txt
B="bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-api-wall --skip-csr"

timeout -k 5 200 python $B > $O/bench_dict.json 2> $O/bench_dict.err && cut -c1-400 $O/bench_dict.json
