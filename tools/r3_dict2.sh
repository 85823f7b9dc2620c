cd $GRAFT_REPO_ROOT
O=gpurun_out/r3dict2; mkdir -p $O
timeout -k 5 900 python -m pytest tests -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest exit $?" >> $O/pytest.log; tail -8 $O/pytest.log
grep -q "pytest exit 0" $O/pytest.log || exit 1
B="bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-api-wall --skip-csr"
timeout -k 5 200 python $B > $O/bench_dict.json 2> $O/bench_dict.err && cut -c1-300 $O/bench_dict.json
timeout -k 5 200 python $B --cells 128 > $O/bench128.json 2> $O/bench128.err && cut -c1-300 $O/bench128.json
timeout -k 5 200 python $B --cells 128 --set sell_dict=0 > $O/bench128p.json 2> $O/bench128p.err && cut -c1-300 $O/bench128p.json
timeout -k 5 200 python $B --cells 64 > $O/bench64.json 2> $O/bench64.err && cut -c1-300 $O/bench64.json
