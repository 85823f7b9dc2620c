# round-3 final evidence pass (row dictionaries on): GPU suite, default bench line, kernel stats, PMC traffic of the SpMV mix
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r3final2; rm -rf $O; mkdir -p $O
timeout -k 5 900 python -m pytest tests -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest exit $?" >> $O/pytest.log; tail -4 $O/pytest.log
grep -q "pytest exit 0" $O/pytest.log || exit 1
timeout -k 5 600 python bench.py > $O/bench.json 2> $O/bench.err || { tail -5 $O/bench.err; exit 1; }
cut -c1-260 $O/bench.json
B="bench.py --cells 256 --steps 1 --warmup 0 --no-cpu-baseline --no-api-wall --skip-fine-bench --skip-csr"
timeout -k 5 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/ks -o ks -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-api-wall --skip-fine-bench --skip-csr > $O/under_rocprof.json 2> $O/ks.err || { tail -5 $O/ks.err; exit 1; }
cp $(find $O/ks -name '*kernel_stats.csv' | head -1) $O/kernel_stats.csv
python3 tools/gap_hist.py $O/ks > $O/gaps256.txt
timeout -k 5 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pf -- python3 $B > $O/pf.json 2> $O/pf.err || { tail -5 $O/pf.err; exit 1; }
timeout -k 5 200 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pw -- python3 $B > $O/pw.json 2> $O/pw.err || { tail -5 $O/pw.err; exit 1; }
python3 tools/pmc_summarize.py $O/pf $O/pw $O/pmc_spmv_dict.json k_spmv 2 k_spmv sym dict > /dev/null
python3 tools/pmc_summarize.py $O/pf $O/pw $O/pmc_asm.json k_asm_node 2 k_asm_node > /dev/null
python3 tools/pmc_summarize.py $O/pf $O/pw $O/pmc_dict_verify.json k_dict_verify 2 k_dict_verify > /dev/null
rm -rf $O/ks $O/pf $O/pw
ls -la $O
