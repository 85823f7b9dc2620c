# round-4 evidence pass: default bench line, kernel stats + gaps, PMC traffic of the SpMV and assembly kernels
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r4final; rm -rf $O; mkdir -p $O
timeout -k 5 400 python3 bench.py > $O/bench.json 2> $O/bench.err; tail -c 300 $O/bench.err
B="bench.py --cells 256 --steps 1 --warmup 1 --no-cpu-baseline --no-api-wall --skip-fine-bench --skip-csr --no-configs"
timeout -k 5 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/ks -o ks -- python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-api-wall --skip-fine-bench --skip-csr --no-configs > $O/under_rocprof.json 2> $O/ks.err || { tail -5 $O/ks.err; exit 1; }
cp $(find $O/ks -name '*kernel_stats.csv' | head -1) $O/kernel_stats.csv
python3 tools/gap_hist.py $O/ks > $O/gaps256.txt
rm -rf $O/ks
timeout -k 5 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pf -- python3 $B > $O/pf.json 2> $O/pf.err || { tail -5 $O/pf.err; exit 1; }
timeout -k 5 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pw -- python3 $B > $O/pw.json 2> $O/pw.err || { tail -5 $O/pw.err; exit 1; }
# (the run holds a warm-up, a timed and an instrumented step: 3 steps)
python3 tools/pmc_summarize.py $O/pf $O/pw $O/pmc_spmv_dict.json k_spmv 3 k_spmv sym dict > /dev/null
python3 tools/pmc_summarize.py $O/pf $O/pw $O/pmc_asm.json k_asm_node2 3 k_asm_node2 > /dev/null
python3 tools/pmc_summarize.py $O/pf $O/pw $O/pmc_check.json k_n2_check 3 k_n2_check_general > /dev/null
rm -rf $O/pf $O/pw
ls $O
python3 - <<PY
import json
d=json.load(open('$O/bench.json'))
print('ms_per_step', d['ms_per_step'], 'value', d['value'], 'asm', d['config']['assemble_ms'], 'solve', d['config']['solve_ms'], 'cold', d['config']['cold_step_ms'], 'dict_build', d['config']['dict_build_ms'])
print('parity', d.get('parity_vs_port'), 'frac', d['roofline']['frac'], d['roofline']['traffic'], d['roofline']['fine_level'])
for k in ('pmc_spmv_dict','pmc_asm','pmc_check'):
    p=json.load(open('$O/%s.json' % k)); print(k, p['launches_per_step'], round(p['traffic_bytes_per_launch']/1e6,1), 'MB per launch')
PY
