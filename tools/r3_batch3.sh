cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r3b3; rm -rf $O; mkdir -p $O
timeout -k 5 120 python3 tools/r3_stagger_probe.py 256 30 > $O/stagger.txt 2>&1; cat $O/stagger.txt
timeout -k 5 150 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum --kernel-trace --output-format csv -d $O/tc -- python3 tools/r3_stagger_probe.py 256 3 > $O/tc.out 2> $O/tc.err || tail -5 $O/tc.err
f=$(find $O/tc -name '*counter_collection.csv' | head -1); python3 tools/r3_pmc_by_order.py $f "k_spmv_sell" 23 > $O/stagger_tc.txt; cat $O/stagger_tc.txt; rm -rf $O/tc
timeout -k 5 120 python3 tools/r3_asm_probe.py 256 3 > $O/asm_xmap.txt 2>&1; cat $O/asm_xmap.txt
timeout -k 5 120 python3 tools/setup_probe.py > $O/setup.txt 2>&1; cat $O/setup.txt
timeout -k 5 200 python3 bench.py --cells 512 --steps 2 --warmup 1 --no-cpu-baseline --skip-csr --no-api-wall > $O/bench512.json 2> $O/bench512.err; tail -c 400 $O/bench512.err; python3 - <<PY
import json
d=json.load(open('$O/bench512.json'))
print('512^3: ms_per_step', d['ms_per_step'], 'DoF/s', d['value'], 'dofs', d['config']['dofs'], 'sweeps', d['config']['picard_sweeps'], d['config']['inner_cg_iterations'], 'setup', d['config']['setup_ms'], 'cold', d['config']['cold_step_ms'], 'frac', d['roofline']['frac'], d['roofline']['fine_level'])
PY
