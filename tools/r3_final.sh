cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r3final; rm -rf $O; mkdir -p $O
timeout -k 5 700 python3 -m pytest tests -m gpu -x -q --durations=10 > $O/pytest.log 2>&1; tail -16 $O/pytest.log
timeout -k 5 300 python3 bench.py > $O/bench.json 2> $O/bench.err; tail -c 300 $O/bench.err; python3 - <<PY
import json
d=json.load(open('$O/bench.json'))
print('ms_per_step', d['ms_per_step'], 'value', d['value'], 'context', d['config']['context_ms'], 'setup', d['config']['setup_ms'], 'cold', d['config']['cold_step_ms'], 'asm', d['config']['assemble_ms'], 'solve', d['config']['solve_ms'])
print('api', d['config'].get('api'))
print('parity', d.get('parity_vs_port'), 'frac', d['roofline']['frac'], d['roofline']['traffic'], d['roofline']['fine_level'], d['roofline']['fine_level_in_solver'])
print('cpu', d['cpu_baseline']['value'], d['cpu_baseline']['cores'])
PY
bash tools/r3_profile.sh > $O/profile.log 2>&1; tail -3 $O/profile.log
timeout -k 5 400 python3 tools/r3_conditioning_3d.py $O > $O/cond3d.txt 2>&1; cat $O/cond3d.txt
