cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r3full2; rm -rf $O; mkdir -p $O
timeout -k 5 900 python3 -m pytest tests -m gpu -x -q --durations=12 > $O/pytest.log 2>&1; tail -22 $O/pytest.log
timeout -k 5 300 python3 bench.py > $O/bench.json 2> $O/bench.err; python3 - <<PY
import json
d=json.load(open('$O/bench.json'))
print('ms_per_step', d['ms_per_step'], 'setup', d['config']['setup_ms'], 'cold', d['config']['cold_step_ms'], 'asm', d['config']['assemble_ms'], 'solve', d['config']['solve_ms'])
print('api', d['config'].get('api'))
print('parity', d.get('parity_vs_port'), 'frac', d['roofline']['frac'], d['roofline']['fine_level'], d['roofline']['fine_level_in_solver'])
PY
timeout -k 5 300 python3 bench.py --cells 512 --steps 2 --warmup 1 --no-cpu-baseline --skip-csr --no-api-wall > $O/bench512.json 2> $O/bench512.err; tail -c 600 $O/bench512.err; python3 - <<PY
import json
d=json.load(open('$O/bench512.json'))
print('512^3: ms_per_step', d['ms_per_step'], 'DoF/s', d['value'], 'dofs', d['config']['dofs'], 'sweeps', d['config']['picard_sweeps'], d['config']['inner_cg_iterations'], 'setup', d['config']['setup_ms'], 'frac', d['roofline']['frac'])
PY
