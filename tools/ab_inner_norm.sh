cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for cfg in "1 1e-1" "1 2e-1" "1 3e-1" "1 5e-1" "1 1e-1" "1 1.5e-1"; do
  set -- $cfg
  timeout -k 10 120 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --skip-fine-bench --inner-norm $1 --inner-reduction $2 > gpurun_out/abn.json 2>gpurun_out/abn.err || exit 1
  python -c "
import json; d=json.load(open('gpurun_out/abn.json')); c=d['config']; print('norm $1 red $2:', round(d['ms_per_step'],1), 'ms; sweeps', c['picard_sweeps'], 'inner', c['inner_cg_iterations'], 'res', c['final_residual'], 'fine launches', d['roofline']['fine_level_in_solver']['launches_per_step'])"
done
