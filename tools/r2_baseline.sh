# round-2 baseline: step times at 64/128/256 and a kernel trace at 128 (launch counts, busy fraction)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r2base; mkdir -p $O
for n in 64 128 256; do
  timeout -k 10 200 python bench.py --steps 5 --warmup 2 --cells $n --no-cpu-baseline > $O/bench_$n.json 2> $O/bench_$n.err || exit 1
  python -c "
import json; d=json.load(open('$O/bench_$n.json')); c=d['config']; r=d['roofline']; print($n, round(d['ms_per_step'],2),'ms', c['picard_sweeps'], c['inner_cg_iterations'], 'asm', c['assemble_ms'], 'solve', c['solve_ms'], 'launches', r['launches_per_step'], 'frac', r['frac'], r.get('fine_level_in_solver'))"
done
