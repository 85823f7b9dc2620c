# A/B of one option at 64/128/256: r2_ab.sh "<bench args A>" "<bench args B>"
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/ab
for n in 64 128 256; do
  for v in A B; do
    if [ $v = A ]; then extra="$1"; else extra="$2"; fi
    timeout -k 10 200 python bench.py --steps 8 --warmup 2 --cells $n --no-cpu-baseline --skip-fine-bench --skip-csr $extra > gpurun_out/ab/$v$n.json 2> gpurun_out/ab/$v$n.err || { tail -5 gpurun_out/ab/$v$n.err; exit 1; }
    python -c "
import json; d=json.load(open('gpurun_out/ab/$v$n.json')); c=d['config']; print('$v [$extra]', $n, round(d['ms_per_step'],3),'ms', c['picard_sweeps'], c['inner_cg_iterations'], 'asm', c['assemble_ms'], 'solve', c['solve_ms'])"
  done
done
