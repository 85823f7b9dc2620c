cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for k in 3 17 3 17 3 17; do
  timeout -k 10 120 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --skip-fine-bench --set spmv_kernel=$k > gpurun_out/ab_$k.json 2>gpurun_out/ab_$k.err || exit 1
  python -c "
import json; d=json.load(open('gpurun_out/ab_$k.json')); r=d['roofline']; print('kernel $k', round(d['ms_per_step'],1), 'ms; spmv frac', r['frac'], 'avg us', r['avg_launch_us'])"
done
