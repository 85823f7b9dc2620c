cd $GRAFT_REPO_ROOT
export PERPHIL_DIST_BACKEND=gloo
for thr in 0 40000 80000; do
  timeout -k 10 500 python bench.py --gpus 4 --cells 256 --steps 2 --warmup 1 --no-cpu-baseline --skip-fine-bench --skip-csr --set mg_replicate_rows_per_rank=$thr > gpurun_out/r4_rehearse4_$thr.json 2> gpurun_out/r4_rehearse4_$thr.err || { tail -5 gpurun_out/r4_rehearse4_$thr.err; exit 1; }
  python -c "
import json; d=json.loads([l for l in open('gpurun_out/r4_rehearse4_$thr.json') if l.startswith('{')][0]); c=d['config']
print('rows_per_rank threshold $thr:', 'halo', c['halo_exchanges_per_step'], 'allreduce', c['allreduces_per_step'], 'split', c['split_products_per_step'], 'sweeps', c['picard_sweeps'], c['inner_cg_iterations'], 'ms', round(d['ms_per_step'],1), 'comm', c['comm'])
"
done
