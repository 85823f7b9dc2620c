# default bench line (with the CPU baseline) and a 4-rank gloo rehearsal of the 256^3 step on one GPU (exchange counts)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/def2; mkdir -p $O
timeout -k 10 500 python bench.py > $O/bench_default.json 2> $O/bench_default.err || { tail -5 $O/bench_default.err; exit 1; }
python -c "
import json; d=json.load(open('$O/bench_default.json')); print(round(d['ms_per_step'],2), d['value']/1e6, d['roofline']['frac'], d['cpu_baseline']['value']/1e6, d['config']['setup_ms'], d['config']['cold_step_ms'])"
PERPHIL_DIST_BACKEND=gloo timeout -k 10 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 4 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 4 --cells 256 --steps 1 --warmup 1 --no-cpu-baseline --skip-fine-bench --skip-csr > $O/bench_gloo4.json 2> $O/bench_gloo4.err || { tail -15 $O/bench_gloo4.err; exit 1; }
python -c "
import json; d=[json.loads(l) for l in open('$O/bench_gloo4.json') if l.startswith('{')][-1]; c=d['config']; print('gloo4', round(d['ms_per_step'],1), c['transport'], c['ranks_seen'], 'halo', c['halo_exchanges_per_step'], 'allreduce', c['allreduces_per_step'], c['picard_sweeps'], c['inner_cg_iterations'])"
