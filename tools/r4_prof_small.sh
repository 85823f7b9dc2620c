# round 4: kernel trace of the 64^3 / 128^3 steps (BASELINE configs 2-3 sizes, Picard algorithm): busy time vs span, gaps
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for N in 64 128; do
O=gpurun_out/prof4s_$N; rm -rf $O; mkdir -p $O
timeout -k 5 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/ks -o ks -- python3 bench.py --cells $N --steps 10 --warmup 2 --no-cpu-baseline --no-api-wall --skip-fine-bench --skip-csr --no-configs "$@" > $O/line.json 2> $O/ks.err || { tail -5 $O/ks.err; exit 1; }
cp $(find $O/ks -name '*kernel_stats.csv' | head -1) $O/kernel_stats.csv
python3 tools/gap_hist.py $O/ks > $O/gaps.txt
rm -rf $O/ks
python3 -c "
import json,csv
d=json.load(open('$O/line.json')); print('N=$N ms_per_step',round(d['ms_per_step'],3),'sweeps',d['config']['picard_sweeps'],'inner',d['config']['inner_cg_iterations'])
rows=list(csv.DictReader(open('$O/kernel_stats.csv')))
tot=sum(float(r['TotalDurationNs']) for r in rows); calls=sum(int(r['Calls']) for r in rows)
print(' kernels busy ms per step (13 steps in trace):', round(tot/1e6/13,3), 'launches per step', calls//13)
for r in rows[:12]: print('  ', r['Name'][:60], r['Calls'], round(float(r['TotalDurationNs'])/1e6/13,3))
"
head -12 $O/gaps.txt
done
