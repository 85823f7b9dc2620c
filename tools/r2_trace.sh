# kernel trace of the bench at one size: per-kernel stats and the idle-gap histogram of the timed steps
# usage: r2_trace.sh <tag> <cells> [bench args...]
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
tag=$1; n=$2; shift 2
O=gpurun_out/$tag; rm -rf $O; mkdir -p $O
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o tr -- python3 bench.py --steps 3 --warmup 2 --cells $n --no-cpu-baseline --skip-fine-bench --skip-csr "$@" > $O/bench.json 2> $O/bench.err || { tail -5 $O/bench.err; exit 1; }
python3 -c "
import json; d=json.load(open('$O/bench.json')); c=d['config']; print($n, round(d['ms_per_step'],2),'ms', c['picard_sweeps'], c['inner_cg_iterations'])"
f=$(find $O/prof -name '*kernel_stats.csv' | head -1)
cp $f $O/kernel_stats.csv
head -25 $O/kernel_stats.csv | cut -c1-160
python3 tools/gap_hist.py $O/prof > $O/gaps.txt; cat $O/gaps.txt
rm -rf $O/prof
