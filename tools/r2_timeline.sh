# kernel timeline of one bench step: r2_timeline.sh <tag> <cells> [bench args...]
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
tag=$1; n=$2; shift 2
O=gpurun_out/$tag; rm -rf $O; mkdir -p $O
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/prof -o tr -- python3 bench.py --steps 2 --warmup 1 --cells $n --no-cpu-baseline --skip-fine-bench --skip-csr "$@" > $O/bench.json 2> $O/bench.err || { tail -5 $O/bench.err; exit 1; }
python3 tools/step_timeline.py $O/prof 0 ${FINE_NS:-2e6} ${STEP:-2} > $O/timeline.txt   # step 2 = the last timed step of --steps 2 --warmup 1
tail -35 $O/timeline.txt
rm -rf $O/prof
