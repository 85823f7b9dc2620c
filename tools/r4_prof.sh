# round-4 evidence pass: bench line under rocprofv3 --kernel-trace --stats (5 timed steps), kernel stats + launch gaps
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-prof4}; rm -rf $O; mkdir -p $O
shift
timeout -k 5 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/ks -o ks -- python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-api-wall --skip-fine-bench --skip-csr --no-configs "$@" > $O/under_rocprof.json 2> $O/ks.err || { tail -5 $O/ks.err; exit 1; }
cp $(find $O/ks -name '*kernel_stats.csv' | head -1) $O/kernel_stats.csv
python3 tools/gap_hist.py $O/ks > $O/gaps256.txt
rm -rf $O/ks
head -25 $O/kernel_stats.csv | cut -c1-200
