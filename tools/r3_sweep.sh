cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r3sweep; rm -rf $O; mkdir -p $O
timeout -k 5 200 python3 -m pytest tests -m gpu -x -q -k "breakdown or ilu or node_assembly" > $O/pytest_small.log 2>&1; tail -3 $O/pytest_small.log
timeout -k 5 600 python3 -m pytest tests/test_reference_sweep.py -m gpu -q --durations=8 > $O/pytest_sweep.log 2>&1; tail -25 $O/pytest_sweep.log
timeout -k 5 500 python3 tools/r3_reference_sweep.py $O both > $O/sweep.txt 2>&1; tail -5 $O/sweep.txt
