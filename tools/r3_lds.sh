cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r3lds; rm -rf $O; mkdir -p $O
timeout -k 10 600 python3 -m pytest tests -m gpu -x -q -k "sell_lds or stencil_ell" > $O/pytest.log 2>&1; tail -15 $O/pytest.log
timeout -k 10 300 python3 tools/r3_lds_probe.py > $O/times.txt 2>&1; cat $O/times.txt
