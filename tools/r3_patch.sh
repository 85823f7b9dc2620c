cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r3patch; rm -rf $O; mkdir -p $O
timeout -k 5 200 python3 -m pytest tests -m gpu -x -q -k "sell_patch" > $O/pytest.log 2>&1; tail -5 $O/pytest.log
timeout -k 5 120 python3 tools/r3_patch_probe.py 256 30 > $O/times.txt 2>&1; cat $O/times.txt
timeout -k 5 150 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum --kernel-trace --output-format csv -d $O/tc -- python3 tools/r3_patch_probe.py 256 3 > $O/tc.out 2> $O/tc.err || tail -5 $O/tc.err
f=$(find $O/tc -name '*counter_collection.csv' | head -1); python3 tools/r3_pmc_by_order.py $f "k_spmv_sell" 23 > $O/tc.txt; cat $O/tc.txt
rm -rf $O/tc
