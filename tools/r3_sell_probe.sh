cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r3sell; rm -rf $O; mkdir -p $O
timeout -k 10 300 python3 tools/r3_sell_probe.py 30 > $O/times.txt 2>&1; cat $O/times.txt
timeout -k 10 400 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum --kernel-trace --output-format csv -d $O/tc -- python3 tools/r3_sell_probe.py 3 > $O/tc.out 2> $O/tc.err || tail -5 $O/tc.err
f=$(find $O/tc -name '*counter_collection.csv' | head -1); [ -n "$f" ] && python3 tools/r3_pmc_by_order.py $f k_spmv_sell 23 > $O/tc_by_cfg.txt; cat $O/tc_by_cfg.txt
rm -rf $O/tc
