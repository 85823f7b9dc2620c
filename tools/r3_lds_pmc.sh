cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r3ldspmc; rm -rf $O; mkdir -p $O
timeout -k 10 300 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum --kernel-trace --output-format csv -d $O/tc -- python3 tools/r3_lds_pmc.py > $O/tc.out 2> $O/tc.err || tail -5 $O/tc.err
f=$(find $O/tc -name '*counter_collection.csv' | head -1); python3 tools/r3_pmc_by_order.py $f "k_spmv_sell" 23 > $O/tc.txt; cat $O/tc.txt
timeout -k 10 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS --kernel-trace --output-format csv -d $O/sq -- python3 tools/r3_lds_pmc.py > $O/sq.out 2> $O/sq.err || tail -5 $O/sq.err
f=$(find $O/sq -name '*counter_collection.csv' | head -1); python3 tools/r3_pmc_by_order.py $f "k_spmv_sell" 23 > $O/sq.txt; cat $O/sq.txt
rm -rf $O/tc $O/sq
