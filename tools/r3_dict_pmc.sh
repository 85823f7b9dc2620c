cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/dictpmc; rm -rf $O; mkdir -p $O
run() { timeout -k 5 200 rocprofv3 --pmc $2 --kernel-trace --output-format csv -d $O/$1 -- python3 tools/r3_dict_pmc.py > $O/$1.out 2> $O/$1.err || { tail -5 $O/$1.err; return 1; }; f=$(find $O/$1 -name '*counter_collection.csv' | head -1); cp $f $O/$1.csv; rm -rf $O/$1; }
run fs "FETCH_SIZE" && run ws "WRITE_SIZE" && run tc "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum" && run sq1 "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU"
for c in fs ws tc sq1; do echo "== $c"; [ -f $O/$c.csv ] && python3 tools/r3_pmc_by_order.py $O/$c.csv "k_spmv" 12; done
