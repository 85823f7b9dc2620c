cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r3full; rm -rf $O; mkdir -p $O
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; tail -5 $O/pytest.log
timeout -k 10 300 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum --kernel-trace --output-format csv -d $O/tc -- python3 tools/r3_asm_pmc.py > $O/tc.out 2> $O/tc.err || tail -5 $O/tc.err
f=$(find $O/tc -name '*counter_collection.csv' | head -1); [ -n "$f" ] && { python3 tools/r3_pmc_by_order.py $f "k_asm_node" 1 > $O/pmc_asm_node.txt; python3 tools/r3_pmc_by_order.py $f "k_asm_tile" 1 > $O/pmc_asm_tile.txt; }; cat $O/pmc_asm_node.txt $O/pmc_asm_tile.txt
rm -rf $O/tc
timeout -k 10 300 python3 bench.py > $O/bench.json 2> $O/bench.err; tail -c 1500 $O/bench.json
