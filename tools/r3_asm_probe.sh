cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r3asm; rm -rf $O; mkdir -p $O
timeout -k 10 300 python3 tools/r3_asm_probe.py 256 3 > $O/times.txt 2>&1; cat $O/times.txt
timeout -k 10 400 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum --kernel-trace --output-format csv -d $O/tc -- python3 tools/r3_asm_probe.py 256 1 > $O/tc.out 2> $O/tc.err || tail -5 $O/tc.err
f=$(find $O/tc -name '*counter_collection.csv' | head -1); [ -n "$f" ] && python3 tools/r3_pmc_by_order.py $f "k_asm_tile<3>" 4 > $O/tc_by_cfg.txt; cat $O/tc_by_cfg.txt
rm -rf $O/tc
timeout -k 10 600 python3 -m pytest tests -m gpu -x -q -k "fused_assembly or bench_gpus or stencil_ell or mid_size" > $O/pytest.log 2>&1; tail -3 $O/pytest.log
