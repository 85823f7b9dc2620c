# round-3 evidence pass: default bench line, kernel stats, PMC traffic of the SpMV and assembly kernels, SQ counters
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/prof3; rm -rf $O; mkdir -p $O
B="bench.py --cells 256 --steps 1 --warmup 0 --no-cpu-baseline --no-api-wall --skip-fine-bench --skip-csr"
timeout -k 5 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/ks -o ks -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-api-wall --skip-fine-bench --skip-csr > $O/under_rocprof.json 2> $O/ks.err || { tail -5 $O/ks.err; exit 1; }
cp $(find $O/ks -name '*kernel_stats.csv' | head -1) $O/kernel_stats.csv
python3 tools/gap_hist.py $O/ks > $O/gaps256.txt
timeout -k 5 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pf -- python3 $B > $O/pf.json 2> $O/pf.err || { tail -5 $O/pf.err; exit 1; }
timeout -k 5 200 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pw -- python3 $B > $O/pw.json 2> $O/pw.err || { tail -5 $O/pw.err; exit 1; }
python3 tools/pmc_summarize.py $O/pf $O/pw $O/pmc_spmv.json k_spmv_sell 2 k_spmv_sell sym > /dev/null
python3 tools/pmc_summarize.py $O/pf $O/pw $O/pmc_asm.json k_asm_node 2 k_asm_node > /dev/null
rocprofv3 -L > $O/counters.txt 2>&1
timeout -k 5 200 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD --kernel-trace --output-format csv -d $O/sq -- python3 tools/sell_probe_short.py > $O/sq.out 2> $O/sq.err || tail -5 $O/sq.err
timeout -k 5 200 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum TCC_EA0_RDREQ_sum --kernel-trace --output-format csv -d $O/tc -- python3 tools/sell_probe_short.py > $O/tc.out 2> $O/tc.err || tail -5 $O/tc.err
for d in sq tc; do f=$(find $O/$d -name '*counter_collection.csv' | head -1); [ -n "$f" ] && cp $f $O/${d}_counters.csv; done
rm -rf $O/ks $O/pf $O/pw $O/sq $O/tc
ls -la $O
