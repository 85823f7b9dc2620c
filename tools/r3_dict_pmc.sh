cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/dictpmc; rm -rf $O; mkdir -p $O
run() { timeout -k 5 200 rocprofv3 --pmc $2 --kernel-trace --output-format csv -d $O/$1 -- python3 tools/r3_dict_pmc.py > $O/$1.out 2> $O/$1.err || { tail -5 $O/$1.err | cut -c1-300; return 1; }; f=$(find $O/$1 -name '*counter_collection.csv' | head -1); cp $f $O/$1.csv; rm -rf $O/$1; }
run lv1 "SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES"
run lv2 "SQ_IFETCH SQ_IFETCH_LEVEL SQ_INST_LEVEL_LDS SQ_INSTS_LDS SQ_INST_LEVEL_SMEM SQ_INSTS_SMEM SQ_LEVEL_WAVES"
run lv3 "SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_LDS_UNALIGNED_STALL SQ_LDS_BANK_CONFLICT SQ_BUSY_CU_CYCLES"
run lv4 "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_BRANCH SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_FLAT SQ_ACTIVE_INST_MISC"
for c in lv1 lv2 lv3 lv4; do echo "== $c"; [ -f $O/$c.csv ] && python3 tools/r3_pmc_by_order.py $O/$c.csv "k_spmv" 12; done
