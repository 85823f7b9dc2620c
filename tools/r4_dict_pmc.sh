# round 4: counter passes over the isolated dictionary product k_spmv_dict_walk<0> (256^3 block, y = A x) - what the ~3 300
# cycles per step and wave are spent on: latencies seen by the L1, stalls of the address / data paths, LDS activity
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/dictpmc4; rm -rf $O; mkdir -p $O
cat > $O/w.py <<'PY'
import os, sys
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
from perphil_amd import _ffi
import bench
ctx = _ffi.Context(0); ctx.mesh_build(3, _ffi.CELL_HEX, 256, 256, 256)
b, g1, g2 = bench.mms_boundary(256, 1.0, 1e-2, 1.0, 1.0)
ctx.set_dirichlet(0, b, g1); ctx.set_dirichlet(1, b, g2); ctx.assemble(1.0, 1e-2, 1.0, 1.0, monolithic=False)
print("walk ms", ctx.spmv_bench(_ffi.MAT_A11, 12))
PY
run() { timeout -k 5 200 rocprofv3 --pmc $2 --kernel-trace --output-format csv -d $O/$1 -- python3 $O/w.py > $O/$1.out 2> $O/$1.err || { tail -5 $O/$1.err | cut -c1-300; return 1; }; f=$(find $O/$1 -name '*counter_collection.csv' | head -1); cp $f $O/$1.csv; rm -rf $O/$1; }
run p1 "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_TCP_LATENCY_sum TCP_TOTAL_CACHE_ACCESSES_sum"
run p2 "TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_TOTAL_WAVEFRONTS_sum"
run p3 "TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_GATE_EN1_sum"
run p4 "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVE_CYCLES"
run p5 "SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_CYCLES"
run p6 "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_REQ_sum"
for c in p1 p2 p3 p4 p5 p6; do echo "== $c"; [ -f $O/$c.csv ] && python3 tools/r3_pmc_by_order.py $O/$c.csv "k_spmv_dict_walk" 17; done
