cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r3asm2; rm -rf $O; mkdir -p $O
timeout -k 10 600 python3 -m pytest tests -m gpu -x -q -k "node_assembly or fused_assembly or stencil_ell or mid_size or K_M_blocks or mesh_dofmap" > $O/pytest.log 2>&1; tail -5 $O/pytest.log
timeout -k 10 300 python3 tools/r3_asm_probe2.py > $O/times.txt 2>&1; cat $O/times.txt
timeout -k 10 300 python3 bench.py --no-cpu-baseline --skip-fine-bench --skip-csr > $O/bench.json 2> $O/bench.err; python3 -c "
import json; d=json.load(open('$O/bench.json')); print(d['ms_per_step'], d['config']['assemble_ms'], d['config']['solve_ms'], d['config']['picard_sweeps'], d['config']['inner_cg_iterations'], d['config']['setup_ms'])"
