#!/bin/bash
# round-2 first GPU call: mtimes, tests, issue table, SQ counter passes (2D 16M, 3D 8M), bench line
set -o pipefail
O=gpurun_out/r02a; mkdir -p $O
ls -la --time-style=full-iso gpu-fluid-simulation_amd/libfluidsim_hip.so gpu-fluid-simulation_amd/csrc/engine.hip gpu-fluid-simulation_amd/build.py > $O/mtimes.txt 2>&1
python -c "import __graft_entry__ as g; print('stale', g.product_is_stale())" >> $O/mtimes.txt 2>&1
python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1 || { tail -30 $O/pytest.log; exit 1; }
tail -3 $O/pytest.log
./tools/valu_issue_table $O/valu_issue_table.csv > $O/valu_issue_table.txt 2>&1 || exit 1
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
A="SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE"
B="SQ_WAVES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_THREAD_CYCLES_VALU SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE"
C="SQ_WAVES SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_CVT SQ_ACTIVE_INST_SCA GRBM_GUI_ACTIVE"
for scene in 2d 3d; do
  for p in A B C; do
    eval "ctr=\$$p"
    rocprofv3 --pmc $ctr -d $O/pmc_${scene}_$p -o p --output-format csv -- python3 tools/pmc_run.py $scene 10 20 > $O/pmc_${scene}_$p.log 2>&1 || { tail -5 $O/pmc_${scene}_$p.log; exit 1; }
    echo "pmc $scene $p done"
  done
done
python bench.py > $O/bench.json 2> $O/bench.err || { tail -20 $O/bench.err; exit 1; }
cat $O/bench.json | head -c 3000
