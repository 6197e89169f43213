#!/bin/bash
# Hardware counters of the three attention-aggregate kernels on the benchmark's layer-1 launches (DESIGN.md section 5,
# profiles/r2/r2_c_agg_counters.txt).  One rocprofv3 process per counter group (--pmc with --kernel-trace only, as the
# MI355X guide prescribes), each running tools/agg_layer_runner.py (12 eager training batches of the benchmark graph).
# usage (repo root, via gpurun; ~2.5 GPU-minutes per group):  bash tools/agg_counters.sh [out_dir]
out=${1:-gpurun_out/pmc_src}
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/$out
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAVE_CYCLES" \
           "SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU" "SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM" \
           "SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU" \
           "TA_BUSY_sum TA_TA_BUSY_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum" \
           "TCP_TA_TCP_STATE_READ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_WRITE_REQ_sum TCP_GATE_EN1_sum" \
           "TCC_REQ_sum TCC_READ_sum TCC_WRITE_sum TCC_EA0_RDREQ_sum" "TCC_TAG_STALL_sum TCC_BUSY_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_STALL_sum"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $R/$out -o g$i -- python $R/tools/agg_layer_runner.py > $R/$out/g$i.log 2>&1
  echo "group $i rc=$?"
done
cd $R && python tools/summarize_counters.py $out > $out/summary.txt && cat $out/summary.txt
