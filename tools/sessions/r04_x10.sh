#!/bin/bash
R=$PWD; O=$R/gpurun_out/r04_x10; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES" \
           "SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_SCA" \
           "SQ_WAVE_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_VMEM_TA_ADDR_FIFO_FULL"; do
  i=$((i+1))
  GILL_NO_GRAPH=1 timeout 600 rocprofv3 --pmc $set -d $O/p$i -o m --output-format csv -- python $R/bench.py --steps 1 --warmup 0 --infer-steps 3 --no-cpu-baseline --no-pmc --no-scale-origin > $O/run$i.log 2>&1
  echo "pass $i rc=$?"
  python $R/tools/pmc_sq.py $O/p$i > $O/table$i.txt 2>&1
  rm -rf $O/p$i
done
