#!/bin/bash
O=gpurun_out/r04_x21; mkdir -p $O
timeout 900 python -m pytest tests/test_configs_gpu.py -q -k "GNFOLD" -s > $O/t.log 2>&1; echo "switch rc=$?"; tail -n 3 $O/t.log
timeout 900 python -m pytest tests/test_ops_gpu.py -q -k "lnproj or groupnorm" > $O/ops.log 2>&1; echo "ops rc=$?"; tail -n 1 $O/ops.log
bash tools/ab_env.sh GILL_UNET_GNFOLD 3
