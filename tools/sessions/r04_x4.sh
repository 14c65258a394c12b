#!/bin/bash
# GILL_UNET_XALG as the default: oracle tests of the full-size engine, soak, switch test, multi-rank tests, loop A/B
O=gpurun_out/r04_x4; mkdir -p $O
timeout 1500 python -m pytest tests/test_configs_gpu.py -q -s -k "teacher or XALG or oracle or full_size" > $O/configs.log 2>&1; echo "configs rc=$?"; tail -n 2 $O/configs.log
timeout 900 python -m pytest tests/test_soak_gpu.py -q > $O/soak.log 2>&1; tail -n 1 $O/soak.log
timeout 900 python -m pytest tests/test_configs_gpu.py -q -k "eight_ranks or two_ranks" > $O/ranks.log 2>&1; tail -n 1 $O/ranks.log
bash tools/ab_env.sh GILL_UNET_XALG 3 > $O/ab.log 2>&1; cat $O/ab.log
