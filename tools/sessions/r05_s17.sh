#!/bin/bash
# r05 session 17: 9-tap input patch conv loop prototype (tools/ubench/conv_patch.hip) vs the shipped ping-pong loop; fp8-vs-oracle test with its corrected bars
O=gpurun_out/r05_s17; mkdir -p $O
timeout 300 ./tools/ubench/conv_patch > $O/conv_patch.log 2>&1; cat $O/conv_patch.log
timeout 900 python -m pytest tests/test_fp8_gpu.py -x -q -k "oracle" > $O/fp8.log 2>&1; tail -1 $O/fp8.log
