#!/bin/bash
# r05 session 23: soak of the final build (every stage output of 40 back-to-back full-size calls vs call 0)
O=gpurun_out/r05_s23; mkdir -p $O
timeout 1200 python tools/soak.py --iters 40 > $O/soak.log 2>&1; tail -6 $O/soak.log
