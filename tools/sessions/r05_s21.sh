#!/bin/bash
# r05 session 21: KV-cached decode (M = 1 row) with the OPT matrices blocked (STREAM64) vs row-major on the general tiles
for rep in 1 2; do for v in 0 1; do echo -n "STREAM64=$v: "; GILL_GEMM_STREAM64=$v timeout 300 python tools/opt_decode.py 24 2>/dev/null | tail -1; done; done
