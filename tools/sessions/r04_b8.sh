#!/bin/bash
# bisect: which round-4 change makes the 8-ranks-on-one-GPU bench step-to-step non-identical?
O=gpurun_out/r04_b8; mkdir -p $O
run() { tag=$1; shift; for i in 1 2 3; do env "$@" python bench.py --gpus 8 --backend gloo --share-gpu --small --total-prompts 61 --infer-steps 2 --steps 3 --warmup 1 --no-cpu-baseline --no-pmc > $O/$tag.$i.log 2>&1; echo "$tag run $i rc=$? $(grep -c 'BAD' $O/$tag.$i.log) bad lines"; done; }
run base X=1
run att0 GILL_ATT_DMA=0
run redgn0 GILL_GEMM_RED_GN=0
run pc1 DEBUG_CLR_GRAPH_PACKET_CAPTURE=1
run nograph GILL_NO_GRAPH=1
