cd $GRAFT_REPO_ROOT
O=gpurun_out/r06_s08; mkdir -p $O
timeout 1200 python -m pytest tests/test_fp8_gpu.py -q -s > $O/fp8_tests.log 2>&1; echo "fp8 tests rc=$?"
grep -E "^\.?\[fp8|^\.?\[bf16|passed|failed|FAILED" $O/fp8_tests.log
