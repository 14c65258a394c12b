set +x
# groupnorm_total_kernel on 1024 threads (16 columns x 64 slab slices): parity (VAE, SD-2.1 geometry, ops), VAE stage time, kernel stats
O=gpurun_out/r06_s15; mkdir -p $O
python -m pytest tests -m gpu -q -x -k "vae or groupnorm or sd2 or gn" > $O/tests.log 2>&1; tail -3 $O/tests.log
python bench.py --steps 4 --warmup 2 --no-pmc --no-cpu-baseline --no-scale-origin 2>/dev/null | tail -1 | python -c "import sys,json; r=json.loads(sys.stdin.readline()); print(r['value'], r['ms_per_step'], r['stages']['vae'], r['output_check'])" | tee $O/bench.log
