set +x
# round-end rehearsal: what the driver runs — smoke(), the default bench command (no flags), timing of both
O=gpurun_out/r06_s20; mkdir -p $O
( time python -c "import __graft_entry__ as g; g.smoke(); print('SMOKE OK')" ) > $O/smoke.log 2>&1; tail -6 $O/smoke.log
( time python bench.py ) > $O/bench_default.log 2>&1; tail -5 $O/bench_default.log | cut -c1-300
