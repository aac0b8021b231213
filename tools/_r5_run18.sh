mkdir -p gpurun_out/r5v
python -m pytest tests -m gpu -x -q > gpurun_out/r5v/tests.log 2>&1; echo "rc=$?" >> gpurun_out/r5v/tests.log; grep -E "passed|failed|rc=" gpurun_out/r5v/tests.log | tail -3
python -c "import __graft_entry__ as g; g.build(); g.smoke()" 2>&1 | grep "smoke ok"
