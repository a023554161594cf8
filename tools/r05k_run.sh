#!/bin/bash
# Developer tool (GPU box), round 5: the whole GPU suite at the round's last commit, then the eight-rank tests three more times each.
OUT=$PWD/gpurun_out/r05k; mkdir -p $OUT
export OMP_NUM_THREADS=8
(timeout 2400 python -m pytest tests -m gpu -q -s 2>&1 | grep -a "held-out\|training frames\|f16 vs f32\|passed\|failed\|FAILED\|Error" | tail -30) > $OUT/gpu_suite.txt
for i in 1 2 3; do
  (timeout 1500 python -m pytest "tests/test_gpu_bench.py::test_workload_all_eight_ranks_on_one_gpu" "tests/test_gpu_multirank.py" "tests/test_gpu_driver.py::test_cli_with_several_ranks_on_one_gpu" -q 2>&1 | tail -3) >> $OUT/world8_soak.txt
done
python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.txt 2>&1
cat $OUT/gpu_suite.txt | cut -c1-250; cat $OUT/world8_soak.txt; tail -3 $OUT/smoke.txt
