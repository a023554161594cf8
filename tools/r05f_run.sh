#!/bin/bash
# Developer tool (GPU box), round 5: the long-form convergence run, and the CLI at world 8 on one GPU three times over.
OUT=$PWD/gpurun_out/r05f; mkdir -p $OUT
(timeout 1500 python tools/convergence.py 12000 3000) > $OUT/convergence.txt 2> $OUT/convergence.err
export OMP_NUM_THREADS=4
for i in 1 2 3; do timeout 900 python -m pytest "tests/test_gpu_driver.py::test_cli_with_several_ranks_on_one_gpu[8]" -x -q 2>&1 | tail -2 >> $OUT/cli_world8.txt; done
grep -v "^JSON" $OUT/convergence.txt | cut -c1-300; tail -3 $OUT/convergence.err; cat $OUT/cli_world8.txt
