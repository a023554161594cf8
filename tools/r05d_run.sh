#!/bin/bash
# Developer tool (GPU box), round 5, fourth session: the GPU suite on the final library, the world-8 paths several times over
# (one run of the second session had diverged replicas), the convergence scene with the live-density student start.
OUT=gpurun_out/r05d; mkdir -p $OUT
export OMP_NUM_THREADS=8
(timeout 1500 python -m pytest tests -m gpu -q --deselect tests/test_gpu_convergence.py 2>&1 | tail -40) > $OUT/gpu_suite.txt
export OMP_NUM_THREADS=4 MASTER_ADDR=127.0.0.1
for i in 1 2 3 4; do
  echo "=== world 8 run $i" >> $OUT/multirank8.txt
  timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port $((29600 + i)) \
    tests/multirank_worker.py 2>&1 | grep -v "amdgpu.ids\|socket.cpp\|OMP_NUM\|UserWarning\|tensor.detach\|losses.append\|Gloo" | tail -12 >> $OUT/multirank8.txt
done
for i in 1 2; do
  (timeout 1500 python -m pytest "tests/test_gpu_driver.py::test_cli_with_several_ranks_on_one_gpu[8]" -x -q 2>&1 | tail -150) > $OUT/cli_world8_$i.txt
done
(timeout 600 python tools/convergence.py 2000 0 scan) > $OUT/conv_scan3.txt 2>&1
tail -8 $OUT/gpu_suite.txt; grep -a "===\|MULTIRANK\|Error\|error" $OUT/multirank8.txt | head -40; tail -5 $OUT/cli_world8_1.txt; tail -5 $OUT/cli_world8_2.txt; grep "^lr\|teacher" $OUT/conv_scan3.txt | cut -c1-400
