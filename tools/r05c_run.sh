#!/bin/bash
# Developer tool (GPU box), round 5, third session: SIMD placement probe, A/B of the de-phasing groupings and of each trim alone,
# the world-8 failures with round 4's kernels, the convergence scene scan.
OUT=gpurun_out/r05c; mkdir -p $OUT
tools/hwid_probe.bin > $OUT/hwid_probe.txt 2>&1
LIBS="exp_libs/base.so exp_libs/dephonly.so exp_libs/dephB.so exp_libs/dephC.so exp_libs/t_amax.so exp_libs/t_pp.so exp_libs/t_unroll.so" \
  ROUNDS_C4=3 ROUNDS_C4H=0 tools/r05_ab_train.sh $OUT > /dev/null 2>&1
export OMP_NUM_THREADS=4 MASTER_ADDR=127.0.0.1
for w in 8 4; do
  for lib in exp_libs/base.so intree; do
    if [ "$lib" = intree ]; then unset DFN_LIB; else export DFN_LIB=$lib; fi
    echo "=== world $w $lib" >> $OUT/multirank.txt
    timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node $w --master-addr 127.0.0.1 --master-port $((29500 + w)) \
      tests/multirank_worker.py 2>&1 | grep -v "amdgpu.ids\|socket.cpp\|OMP_NUM" | tail -15 >> $OUT/multirank.txt
  done
done
export DFN_LIB=exp_libs/base.so
(timeout 1200 python -m pytest tests/test_gpu_driver.py -k "several_ranks" -x -q 2>&1 | tail -120) > $OUT/cli_world8_base.txt
(timeout 600 python tools/convergence.py 1500 0 scan) > $OUT/conv_scan2.txt 2>&1
unset DFN_LIB
cat $OUT/hwid_probe.txt; grep -v "^$" $OUT/ab_train.txt | tail -60; cat $OUT/multirank.txt; tail -60 $OUT/cli_world8_base.txt; grep -v "^    " $OUT/conv_scan2.txt | tail -12
