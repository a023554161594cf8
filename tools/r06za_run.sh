#!/bin/bash
# Round 6, GPU session ZA (developer tool): the f32 kernels' weight-fragment reads pinned where the source issues them
# (-DDFN_GEMM_SCHEDBAR=0: four fragments ahead; hipcc sinks each ds_read_b128 next to the MFMAs that use it and waits for it at once).
cd "$(dirname "$0")/.."
OUT=gpurun_out/r06za; mkdir -p $OUT
{
for v in sb0 sb0pf2; do echo "$v tests: $(DFN_LIB=exp_libs/$v.so python -m pytest tests/test_gpu_train.py tests/test_gpu_parity.py tests/test_gpu_api.py -x -q -k 'f32' 2>&1 | tail -1)"; done
for r in 1 2; do
for v in base sb0 sb0pf2; do echo "$v fwd f32: $(DFN_LIB=exp_libs/$v.so python tools/time_fwd.py f32 2>&1 | tail -1)"; done
for v in base sb0 sb0pf2; do echo "$v dx f32: $(DFN_LIB=exp_libs/$v.so python tools/time_dx.py f32 2>&1 | tail -1)"; done
done
ms() { python -c "import sys,json; print('%.4f ms' % json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'])"; }
B="python bench.py --workload c4 --tier f32 --steps 150 --warmup 10 --no-extra --no-cpu-baseline --sustain-seconds 0 --no-parity-check"
for r in 1 2 3; do
for v in base sb0 sb0pf2; do echo "step f32 $v: $(DFN_LIB=exp_libs/$v.so $B 2>/dev/null | ms)"; done
done
B="python bench.py --workload c2 --tier f32 --steps 5 --warmup 1 --no-extra --no-cpu-baseline --sustain-seconds 0"
for r in 1 2; do for v in base sb0 sb0pf2; do echo "c2_f32 $v: $(DFN_LIB=exp_libs/$v.so $B 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['roofline']['frac'], d.get('parity_check',{}).get('psnr_db'))")"; done; done
} 2>&1 | grep -v amdgpu.ids | tee $OUT/log.txt
