#!/bin/bash
# Developer tool (GPU box): interleaved A/B of the training step (bench.py --workload c4 | c4h) between variants given as
# "NAME:ENV=VAL,ENV=VAL" arguments (NAME only = no environment change), ROUNDS rounds (default 3).
#   tools/ab_c4.sh base:DFN_LIB=exp_libs/base.so new
WL="${WL:-c4}"; ROUNDS="${ROUNDS:-3}"; STEPS="${STEPS:-600}"
for r in $(seq $ROUNDS); do
  for v in "$@"; do
    name="${v%%:*}"; envs=""; [ "$v" != "$name" ] && envs="${v#*:}"
    ms=$(env $(echo "$envs" | tr ',' ' ') python bench.py --workload $WL --steps $STEPS --warmup 50 --no-extra --no-cpu-baseline --sustain-seconds 0 2>/dev/null \
         | python -c "import sys,json; print('%.4f' % json.loads(sys.stdin.read())['ms_per_step'])")
    echo "$WL round $r $name $ms"
  done
done
