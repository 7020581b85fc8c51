#!/bin/bash
# serial / in-flight frame times of the triangle configurations (development: tile order)
mkdir -p gpurun_out/r03
for cfg in REF TRI TRI4K; do
  echo "== $cfg" >> gpurun_out/r03/tri_order2.log
  python bench.py --config $cfg --no-cpu-baseline --steps 200 --warmup 10 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('in flight %.4f serial %s check %s' % (d['ms_per_step'], d.get('serial_ms_per_step'), d.get('frame_check')))" >> gpurun_out/r03/tri_order2.log
done
cat gpurun_out/r03/tri_order2.log
