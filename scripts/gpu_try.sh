#!/bin/bash
set -u
mkdir -p gpurun_out
for rep in 1 2; do
for pct in 0 35 50 65 100; do
  ANIREC_CATCHUP_HEAD_PCT=$pct timeout -k 10 300 python bench.py --no-also --no-cpu-baseline > gpurun_out/try_p${pct}_$rep.json 2> gpurun_out/try_p${pct}_$rep.err || exit 1
done
done
python - <<'PY'
import json
for rep in (1,2):
  for pct in (0,35,50,65,100):
    d=json.loads(open('gpurun_out/try_p%d_%d.json'%(pct,rep)).read().strip().splitlines()[-1])
    k=d['lazy_kernels_ms']
    print(pct, rep, round(d['value']/1e6,2), 'M/s', round(d['ms_per_step'],4), 'ms', 'fwd %.1f head %.1f bwd %.1f adam %.1f flush %.1f sum4 %.1f' % (k['fwd']*1e3,k['head']*1e3,k['bwd']*1e3,k['lazy_adam']*1e3,k['lazy_flush']*1e3,(k['fwd']+k['head']+k['bwd']+k['lazy_adam'])*1e3))
PY
