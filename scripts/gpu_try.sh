#!/bin/bash
set -u
mkdir -p gpurun_out
for rep in 1 2; do
for pct in off 0 30 50 70 100; do
  if [ $pct = off ]; then unset ANIREC_EMU_ROLLING; else export ANIREC_EMU_ROLLING=$pct; fi
  timeout -k 10 300 python bench.py --no-also --no-cpu-baseline > gpurun_out/try_e${pct}_$rep.json 2> gpurun_out/try_e${pct}_$rep.err || exit 1
done
done
python - <<'PY'
import json
for rep in (1,2):
  for pct in ("off","0","30","50","70","100"):
    d=json.loads(open('gpurun_out/try_e%s_%d.json'%(pct,rep)).read().strip().splitlines()[-1])
    k=d['lazy_kernels_ms']
    print(pct, rep, round(d['value']/1e6,2), 'M/s', round(d['ms_per_step']*1e3,2), 'us', 'fwd %.1f head %.1f bwd %.1f adam %.1f flush/8 %.1f' % (k['fwd']*1e3,k['head']*1e3,k['bwd']*1e3,k['lazy_adam']*1e3,k['lazy_flush']*1e3/8))
PY
