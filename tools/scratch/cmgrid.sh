rm -f gpurun_out/cmg.log
for m in 0 768 512 640 896 0 768; do
if [ $m = 0 ]; then E=""; else E="CRUSE_CM_GRID=$m"; fi
env $E python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-secondary --no-parity 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); k=d['kernel_ms_per_step']; print('cm_grid=$m', d['ms_per_step'], d['ms_per_step_median'], 'convs', round(k.get('conv_gather',0)+k.get('conv_gather_bnin',0)+k.get('conv_scatter2',0)+k.get('conv_scatter2_bnin',0),3))
" >> gpurun_out/cmg.log
done
