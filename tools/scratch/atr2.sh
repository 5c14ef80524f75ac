rm -f gpurun_out/atr2.log
for rep in 1 2 3; do
for cfg in "CRUSE_DX_ATR=0" "CRUSE_DX_ATR=1"; do
env $cfg python bench.py --steps 80 --warmup 10 --no-cpu-baseline --no-secondary --no-parity --no-kernel-timing 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('$cfg', d['ms_per_step'], d['ms_per_step_median'], d.get('launch_choice'))
" >> gpurun_out/atr2.log
done
done
