python tools/scratch/atr_ab.py > gpurun_out/atr.log 2>&1
for cfg in "CRUSE_DX_ATR=0" "CRUSE_DX_ATR=1" "CRUSE_DX_ATR=0" "CRUSE_DX_ATR=1"; do
env $cfg python bench.py --steps 40 --warmup 8 --no-cpu-baseline --no-secondary --no-parity 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); k=d['kernel_ms_per_step']; print('$cfg', d['ms_per_step'], d['ms_per_step_median'], 'loss', d['final_loss'], {x: k[x] for x in k if 'gate_grads' in x or 'gemm' in x})
" >> gpurun_out/atr.log
done
