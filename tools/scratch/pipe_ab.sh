python tools/scratch/pipe_ab.py > gpurun_out/pipe.log 2>&1
for cfg in "CRUSE_GB_PIPE=0" "CRUSE_GB_PIPE=1" "CRUSE_GB_PIPE=0" "CRUSE_GB_PIPE=1" "CRUSE_GB_PIPE=0" "CRUSE_GB_PIPE=1"; do
env $cfg python bench.py --steps 40 --warmup 8 --no-cpu-baseline --no-secondary --no-parity 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); k=d['kernel_ms_per_step']; print('$cfg', d['ms_per_step'], d['ms_per_step_median'], 'loss', d['final_loss'], {x: k[x] for x in k if 'gemm' in x})
" >> gpurun_out/pipe.log
done
