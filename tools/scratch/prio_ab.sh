rm -f gpurun_out/prio.log
for cfg in "CRUSE_GRU_PRIO=0" "CRUSE_GRU_PRIO=3" "CRUSE_GRU_PRIO=1" "CRUSE_GRU_PRIO=0" "CRUSE_GRU_PRIO=3" "CRUSE_GRU_PRIO=2"; do
env $cfg python bench.py --steps 40 --warmup 8 --no-cpu-baseline --no-secondary --no-parity 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); k=d['kernel_ms_per_step']; print('$cfg', d['ms_per_step'], d['ms_per_step_median'], 'loss', d['final_loss'], 'status', d.get('gru_status'), {x: k[x] for x in k if 'gru_seq' in x})
" >> gpurun_out/prio.log
done
