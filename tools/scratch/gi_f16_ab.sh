rm -f gpurun_out/gif16.log
for cfg in "CRUSE_GI_F16=0" "CRUSE_GI_F16=2" "CRUSE_GI_F16=3" "CRUSE_GI_F16=0" "CRUSE_GI_F16=2" "CRUSE_GI_F16=3"; do
env $cfg python bench.py --steps 40 --warmup 8 --no-cpu-baseline --no-secondary 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('$cfg', d['ms_per_step'], d['ms_per_step_median'], 'parity', d['parity_rel_l2'], 'loss', d['final_loss'])
" >> gpurun_out/gif16.log
done
python -m pytest tests -m gpu -q 2>&1 | tail -8 >> gpurun_out/gif16.log
