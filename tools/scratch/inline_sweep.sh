rm -f gpurun_out/inl.log
for m in 8 1032 520 1544 264 776 8 1032 520 1544; do
CRUSE_INLINE=$m python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-secondary --no-parity --no-kernel-timing 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('inline=$m', d['ms_per_step'], d['ms_per_step_median'], d['config']['launch_form_timing'])
" >> gpurun_out/inl.log
done
