run() { # label, env...
  label="$1"; shift
  env "$@" python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-secondary --no-parity --no-kernel-timing 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('$label', d['ms_per_step'], d['ms_per_step_median'], d['config']['launch_form_timing'])
" >> gpurun_out/sweep.log
}
rm -f gpurun_out/sweep.log
run base X=1
run pb3 CRUSE_GRU_POLL_BWD=3
run pb7 CRUSE_GRU_POLL_BWD=7
run pf2 CRUSE_GRU_POLL_FWD=2
run defer1 CRUSE_DEFER=1
run defer3 CRUSE_DEFER=3
run defer7 CRUSE_DEFER=7
run defer15 CRUSE_DEFER=15
run inline0 CRUSE_INLINE=0
run inline1 CRUSE_INLINE=1
run inline2 CRUSE_INLINE=2
run inline4 CRUSE_INLINE=4
run inline5 CRUSE_INLINE=5
run early0 CRUSE_EARLY_T=0
run early2 CRUSE_EARLY_T=2
run base2 X=1
