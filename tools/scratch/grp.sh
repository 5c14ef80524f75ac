python -m pytest tests/test_gpu_kernels.py -m gpu -q -x -k "all_groups" 2>&1 | tail -6 > gpurun_out/grp.log
for cfg in "CRUSE_GEMM_GROUPS=0" "CRUSE_GEMM_GROUPS=1" "CRUSE_GEMM_GROUPS=0" "CRUSE_GEMM_GROUPS=1"; do
env $cfg python bench.py --groups 4 --steps 30 --warmup 6 --no-cpu-baseline --no-secondary --no-parity 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); k=d['kernel_ms_per_step']; print('g4 $cfg', d['ms_per_step'], d['ms_per_step_median'], 'loss', d['final_loss'])
" >> gpurun_out/grp.log
done
for cfg in "CRUSE_GEMM_GROUPS=0" "CRUSE_GEMM_GROUPS=1"; do
env $cfg python bench.py --groups 4 --df --batch 32 --steps 30 --warmup 6 --no-cpu-baseline --no-secondary --no-parity 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); k=d['kernel_ms_per_step']; print('df $cfg', d['ms_per_step'], d['ms_per_step_median'], 'loss', d['final_loss'])
" >> gpurun_out/grp.log
done
python -m pytest tests -m gpu -q -x 2>&1 | tail -4 >> gpurun_out/grp.log
