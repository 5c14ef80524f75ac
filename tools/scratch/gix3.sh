rm -f gpurun_out/gix3.log
for x in 3 2 1 0; do
CRUSE_GI_X3=$x python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-secondary 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('gi_x3=$x', d['ms_per_step'], d['ms_per_step_median'], 'parity', d['parity_rel_l2'], 'loss', d['final_loss'], d['kernel_ms_per_step'].get('gemm_bf16x3_nt'), d['kernel_ms_per_step'].get('gemm_bf16_nt'))
" >> gpurun_out/gix3.log
done
for i in 1 2 3 4 5; do python -m pytest tests/test_gpu_model.py -m gpu -q -k "time_chunk_pipeline" 2>&1 | grep -o "assert [0-9.e-]* < 5e-05\|passed\|failed" | tr '\n' ' ' >> gpurun_out/gix3.log; echo >> gpurun_out/gix3.log; done
