#!/usr/bin/env bash
# kernel trace of tools/mtfaa_stress.py (config 5) -> gpurun_out/mtfaa_kernel_stats2.csv
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d gpurun_out/mtfaa_prof -o p -- python tools/mtfaa_stress.py --steps 5 > gpurun_out/mtfaa_prof.log 2>&1
DB=$(ls gpurun_out/mtfaa_prof/*/p_results.db gpurun_out/mtfaa_prof/p_results.db 2>/dev/null | head -1)
python tools/rocpd_stats.py "$DB" gpurun_out/mtfaa_kernel_stats2.csv > /dev/null 2>&1
rm -rf gpurun_out/mtfaa_prof
grep "ms/step" gpurun_out/mtfaa_prof.log
