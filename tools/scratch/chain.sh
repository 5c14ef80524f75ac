#!/usr/bin/env bash
# serial-chain view of one eager step of the default bench (tools/rocpd_chain.py) -> gpurun_out/chain.txt
export TMPDIR=/tmp
rocprofv3 --kernel-trace -d gpurun_out/chain_prof -o p -- python bench.py --steps 12 --warmup 4 --no-graph --no-cpu-baseline --no-kernel-timing --no-parity --no-secondary > gpurun_out/chain_prof.log 2>&1
DB=$(ls gpurun_out/chain_prof/*/p_results.db gpurun_out/chain_prof/p_results.db 2>/dev/null | head -1)
python tools/rocpd_chain.py "$DB" 3 --all > gpurun_out/chain.txt 2>&1
rm -rf gpurun_out/chain_prof
tail -5 gpurun_out/chain.txt
