"""Runs a Python script with torch's device allocator replaced by tools/guard_alloc (every tensor end-aligned in its own hipMalloc block): kernels that
read or write past a tensor die with a memory access fault.  Eager launches only (no HIP graphs under this allocator).
    usage: python tools/engine_guard_run.py script.py [args ...]"""
import os
import runpy
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
so = os.path.join(HERE, "guard_alloc", "libguard_alloc.so")
if not os.path.exists(so):
    import subprocess
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-O2", "-o", so, os.path.join(HERE, "guard_alloc", "guard_alloc.cpp")])
alloc = torch.cuda.memory.CUDAPluggableAllocator(so, "guard_malloc", "guard_free")
torch.cuda.memory.change_current_allocator(alloc)
sys.argv = sys.argv[1:]
runpy.run_path(sys.argv[0], run_name="__main__")
