"""Out-of-bounds hunt through the C ABI: runs tests/test_gpu_abi_ref.py's comparisons with every device mirror placed at the very END of its own
20 MB allocator segment, so that a kernel that reads or writes past a tensor leaves the mapping and the process dies with a memory access
fault (the placement of ordinary test tensors hides such accesses: they land in the allocator's neighbouring blocks).  A tool, not a test -- a fault
cannot be caught.  Found in r5: conv_mfma's staging read tile row r of the TENSOR for rows outside the clip (a tensor shorter than a tile).
    usage: python tools/abi_guard_sweep.py [name-substring ...]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_gpu_abi_ref as T  # noqa: E402
import test_gpu_abi_sweeps as T2  # noqa: E402
import ref_lib as R  # noqa: E402

SEG = 20 * 1024 * 1024
_keep = []


def guard_place(a):
    nbytes = max(int(a.nbytes), 16)
    buf = torch.empty(SEG, dtype=torch.uint8, device="cuda")
    _keep.append(buf)
    if len(_keep) > 48:
        torch.cuda.synchronize(); del _keep[:24]
    off = SEG - (nbytes + 15) // 16 * 16
    t = buf[off:off + a.nbytes].view(torch.from_numpy(a).dtype).view(a.shape)
    t.copy_(torch.from_numpy(a))
    return t


def main():
    from cruse_amd._lib import lib
    T.PLACE = guard_place
    T2.PLACE = guard_place
    ref = R.load()
    pick = sys.argv[1:]
    import inspect
    for mod in (T, T2):
        for n in [n for n in dir(mod) if n.startswith("test_")]:
            if pick and not any(p in n for p in pick):
                continue
            print("running", n, flush=True)
            fn = getattr(mod, n)
            fn(*[{"hip": lib, "ref": ref}[a] for a in inspect.signature(fn).parameters])
            torch.cuda.synchronize()
    print("no out-of-bounds access reached an unmapped page")


if __name__ == "__main__":
    main()
