"""Which CUs does a CU-masked stream reach?  Launches the census kernel on (a) the default stream, (b) masked streams with
various bit patterns, and prints the (XCC, SE, SH, CU) sets the blocks landed on -- the map from mask bit to physical CU
that the leaf / recurrence partition in cruse_net._SideStream needs."""
import ctypes
import os
import sys
from collections import Counter

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from cruse_amd._lib import check, lib  # noqa: E402


def masked_stream(bits):
    words = (ctypes.c_uint * 8)()
    for b in bits:
        words[b // 32] |= 1 << (b % 32)
    out = ctypes.c_void_p()
    check(lib.cruse_stream_create_masked(ctypes.byref(out), words, 8))
    return torch.cuda.ExternalStream(out.value)


def census(stream, nblocks=512, spin=200000):
    out = torch.zeros(2 * nblocks, dtype=torch.int32, device="cuda")
    with torch.cuda.stream(stream):
        check(lib.cruse_cu_census(out.data_ptr(), nblocks, spin, torch.cuda.current_stream().cuda_stream))
    torch.cuda.synchronize()
    w = out.cpu().view(nblocks, 2).tolist()
    locs = []
    for xcc, hw in w:
        xcc &= 0xf
        cu, sh, se = (hw >> 8) & 0xf, (hw >> 12) & 0x1, (hw >> 13) & 0x7
        locs.append((xcc, se, sh, cu))
    return locs


def describe(name, locs):
    c = Counter(locs)
    per_xcc = Counter(l[0] for l in c)
    print(f"{name}: {len(c)} distinct CUs; per XCC {dict(sorted(per_xcc.items()))}")
    return set(c)


def main():
    torch.cuda.init()
    base = describe("unmasked", census(torch.cuda.current_stream()))
    print("  example ids (xcc, se, sh, cu):", sorted(base)[:12])
    for name, bits in (("bits 0..31", range(32)), ("bits 0..7", range(8)), ("bits 0,8,16,..", range(0, 256, 8)),
                       ("bits 0..159", range(160)), ("bits 160..255", range(160, 256)),
                       ("bits with (b%32) < 20", [b for b in range(256) if b % 32 < 20]),
                       ("bits with (b%32) >= 20", [b for b in range(256) if b % 32 >= 20]),
                       ("bits with (b//8)%32 < 20", [b for b in range(256) if (b // 8) % 32 < 20])):
        s = masked_stream(list(bits))
        got = describe(name, census(s))
        print("   ", sorted(got)[:10], "...")


if __name__ == "__main__":
    main()
