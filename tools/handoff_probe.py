"""One-hop latency of the recurrences' hand-off protocol, measured between two workgroups (tools/probes/handoff_probe.hip,
compiled here with hipcc): A publishes nst x 1 KB of tagged granules per wave (plain or write-through stores), B sweeps them
with sc1 loads until the tags match and answers with one 8-byte granule per lane.  Prints cycles per round trip."""
import ctypes, os, subprocess, sys, torch
here = os.path.dirname(os.path.abspath(__file__))
so = "/tmp/handoff_probe.so"
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-shared", "-fPIC", "-o", so,
                       os.path.join(here, "probes", "handoff_probe.hip")])
lib = ctypes.CDLL(so)
lib.handoff_run.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
buf = torch.zeros(1 << 20, dtype=torch.int32, device="cuda")
out = torch.zeros(4, dtype=torch.int64, device="cuda")
iters = 2000
for other, where in ((8, "same XCD (blocks 0 and 8)"), (1, "neighbouring XCDs (blocks 0 and 1)")):
    for plain in (1, 0):
        if plain and other == 1:
            continue                       # plain stores are only visible inside one XCD's L2
        for nw, nst in ((1, 1), (4, 1), (4, 5), (1, 5)):
            buf.zero_(); out.zero_(); torch.cuda.synchronize()
            rc = lib.handoff_run(buf.data_ptr(), out.data_ptr(), iters, nst, nw, other, plain, torch.cuda.current_stream().cuda_stream)
            torch.cuda.synchronize()
            cyc, polls = out[0].item() / iters, out[1].item() / iters
            print(f"{where}, {'plain' if plain else 'sc1  '} stores, {nw} wave(s) x {nst} KB published: {cyc:7.0f} cycles per round trip "
                  f"({polls:.2f} sweeps per hop on the consumer side)", flush=True)
