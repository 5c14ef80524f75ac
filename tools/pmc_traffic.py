"""Turn the two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE; separate runs, csv output) into profiles/rNN_pmc_hbm_traffic.csv.

    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d DIR_F -o p -- python bench.py --steps 3 --warmup 1 \
              --no-graph --no-cpu-baseline --no-kernel-timing --no-parity --no-secondary        (CRUSE_OVERLAP=0; same for WRITE_SIZE)
    python tools/pmc_traffic.py DIR_F/p_counter_collection.csv DIR_W/p_counter_collection.csv profiles/r02_pmc_hbm_traffic.csv

MI355X_MICROARCH.md (HBM section): FETCH_SIZE / WRITE_SIZE are reported in KiB; on gfx950 FETCH_SIZE counts 64 B per 128-B
request of a wide coalesced read, so hbm_bytes = 2 * fetch + write.  Infinity-Cache hits are counted (re-reads that miss L2
show up)."""
import csv
import re
import sys

FAMILY = [("gru_bwd", "gru_seq_bwd"), ("gru_fwd", "gru_seq_fwd"), ("gru_gate_grads", "gru_gate_grads_bf16"),
          ("gemm_bf16_nt_kernel", "gemm_bf16_nt"), ("conv_mfma", "conv"), ("conv_gather", "conv"), ("conv_scatter2", "conv"),
          ("wgrad_mfma", "conv_wgrad"), ("wgrad_reduce", "conv_wgrad"), ("bn_act_bwd", "bn_act_bwd"), ("bn_act_fwd", "bn_act_fwd"),
          ("bn_stats", "bn_stats"), ("bn_finalize", "bn_stats"), ("ln_bwd", "ln_bwd"), ("ln_fwd", "ln_fwd"), ("transpose_bf16", "transpose_bf16"),
          ("cast_bf16", "cast_bf16"), ("ktile_bf16", "cast_bf16"), ("adam", "adam"), ("stft320", "stft"), ("mask_loss", "mask_loss"),
          ("cruse_zero", "zero"), ("channel_sum", "bias_sums"), ("sumsq", "adam"), ("accum_f64", "adam"), ("counters_add", "bn_stats")]


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    name = re.sub(r"\(.*$", "", name)
    m = re.match(r"_ZN\d+_GLOBAL__N_1(\d+)([A-Za-z_0-9]+)", name)
    if m:
        name = m.group(2)[:int(m.group(1))]
    return name


def load(path, counter):
    acc = {}
    with open(path) as f:
        for r in csv.DictReader(f):
            if r["Counter_Name"] != counter:
                continue
            a = acc.setdefault(short(r["Kernel_Name"]), [0, 0.0])
            a[0] += 1; a[1] += float(r["Counter_Value"])
    return acc


def main():
    fpath, wpath, out = sys.argv[1:4]
    steps = int(sys.argv[4]) if len(sys.argv) > 4 else 4
    F, W = load(fpath, "FETCH_SIZE"), load(wpath, "WRITE_SIZE")
    rows = []
    for k in sorted(set(F) | set(W)):
        fam = next((fam for key, fam in FAMILY if key in k), None)
        if fam is None:
            continue                                          # torch / rocclr kernels of the data generator
        nf, f = F.get(k, [0, 0.0]); nw, w = W.get(k, [0, 0.0])
        n = nw or nf
        fk, wk = (f / nf if nf else 0.0), (w / nw if nw else 0.0)
        rows.append((k, fam, n, fk, wk, (2 * fk + wk) * 1024))
    rows.sort(key=lambda r: -r[2] * r[5])
    total = sum(r[2] * r[5] for r in rows) / steps
    with open(out, "w") as f:
        f.write(f"# rocprofv3 --pmc FETCH_SIZE --kernel-trace (pass 1) / --pmc WRITE_SIZE --kernel-trace (pass 2) -- python bench.py --steps 3 --warmup 1 --no-graph --no-cpu-baseline --no-kernel-timing, CRUSE_OVERLAP=0\n")
        f.write("# per-launch averages, KiB as reported.  MI355X_MICROARCH.md (HBM): on gfx950 FETCH_SIZE reports half the bytes of wide coalesced reads -> hbm_bytes_corrected = 2*fetch + write;\n")
        f.write(f"# Infinity-Cache hits are counted.  launches = over the {steps} steps of the run; whole step: {total / 1e9:.2f} GB\n")
        f.write("kernel,bench_family,launches,fetch_kib_per_launch_raw,write_kib_per_launch_raw,hbm_bytes_per_launch_corrected\n")
        for k, fam, n, fk, wk, b in rows:
            f.write(f"\"{k}\",{fam},{n},{fk:.1f},{wk:.1f},{b:.0f}\n")
    print(f"{len(rows)} kernels, {total / 1e9:.2f} GB per step -> {out}")
    fams = {}
    for k, fam, n, fk, wk, b in rows:
        fams[fam] = fams.get(fam, 0.0) + n * b / steps
    for fam, v in sorted(fams.items(), key=lambda kv: -kv[1]):
        print(f"  {fam:22s} {v / 1e9:6.2f} GB/step")


if __name__ == "__main__":
    main()
