"""Turn the rocprofv3 PMC passes (HBM traffic: two passes; MFMA utilisation: a third, `--mfma`, see mfma_util) into profiles/ CSVs.

Turn the two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE; separate runs, csv output) into profiles/rNN_pmc_hbm_traffic.csv.

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
          ("gemm_bf16_nt_kernel", "gemm_bf16_nt"), ("gemm_slab_reduce", "gemm_bf16_nt"), ("conv_mfma", "conv"), ("conv_gather", "conv"), ("conv_scatter2", "conv"),
          ("wgrad_mfma", "conv_wgrad"), ("wgrad_rd", "conv_wgrad"), ("wgrad_reduce", "conv_wgrad"), ("bn_act_bwd", "bn_act_bwd"), ("bn_act_fwd", "bn_act_fwd"),
          ("bn_fin_act_fwd", "bn_act_fwd"),      # (r03's table had no key for this one: its 7 launches per step, 1.15 GB, were dropped)
          ("bn_stats", "bn_stats"), ("bn_finalize", "bn_stats"), ("channel_pair_reduce", "bn_stats"), ("ln_bwd", "ln_bwd"), ("ln_fwd", "ln_fwd"), ("transpose_bf16", "transpose_bf16"),
          ("cast_bf16", "cast_bf16"), ("ktile_bf16", "cast_bf16"), ("adam", "adam"), ("stft320", "stft"), ("mask_loss", "mask_loss"),
          ("cruse_zero", "zero"), ("zero_kernel", "zero"), ("channel_sum", "bias_sums"), ("sumsq", "adam"), ("accum_f64", "adam"),
          ("counters_add", "bn_stats"), ("step_health", "adam"), ("onepole", "data"), ("snr_mix", "data"), ("col_sum", "bias_sums"),
          ("gate_bias_sums", "bias_sums"), ("mask_apply", "mask_loss"), ("istft", "stft"), ("sisnr", "mask_loss"), ("wave_l1mse", "mask_loss"),
          ("deepfilter", "deepfilter"), ("wo_male_spec", "mask_loss"), ("sdnr", "mask_loss"), ("axpby", "adam"), ("cast_f16", "cast_bf16")]
# kernels that are NOT this library's (the data generator's torch kernels, rocclr copies): everything else must be classified
FOREIGN = ("at::native", "at_cuda_detail", "rocprim", "__amd_rocclr", "hipcub", "distribution_elementwise", "vectorized_elementwise",
           "elementwise_kernel", "reduce_kernel", "fillBuffer", "copyBuffer", "Cijk_", "rccl", "nccl")


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


def mfma_util(path, out, n_cu=256, n_xcd=8, steps=4):
    """Third PMC pass: rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE SQ_WAVES --kernel-trace.
    Per kernel: MFMA utilisation = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE * CUs * 4 SIMDs) -- the share of all matrix
    pipes' cycles that issued MFMA work while the kernel ran (MI355X_MICROARCH.md: the counter counts cycles, e.g. 32 per
    v_mfma_f32_32x32x16_bf16; ROCm 7.2 has no gfx950 derived-counter section, so the gfx94x MfmaUtil formula is applied by
    hand) -- and CU occupancy in time = SQ_BUSY_CU_CYCLES / (GRBM_GUI_ACTIVE * CUs).
    The csv reports GRBM_GUI_ACTIVE summed over the 8 XCDs (14.9 M "cycles" for the 0.81 ms backward recurrence = 8 x 1.86 M;
    the derived formulas take reduce(.., max)), so it is divided by n_xcd here.  Cross-checks on the r03 pass: the recurrences
    show cu_busy 0.60 (160 of 256 CUs hold a workgroup) and 122.88 M MFMA-busy cycles = 30 MFMAs x 401 steps x 640 waves x 16."""
    acc = {}
    with open(path) as f:
        for r in csv.DictReader(f):
            k = short(r["Kernel_Name"])
            a = acc.setdefault(k, {})
            c = a.setdefault(r["Counter_Name"], [0, 0.0])
            c[0] += 1; c[1] += float(r["Counter_Value"])
    rows = []
    for k, a in acc.items():
        fam = next((fam for key, fam in FAMILY if key in k), None)
        if fam is None or "GRBM_GUI_ACTIVE" not in a:
            continue
        n = a["GRBM_GUI_ACTIVE"][0]
        gui = a["GRBM_GUI_ACTIVE"][1] / n_xcd
        mf = a.get("SQ_VALU_MFMA_BUSY_CYCLES", [0, 0.0])[1]
        bc = a.get("SQ_BUSY_CU_CYCLES", [0, 0.0])[1]
        wv = a.get("SQ_WAVES", [0, 0.0])[1]
        rows.append((k, fam, n, gui / n, mf / n, mf / (gui * n_cu * 4) if gui else 0.0, bc / (gui * n_cu) if gui else 0.0, wv / n))
    rows.sort(key=lambda r: -r[2] * r[3])
    with open(out, "w") as f:
        f.write("# rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE SQ_WAVES --kernel-trace -- python bench.py --steps 3 --warmup 1 --no-graph "
                "--no-cpu-baseline --no-kernel-timing --no-parity --no-secondary (CRUSE_OVERLAP=0: one kernel at a time)\n")
        f.write(f"# mfma_util = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / {n_xcd} XCDs * {n_cu} CUs * 4 SIMDs); cu_busy = SQ_BUSY_CU_CYCLES / (GRBM_GUI_ACTIVE / {n_xcd} * {n_cu}); per-launch averages\n")
        f.write("kernel,bench_family,launches,gui_active_cycles_per_launch,mfma_busy_cycles_per_launch,mfma_util,cu_busy,waves_per_launch\n")
        for k, fam, n, gui, mf, u, cb, wv in rows:
            f.write(f"\"{k}\",{fam},{n},{gui:.0f},{mf:.0f},{u:.4f},{cb:.3f},{wv:.0f}\n")
    fams = {}
    for k, fam, n, gui, mf, u, cb, wv in rows:
        a = fams.setdefault(fam, [0.0, 0.0])
        a[0] += n * mf; a[1] += n * gui * n_cu * 4
    print(f"{len(rows)} kernels -> {out}")
    for fam, (mf, cap) in sorted(fams.items(), key=lambda kv: -kv[1][1]):
        print(f"  {fam:22s} mfma_util {mf / cap if cap else 0.0:7.4f}   share of GPU-active cycles {cap / sum(v[1] for v in fams.values()):6.3f}")


def main():
    if sys.argv[1] == "--mfma":
        return mfma_util(sys.argv[2], sys.argv[3])
    fpath, wpath, out = sys.argv[1:4]
    F, W = load(fpath, "FETCH_SIZE"), load(wpath, "WRITE_SIZE")
    # steps of the run: given, or counted -- every step launches the noisy and the clean STFT (bench.py adds untimed steps of its own)
    n_stft = sum(v[0] for k, v in (W or F).items() if "stft320_kernel<0>" in k)
    steps = int(sys.argv[4]) if len(sys.argv) > 4 else (n_stft // 2 if n_stft >= 2 else 4)
    rows = []
    for k in sorted(set(F) | set(W)):
        fam = next((fam for key, fam in FAMILY if key in k), None)
        if fam is None:
            if any(t in k for t in FOREIGN):
                continue                                      # torch / rocclr kernels of the data generator
            # a kernel of the library without a family would silently vanish from the step total (r03: bn_fin_act_fwd, 1.2 GB)
            raise SystemExit(f"pmc_traffic: kernel '{k}' has no family in FAMILY (and is not a known foreign kernel): add it")
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
