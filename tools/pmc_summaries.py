"""gpurun_out/<tag>/pmc_{fetch,write,mfma}.csv (tools/prof_pmc.sh) -> profiles/<prefix>_pmc_hbm_traffic.json + <prefix>_pmc_mfma_busy.json
   python tools/pmc_summaries.py gpurun_out/r2w_pmc profiles/r02"""
import csv, json, sys
src, prefix = sys.argv[1], sys.argv[2]


def load(f):
    d = {}
    for r in csv.DictReader(open(f)):
        d.setdefault(r["Kernel_Name"], {})[r["Counter_Name"]] = (int(r["Launches"]), float(r["Sum"]), float(r["Per_Launch"]))
    return d


fe, wr, mf = load(src + "/pmc_fetch.csv"), load(src + "/pmc_write.csv"), load(src + "/pmc_mfma.csv")
cmd = ("rocprofv3 --pmc <FETCH_SIZE | WRITE_SIZE | SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE> --kernel-trace --output-format csv -- python3 bench.py "
       "--steps 2 --warmup 1 --no-cpu-baseline --no-probe --single-variant --no-side-stream   (three separate passes: tools/prof_pmc.sh)")
out = {"command": cmd, "units": "FETCH_SIZE / WRITE_SIZE in KiB; FETCH_SIZE x2 on gfx950 (MI355X_MICROARCH.md); workload: bench.py defaults (batch 64 x 4 s, as-executed variant)",
       "kernels": {}}
for k in sorted(fe, key=lambda k: -(2 * fe[k]["FETCH_SIZE"][1] + wr.get(k, {}).get("WRITE_SIZE", (0, 0, 0))[1])):
    out["kernels"][k[:120]] = {"launches": fe[k]["FETCH_SIZE"][0], "fetch_bytes_per_launch": int(2 * 1024 * fe[k]["FETCH_SIZE"][2]),
                              "write_bytes_per_launch": int(1024 * wr.get(k, {}).get("WRITE_SIZE", (0, 0, 0))[2])}
json.dump(out, open(prefix + "_pmc_hbm_traffic.json", "w"), indent=1)
m = {"command": cmd,
     "definition": "mfma_util = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 * 1024): the counter adds 16 cycles per v_mfma_f32_16x16x32_bf16 over all SIMDs (checked on "
                   "fusion_xattn_fwd_kernel: 512 workgroups x 3520 MFMAs x 16 = 2.884e7 = the counter), GRBM_GUI_ACTIVE is summed over the 8 XCDs, 1024 = 256 CUs x 4 SIMDs",
     "kernels": {}}
for k in sorted(mf, key=lambda k: -mf[k]["SQ_VALU_MFMA_BUSY_CYCLES"][1]):
    b, g = mf[k]["SQ_VALU_MFMA_BUSY_CYCLES"], mf[k]["GRBM_GUI_ACTIVE"]
    if b[1] <= 0:
        continue
    m["kernels"][k[:120]] = {"launches": g[0], "mfma_busy_cycles_per_launch": round(b[2]), "kernel_cycles_per_launch": round(g[2] / 8),
                            "mfma_util": round(b[2] / (g[2] / 8 * 1024), 4)}
json.dump(m, open(prefix + "_pmc_mfma_busy.json", "w"), indent=1)
for k, v in list(m["kernels"].items())[:12]:
    print(f"{v['mfma_util']:.3f} {v['launches']:5d} {k[:100]}")
for k, v in list(out["kernels"].items())[:4]:
    print(f"{v['launches']:5d} rd {v['fetch_bytes_per_launch'] / 1e6:8.1f} MB wr {v['write_bytes_per_launch'] / 1e6:8.1f} MB {k[:90]}")
