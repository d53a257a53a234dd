"""Post-process two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE; separate runs, kernel trace only) of the same bench command into
per-kernel HBM-side bytes per launch -> profiles/*_pmc_hbm_traffic.json.  Units per /opt/skills/guides/MI355X_MICROARCH.md: both
counters are in KiB; FETCH_SIZE is doubled on gfx950 (it reports half of 16-B/lane streaming reads)."""
import collections, csv, glob, json, sys


def load(dirname, counter):
    f = dirname if dirname.endswith(".csv") else sorted(glob.glob(dirname + "/*/*_counter_collection.csv"))[-1]
    tot, n = collections.defaultdict(float), collections.Counter()
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != counter:
            continue
        k = r["Kernel_Name"]
        tot[k] += float(r["Counter_Value"]); n[k] += 1
    return tot, n


fetch, nf = load(sys.argv[1], "FETCH_SIZE")
write, nw = load(sys.argv[2], "WRITE_SIZE")
out = {"command": sys.argv[4] if len(sys.argv) > 4 else "", "units": "FETCH_SIZE / WRITE_SIZE in KiB; FETCH_SIZE x2 on gfx950 (MI355X_MICROARCH.md)",
       "kernels": {}}
for k in sorted(fetch, key=lambda k: -(2 * fetch[k] + write.get(k, 0.0))):
    launches = nf[k]
    out["kernels"][k[:120]] = {"launches": launches, "fetch_bytes_per_launch": int(2 * 1024 * fetch[k] / launches),
                              "write_bytes_per_launch": int(1024 * write.get(k, 0.0) / max(nw.get(k, 1), 1))}
json.dump(out, open(sys.argv[3], "w"), indent=1)
for k, v in list(out["kernels"].items())[:14]:
    print(f"{v['launches']:5d} {v['fetch_bytes_per_launch'] / 1e6:9.1f} MB rd {v['write_bytes_per_launch'] / 1e6:9.1f} MB wr  {k[:90]}")
