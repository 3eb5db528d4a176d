#!/usr/bin/env python3
"""Turn two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate runs, as MI355X_MICROARCH.md
prescribes) into profiles/pmc_traffic.json: mean HBM-side bytes per launch of each kernel class.

  rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_fetch -- python3 bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-kernel-timing
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_write -- python3 bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-kernel-timing
  python tools/pmc_traffic.py gpurun_out/pmc_fetch gpurun_out/pmc_write profiles/pmc_traffic.json [profiles/rNN_pmc_hbm_traffic.csv]

HBM bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024: on gfx950 FETCH_SIZE counts 128-byte requests of wide
streaming reads at 64 bytes (the guide's correction); WRITE_SIZE is exact for 16-byte-per-lane stores.
"""
import csv, glob, json, os, re, sys
from collections import defaultdict


def collect(d, counter):
    acc = defaultdict(lambda: [0.0, set()])
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            if row.get("Counter_Name") != counter:
                continue
            a = acc[row["Kernel_Name"]]
            a[0] += float(row["Counter_Value"])
            a[1].add(row.get("Dispatch_Id") or row.get("Correlation_Id"))
    return {k: (v[0], len(v[1])) for k, v in acc.items()}


def klass(name):
    m = re.match(r"(?:void )?(k_[a-z_0-9]+)", name)
    base = m.group(1) if m else name
    if base == "k_dwp":
        return "k_dw"
    if base == "k_fwd":
        return "k_fwd_slab" if "<1," in name.replace(" ", "") else "k_fwd"
    return base


def main():
    fdir, wdir, out = sys.argv[1:4]
    fetch, write = collect(fdir, "FETCH_SIZE"), collect(wdir, "WRITE_SIZE")
    res, rows = {}, []
    for name in sorted(set(fetch) | set(write)):
        if not re.match(r"(?:void )?k_", name):
            continue
        fs, fn = fetch.get(name, (0.0, 0))
        ws, wn = write.get(name, (0.0, 0))
        n = max(fn, wn, 1)
        fkb, wkb = fs / max(fn, 1), ws / max(wn, 1)
        e = {"kernel": name, "launches": n, "FETCH_SIZE_KB": round(fkb, 1), "WRITE_SIZE_KB": round(wkb, 1),
             "hbm_bytes_per_launch": int(round((2 * fkb + wkb) * 1024))}
        rows.append(e)
        k = klass(name)
        if k not in res or res[k]["launches"] < n:
            res[k] = e
    json.dump(res, open(out, "w"), indent=1)
    if len(sys.argv) > 4:
        with open(sys.argv[4], "w") as f:
            w = csv.writer(f)
            w.writerow(["kernel", "launches", "FETCH_SIZE_KB_per_launch", "WRITE_SIZE_KB_per_launch", "hbm_bytes_per_launch"])
            for e in rows:
                w.writerow([e["kernel"], e["launches"], e["FETCH_SIZE_KB"], e["WRITE_SIZE_KB"], e["hbm_bytes_per_launch"]])
    for e in rows:
        print("%-60s n=%4d  fetch %10.1f KB  write %10.1f KB  hbm %12d B" % (e["kernel"][:60], e["launches"], e["FETCH_SIZE_KB"], e["WRITE_SIZE_KB"], e["hbm_bytes_per_launch"]))


if __name__ == "__main__":
    main()
