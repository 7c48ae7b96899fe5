#!/usr/bin/env python3
"""Turn the two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) of tools/pmc_run.py into a JSON summary
and, with --table, merge it into profiles/pmc_traffic.json -- the table bench.py reads
`roofline.traffic` from, keyed by workload | storage | kernel so that a measurement is only ever
quoted for the case it was taken on.

usage: python tools/pmc_parse.py <fetch_dir> <write_dir> <out.json> [--workload W --storage sym|plain --round rNN_x --table profiles/pmc_traffic.json
                                  --layout layout.json]
  --layout: the fingerprint tools/pmc_run.py --layout-out wrote (stored values, residual size and form, work items, format
  bytes of the plan the counters were taken on); bench.py quotes an entry only for a plan with the same fingerprint.

Units and corrections as MI355X_MICROARCH.md (HBM / rocprofv3) prescribes: the counters are in
KiB; on gfx950 FETCH_SIZE reports exactly half of a wide coalesced streaming read, so it is
doubled -- the factor is re-measured here on ehyb_read_kernel (1 GiB of 16-byte-per-lane loads
per launch) and that measured factor is what is applied to the SpMV kernels.
"""
import csv
import glob
import json
import os
import sys


def per_kernel(dirpath, counter):
    out = {}
    files = glob.glob(os.path.join(dirpath, "**", "*counter_collection.csv"), recursive=True)
    for f in files:
        for row in csv.DictReader(open(f)):
            if row.get("Counter_Name") != counter:
                continue
            name = row["Kernel_Name"].split("(")[0].replace("void ", "")
            out.setdefault(name, []).append(float(row["Counter_Value"]))
    return out


def main():
    fetch_dir, write_dir, out_path = sys.argv[1:4]
    opts = dict(zip(sys.argv[4::2], sys.argv[5::2]))
    fetch = per_kernel(fetch_dir, "FETCH_SIZE")
    write = per_kernel(write_dir, "WRITE_SIZE")
    res = {"units": "bytes per launch; counters are KiB", "kernels": {}}
    probe = [k for k in fetch if "ehyb_read_kernel" in k]
    factor = 2.0
    if probe:
        vals = fetch[probe[0]]
        mean_kib = sum(vals) / len(vals)
        factor = (1 << 30) / (mean_kib * 1024.0)
        res["fetch_calibration"] = {"kernel": probe[0], "known_bytes": 1 << 30, "FETCH_SIZE_KiB_mean": mean_kib,
                                    "factor": factor, "guide_factor": 2.0}
    for name in sorted(set(fetch) | set(write)):
        fv, wv = fetch.get(name, []), write.get(name, [])
        f_raw = sum(fv) / len(fv) * 1024 if fv else None
        w_raw = sum(wv) / len(wv) * 1024 if wv else None
        res["kernels"][name] = {
            "launches": max(len(fv), len(wv)),
            "FETCH_SIZE_bytes_raw": f_raw, "WRITE_SIZE_bytes": w_raw,
            "fetch_bytes_corrected": None if f_raw is None else f_raw * factor,
            "hbm_bytes_per_launch": None if f_raw is None else f_raw * factor + (w_raw or 0.0),
        }
    for name, k in res["kernels"].items():
        if "ehyb_ell_kernel" in name and k["hbm_bytes_per_launch"]:
            res["ehyb_ell_kernel_hbm_bytes_per_launch"] = k["hbm_bytes_per_launch"]
        if "ehyb_er_kernel" in name and k["hbm_bytes_per_launch"]:
            res["ehyb_er_kernel_hbm_bytes_per_launch"] = k["hbm_bytes_per_launch"]
    res["workload"], res["storage"] = opts.get("--workload"), opts.get("--storage")
    layout = None
    if "--layout" in opts:
        try:
            layout = json.load(open(opts["--layout"]))
        except (OSError, ValueError):
            layout = None
    res["layout"] = layout
    json.dump(res, open(out_path, "w"), indent=1)
    if "--table" in opts and res["workload"] and res["storage"]:
        try:
            tab = json.load(open(opts["--table"]))
        except (OSError, ValueError):
            tab = {"units": "HBM bytes per launch = FETCH_SIZE x calibrated factor + WRITE_SIZE (separate --pmc passes)", "entries": {}}
        # the two passes of the panel residual are one "residual launch" to bench.py: their bytes add up
        pb = [k for name, k in res["kernels"].items() if "ehyb_pb_" in name and k["hbm_bytes_per_launch"]]
        if len(pb) == 2:
            res["kernels"]["ehyb_pb_scale_kernel+ehyb_pb_reduce_kernel"] = {
                "launches": min(k["launches"] for k in pb), "FETCH_SIZE_bytes_raw": sum(k["FETCH_SIZE_bytes_raw"] for k in pb),
                "WRITE_SIZE_bytes": sum(k["WRITE_SIZE_bytes"] or 0.0 for k in pb),
                "fetch_bytes_corrected": sum(k["fetch_bytes_corrected"] for k in pb),
                "hbm_bytes_per_launch": sum(k["hbm_bytes_per_launch"] for k in pb)}
        for name, k in res["kernels"].items():
            base = ("ehyb_pb_scale_kernel+ehyb_pb_reduce_kernel" if name.startswith("ehyb_pb_scale_kernel+") else
                    "ehyb_ell_kernel" if "ehyb_ell_kernel" in name else ("ehyb_er_kernel" if "ehyb_er_kernel" in name else None))
            if base and k["hbm_bytes_per_launch"]:
                tab["entries"][f"{res['workload']}|{res['storage']}|{base}"] = {
                    "hbm_bytes_per_launch": k["hbm_bytes_per_launch"], "fetch_bytes_corrected": k["fetch_bytes_corrected"],
                    "write_bytes": k["WRITE_SIZE_bytes"], "fetch_factor": factor, "instantiation": name,
                    "launches": k["launches"], "evidence": opts.get("--round"), "layout": layout}
        json.dump(tab, open(opts["--table"], "w"), indent=1, sort_keys=True)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
