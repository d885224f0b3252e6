#!/usr/bin/env python3
"""Fold the two rocprofv3 counter passes of tools/profile_round.sh into profiles/rNN_hbm_traffic_pmc.json.

HBM bytes per launch = (FETCH_SIZE * 2 + WRITE_SIZE) KB: on gfx950 FETCH_SIZE tallies 128-byte requests at 64 B
(MI355X_MICROARCH.md, HBM / rocprofv3 section), both counters are in KB, and each is collected in a pass of its own.
usage: pmc_traffic.py <fetch_counter_collection.csv> <write_counter_collection.csv> <steps_profiled> <out.json>
"""
import csv, json, sys, collections

CLASSES = [("gemm_nt", ("gemm_nt_",)), ("gemm_tn", ("gemm_tn_",)), ("attn", ("attn_", "flash_")), ("dwconv", ("dwconv", "dw3x3")),
           ("im2col", ("im2col",)), ("col2im", ("col2im",)), ("bn", ("bn_",)), ("ln", ("layernorm",))]


def klass(name):
    for k, pats in CLASSES:
        if any(p in name for p in pats):
            return k
    return "other"


def load(path, counter):
    tot, n = collections.Counter(), collections.Counter()
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        k = klass(r["Kernel_Name"])
        tot[k] += float(r["Counter_Value"])
        n[k] += 1
    return tot, n


def main():
    fpath, wpath, steps, out = sys.argv[1], sys.argv[2], int(sys.argv[3]), sys.argv[4]
    f, nf = load(fpath, "FETCH_SIZE")
    w, nw = load(wpath, "WRITE_SIZE")
    res = {"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, tools/profile_round.sh) on bench.py --steps 2 --warmup 1 "
                     "--panoramas 256 --precision <fp32|bf16>; FETCH_SIZE doubled (gfx950 correction, MI355X_MICROARCH.md HBM section), KB units; "
                     "launches of the warm-up step are included in steps_profiled",
           "steps_profiled": steps, "per_kernel_class": {}}
    for k in sorted(set(f) | set(w), key=lambda k: -(2 * f[k] + w[k])):
        fetch, write, launches = 2.0 * f[k] * 1024, w[k] * 1024, max(nf[k], 1)
        res["per_kernel_class"][k] = {"launches_per_step": round(nf[k] / steps, 1), "fetch_GB_per_step": round(fetch / steps / 1e9, 2),
                                      "write_GB_per_step": round(write / steps / 1e9, 2),
                                      "traffic_bytes_per_launch": int((fetch + write) / launches)}
    res["total_GB_per_step"] = round(sum(2.0 * f[k] * 1024 + w[k] * 1024 for k in set(f) | set(w)) / steps / 1e9, 2)
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps(res["per_kernel_class"].get("gemm_nt")), res["total_GB_per_step"])


if __name__ == "__main__":
    main()
