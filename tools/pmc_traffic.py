#!/usr/bin/env python3
"""Fold the two rocprofv3 counter passes of tools/profile_round.sh into profiles/rNN_hbm_traffic_pmc.json.

HBM bytes per launch = (FETCH_SIZE * 2 + WRITE_SIZE) KB: on gfx950 FETCH_SIZE tallies 128-byte requests at 64 B
(MI355X_MICROARCH.md, HBM / rocprofv3 section), both counters are in KB, and each is collected in a pass of its own.
usage: pmc_traffic.py <fetch_counter_collection.csv> <write_counter_collection.csv> <steps_profiled> <out.json>
"""
import csv, json, os, sys, collections

CLASSES = [("gemm_nt", ("gemm_nt_",)), ("gemm_tn", ("gemm_tn_",)), ("attn", ("attn_", "flash_")), ("dwconv", ("dwconv", "dw3x3")),
           ("im2col", ("im2col",)), ("col2im", ("col2im",)), ("bn", ("bn_",)), ("ln", ("layernorm",))]


def klass(name):
    for k, pats in CLASSES:
        if any(p in name for p in pats):
            return k
    return "other"


def short(name):
    return name.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:90]


def load(path, counter):
    """per class and per (kernel, grid) totals of one counter; rocprofv3 writes one row per dispatch and counter"""
    tot, n = collections.Counter(), collections.Counter()
    shape_tot, shape_n = collections.Counter(), collections.Counter()
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        k = klass(r["Kernel_Name"])
        v = float(r["Counter_Value"])
        tot[k] += v
        n[k] += 1
        key = (short(r["Kernel_Name"]), r.get("Grid_Size", r.get("Grid_Size_X", "?")))
        shape_tot[key] += v
        shape_n[key] += 1
    return tot, n, shape_tot, shape_n


def source_hash():
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from geoguessr_ai_amd import _lib
    return _lib.source_hash()


def main():
    fpath, wpath, steps, out = sys.argv[1], sys.argv[2], int(sys.argv[3]), sys.argv[4]
    f, nf, sf, snf = load(fpath, "FETCH_SIZE")
    w, nw, sw, _ = load(wpath, "WRITE_SIZE")
    res = {"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, tools/profile_round.sh) on the command below; FETCH_SIZE doubled "
                     "(gfx950 correction, MI355X_MICROARCH.md HBM section), KB units; launches of the warm-up step are included in steps_profiled",
           "command": os.environ.get("GG_PMC_COMMAND", "python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --precision <fp32|bf16>"),
           "git_head": os.environ.get("GG_GIT_HEAD", "unknown (no .git on the GPU box; pass GG_GIT_HEAD)"),
           "source_hash": source_hash(),          # sha256 of csrc/* + include/gg.h: bench.py marks the traffic stale when the running build differs
           "steps_profiled": steps, "per_kernel_class": {}, "per_kernel_and_grid": []}
    for k in sorted(set(f) | set(w), key=lambda k: -(2 * f[k] + w[k])):
        fetch, write, launches = 2.0 * f[k] * 1024, w[k] * 1024, max(nf[k], 1)
        res["per_kernel_class"][k] = {"launches_per_step": round(nf[k] / steps, 1), "fetch_GB_per_step": round(fetch / steps / 1e9, 2),
                                      "write_GB_per_step": round(write / steps / 1e9, 2),
                                      "traffic_bytes_per_launch": int((fetch + write) / launches)}
    # one row per (kernel instantiation, grid) = per GEMM shape / epilogue class: the table behind "which launches over-fetch"
    for key in sorted(set(sf) | set(sw), key=lambda k: -(2 * sf[k] + sw[k]))[:60]:
        nl = max(snf[key], 1)
        res["per_kernel_and_grid"].append({"kernel": key[0], "grid": key[1], "launches_per_step": round(snf[key] / steps, 2),
                                           "fetch_MB_per_launch": round(2.0 * sf[key] * 1024 / nl / 1e6, 1),
                                           "write_MB_per_launch": round(sw[key] * 1024 / nl / 1e6, 1)})
    res["total_GB_per_step"] = round(sum(2.0 * f[k] * 1024 + w[k] * 1024 for k in set(f) | set(w)) / steps / 1e9, 2)
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps(res["per_kernel_class"].get("gemm_nt")), res["total_GB_per_step"], res["source_hash"][:16])


if __name__ == "__main__":
    main()
