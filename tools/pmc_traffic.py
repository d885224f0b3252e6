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


def gemm_forms(fpath, wpath, steps, launches_json, out_txt):
    """Per GEMM launch FORM (kernel instantiation x grid): measured HBM bytes (FETCH_SIZE x 2 + WRITE_SIZE, KB units) against the algorithmic bytes the
    launch declares (bench.py --dump-launches), joined by issue order within a step: the i-th GEMM-category launch of the instrumented step is the
    i-th gemm_nt_* / gemm_tn_* dispatch of a profiled step."""
    def rows(path, counter):
        r = [x for x in csv.DictReader(open(path)) if x["Counter_Name"] == counter and ("gemm_nt_" in x["Kernel_Name"] or "gemm_tn_" in x["Kernel_Name"])]
        if r and "Dispatch_Id" in r[0]:
            r.sort(key=lambda x: int(x["Dispatch_Id"]))
        return r
    F, W = rows(fpath, "FETCH_SIZE"), rows(wpath, "WRITE_SIZE")
    d = json.load(open(launches_json))
    gemm = [l for l in d["launches"] if l[0] == 0]
    per = len(F) // steps
    if per != len(gemm) or len(W) != len(F):
        print(f"gemm_forms: {len(F)} / {len(W)} GEMM dispatches over {steps} steps vs {len(gemm)} instrumented launches per step: cannot join", file=sys.stderr)
        return
    agg = collections.OrderedDict()
    for i, (cat, ms, fl, by) in enumerate(gemm):
        f, w = F[(steps - 1) * per + i], W[(steps - 1) * per + i]
        assert short(f["Kernel_Name"]) == short(w["Kernel_Name"])
        key = (short(f["Kernel_Name"]), f.get("Grid_Size", "?"))
        a = agg.setdefault(key, [0, 0.0, 0.0, 0.0, 0.0, 0.0])
        a[0] += 1; a[1] += 2048.0 * float(f["Counter_Value"]); a[2] += 1024.0 * float(w["Counter_Value"]); a[3] += by; a[4] += fl; a[5] += ms
    tot_meas = sum(a[1] + a[2] for a in agg.values()); tot_alg = sum(a[3] for a in agg.values())
    with open(out_txt, "w") as o:
        o.write(f"# HBM traffic of every GEMM launch form of one {d['precision']} step (TinyViT-21M-224, 1024 images): PMC (FETCH_SIZE x 2 + WRITE_SIZE) against the\n"
                f"# algorithmic bytes each launch declares; joined by issue order.  total measured {tot_meas / 1e9:.1f} GB, algorithmic {tot_alg / 1e9:.1f} GB, ratio {tot_meas / tot_alg:.3f}\n")
        o.write(f"{'kernel':92s} {'grid':>9s} {'n':>3s} {'alg GB':>8s} {'fetch GB':>9s} {'write GB':>9s} {'ratio':>6s} {'excess GB':>9s} {'ms':>7s} {'TFLOP/s':>8s}\n")
        for (k, g), a in sorted(agg.items(), key=lambda kv: -((kv[1][1] + kv[1][2]) - kv[1][3])):
            o.write(f"{k:92s} {g:>9s} {a[0]:3d} {a[3] / 1e9:8.2f} {a[1] / 1e9:9.2f} {a[2] / 1e9:9.2f} {(a[1] + a[2]) / max(a[3], 1):6.2f} {((a[1] + a[2]) - a[3]) / 1e9:9.2f} "
                    f"{a[5]:7.2f} {a[4] / max(a[5], 1e-9) / 1e9:8.1f}\n")
    print(open(out_txt).read()[:3000])


def source_hash():
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from geoguessr_ai_amd import _lib
    return _lib.source_hash()


def main():
    fpath, wpath, steps, out = sys.argv[1], sys.argv[2], int(sys.argv[3]), sys.argv[4]
    if len(sys.argv) > 6:
        gemm_forms(fpath, wpath, steps, sys.argv[5], sys.argv[6])
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
