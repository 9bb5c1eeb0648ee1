#!/usr/bin/env python3
"""Summarises a tools/profile_gpu.sh output directory: per-kernel time (kernel trace) and per-kernel PMC averages."""
import csv
import glob
import os
import re
import sys
from collections import defaultdict


def find(d, pat):
    r = glob.glob(os.path.join(d, "**", pat), recursive=True)
    return r[0] if r else None


def short(name):
    # "(anonymous namespace)::" sits in FRONT of a set-up kernel's name: strip it before cutting at the argument list's "("
    name = name.replace("(anonymous namespace)::", "")
    m = re.search(r"rocprim::ROCPRIM_\w+::detail::trampoline_kernel<rocprim::ROCPRIM_\w+::detail::wrapped_(\w+)_config<[^,]*, ([^>]*?)>, ", name)
    if m:   # the device primitives of the set-up (hipCUB -> rocPRIM): which primitive, on which key / value types
        types = m.group(2).replace("rocprim::", "").replace("ROCPRIM_400200_NS::", "").replace("unsigned long", "u64").replace("empty_type", "-")
        return f"rocprim {m.group(1)} <{types}>"
    name = re.sub(r"rocprim::ROCPRIM_\w+::detail::", "rocprim ", name)
    name = name.split("(")[0]
    for ns in ("fdapde_hip::", "fdapde_engine::", "void "):
        name = name.replace(ns, "")
    return name.strip()


def main(out):
    kt = find(os.path.join(out, "trace"), "*kernel_trace.csv")
    if kt:
        dur = defaultdict(list)
        for r in csv.DictReader(open(kt)):
            dur[short(r["Kernel_Name"])].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
        tot = sum(sum(v) for v in dur.values())
        print("== kernel trace (us) ==")
        print(f"{'kernel':60s} {'calls':>7s} {'total_us':>12s} {'avg_us':>10s} {'min_us':>10s} {'max_us':>10s} {'%':>6s}")
        for k, v in sorted(dur.items(), key=lambda kv: -sum(kv[1])):
            print(f"{k[:60]:60s} {len(v):7d} {sum(v):12.1f} {sum(v)/len(v):10.2f} {min(v):10.2f} {max(v):10.2f} {100*sum(v)/tot:6.2f}")
    for sub in ("pmc_fetch", "pmc_write", "pmc_l2", "pmc_sq", "pmc_sq2"):
        f = find(os.path.join(out, sub), "*counter_collection.csv")
        if not f:
            print(f"== {sub}: no counter file ==")
            continue
        acc = defaultdict(lambda: defaultdict(list))
        for r in csv.DictReader(open(f)):
            acc[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
        print(f"== {sub} (average per dispatch) ==")
        for k, cs in sorted(acc.items()):
            if not os.environ.get("SUMMARY_ALL_KERNELS") and not any(s in k for s in ("k_spmv", "k_cg", "k_assemble", "k_bicg", "k_scale", "k_persist")):
                continue
            print(f"{k[:60]:60s} " + "  ".join(f"{c}={sum(v)/len(v):.4g} (n={len(v)})" for c, v in sorted(cs.items())))


def pmc_json(out):
    """-> dict with the SpMV kernel's per-dispatch PMC averages (KB, as rocprofv3 reports them) and trace average"""
    import json

    res = {}
    kt = find(os.path.join(out, "trace"), "*kernel_trace.csv")
    if kt:
        d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in csv.DictReader(open(kt)) if "k_spmv" in r["Kernel_Name"]]
        d = [x for x in d if x > 10.0]   # launches after convergence return at once
        if d:
            res["spmv_trace_avg_us"] = sum(d) / len(d)
            res["spmv_trace_calls"] = len(d)
    for sub, ctr in (("pmc_fetch", "FETCH_SIZE"), ("pmc_write", "WRITE_SIZE")):
        f = find(os.path.join(out, sub), "*counter_collection.csv")
        if f:
            v = [float(r["Counter_Value"]) for r in csv.DictReader(open(f)) if "k_spmv" in r["Kernel_Name"] and r["Counter_Name"] == ctr]
            v = [x for x in v if x > 1000.0]
            if v:
                res[ctr + "_KB"] = sum(v) / len(v)
    # the persistent single-launch CG: one dispatch per solve; traffic per solve (bench.py divides by its iteration count)
    if kt:
        d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in csv.DictReader(open(kt)) if "k_cg_persist" in r["Kernel_Name"]]
        if d:
            res["persist_trace_avg_us"] = sum(d) / len(d)
            res["persist_trace_calls"] = len(d)
    for sub, ctr in (("pmc_fetch", "FETCH_SIZE"), ("pmc_write", "WRITE_SIZE")):
        f = find(os.path.join(out, sub), "*counter_collection.csv")
        if f:
            v = [float(r["Counter_Value"]) for r in csv.DictReader(open(f)) if "k_cg_persist" in r["Kernel_Name"] and r["Counter_Name"] == ctr]
            if v:
                res["persist_" + ctr + "_KB"] = sum(v) / len(v)
    if "persist_FETCH_SIZE_KB" in res and "persist_WRITE_SIZE_KB" in res:
        res["persist_hbm_bytes_per_solve"] = (2.0 * res["persist_FETCH_SIZE_KB"] + res["persist_WRITE_SIZE_KB"]) * 1024.0
        res["correction"] = "2 x FETCH_SIZE + WRITE_SIZE (KB x 1024), separate --pmc passes"
    if "FETCH_SIZE_KB" in res and "WRITE_SIZE_KB" in res:
        # MI355X_MICROARCH.md (HBM): on gfx950 FETCH_SIZE reads 1/2 of the bytes of wide coalesced reads; WRITE_SIZE exact
        res["hbm_bytes_per_launch"] = (2.0 * res["FETCH_SIZE_KB"] + res["WRITE_SIZE_KB"]) * 1024.0
        res["correction"] = "2 x FETCH_SIZE + WRITE_SIZE (KB x 1024), separate --pmc passes"
    # which workload these passes ran (bench.py takes the counter bytes only for the workload they were collected on): from the bench line
    # the traced run printed
    try:
        import re

        for line in open(os.path.join(out, "trace.log")):
            if line.startswith("{") and '"metric"' in line:
                b = json.loads(line)
                m = re.search(r"(\d+)\^3 x 6", b["config"]["workload"])
                if m:
                    res["nx"] = int(m.group(1))
                if b["config"].get("persistent_launch"):
                    res["persist_iterations"] = int(b["config"]["cg_iterations"])
                    if "persist_hbm_bytes_per_solve" in res:
                        res["note"] = ("persist_*: the single-launch CG, one dispatch per solve of %d iterations; hbm bytes per iteration = %.1f MB.  "
                                       "spmv_* / FETCH / WRITE: the multi-launch SpMV (k_spmv_team2) of the lift and the fall-back path."
                                       % (res["persist_iterations"], res["persist_hbm_bytes_per_solve"] / res["persist_iterations"] / 1e6))
    except Exception as e:   # (the json stays usable by hand)
        res["workload_note"] = "bench line of the traced run not found: %s" % e
    # what the counters were collected ON: the commit the caller names (FDAPDE_HEAD: there is no .git on the GPU box) and a digest of the solve's kernel
    # sources as they lie in this snapshot -- bench.py prints both (roofline.traffic_head) and flags a file that no longer matches the kernels it runs
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    try:
        import bench as _bench

        res["kernel_source_sha16"] = _bench.solve_kernel_sha16()
    except Exception as e:
        res["kernel_source_sha16"] = None
        res["kernel_source_sha16_error"] = str(e)[:200]
    res["head"] = os.environ.get("FDAPDE_HEAD")
    json.dump(res, open(os.path.join(out, "spmv_pmc.json"), "w"), indent=1)
    print("== spmv_pmc.json ==")
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main(sys.argv[1])
    pmc_json(sys.argv[1])
