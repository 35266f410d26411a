"""Turn the counter CSVs of tools/pmc_bench.sh into the committed summaries.

    python tools/pmc_traffic.py gpurun_out/<tag>_pmc_fetch/fetch_counter_collection.csv \
        gpurun_out/<tag>_pmc_write/write_counter_collection.csv gpurun_out/<tag>_pmc_sq/sq_counter_collection.csv <out prefix>

writes <prefix>_pmc_traffic.json (what bench.py reads into roofline.traffic: keyed by the sha256 of awq.hip, so a later
change of the kernel source retires it), <prefix>_pmc_awq_gemm_sq.json and <prefix>_pmc_per_kernel.csv (every kernel's
per-launch averages of all counters: the raw material).  FETCH_SIZE and WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE
counts the 128-byte requests of wide streaming reads at 64 bytes and is doubled here.
"""
import collections
import csv
import hashlib
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PROJ = ("qkv", "o", "gate_up", "down")  # launch order of the fused GEMM inside a decoder layer


def short(name):
    name = re.sub(r"^void ", "", name)
    name = re.sub(r"\(.*", "", name)
    return re.sub(r"\s+", " ", name)


GRIDS = {}  # dispatch id -> grid size (threads), from the first file loaded


def load(path):
    """per kernel (short name) -> counter -> list of (dispatch id, value); one row per dispatch (dimension rows added up)"""
    acc = collections.defaultdict(lambda: collections.defaultdict(lambda: collections.defaultdict(float)))
    for r in csv.DictReader(open(path)):
        acc[short(r["Kernel_Name"])][r["Counter_Name"]][int(r["Dispatch_Id"])] += float(r["Counter_Value"])
        GRIDS.setdefault(int(r["Dispatch_Id"]), int(r["Grid_Size"]))
    return {k: {c: sorted(d.items()) for c, d in v.items()} for k, v in acc.items()}


def classify(fetch_rows):
    """dispatch id -> projection.  qkv and o have grids of their own (48 x 4 and 32 x 8 blocks of 512 threads); gate_up
    and down both launch 224 blocks and are told apart by what they fetch (58.7 against 29.4 MB of weights).  The three
    passes run the same program, so the dispatch ids of one pass name the same launches in the others."""
    big = [v for d, v in fetch_rows if GRIDS[d] == 224 * 512]
    mid = (min(big) + max(big)) / 2 if big else 0
    out = {}
    for d, v in fetch_rows:
        g = GRIDS[d]
        out[d] = "qkv" if g == 192 * 512 else "o" if g == 256 * 512 else ("gate_up" if v > mid else "down") if g == 224 * 512 else None
    return out


def main(fetch_csv, write_csv, sq_csv, prefix):
    fetch, write, sq = load(fetch_csv), load(write_csv), load(sq_csv)
    gemm = next(k for k in fetch if k.startswith("sglk::awq_gemm_wide_kernel"))
    per = {}
    which = classify(fetch[gemm]["FETCH_SIZE"])
    for name, table, counter in (("fetch", fetch, "FETCH_SIZE"), ("write", write, "WRITE_SIZE")):
        rows = table[gemm][counter]
        for i, p in enumerate(PROJ):
            vals = [v for d, v in rows if which.get(d) == p]
            per.setdefault(p, {})[name + "_KiB_raw_avg"] = sum(vals) / len(vals)
            per[p]["launches"] = len(vals)
    total = 0.0
    for p in PROJ:
        per[p]["fetch_bytes_corrected"] = int(per[p]["fetch_KiB_raw_avg"] * 1024 * 2)
        per[p]["write_bytes"] = int(per[p]["write_KiB_raw_avg"] * 1024)
        total += per[p]["fetch_bytes_corrected"] + per[p]["write_bytes"]
    digest = hashlib.sha256(open(os.path.join(ROOT, "sglang_security_amd", "csrc", "awq.hip"), "rb").read()).hexdigest()
    others = {}
    for k in fetch:
        if k == gemm or not k.startswith("sglk::"):
            continue
        f = [v for _, v in fetch[k]["FETCH_SIZE"]]
        w = [v for _, v in write.get(k, {}).get("WRITE_SIZE", [])]
        others[k] = {"launches": len(f), "fetch_bytes_corrected": int(sum(f) / len(f) * 2048),
                     "write_bytes": int(sum(w) / max(len(w), 1) * 1024)}
    out = {
        "what": "HBM-side (L2 fabric) bytes per launch of the fused AWQ GEMM inside the decode step (Llama-3-8B, M=64, "
                "layer-fusion forms), rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in two separate runs of "
                "`python3 bench.py --no-cpu-baseline --no-prefix-hit --no-graph --steps 2 --warmup 2` (tools/pmc_bench.sh)",
        "correction": "FETCH_SIZE doubled (gfx950 counts the 128-byte requests of wide streaming reads at 64 bytes); "
                      "WRITE_SIZE as is; both in KiB",
        "awq_hip_sha256": digest, "hidden": 4096, "batch": 64,
        "per_projection": per,
        "awq_gemm_hbm_bytes_per_launch": int(total / 4),
        "other_kernels": others,
    }
    json.dump(out, open(prefix + "_pmc_traffic.json", "w"), indent=1)
    # SQ counters of the GEMM: per projection
    sqo = {"note": "rocprofv3 --pmc (one pass, eight SQ counters) on the same command; SQ_WAVE/WAIT/ACTIVE counters are "
                   "quad-cycles summed over waves, SQ_VALU_MFMA_BUSY_CYCLES is cycles summed over SIMDs; per-launch averages",
           "awq_hip_sha256": digest, "per_projection": {}}
    for i, p in enumerate(PROJ):
        d = {}
        for c, rows in sq[gemm].items():
            vals = [v for dd, v in rows if which.get(dd) == p]
            d[c] = sum(vals) / len(vals)
        wc = d.get("SQ_WAVE_CYCLES", 0) or 1
        d["frac_parked_WAIT_ANY"] = round(d.get("SQ_WAIT_ANY", 0) / wc, 3)
        d["frac_issue_stall_WAIT_INST_ANY"] = round(d.get("SQ_WAIT_INST_ANY", 0) / wc, 3)
        d["frac_issuing_ACTIVE_INST_ANY"] = round(d.get("SQ_ACTIVE_INST_ANY", 0) / wc, 3)
        # SQ_BUSY_CYCLES: cycles summed over the 32 shader engines' sequencers (8 XCDs x 4); 1024 SIMDs on the chip
        if d.get("SQ_BUSY_CYCLES"):
            d["mfma_pipe_busy_frac"] = round(d["SQ_VALU_MFMA_BUSY_CYCLES"] / (d["SQ_BUSY_CYCLES"] / 32 * 1024), 3)
        sqo["per_projection"][p] = d
    json.dump(sqo, open(prefix + "_pmc_awq_gemm_sq.json", "w"), indent=1)
    with open(prefix + "_pmc_per_kernel.csv", "w") as fh:
        fh.write("kernel,counter,launches,avg,min,max\n")
        for table in (fetch, write, sq):
            for k in sorted(table):
                if not k.startswith("sglk::"):
                    continue
                for c, rows in sorted(table[k].items()):
                    v = [x for _, x in rows]
                    fh.write(f'"{k}",{c},{len(v)},{sum(v) / len(v):.1f},{min(v):.1f},{max(v):.1f}\n')
    print(json.dumps({"awq_gemm_hbm_bytes_per_launch": out["awq_gemm_hbm_bytes_per_launch"], "per_projection": per}, indent=1))
    print(json.dumps(sqo["per_projection"], indent=1))


if __name__ == "__main__":
    main(*sys.argv[1:5])
