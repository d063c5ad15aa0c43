#!/usr/bin/env python3
"""Merge rocprofv3 --pmc passes (gpurun_out/<prefix>_<COUNTER>/.../*_counter_collection.csv) into one
JSON summary for profiles/.  usage: make_pmc_json.py <prefix> <out.json> B D mode"""
import collections, csv, glob, json, re, sys
prefix, out, B, D, mode = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), sys.argv[5]
agg = collections.defaultdict(dict)
used = collections.defaultdict(dict)     # kernel -> counter -> (dispatches used, dispatches seen)
def _median(v):
    v = sorted(v)
    n = len(v)
    return v[n // 2] if n % 2 else 0.5 * (v[n // 2 - 1] + v[n // 2])
for path in glob.glob(f"gpurun_out/{prefix}*/*/*_counter_collection.csv"):
    per = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(path)):
        k = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void crossclr::", "").replace("crossclr::", "")
        if "at::" in k or "kernel" not in k: continue
        per[k][r["Counter_Name"]].append((int(r["Grid_Size"]), float(r["Counter_Value"])))
    for k, cs in per.items():
        for c, v in cs.items():
            # a kbench pass also contains the module's one-off self-test launches of the same kernel at 640 / 1152 rows (loss._xf_selftest) and
            # the small synthetic plans of the other stages: only the dispatches of the grid launched MOST OFTEN under the kernel name are the workload,
            # and of those the MEDIAN is reported (the first dispatches of a pass run cold)
            cnt = collections.Counter(g for g, _ in v)
            gmax = max(cnt, key=lambda g: (cnt[g], g))       # the grid launched most often = the timed workload (ties: the larger)
            full = [x for g, x in v if g == gmax]
            agg[k][c] = _median(full)
            used[k][c] = (len(full), len(v))
import hashlib, os
_root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_h = hashlib.sha256()
_d = os.path.join(_root, "crossmodal-contrastive-learning_amd", "csrc")
for _f in sorted(os.listdir(_d)):
    if _f.endswith((".h", ".cpp")):
        _h.update(_f.encode()); _h.update(open(os.path.join(_d, _f), "rb").read())
res = {"csrc_sha": _h.hexdigest()[:16],     # bench.py compares it with the sources it runs: roofline.traffic_stale
       "config": {"B": B, "D": D, "mode": mode, "command": "rocprofv3 --kernel-trace --pmc <one counter group per pass> -- python tools/kbench.py"},
       "note": "MEDIAN over the dispatches of the grid size launched most often per kernel name (dispatches_used / dispatches_seen per kernel: smaller launches of the same kernel -- the module's self-test at 640 / 1152 rows -- are dropped; round 4's summaries averaged them in and read 9 % low). FETCH_SIZE/WRITE_SIZE in KiB. Per MI355X_MICROARCH.md (HBM section) FETCH_SIZE on gfx950 reports half the bytes of a wide (16 B/lane) coalesced read stream, so hbm_bytes_corrected = (2*FETCH_SIZE + WRITE_SIZE)*1024; FETCH_SIZE and WRITE_SIZE need separate passes. SQ_WAVE_CYCLES counts quad-cycles, SQ_VALU_MFMA_BUSY_CYCLES cycles. SQ_INSTS_VALU includes the MFMA instructions (non_mfma_issues_per_mfma subtracts them; VMEM not counted).",
       "kernels": {}}
for k, e in agg.items():
    e = dict(e)
    if "FETCH_SIZE" in e and "WRITE_SIZE" in e:
        e["hbm_bytes_raw"] = (e["FETCH_SIZE"] + e["WRITE_SIZE"]) * 1024
        e["hbm_bytes_corrected"] = (2 * e["FETCH_SIZE"] + e["WRITE_SIZE"]) * 1024
    if "TCC_HIT_sum" in e: e["l2_hit_rate"] = e["TCC_HIT_sum"] / max(1.0, e["TCC_HIT_sum"] + e["TCC_MISS_sum"])
    if e.get("SQ_WAVE_CYCLES", 0) > 0 and "SQ_VALU_MFMA_BUSY_CYCLES" in e:
        e["mfma_busy_over_wave_cycles"] = e["SQ_VALU_MFMA_BUSY_CYCLES"] / (4 * e["SQ_WAVE_CYCLES"])
    if e.get("SQ_INSTS_MFMA", 0) > 0 and all(c in e for c in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS")):
        # SQ_INSTS_VALU COUNTS the MFMA instructions (a pure MFMA loop reads VALU = 1.014 x MFMA: profiles/r04_pmc_valu_counts_mfma.txt)
        e["non_mfma_issues_per_mfma"] = (e["SQ_INSTS_VALU"] - e["SQ_INSTS_MFMA"] + e["SQ_INSTS_SALU"] + e["SQ_INSTS_LDS"]) / e["SQ_INSTS_MFMA"]
    du = used.get(k, {})
    if du:
        e["dispatches_used"] = min(a for a, _ in du.values())
        e["dispatches_seen"] = max(b for _, b in du.values())
    res["kernels"][k] = e
json.dump(res, open(out, "w"), indent=1)
for k, e in res["kernels"].items():
    print(f"{k[:44]:44s} hbm {e.get('hbm_bytes_corrected',0)/1e6:7.1f} MB  L2hit {e.get('l2_hit_rate',0):.2f}  mfma/wave {e.get('mfma_busy_over_wave_cycles',0):.3f}  ldsconf {e.get('SQ_LDS_BANK_CONFLICT')}")
