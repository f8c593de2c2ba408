#!/usr/bin/env python3
"""Turn the raw output of scripts/profile_round.sh (gpurun_out/<tag>/) into the small summaries kept under profiles/<round>/.
  summarize_profile.py <tag> <dest> pmc    -> pmc_per_launch.csv, pmc_frame.json (one frame alone: HBM bytes, VALU issue, lane use)
  summarize_profile.py <tag> <dest> stats  -> kernel_stats.csv (frames in flight), kernel_stats_alone.csv (one at a time), bench lines"""
import collections
import csv
import glob
import json
import os
import shutil
import sys

tag, dest, what = sys.argv[1], sys.argv[2], sys.argv[3]
src = os.path.join("gpurun_out", tag)
os.makedirs(dest, exist_ok=True)


def load(path):
    rows = list(csv.DictReader(open(glob.glob(path)[0])))
    d = collections.OrderedDict()
    for r in rows:
        d.setdefault((int(r["Dispatch_Id"]), r["Kernel_Name"]), {})[r["Counter_Name"]] = float(r["Counter_Value"])
    return d


def last_frame(d):
    """dispatches of the last frame: from its first launch of this library (k_frame, or the camera-ray launch of the
    per-depth path, k_trace<MODE, OVF, PRIMARY = true, ...>) to the end"""
    import re
    keys = list(d.keys())
    start = max(i for i, k in enumerate(keys) if "k_frame" in k[1] or re.search(r"k_trace<\d+, (true|false), true,", k[1]))
    return [(k, d[k]) for k in keys[start:] if k[1].startswith(("void k_", "k_"))]


if what == "pmc":
    fetch, write, sq = (load(src + "/pmc_%s/*/*counter_collection.csv" % n) for n in ("fetch", "write", "sq"))
    rows = []
    tot_f = tot_w = valu = thr = act = wait = wcyc = 0.0
    for (kf, cf), (kw, cw), (ks, cs) in zip(last_frame(fetch), last_frame(write), last_frame(sq)):
        name = kf[1].split("(")[0].replace("void ", "")
        f_kb, w_kb = cf.get("FETCH_SIZE", 0.0), cw.get("WRITE_SIZE", 0.0)
        hit, miss = cw.get("TCC_HIT_sum", 0.0), cw.get("TCC_MISS_sum", 0.0)
        wc = cs.get("SQ_WAVE_CYCLES", 1.0)
        rows.append(dict(kernel=name, fetch_MB_raw=f_kb / 1024, fetch_MB_x2=2 * f_kb / 1024, write_MB=w_kb / 1024,
                         l2_hit=hit / max(1.0, hit + miss), valu_insts=cs.get("SQ_INSTS_VALU", 0.0), salu_insts=cs.get("SQ_INSTS_SALU", 0.0),
                         vmem_rd_insts=cs.get("SQ_INSTS_VMEM_RD", 0.0), wait_frac=cs.get("SQ_WAIT_ANY", 0.0) / wc,
                         issue_frac=cs.get("SQ_ACTIVE_INST_ANY", 0.0) / wc,
                         valu_lane_util=cs.get("SQ_THREAD_CYCLES_VALU", 0.0) / max(1.0, 64 * cs.get("SQ_ACTIVE_INST_VALU", 0.0))))
        tot_f += 2 * f_kb * 1024
        tot_w += w_kb * 1024
        valu += cs.get("SQ_INSTS_VALU", 0.0)
        thr += cs.get("SQ_THREAD_CYCLES_VALU", 0.0)
        act += cs.get("SQ_ACTIVE_INST_VALU", 0.0)
        wait += cs.get("SQ_WAIT_ANY", 0.0)
        wcyc += cs.get("SQ_WAVE_CYCLES", 0.0)
    with open(os.path.join(dest, "pmc_per_launch.csv"), "w", newline="") as f:
        wr = csv.DictWriter(f, fieldnames=list(rows[0].keys()))
        wr.writeheader()
        for r in rows:
            wr.writerow({k: (round(v, 4) if isinstance(v, float) else v) for k, v in r.items()})
    json.dump({"workload": "dragon 3840x2160, one frame alone (scripts/prof_frame.py --workload dragon --reps 1)",
               "hbm_bytes_per_frame": tot_f + tot_w, "fetch_bytes_per_frame_x2_corrected": tot_f, "write_bytes_per_frame": tot_w,
               "valu_insts_per_frame": valu, "valu_lane_util": thr / max(1.0, 64 * act), "wait_frac": wait / max(1.0, wcyc), "kernel_ms": True,
               "launches": [r["kernel"] for r in rows],
               "how": "rocprofv3 --pmc in three separate passes (FETCH_SIZE; WRITE_SIZE TCC_HIT_sum TCC_MISS_sum; eight SQ_ counters) over one frame; "
                      "FETCH_SIZE doubled as the MI355X guide prescribes for gfx950 (it tallies 128-B requests at 64 B); KB -> bytes x1024; "
                      "valu_insts = SQ_INSTS_VALU (wave-level); lane use = SQ_THREAD_CYCLES_VALU / (64 x SQ_ACTIVE_INST_VALU)"},
              open(os.path.join(dest, "pmc_frame.json"), "w"), indent=1)
    print(open(os.path.join(dest, "pmc_per_launch.csv")).read())
else:
    shutil.copyfile(glob.glob(src + "/stats/*/*kernel_stats.csv")[0], os.path.join(dest, "kernel_stats.csv"))
    shutil.copyfile(glob.glob(src + "/stats_alone/*/*kernel_stats.csv")[0], os.path.join(dest, "kernel_stats_alone.csv"))
    for n in ("bench.json", "bench_under_rocprof.json", "bench_alone_under_rocprof.json"):
        shutil.copyfile(os.path.join(src, n), os.path.join(dest, n))
    print(open(os.path.join(dest, "kernel_stats_alone.csv")).read()[:1500])
