#!/usr/bin/env python3
"""Turn the raw output of scripts/profile_config.sh (gpurun_out/<tag>/<workload>/) into the summaries kept under profiles/<round>/.
  summarize_config.py <tag> <workload> <dest> pmc    -> pmc_<workload>.json + pmc_<workload>_per_launch.csv: ONE frame alone, every
                                                        launch of it: HBM bytes, L2 hit rate, VALU issue and lane use, L1 / texture
                                                        path busy and stall fractions
  summarize_config.py <tag> <workload> <dest> stats  -> <workload>.json: the bench line, the kernel-trace averages of the same command
                                                        (frames in flight, and one at a time) and the counters above in one file"""
import collections
import csv
import glob
import json
import os
import re
import sys

tag, wl, dest, what = sys.argv[1], sys.argv[2], sys.argv[3], sys.argv[4]
src = os.path.join("gpurun_out", tag, wl)
os.makedirs(dest, exist_ok=True)


def load(name):
    paths = glob.glob(src + "/pmc_%s/*/*counter_collection.csv" % name)
    if not paths:
        return None
    d = collections.OrderedDict()
    for r in csv.DictReader(open(paths[0])):
        d.setdefault((int(r["Dispatch_Id"]), r["Kernel_Name"]), {})[r["Counter_Name"]] = float(r["Counter_Value"])
    return d


def last_frame(d):
    """dispatches of the last frame: from its first launch of this library (k_frame, or the camera-ray launch of the
    per-depth path, k_trace<MODE, OVF, PRIMARY = true, ...>) to the end"""
    keys = list(d.keys())
    start = max(i for i, k in enumerate(keys) if "k_frame" in k[1] or re.search(r"k_trace<\d+, (true|false), true,", k[1]))
    return [(k, d[k]) for k in keys[start:] if k[1].startswith(("void k_", "k_"))]


def short(name):
    return name.split("(")[0].replace("void ", "")


if what == "pmc":
    passes = {n: load(n) for n in ("fetch", "write", "sq", "l1a", "l1b")}
    frames = {n: last_frame(d) for n, d in passes.items() if d}
    n = len(frames["sq"])
    assert all(len(f) == n for f in frames.values()), {k: len(v) for k, v in frames.items()}
    rows = []
    tot = collections.Counter()
    for i in range(n):
        c = {}
        for f in frames.values():
            c.update(f[i][1])
        name = short(frames["sq"][i][0][1])
        wc = max(1.0, c.get("SQ_WAVE_CYCLES", 0.0))
        gui = c.get("GRBM_GUI_ACTIVE", 0.0)            # summed over the 8 XCDs
        cu_cycles = gui / 8.0                            # cycles of the launch; per-CU blocks (TA, TD, TCP) are summed over 256 CUs
        per_cu = max(1.0, cu_cycles * 256.0)
        row = collections.OrderedDict(
            kernel=name, fetch_MB_x2=2 * c.get("FETCH_SIZE", 0.0) / 1024, write_MB=c.get("WRITE_SIZE", 0.0) / 1024,
            l2_hit=c.get("TCC_HIT_sum", 0.0) / max(1.0, c.get("TCC_HIT_sum", 0.0) + c.get("TCC_MISS_sum", 0.0)),
            valu_insts=c.get("SQ_INSTS_VALU", 0.0), salu_insts=c.get("SQ_INSTS_SALU", 0.0), vmem_rd_insts=c.get("SQ_INSTS_VMEM_RD", 0.0), lds_insts=c.get("SQ_INSTS_LDS", 0.0),
            valu_active_quad_cycles=c.get("SQ_ACTIVE_INST_VALU", 0.0),
            wait_frac=c.get("SQ_WAIT_ANY", 0.0) / wc, issue_frac=c.get("SQ_ACTIVE_INST_ANY", 0.0) / wc, issue_stall_frac=c.get("SQ_WAIT_INST_ANY", 0.0) / wc,
            valu_lane_util=c.get("SQ_THREAD_CYCLES_VALU", 0.0) / max(1.0, 64 * c.get("SQ_ACTIVE_INST_VALU", 0.0)),
            gui_active_cycles=cu_cycles,
            ta_busy=c.get("TA_TA_BUSY_sum", 0.0) / per_cu, ta_addr_stalled_by_tc=c.get("TA_ADDR_STALLED_BY_TC_CYCLES_sum", 0.0) / per_cu,
            td_busy=c.get("TD_TD_BUSY_sum", 0.0) / per_cu, td_tc_stall=c.get("TD_TC_STALL_sum", 0.0) / per_cu,
            tcp_busy=c.get("TCP_GATE_EN1_sum", 0.0) / per_cu, tcp_pending_stall=c.get("TCP_PENDING_STALL_CYCLES_sum", 0.0) / per_cu,
            tcp_ta_data_stall=c.get("TCP_TCP_TA_DATA_STALL_CYCLES_sum", 0.0) / per_cu, tcp_tagconflict_stall=c.get("TCP_READ_TAGCONFLICT_STALL_CYCLES_sum", 0.0) / per_cu,
            tcp_tcr_stall=c.get("TCP_TCR_TCP_STALL_CYCLES_sum", 0.0) / per_cu,
            tcp_hit=1.0 - c.get("TCP_TCC_READ_REQ_sum", 0.0) / max(1.0, c.get("TCP_TOTAL_CACHE_ACCESSES_sum", 0.0)),
            tcp_tcc_read_latency_cycles=c.get("TCP_TCC_READ_REQ_LATENCY_sum", 0.0) / max(1.0, c.get("TCP_TCC_READ_REQ_sum", 0.0)),
            sq_busy_cu_cycles=c.get("SQ_BUSY_CU_CYCLES", 0.0), lds_bank_conflict_frac=c.get("SQ_LDS_BANK_CONFLICT", 0.0) / max(1.0, c.get("SQ_LDS_IDX_ACTIVE", 0.0)),
            lds_active_frac_of_wave_cycles=c.get("SQ_ACTIVE_INST_LDS", 0.0) / wc, lds_issue_stall_frac=c.get("SQ_WAIT_INST_LDS", 0.0) / wc)
        rows.append(row)
        for k in ("FETCH_SIZE", "WRITE_SIZE", "TCC_HIT_sum", "TCC_MISS_sum", "SQ_INSTS_VALU", "SQ_THREAD_CYCLES_VALU", "SQ_ACTIVE_INST_VALU", "SQ_WAIT_ANY", "SQ_WAVE_CYCLES",
                  "GRBM_GUI_ACTIVE", "TA_TA_BUSY_sum", "TD_TD_BUSY_sum", "TCP_GATE_EN1_sum", "TCP_PENDING_STALL_CYCLES_sum", "TA_ADDR_STALLED_BY_TC_CYCLES_sum",
                  "TCP_TOTAL_CACHE_ACCESSES_sum", "TCP_TCC_READ_REQ_sum", "SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE"):
            tot[k] += c.get(k, 0.0)
    with open(os.path.join(dest, "pmc_%s_per_launch.csv" % wl), "w", newline="") as f:
        wr = csv.DictWriter(f, fieldnames=list(rows[0].keys()))
        wr.writeheader()
        for r in rows:
            wr.writerow({k: (round(v, 4) if isinstance(v, float) else v) for k, v in r.items()})
    per_cu = max(1.0, tot["GRBM_GUI_ACTIVE"] / 8.0 * 256.0)
    out = {"workload": "%s, one frame alone (scripts/prof_frame.py --workload %s --reps 1)" % (wl, wl),
           "hbm_bytes_per_frame": 2 * tot["FETCH_SIZE"] * 1024 + tot["WRITE_SIZE"] * 1024,
           "fetch_bytes_per_frame_x2_corrected": 2 * tot["FETCH_SIZE"] * 1024, "write_bytes_per_frame": tot["WRITE_SIZE"] * 1024,
           "l2_hit": tot["TCC_HIT_sum"] / max(1.0, tot["TCC_HIT_sum"] + tot["TCC_MISS_sum"]),
           "valu_insts_per_frame": tot["SQ_INSTS_VALU"], "valu_active_quad_cycles": tot["SQ_ACTIVE_INST_VALU"],
           "valu_lane_util": tot["SQ_THREAD_CYCLES_VALU"] / max(1.0, 64 * tot["SQ_ACTIVE_INST_VALU"]), "wait_frac": tot["SQ_WAIT_ANY"] / max(1.0, tot["SQ_WAVE_CYCLES"]),
           "gui_active_cycles": tot["GRBM_GUI_ACTIVE"] / 8.0,
           "ta_busy": tot["TA_TA_BUSY_sum"] / per_cu, "ta_addr_stalled_by_tc": tot["TA_ADDR_STALLED_BY_TC_CYCLES_sum"] / per_cu, "td_busy": tot["TD_TD_BUSY_sum"] / per_cu,
           "tcp_busy": tot["TCP_GATE_EN1_sum"] / per_cu, "tcp_pending_stall": tot["TCP_PENDING_STALL_CYCLES_sum"] / per_cu,
           "tcp_hit": 1.0 - tot["TCP_TCC_READ_REQ_sum"] / max(1.0, tot["TCP_TOTAL_CACHE_ACCESSES_sum"]),
           "lds_bank_conflict_frac": tot["SQ_LDS_BANK_CONFLICT"] / max(1.0, tot["SQ_LDS_IDX_ACTIVE"]),
           "launches": [r["kernel"] for r in rows],
           "how": "rocprofv3 --pmc, five separate passes over one frame rendered alone (FETCH_SIZE; WRITE_SIZE TCC_HIT_sum TCC_MISS_sum; eight SQ_ counters; "
                  "TCP/TA/TD busy and stall counters with GRBM_GUI_ACTIVE; TCP stall and LDS counters).  FETCH_SIZE doubled as the MI355X guide prescribes for gfx950 "
                  "(it tallies 128-B requests at 64 B); KB -> bytes x 1024.  valu_insts = SQ_INSTS_VALU (wave-level); valu_active_quad_cycles = SQ_ACTIVE_INST_VALU "
                  "(x 4 = SIMD cycles with a VALU instruction in flight); lane use = SQ_THREAD_CYCLES_VALU / (64 x SQ_ACTIVE_INST_VALU); wait_frac = SQ_WAIT_ANY / SQ_WAVE_CYCLES.  "
                  "Per-CU blocks (TA, TD, TCP: *_sum over 256 CUs) are divided by 256 x the launch's cycles, GRBM_GUI_ACTIVE / 8 (summed over 8 XCDs): fractions of the launch."}
    json.dump(out, open(os.path.join(dest, "pmc_%s.json" % wl), "w"), indent=1)
    print(json.dumps({k: v for k, v in out.items() if k not in ("how", "launches")}, indent=1))
else:
    def stats(name):
        p = glob.glob(src + "/%s/*/*kernel_stats.csv" % name)
        if not p:
            return None
        rows = []
        for r in csv.DictReader(open(p[0])):
            if r["Name"].startswith(("void k_", "k_", "__amd_rocclr")):
                rows.append({"kernel": short(r["Name"]), "calls": int(r["Calls"]), "avg_us": float(r["AverageNs"]) / 1e3, "total_ms": float(r["TotalDurationNs"]) / 1e6, "pct": float(r["Percentage"])})
        return rows

    def line(name):
        p = os.path.join(src, name)
        try:
            return json.loads(open(p).read().strip().splitlines()[-1])
        except Exception as e:
            return {"error": str(e)}
    bench = line("bench.json")
    out = {"workload": wl, "bench": bench,
           "kernel_trace_frames_in_flight": stats("stats"), "kernel_trace_one_at_a_time": stats("stats_alone"),
           "bench_under_rocprof": {k: line("bench_under_rocprof.json").get(k) for k in ("value", "ms_per_step", "frame_alone_ms")},
           "bench_alone_under_rocprof": {k: line("bench_alone_under_rocprof.json").get(k) for k in ("value", "ms_per_step", "frame_alone_ms")},
           "counters_one_frame_alone": json.load(open(os.path.join(dest, "pmc_%s.json" % wl))),
           "how": "scripts/profile_config.sh <tag> %s: counter passes, then `python3 bench.py --workload %s`, then `rocprofv3 --kernel-trace --stats` of the same command "
                  "(frames in flight) and of the same with --pipeline 1 (one frame at a time: its average launch duration is what roofline.launch_ms_alone must agree with)" % (wl, wl)}
    json.dump(out, open(os.path.join(dest, "%s.json" % wl), "w"), indent=1)
    r = bench.get("roofline", {})
    print(wl, "value", bench.get("value"), "ms/step", bench.get("ms_per_step"), "alone", bench.get("frame_alone_ms"), "roofline", r.get("bound"), r.get("frac"), "clock", bench.get("shader_clock_hz"))
