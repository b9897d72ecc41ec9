#!/usr/bin/env python3
"""Assemble profiles/traffic.json and profiles/issue.json (and copy the per-workload summaries and kernel stats)
from the gpurun_out/prof_<workload>/ directories of a profile stage whose own table step did not run (a gpurun
call that hit its time limit).  Same content as the table step of tools/collect_profiles.sh."""
import glob, json, os, shutil, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
RT = os.environ.get("NSK_ROUND_TAG", "r4")
traffic = {"_source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of tools/collect_profiles.sh (tools/profile_gpu.sh per workload), "
                      "corrected with the known-byte stream-copy calibration of the same run (commit: profiles/%s_COMMIT.txt); "
                      "separate passes from the bench run" % RT}
issue = {"_source": "rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD pass of "
                    "tools/profile_gpu.sh per workload: means per launch of the dominant kernel"}
for d in sorted(glob.glob(os.path.join(R, "gpurun_out", "prof_*"))):
    wl = os.path.basename(d)[5:]
    for name, table in (("traffic_%s.json" % wl, traffic), ("issue_%s.json" % wl, issue)):
        f = os.path.join(d, name)
        if os.path.exists(f):
            table.update(json.load(open(f)))
    for src, dst in (("summary.txt", "%s_%s_summary.txt" % (RT, wl)), ("summary.json", "%s_%s_summary.json" % (RT, wl))):
        if os.path.exists(os.path.join(d, src)):
            shutil.copy(os.path.join(d, src), os.path.join(R, "profiles", dst))
    ks = glob.glob(os.path.join(d, "trace", "**", "*kernel_stats.csv"), recursive=True)
    if ks:
        shutil.copy(ks[0], os.path.join(R, "profiles", "%s_%s_kernel_stats.csv" % (RT, wl)))
json.dump(traffic, open(os.path.join(R, "profiles", "traffic.json"), "w"), indent=1)
json.dump(issue, open(os.path.join(R, "profiles", "issue.json"), "w"), indent=1)
print(json.dumps(traffic)[:800])
print(json.dumps(issue)[:400])
