"""Whole-pass / single-block launch statistics of psmf_blk_filter3 from a rocprofv3 kernel trace CSV
(-> profiles/r2_kernel_trace_chain_summary.json).  usage: python trace_summary.py <dir with *_kernel_trace.csv> <note>"""
import csv, glob, json, os, statistics, sys

files = glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True)
dur = []
for f in files:
    with open(f) as fh:
        for row in csv.DictReader(fh):
            if "psmf_blk_filter3" in row["Kernel_Name"] and "filter3s" not in row["Kernel_Name"]:
                dur.append((int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) * 1e-6)      # ms
whole = sorted(d for d in dur if d > 5.0)
single = [d for d in dur if d <= 5.0]
print(json.dumps({
    "kernel": "psmf::psmf_blk_filter3", "launches": len(dur), "whole_pass_launches": len(whole),
    "whole_pass_ms": {"min": whole[0], "median": statistics.median(whole), "max": whole[-1], "mean": statistics.mean(whole)} if whole else None,
    "single_block_launches": len(single), "single_block_us_mean": 1e3 * statistics.mean(single) if single else None,
    "note": sys.argv[2] if len(sys.argv) > 2 else ""}, indent=1))
