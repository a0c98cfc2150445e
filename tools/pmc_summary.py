"""Summarise rocprofv3 --pmc counter_collection CSVs (one pass per counter) into per-kernel means per launch.
usage: python tools/pmc_summary.py out.json FETCH_SIZE=<dir> WRITE_SIZE=<dir>"""
import sys, os, glob, csv, json, collections

def summarise(d, counter):
    acc = collections.defaultdict(list)
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            if row.get("Counter_Name") != counter:
                continue
            name = row["Kernel_Name"].split("(")[0]
            acc[name].append(float(row["Counter_Value"]))
    return {k: dict(launches=len(v), mean_KB=sum(v) / len(v), min_KB=min(v), max_KB=max(v)) for k, v in acc.items()}

if __name__ == "__main__":
    out = {}
    for a in sys.argv[2:]:
        c, d = a.split("=")
        out[c] = summarise(d, c)
    json.dump(out, open(sys.argv[1], "w"), indent=1)
    print(json.dumps(out, indent=1))
