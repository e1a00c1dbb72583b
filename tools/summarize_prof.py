"""rocprofv3 CSV output -> the small summaries committed under profiles/.

    python3 tools/summarize_prof.py stats gpurun_out/prof_X/run_kernel_stats.csv "<command>" > profiles/NAME.md
    python3 tools/summarize_prof.py pmc gpurun_out/pmc_fetch_X/run_counter_collection.csv \
                                        gpurun_out/pmc_write_X/run_counter_collection.csv > profiles/NAME.json
"""
import collections
import csv
import json
import re
import sys


def short(name):
    name = name.replace("(anonymous namespace)::", "")
    return re.sub(r"\s+", " ", name)[:110]


def stats(path, command):
    rows = list(csv.DictReader(open(path)))
    print("# rocprofv3 --kernel-trace --stats\n")
    print("Command: `%s`\n" % command)
    print("| kernel | calls | avg us | min us | max us | total ms | % |")
    print("|---|---|---|---|---|---|---|")
    for r in rows:
        print("| %s | %s | %.2f | %.2f | %.2f | %.2f | %.2f |" % (
            short(r["Name"]), r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3,
            float(r["MaxNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6, float(r["Percentage"])))


def pmc(fetch_csv, write_csv):
    """FETCH_SIZE / WRITE_SIZE are in KiB per dispatch.  gfx950 correction (MI355X_MICROARCH.md, HBM):
    FETCH_SIZE counts 128-B read requests as 64 B -> doubled for wide coalesced reads; WRITE_SIZE is exact
    for 16-B/lane stores and for float atomics."""
    out = collections.OrderedDict()
    for counter, path in (("FETCH_SIZE", fetch_csv), ("WRITE_SIZE", write_csv)):
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(path)):
            if r["Counter_Name"] == counter and "emb_" in r["Kernel_Name"]:
                agg[short(r["Kernel_Name"]).split("(")[0]].append(float(r["Counter_Value"]))
        for k, v in agg.items():
            out.setdefault(k, {})[counter + "_KiB_avg"] = sum(v) / len(v)
            out[k]["launches"] = len(v)
    for k, d in out.items():
        d["read_bytes_corrected"] = 2.0 * d["FETCH_SIZE_KiB_avg"] * 1024
        d["write_bytes"] = d["WRITE_SIZE_KiB_avg"] * 1024
        d["traffic_bytes"] = d["read_bytes_corrected"] + d["write_bytes"]
    print(json.dumps(out, indent=1))


def pmc_step(fetch_csv, write_csv, patterns):
    """Per-kernel HBM-side bytes per launch (FETCH_SIZE doubled as in pmc(), WRITE_SIZE as is) for every kernel whose
    name contains one of the comma-separated patterns: the training step's kernels under `bench.py`."""
    pats = patterns.split(",")
    out = collections.OrderedDict()
    for counter, path in (("FETCH_SIZE", fetch_csv), ("WRITE_SIZE", write_csv)):
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(path)):
            if r["Counter_Name"] == counter and any(p in r["Kernel_Name"] for p in pats):
                agg[short(r["Kernel_Name"]).split("(")[0]].append(float(r["Counter_Value"]))
        for k, v in agg.items():
            out.setdefault(k, {})[counter + "_KiB_avg"] = sum(v) / len(v)
            out[k]["launches"] = len(v)
    for k, d in out.items():
        d["read_bytes_corrected"] = 2.0 * d.get("FETCH_SIZE_KiB_avg", 0.0) * 1024
        d["write_bytes"] = d.get("WRITE_SIZE_KiB_avg", 0.0) * 1024
        d["traffic_bytes"] = d["read_bytes_corrected"] + d["write_bytes"]
    print(json.dumps(out, indent=1))


def pmc_scale(out_dir, cases):
    """cases: "B512_V3709_zipf,..." -- reads OUT/pmc_fetch_<case>/ and OUT/pmc_write_<case>/ counter CSVs and
    prints {case: {"gather": {...}, "scatter": {...}}} (the file bench.py's roofline_at_scale reads)."""
    import glob
    import io
    import contextlib
    table = collections.OrderedDict()
    for case in cases.split(","):
        fetch = glob.glob("%s/pmc_fetch_%s/*counter_collection.csv" % (out_dir, case))
        write = glob.glob("%s/pmc_write_%s/*counter_collection.csv" % (out_dir, case))
        if not fetch or not write:
            continue
        buf = io.StringIO()
        with contextlib.redirect_stdout(buf):
            pmc(fetch[0], write[0])
        d = json.loads(buf.getvalue())
        table[case] = {("gather" if "gather" in k else "scatter"): v for k, v in d.items()}
    print(json.dumps(table, indent=1))


if __name__ == "__main__":
    if sys.argv[1] == "pmc_step":
        pmc_step(sys.argv[2], sys.argv[3], sys.argv[4])
        sys.exit(0)
    if sys.argv[1] == "pmc_scale":
        pmc_scale(sys.argv[2], sys.argv[3])
        sys.exit(0)
    if sys.argv[1] == "stats":
        stats(sys.argv[2], sys.argv[3])
    else:
        pmc(sys.argv[2], sys.argv[3])
