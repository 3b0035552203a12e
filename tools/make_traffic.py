"""Writes the per-launch HBM traffic of the fused search kernel from two counter passes:
    python tools/make_traffic.py <bench_line.json> <pmc_FETCH_SIZE.csv> <pmc_WRITE_SIZE.csv> <out.json>
bench.py quotes the result as roofline.traffic when its workload (pairs per launch, cloud sizes) matches."""
import csv
import json
import sys


def mean_of(path, counter):
    for row in csv.DictReader(open(path)):
        if "k_ls_batch<1>" in row["Kernel_Name"] and row["Counter_Name"] == counter:
            return float(row["Mean"]) * 1024.0, int(row["Dispatches"])       # counter unit: KiB
    raise SystemExit("no k_ls_batch<1> row for %s in %s" % (counter, path))


line_path, fetch_csv, write_csv, out = sys.argv[1:5]
line = json.loads([l for l in open(line_path) if l.startswith("{")][0])
cfg = line["config"]
fetch, nf = mean_of(fetch_csv, "FETCH_SIZE")
write, nw = mean_of(write_csv, "WRITE_SIZE")
json.dump({
    "kernel": "k_ls_batch<1>",
    "pairs_per_launch": cfg["pairs_per_gpu_per_step"], "n_a": cfg["n_a"], "n_b": cfg["n_b"],
    "fetch_bytes_per_launch": fetch, "write_bytes_per_launch": write, "dispatches": [nf, nw],
    "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) of bench.py --no-cpu-baseline "
              "--steps 3 --warmup 1; counter unit KiB; raw values, no read-side doubling applied (the reads are "
              "gathers and 24-B strided f64 streams, not 16 B/lane coalesced streams)",
}, open(out, "w"), indent=1)
print(open(out).read())
