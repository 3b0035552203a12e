"""Writes the per-launch HBM traffic of the fused search kernel from two counter passes and their calibration:
    python tools/make_traffic.py <bench_line.json> <pmc_FETCH_SIZE.csv> <pmc_WRITE_SIZE.csv> \
                                 <calib_FETCH_SIZE.csv> <calib_WRITE_SIZE.csv> <out.json>
The calibration files hold the same counters for tools/calib_traffic.hip, whose kernels move known byte counts
with the access widths of the search kernel's streams (3 x 8-byte loads / stores per lane at a 24-byte stride;
4 bytes per lane).  bench.py quotes fetch + write as roofline.traffic when its workload matches."""
import csv
import json
import sys

KNOWN = {"k_calib_move24": (32 << 20) * 24.0, "k_calib_float_rw": (128 << 20) * 4.0}      # bytes read = bytes written per launch


def mean_of(path, kernel, counter):
    for row in csv.DictReader(open(path)):
        if kernel in row["Kernel_Name"] and row["Counter_Name"] == counter:
            return float(row["Mean"]) * 1024.0, int(row["Dispatches"])       # counter unit: KiB
    raise SystemExit("no %s row for %s in %s" % (kernel, counter, path))


line_path, fetch_csv, write_csv, cal_fetch, cal_write, out = sys.argv[1:7]
line = json.loads([l for l in open(line_path) if l.startswith("{")][0])
cfg = line["config"]
fetch, nf = mean_of(fetch_csv, "k_ls_batch<1>", "FETCH_SIZE")
write, nw = mean_of(write_csv, "k_ls_batch<1>", "WRITE_SIZE")
cal = {}
for k, known in KNOWN.items():
    cal[k] = {"known_bytes_each_way": known,
              "FETCH_SIZE_over_known": mean_of(cal_fetch, k, "FETCH_SIZE")[0] / known,
              "WRITE_SIZE_over_known": mean_of(cal_write, k, "WRITE_SIZE")[0] / known}
# the kernel's streamed bytes are dominated by the float64 working copy (24 B read + 24 B written per source point and
# launch): its calibration factors are the ones applied
ff, wf = cal["k_calib_move24"]["FETCH_SIZE_over_known"], cal["k_calib_move24"]["WRITE_SIZE_over_known"]
json.dump({
    "kernel": "k_ls_batch<1>", "workload": "sequence" if "configs[3]" in cfg["workload"] else "moved-copy",
    "pairs_per_launch": cfg["pairs_per_gpu_per_step"], "n_a": cfg["n_a"], "n_b": cfg["n_b"],
    "fetch_bytes_per_launch_raw": fetch, "write_bytes_per_launch_raw": write,
    "fetch_bytes_per_launch": fetch / ff, "write_bytes_per_launch": write / wf,
    "calibration": cal, "dispatches": [nf, nw],
    "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) of bench.py --no-cpu-baseline --no-extras "
              "--no-latency --steps 3 --warmup 1; counter unit KiB; divided by the counter/known-bytes ratio measured on "
              "tools/calib_traffic.hip's 24-byte-stride float64 stream in the same two passes",
}, open(out, "w"), indent=1)
print(open(out).read())
