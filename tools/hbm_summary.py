"""HBM traffic per kernel from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; tools/measure_hbm.sh) -> JSON.

    python tools/hbm_summary.py gpurun_out/hbm_f gpurun_out/hbm_w <steps incl. warm-up> > profiles/rNN_hbm_traffic.json

Units and corrections as MI355X_MICROARCH.md § HBM prescribes: both counters are KB; on gfx950 FETCH_SIZE reports half the bytes of
wide (16 B/lane) coalesced reads, so it is doubled (narrower reads are uncalibrated: the doubled figure is an upper bound there)."""
import collections
import csv
import glob
import json
import sys

CLASS_OF = [("pw_bwd_w4_kernel", "conv1x1x1_bwd_data_x6"), ("pw_bwd_x6_kernel", "conv1x1x1_bwd_data_x6"), ("pw_fwd_w4_kernel", "conv1x1x1_fwd_x6"), ("pw_fwd_x6_kernel", "conv1x1x1_fwd_x6"), ("pw_fwd_h3k_kernel", "conv1x1x1_fwd_x6"), ("conv3_wgrad_w4_kernel", "conv3x3x3_wgrad_x6"), ("conv3_wgrad_x6_kernel", "conv3x3x3_wgrad_x6"),
            ("conv3_w4_kernel<25, 32, 11, false", "conv3x3x3_fwd_x6"), ("conv3_w4_kernel<32, 25, 11, false", "conv3x3x3_bwd_data_x6"),
            ("conv3_pp_kernel<25, false", "conv3x3x3_fwd_x6"), ("conv3_pp_kernel<32, false", "conv3x3x3_bwd_data_x6"),
            ("conv3_pstrip_kernel<25", "conv3x3x3_fwd_x6"), ("conv3_pstrip_kernel<32", "conv3x3x3_bwd_data_x6"), ("conv3_strip_kernel<25", "conv3x3x3_fwd_x6"), ("conv3_strip_kernel<32", "conv3x3x3_bwd_data_x6"),
            ("pw_bwd2_mfma_kernel", "conv1x1x1_bwd_data"), ("pw_fwd_mfma_kernel", "conv1x1x1_fwd"), ("conv3_wgrad_mfma_kernel", "conv3x3x3_wgrad")]


def collect(d, name):
    tot, calls = collections.defaultdict(float), collections.Counter()
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == name:
                k = r["Kernel_Name"].split("(")[0]
                tot[k] += float(r["Counter_Value"]) * 1024.0
                calls[k] += 1
    return tot, calls


def main():
    fdir, wdir, steps = sys.argv[1], sys.argv[2], int(sys.argv[3])
    ft, fc = collect(fdir, "FETCH_SIZE")
    wt, wc = collect(wdir, "WRITE_SIZE")
    kernels = {}
    for k in sorted(set(ft) | set(wt), key=lambda k: -(2 * ft.get(k, 0) + wt.get(k, 0))):
        n = max(fc.get(k, 0), wc.get(k, 0))
        kernels[k] = {"launches": n, "read_bytes_per_launch": round(2 * ft.get(k, 0) / max(1, fc.get(k, 0))),
                      "write_bytes_per_launch": round(wt.get(k, 0) / max(1, wc.get(k, 0)))}
    per_class = {}
    for pat, cls in CLASS_OF:
        for k, v in kernels.items():
            if pat in k and cls not in per_class:
                per_class[cls] = {"kernel": k, "bytes_per_launch": v["read_bytes_per_launch"] + v["write_bytes_per_launch"]}
    out = {"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, --kernel-trace only) around bench.py; KB -> bytes, FETCH_SIZE doubled (gfx950)",
           "steps_profiled": steps,
           "bytes_per_step": round((2 * sum(ft.values()) + sum(wt.values())) / steps),
           "read_bytes_per_step": round(2 * sum(ft.values()) / steps), "write_bytes_per_step": round(sum(wt.values()) / steps),
           "per_class": per_class, "kernels": dict(list(kernels.items())[:14])}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
