"""Copies the summaries tools/round_evidence.sh left under gpurun_out/evidence/ into profiles/ (round-1 names)."""
import csv, glob, os, shutil
E = "gpurun_out/evidence"
def newest(pat):
    return sorted(glob.glob(pat, recursive=True), key=os.path.getmtime)[-1]
def cp(src, dst):
    shutil.copy(src, os.path.join("profiles", dst)); print("->", dst, "<-", os.path.basename(src))
cp(newest(E + "/prof_bench/runc/*_kernel_stats.csv"), "r01_bench_c2_kernel_stats.csv")
cp(newest(E + "/prof_dp/runc/*_kernel_stats.csv"), "r01_bench_c2_dp_native_one_rank_kernel_stats.csv")
cp(newest(E + "/prof_uvt/runc/*_kernel_stats.csv"), "r01_uvt_pass_kernel_stats.csv")
for name in ("bench_c2_under_rocprof.json", "bench_c2_final_run.json", "bench_c2_dp_native_one_rank.json",
             "bench_c4_single.json", "bench_c4_dp_native_one_rank.json", "uvt_pass_roofline.txt",
             "metric_functions_c2.txt"):
    cp(os.path.join(E, name), "r01_" + name)
cp(os.path.join(E, "pmc_traffic.json"), "pmc_traffic.json")
for tag, name in (("pmc_fetch", "r01_resident_pmc_fetch_size.csv"), ("pmc_write", "r01_resident_pmc_write_size.csv")):
    rows = list(csv.reader(open(newest(E + f"/{tag}/**/*counter_collection.csv"))))
    ki = rows[0].index("Kernel_Name")
    keep = [rows[0]] + [r for r in rows[1:] if "resident_train_kernel" in r[ki]]
    csv.writer(open("profiles/" + name, "w")).writerows(keep); print("->", name, len(keep) - 1, "rows")
