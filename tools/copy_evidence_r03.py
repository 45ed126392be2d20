"""Copies the summaries tools/round3_evidence.sh left under gpurun_out/ev3/ into profiles/ (round-3 names)."""
import csv, glob, json, os, shutil, subprocess, sys
E = "gpurun_out/ev3"


def newest(pat):
    return sorted(glob.glob(pat, recursive=True), key=os.path.getmtime)[-1]


def cp(src, dst):
    shutil.copy(src, os.path.join("profiles", dst))
    print("->", dst)


for tag, name in (("prof_driver", "r03_bench_c2_driver_cmd_kernel_stats.csv"), ("prof_bench", "r03_bench_c2_kernel_stats.csv"),
                  ("prof_c3", "r03_bench_c3_kernel_stats.csv"), ("prof_c4", "r03_bench_c4_kernel_stats.csv"),
                  ("prof_uvt", "r03_uvt_pass_kernel_stats.csv")):
    if glob.glob(f"{E}/{tag}/**/*_kernel_stats.csv", recursive=True):      # a section that was not re-collected keeps its file
        cp(newest(f"{E}/{tag}/**/*_kernel_stats.csv"), name)
for name in ("bench_c2_default.json", "bench_c2_driver_cmd.json", "bench_c2_driver_cmd_under_rocprof.json",
             "bench_c2_under_rocprof.json", "bench_c3_f32.json", "bench_c3_bf16.json", "bench_c4_single.json",
             "bench_c2_dp_native_one_rank.json", "bench_c2_shard_one_rank.json", "bench_c4_shard_one_rank.json",
             "bench_c4_shard_one_rank_strict.json", "bench_c4_shard_one_rank_pipelined.json", "samplers.txt",
             "uvt_pass_roofline.txt", "metric_functions_c2.txt", "step_period_by_size.txt", "short_call_breakdown.txt",
             "tiny_problem_forms.txt", "resident_pmc_c2.txt", "resident_pmc_c3.txt", "streaming_pmc_C4.txt",
             "streaming_pmc_C5.txt", "resident_common_path.txt", "uvt_pass_vs_load_history.txt",
             "driver_call_event_pair_cost.txt", "valu_issue_microbench.txt", "metric_functions_c5.txt",
             "resident_wave_accounting.txt", "uvt_pmc.txt", "short_call_wave_timeline.txt", "gpu_tests.log",
             "fuzz_parity.txt", "fuzz_bf16.txt", "fuzz_multi_gpu_rehearsal.txt", "fuzz_uvt.txt"):
    if os.path.exists(os.path.join(E, name)):
        cp(os.path.join(E, name), "r03_" + name)
if os.path.isdir(f"{E}/pmc_c2_fetch") and os.path.isdir(f"{E}/pmc_c2_write"):
    subprocess.run([sys.executable, "tools/pmc_traffic.py", f"{E}/pmc_c2_fetch", f"{E}/pmc_c2_write", "profiles/r03_pmc_traffic.json"],
                   stdout=subprocess.DEVNULL, check=True)
    print("-> r03_pmc_traffic.json")

# per-kernel split of one UV^T pass per shape (medians over the passes of the profiled run)
import re


def short(name):
    m = re.search(r"((?:uvt|colsum|centre|x_rows)\w*(?:<[^>]*>)?)", name)
    return m.group(1) if m else name[:60]


if not glob.glob(f"{E}/prof_uvt/runc/*_kernel_trace.csv"):
    print("(no UV^T kernel trace kept: the per-pass split is skipped)")
    sys.exit(0)
rows = list(csv.DictReader(open(newest(f"{E}/prof_uvt/runc/*_kernel_trace.csv"))))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
passes, cur = [], []
for r in rows:
    if "colsum_partial" in r["Kernel_Name"] and cur:
        passes.append(cur)
        cur = []
    cur.append(r)
passes.append(cur)
by_shape = {}
for p in passes:
    main = [r for r in p if "uvt_tiled_kernel" in r["Kernel_Name"] or "uvt_main_kernel" in r["Kernel_Name"]]
    if main:
        by_shape.setdefault(short(main[0]["Kernel_Name"]), []).append(p)
with open("profiles/r03_uvt_pass_kernel_trace_summary.txt", "w") as f:
    f.write("One UV^T pass per main-kernel instantiation (rocprofv3 --kernel-trace of tools/bench_uvt.py C2 C3 C5; the median "
            "pass of each; us).  Template arguments: <D, waves, stage columns, 16-byte X, waves/SIMD, X prefetch, WHAT> with "
            "WHAT 1 = rows only, 2 = error only, 3 = both.\n")
    for k, ps in by_shape.items():
        p = ps[len(ps) // 2]
        t0 = int(p[0]["Start_Timestamp"])
        f.write(f"\n{k}\n")
        last = t0
        for r in p:
            if "at::native" in r["Kernel_Name"]:
                break
            s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
            last = e
            f.write(f"   +{(s - t0) / 1e3:9.1f}  dur {(e - s) / 1e3:9.1f}  {short(r['Kernel_Name'])}\n")
        f.write(f"   span {(last - t0) / 1e3:.1f}\n")
print("-> r03_uvt_pass_kernel_trace_summary.txt")
