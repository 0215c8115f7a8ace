"""rocprofv3 *_kernel_stats.csv + the bench logs of the same gpurun call -> profiles/<tag>_bench_c3_kernel_stats.{csv,md}
usage: stats_to_md.py <profdir> <tag> <profiled bench log> <plain bench log> "<title>" """
import csv, glob, os, shutil, sys

prof, tag, log_prof, log_plain, title = sys.argv[1:6]
src = sorted(glob.glob(os.path.join(prof, "**", "*_kernel_stats.csv"), recursive=True))[-1]
root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "profiles")
shutil.copy(src, os.path.join(root, f"{tag}_bench_c3_kernel_stats.csv"))


def json_line(path):
    lines = [l.strip() for l in open(path) if l.startswith("{")]
    return lines[-1] if lines else "(no JSON line)"


with open(os.path.join(root, f"{tag}_bench_c3_kernel_stats.md"), "w") as out:
    out.write(f"# {title}\n\n| kernel | calls | total ms | avg us | min us | max us | % |\n|---|---:|---:|---:|---:|---:|---:|\n")
    for r in csv.DictReader(open(src)):
        out.write(f"| `{r['Name'][:80]}` | {r['Calls']} | {float(r['TotalDurationNs'])/1e6:.3f} | {float(r['AverageNs'])/1e3:.2f} | "
                  f"{float(r['MinNs'])/1e3:.2f} | {float(r['MaxNs'])/1e3:.2f} | {float(r['Percentage']):.2f} |\n")
    out.write("\nbench.py JSON line of the profiled run:\n\n```\n" + json_line(log_prof) + "\n```\n")
    out.write("\nbench.py JSON line of the un-profiled run on the same box:\n\n```\n" + json_line(log_plain) + "\n```\n")
