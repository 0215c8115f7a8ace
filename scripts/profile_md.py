"""Turn a rocprofv3 kernel-stats CSV (+ the bench logs of the same box) into profiles/<name>.{csv,md}.

usage: profile_md.py <prof-dir> <name> <title> [profiled-bench-log] [plain-bench-log]
"""
import csv
import glob
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def bench_line(path):
    if not path or not os.path.exists(path):
        return None
    for line in open(path):
        if line.startswith('{"metric"'):
            return line.strip()
    return None


def main():
    prof, name, title = sys.argv[1:4]
    src = sorted(glob.glob(os.path.join(prof, "**", "*_kernel_stats.csv"), recursive=True))[-1]
    dst = os.path.join(ROOT, "profiles", name)
    shutil.copy(src, dst + ".csv")
    rows = list(csv.DictReader(open(src)))
    out = [f"# {title}", "", "| kernel | calls | total ms | avg us | min us | max us | % |", "|---|---:|---:|---:|---:|---:|---:|"]
    for r in rows[:28]:
        out.append(
            f"| `{r['Name'][:80]}` | {r['Calls']} | {float(r['TotalDurationNs']) / 1e6:.3f} | "
            f"{float(r['AverageNs']) / 1e3:.2f} | {float(r['MinNs']) / 1e3:.2f} | {float(r['MaxNs']) / 1e3:.2f} | "
            f"{float(r['Percentage']):.2f} |"
        )
    for label, path in (("profiled run", sys.argv[4] if len(sys.argv) > 4 else None),
                        ("un-profiled run on the same box", sys.argv[5] if len(sys.argv) > 5 else None)):
        line = bench_line(path)
        if line:
            out += ["", f"bench.py JSON line of the {label}:", "", "```", line, "```"]
    open(dst + ".md", "w").write("\n".join(out) + "\n")
    print("wrote", dst + ".md")


if __name__ == "__main__":
    main()
