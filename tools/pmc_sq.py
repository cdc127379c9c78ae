"""Print per-kernel averages of SQ counters from rocprofv3 --pmc csv output directories.
usage: pmc_sq.py DIR [DIR...] [--match REGEX]"""
import csv, glob, re, sys
from collections import defaultdict
dirs = [a for a in sys.argv[1:] if not a.startswith("--")]
match = None
for i, a in enumerate(sys.argv):
    if a == "--match":
        match = re.compile(sys.argv[i + 1]); dirs = [d for d in dirs if d != sys.argv[i + 1]]
acc = defaultdict(lambda: defaultdict(lambda: [0, 0.0]))
for d in dirs:
    for path in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        with open(path) as f:
            for row in csv.DictReader(f):
                name = re.sub(r"\(.*", "", row["Kernel_Name"]).replace("void ", "")
                if match and not match.search(name):
                    continue
                a = acc[name][row["Counter_Name"]]
                a[0] += 1; a[1] += float(row["Counter_Value"])
for k, cs in acc.items():
    print(k[:100])
    for c, (n, v) in sorted(cs.items()):
        print(f"   {c:34s} {v / n:16.1f}  (n={n})")
