"""dev aid: print calls / average ns of the kernels of a rocprofv3 --kernel-trace --stats run whose names contain any of the given substrings
usage: python tools/kstats.py <rocprof output dir> stft istft ..."""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    if len(sys.argv) < 3 or any(k in r["Name"] for k in sys.argv[2:]):
        print(r["Name"][:60], r["Calls"], r["AverageNs"])
