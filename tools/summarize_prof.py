"""dev aid: condense the rocprofv3 outputs of tools/profile.sh into profiles/<tag>_summary.txt"""
import csv, glob, collections, sys, os, json
tag = sys.argv[1]; src = sys.argv[2]; out = sys.argv[3]; rnd = sys.argv[4] if len(sys.argv) > 4 else "round2"
lines = []
P = lambda *a: lines.append(" ".join(str(x) for x in a))
def short(k):
    k = k.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
    m = __import__("re").match(r"_ZN12_GLOBAL__N_1\d+(\w+?_kernel)I(.*?)EEv", k)      # names rocprofv3 left mangled
    if m: k = m.group(1) + "<" + m.group(2).replace("Li", "").replace("E", ",").replace("DF16b", "bf16").replace("f", "float").strip(",") + ">"
    return k[:90]
newest = lambda pat: max(glob.glob(pat), key=os.path.getmtime)      # (gpurun merges into an existing directory: take the latest run)
st = newest(os.path.join(src, "trace/*/*_kernel_stats.csv"))
cmd = open(os.path.join(src, "command.txt")).read().strip() if os.path.exists(os.path.join(src, "command.txt")) else "python bench.py --precision " + tag
P("# rocprofv3 --kernel-trace --stats -- %s --steps 2 --warmup 1   [workload tag: %s]" % (cmd, tag))
P("# (per kernel: calls, total ms, average us, % of GPU kernel time)")
for r in csv.DictReader(open(st)):
    P("%-92s calls %6s  total %10.3f ms  avg %10.2f us  %6s %%" % (short(r["Name"]), r["Calls"], float(r["TotalDurationNs"]) / 1e6,
                                                                  float(r["AverageNs"]) / 1e3, r["Percentage"]))
P("")
P("# bench line of the traced run:")
P("# (tracing slows the host side: the traced step is longer than the kernels' sum; per-kernel launch times in the JSON come from HIP events)")
P([l for l in open(os.path.join(src, "bench_trace.log")).read().splitlines() if l.startswith("{")][-1])
def pmc(name):
    f = newest(os.path.join(src, name, "*/*_counter_collection.csv"))
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); nd = collections.defaultdict(set); dur = collections.defaultdict(float)
    for r in csv.DictReader(open(f)):
        k = short(r["Kernel_Name"])
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); nd[k].add(r["Dispatch_Id"])
        dur[(k, r["Dispatch_Id"])] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    return agg, nd, dur
P("")
P("# HBM traffic (separate --pmc passes, niter=10): FETCH_SIZE and WRITE_SIZE are in KiB-units of 1 KB;")
P("# MI355X_MICROARCH.md: FETCH_SIZE reports 1/2 of the bytes of wide coalesced streaming reads on gfx950 -> doubled below")
fa, fn, _ = pmc("pmc_fetch"); wa, wn, _ = pmc("pmc_write")
for k in fa:
    if not any(s in k for s in ("chain_kernel", "decode_kernel", "stream_kernel", "stream2_kernel", "rot_kernel", "fused_kernel", "w_update", "w_partial")): continue
    n = len(fn[k]); f = fa[k]["FETCH_SIZE"] / n; w = wa.get(k, {}).get("WRITE_SIZE", 0.0) / max(len(wn.get(k, [1])), 1)
    P("%-92s launches %4d  FETCH_SIZE/launch %10.1f KB (x2 = %8.2f MB)  WRITE_SIZE/launch %10.1f KB  => HBM %8.2f MB/launch"
      % (k, n, f, 2 * f / 1024, w, (2 * f + w) / 1024))
traffic = {}
for k in fa:
    n = len(fn[k]); f = fa[k]["FETCH_SIZE"] / n; w = wa.get(k, {}).get("WRITE_SIZE", 0.0) / max(len(wn.get(k, [1])), 1)
    traffic[k] = {"fetch_size_kb_raw": f, "write_size_kb": w, "hbm_bytes_per_launch": (2 * f + w) * 1024, "launches": n}
json.dump({"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) -- %s --steps 1 --warmup 0 --niter 10" % cmd,
           "correction": "FETCH_SIZE doubled (gfx950 reports 1/2 of wide coalesced reads, MI355X_MICROARCH.md); units of 1 KiB",
           "kernels": traffic}, open(os.path.join(os.path.dirname(out), "%s_%s_traffic.json" % (rnd, tag)), "w"), indent=1)
P("")
P("# SQ counters (niter=10), summed over launches; *_CYCLES of waves are quad-cycles, VALU_MFMA_BUSY in cycles")
sa, sn, _ = pmc("pmc_sq")
for k in sa:
    if not any(s in k for s in ("chain_kernel", "decode_kernel", "stream_kernel", "stream2_kernel", "rot_kernel", "fused_kernel")): continue
    v = sa[k]; wc = v["SQ_WAVE_CYCLES"]
    P("%-92s launches %d" % (k, len(sn[k])))
    for c in sorted(v): P("    %-28s %16.0f   (%.3f of SQ_WAVE_CYCLES)" % (c, v[c], v[c] / wc))
try:
    s2, s2n, _ = pmc("pmc_sq2")
    P("")
    P("# second SQ pass (niter=10), per launch")
    for k in s2:
        if not any(s in k for s in ("chain_kernel", "stream_kernel", "stream2_kernel", "rot_kernel", "fused_kernel")): continue
        P("%-92s launches %d" % (k, len(s2n[k])))
        for c in sorted(s2[k]): P("    %-28s %16.0f per launch" % (c, s2[k][c] / len(s2n[k])))
except Exception as e:
    P("# (second SQ pass missing: %s)" % e)
ga, gn, gd = pmc("pmc_grbm")
for k in ga:
    if "chain_kernel" not in k: continue
    t = sum(d for (kk, _), d in gd.items() if kk == k)
    P("")
    P("# clock while the chain kernel runs: GRBM_GUI_ACTIVE / 8 / kernel time = %.2f GHz" % (ga[k]["GRBM_GUI_ACTIVE"] / 8 / t))
open(out, "w").write("\n".join(lines) + "\n")
print("\n".join(lines[:12])); print("..."); print("\n".join(l for l in lines if "HBM" in l and "launches" in l))
