"""dev aid: instruction histogram + issue-cost estimate of the innermost big loop of one kernel in a .s file
usage: python tools/isa_hist.py /tmp/isa/engine.s <mangled-name-substring> [min_loop_len]"""
import re, sys, collections
txt = open(sys.argv[1]).read()
sub = sys.argv[2]
minlen = int(sys.argv[3]) if len(sys.argv) > 3 else 300
names = [m.group(1) for m in re.finditer(r"^(_Z\S+):", txt, re.M) if sub in m.group(1)]
name = names[0]
i = txt.index(name + ":"); j = txt.index(".Lfunc_end", i)
body = txt[i:j].split("\n")
labels = {}
for n, l in enumerate(body):
    m = re.match(r"(\.LBB\d+_\d+):", l)
    if m: labels[m.group(1)] = n
loops = []
for n, l in enumerate(body):
    m = re.search(r"s_cbranch_\w+ (\.LBB\d+_\d+)|s_branch (\.LBB\d+_\d+)", l)
    if m:
        t = m.group(1) or m.group(2)
        if t in labels and labels[t] < n and n - labels[t] >= minlen: loops.append((labels[t], n))
print(name, "loops:", loops)
lo, hi = min(loops, key=lambda x: x[1] - x[0]) if loops else (0, len(body))
c = collections.Counter()
for l in body[lo:hi + 1]:
    l = l.strip()
    if not l or l.startswith((".", ";", "//")) or l.endswith(":"): continue
    c[l.split()[0]] += 1
TRANS = ("v_exp", "v_log", "v_rcp", "v_rsq", "v_sqrt", "v_sin", "v_cos")
cost = 0
for k, v in c.items():
    if k.startswith(TRANS) or k.startswith("v_mfma") or k.startswith("v_pk_") or "f64" in k: cost += 8 * v
    elif k.startswith("v_") or k == "s_nop": cost += 4 * v
print("instructions", sum(c.values()), " est. vector issue cycles", cost)
for k, v in c.most_common(40): print("%5d %s" % (v, k))
