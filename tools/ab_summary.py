import re, collections, sys
acc = collections.OrderedDict()
side = None
for ln in open(sys.argv[1]):
    m = re.match(r"== (\S+) \(rep", ln)
    if m: side = m.group(1); continue
    m = re.match(r"(.*?)\s+([0-9.]+)( us.*)?$", ln.strip())
    if m and side:
        key = m.group(1)[:52]
        try: v = float(m.group(2))
        except: continue
        mm = re.search(r"([0-9.]+) us", ln)
        if "fused rollout (self" in ln: v = float(mm.group(1)); key = ln[:28].strip()
        acc.setdefault(key, collections.OrderedDict()).setdefault(side, []).append(v)
sides = list(next(iter(acc.values())).keys())
print("%-54s" % "row", "  ".join("%10s" % s for s in sides))
for k, d in acc.items():
    print("%-54s" % k, "  ".join("%10.2f" % (sum(d.get(s, [0])) / max(1, len(d.get(s, [])))) for s in sides))
