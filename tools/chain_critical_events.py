import re,sys
lo,hi=float(sys.argv[2]),float(sys.argv[3])
for line in open(sys.argv[1]):
    m = re.match(r"\s+([0-9.]+) us  ticket\s+(\d+) row\s+(\d+) step\s+(\d+)  (.*)", line)
    if not m: continue
    t, tk, r, st, name = float(m.group(1)), int(m.group(2)), int(m.group(3)), int(m.group(4)), m.group(5)
    if lo <= t <= hi and (("leaf" in name) or ("diagonal" in name) or (r - st <= 2 and r > st)):
        print(line.rstrip())
