"""dev tool: static instruction mix of one kernel in a hipcc -S listing.   python scripts/probes/isa_mix.py file.s kernel-substring"""
import collections, re, sys
lines = open(sys.argv[1]).read().split("\n")
key = sys.argv[2]
start = next(i for i, l in enumerate(lines) if l.startswith("_Z") and key in l.split(":")[0])
end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith("s_endpgm"))
ops = collections.Counter(); valu = sg = vcc = 0
for l in lines[start + 1:end]:
    l = l.split(";")[0].strip()
    if not l or l.startswith(".") or l.endswith(":"):
        continue
    p = l.split(None, 1)
    ops[p[0]] += 1
    if p[0].startswith("v_"):
        valu += 1
        a = p[1] if len(p) > 1 else ""
        if re.search(r"\bs\d+\b|\bs\[", a): sg += 1
        if "vcc" in a: vcc += 1
print(f"{lines[start].split(':')[0][:60]}: {end - start} lines, static VALU {valu}, with an SGPR operand {sg}, touching vcc {vcc}")
for k, v in ops.most_common(int(sys.argv[3]) if len(sys.argv) > 3 else 40):
    print(f"  {k:30s} {v}")
