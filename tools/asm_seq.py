"""Compressed instruction sequence of one kernel from a device assembly listing (hipcc -S --cuda-device-only): runs of
M = MFMA, . = other VALU, F = buffer load, G = global load, S = global store / atomic, r / w = LDS read / write-or-atomic,
[vN lN] = s_waitcnt, B = barrier, | = branch / label.   python tools/asm_seq.py file.s <substring of the mangled name> [max chars]"""
import collections
import re
import sys

lines = open(sys.argv[1]).read().split("\n")
pat = sys.argv[2]
start = [i for i, l in enumerate(lines) if pat in l and not l.startswith("\t") and re.match(r"^[A-Za-z_][\w$.]*:", l)][0]
end = [i for i in range(start, len(lines)) if "s_endpgm" in lines[i]][0]
kinds, cnt = [], collections.Counter()
for l in lines[start + 1:end]:
    t = l.strip()
    if not t or t.startswith(";") or t.startswith("."):
        if t.startswith(".LBB") and t.endswith(":"):
            kinds.append("|")
        continue
    op = t.split()[0]
    cnt[op] += 1
    if op.startswith("v_mfma"): k = "M"
    elif op.startswith("buffer_load"): k = "F"
    elif op.startswith("global_load") or op.startswith("flat_load"): k = "G"
    elif op.startswith("global_") or op.startswith("buffer_") or op.startswith("flat_"): k = "S"
    elif op.startswith("ds_read"): k = "r"
    elif op.startswith("ds_"): k = "w"
    elif op.startswith("s_waitcnt"):
        m, m2 = re.search(r"vmcnt\((\d+)\)", t), re.search(r"lgkmcnt\((\d+)\)", t)
        k = "[" + ("v%s" % m.group(1) if m else "") + ("l%s" % m2.group(1) if m2 else "") + "]"
    elif op.startswith("s_barrier"): k = "B"
    elif op.startswith("s_cbranch") or op.startswith("s_branch"): k = "|"
    elif op.startswith("v_"): k = "."
    else: k = ""
    kinds.append(k)
s = "".join(kinds)
for ch in "M.FGSrw":
    s = re.sub("(%s+)" % re.escape(ch), lambda m, ch=ch: "%s%d " % (ch, len(m.group(1))), s)
print(lines[start][:120])
print(s[: int(sys.argv[3]) if len(sys.argv) > 3 else 8000])
print(cnt.most_common(25))
