"""Re-wraps the paragraphs and bullets of markdown files to <= 158 BYTES per line (tables, headings and code blocks untouched).  usage: python tools/reflow_md.py FILE..."""
import textwrap, re, sys
def blen(s): return len(s.encode("utf-8"))
def wrap_bytes(text, width, ind0, sub):
    words = text.split()
    lines, cur, fresh = [], ind0, True
    for w in words:
        cand = cur + w if fresh else cur + " " + w
        if blen(cand) > width and not fresh:
            lines.append(cur); cur = sub + w
        else:
            cur = cand
        fresh = False
    if not fresh: lines.append(cur)
    return lines
def reflow(path, width=158):
    L = open(path).read().split("\n")
    out, para, in_code = [], [], False
    def flush():
        if para:
            text = " ".join(x.strip() for x in para)
            first = para[0]
            m = re.match(r"^(\s*(?:[*-]|\d+\.)\s+)", first)
            ind0 = m.group(1) if m else re.match(r"^(\s*)", first).group(1)
            sub = " " * len(ind0)
            body = text[len(m.group(1).strip()):].strip() if m else text.strip()
            out.extend(wrap_bytes(body, width, ind0, sub))
            para.clear()
    for l in L:
        if l.strip().startswith("```"):
            flush(); in_code = not in_code; out.append(l); continue
        if in_code or l.lstrip().startswith("|") or l.startswith("#") or l.strip() == "":
            flush(); out.append(l); continue
        if re.match(r"^\s*([*-]|\d+\.)\s+", l):
            flush()
        elif para and re.match(r"^(\s*)", l).group(1) == "" and re.match(r"^(\s*)", para[0]).group(1) != "" and not re.match(r"^\s*([*-]|\d+\.)\s+", para[0]):
            flush()
        para.append(l)
    flush()
    open(path, "w").write("\n".join(out))
for f in sys.argv[1:]:
    reflow(f)
