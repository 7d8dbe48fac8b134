#!/usr/bin/env python3
"""Re-derives the number-carrying parts of the documents from the lines committed under profiles/: the "Numbers" list of DESIGN.md section 6
(tools/design_numbers.py) and the bench table of README.md (tools/readme_table.py).   usage: python tools/sync_docs.py [name]   (default r06_e)"""
import os, subprocess, sys, textwrap
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NAME = sys.argv[1] if len(sys.argv) > 1 else "r06_e"
nums = subprocess.check_output([sys.executable, os.path.join(ROOT, "tools", "design_numbers.py"), NAME], text=True).strip().split("\n")
wrap = lambda line: "\n".join(textwrap.wrap(line, width=158, subsequent_indent="  ", break_long_words=False, break_on_hyphens=False))
p = os.path.join(ROOT, "DESIGN.md")
s = open(p).read()
head = "Numbers (source files under `profiles/` in brackets; `tools/design_numbers.py` prints this list from them):"
a, b = s.index(head), s.index("**What bounds the kernel.**")
s = s[:a] + head + "\n\n" + "\n".join(wrap(l) for l in nums) + "\n\n" + s[b:]
open(p, "w").write(s)
subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "readme_table.py"), NAME], stdout=subprocess.DEVNULL)
print("DESIGN.md numbers (%d items) and README.md table refreshed from profiles/%s_*" % (len(nums), NAME))
