#!/usr/bin/env python3
"""Wraps the physical lines of Python files that are longer than WIDTH columns, without changing the program: the AST of the result must equal the AST of
the input, or the file is left alone.  What it does to a long line:
  * a comment line                      -> re-wrapped comment lines of the same indentation
  * code + trailing comment             -> the comment goes onto its own line(s) in front of the code
  * code inside brackets                -> broken after commas / operators at bracket depth >= 1 (continuation indent: the line's + 4 per depth level, at least 4)
  * a string literal that is too long   -> split at a blank into adjacent literals (plain '...' / "..." strings inside brackets only)
usage: python tools/wrap_py.py FILE...       (WIDTH = 160)"""
import ast
import io
import sys
import tokenize

WIDTH = 158


def wrap_comment(indent, text):
    words = text.lstrip("#").strip().split(" ")
    out, cur = [], indent + "#"
    for w in words:
        if len(cur) + 1 + len(w) > WIDTH and cur.strip() != "#":
            out.append(cur)
            cur = indent + "#" + (" " if not text.lstrip("#").startswith("  ") else "  ") + w
        else:
            cur += " " + w
    out.append(cur)
    return out


def split_string(tok, room):
    """split a plain string literal token into two adjacent literals, the first at most `room` columns wide; None if not possible"""
    s = tok
    prefix = ""
    while s and s[0] in "rRbBuUfF":
        prefix += s[0]
        s = s[1:]
    if prefix.lower() not in ("", "f") or s[:3] in ('"""', "'''") or len(s) < 2:
        return None
    q = s[0]
    body = s[1:-1]
    limit = room - len(prefix) - 2
    if limit < 20:
        return None
    cut = body.rfind(" ", 0, limit)
    while cut > 0 and body[cut - 1] == "\\":
        cut = body.rfind(" ", 0, cut)
    if cut <= 0:
        return None
    if prefix.lower() == "f":              # never cut inside a replacement field
        depth = 0
        for ch in body[:cut + 1]:
            depth += ch == "{"
            depth -= ch == "}"
        if depth != 0:
            return None
    return prefix + q + body[:cut + 1] + q, prefix + q + body[cut + 1:] + q


BREAK_BEFORE = {"+", "-", "*", "/", "%", "and", "or", "if", "else", "for", "|", "=="}


def wrap_row(row_text, row_toks, first_row):
    """One physical line (row_text) that is too long, with its tokens [(type, string, start col, end col, bracket depth before the token)].
    Returns the physical lines that replace it, or None if it cannot be wrapped.  Breaks go after the comma of the LOWEST bracket depth that still fits (the
    last one of that depth), else in front of a binary operator / keyword; continuation lines are aligned behind the innermost bracket that was opened on
    this row and is still open at the break (else: the row's own indentation, + 4 for the first row of a statement)."""
    indent = row_text[:len(row_text) - len(row_text.lstrip())]
    base_cont = indent + ("    " if first_row else "")
    comment = None
    if row_toks and row_toks[-1][0] == tokenize.COMMENT:
        comment = row_toks[-1][1]
        row_toks = row_toks[:-1]
    head = []
    if comment is not None:
        code = row_text[:row_toks[-1][3]] if row_toks else ""
        if not code.strip():
            return wrap_comment(indent, comment)
        head = wrap_comment(indent, comment)            # the comment goes in front of the code
        if len(code) <= WIDTH:
            return head + [code.rstrip()]
    out, cur = [], indent
    cands = []                                           # (position in `cur`, depth, kind 0 = after a comma / 1 = in front of an operator, continuation indent)
    opened = []                                          # columns (in `cur`) behind the brackets opened on this physical line and still open
    prev_end = len(indent)
    prev2 = prev1 = None
    for typ, text, c0, c1, depth in row_toks:
        gap = " " * (c0 - prev_end) if cur.strip() else ""
        prev_end = c1
        cont_now = " " * opened[-1] if opened else base_cont
        if depth >= 1 and (text in BREAK_BEFORE) and typ in (tokenize.OP, tokenize.NAME) and cur.strip():
            cands.append((len(cur) + len(gap), depth, 1, cont_now))
        if typ == tokenize.OP and text in ")]}" and opened and len(opened) > 0 and depth < len(opened) + (depth - len(opened) + 1):
            pass
        cur += gap + text
        if typ == tokenize.OP and text in "([{":
            opened.append(len(cur))
        elif typ == tokenize.OP and text in ")]}" and opened:
            opened.pop()
        while len(cur) > WIDTH:
            fit = [c for c in cands if c[0] < len(cur) and len(cur[:c[0]].rstrip()) <= WIDTH and cur[:c[0]].strip()]
            commas = [c for c in fit if c[2] == 0]
            pool = commas if commas else fit
            if pool:
                dmin = min(c[1] for c in pool)
                at, _, _, cont = [c for c in pool if c[1] == dmin][-1]
                if len(cont) + len(cur) - at >= len(cur):            # no progress (the continuation indent is not shorter than what was cut)
                    cont = base_cont
                out.append(cur[:at].rstrip())
                shift = len(cont) - at - (len(cur[at:]) - len(cur[at:].lstrip()))
                cur = cont + cur[at:].lstrip()
                cands = [(p + shift, d, k, cn) for p, d, k, cn in cands if p > at]
                opened = [max(len(cont), o + shift) for o in opened]
                continue
            if typ == tokenize.STRING and depth >= 1 and "\n" not in text:
                start = len(cur) - len(text)
                sp = split_string(text, WIDTH - start)
                if sp:
                    out.append((cur[:start] + sp[0]).rstrip())
                    cont = " " * start if start < 100 else base_cont
                    cur = cont + sp[1]
                    text = sp[1]
                    cands = []
                    continue
            return None
        if text == "," and depth >= 1 and not (prev1 is not None and prev1[0] == tokenize.NAME and prev2 is not None and prev2[1] == "for"):
            cands.append((len(cur), depth, 0, " " * opened[-1] if opened else base_cont))
        prev2, prev1 = prev1, (typ, text)
    if cur.strip():
        out.append(cur.rstrip())
    return head + out


def wrap_logical(lines, first):
    """lines: the physical lines of ONE logical line.  Only the physical lines that are too long are touched."""
    src = "\n".join(lines) + "\n"
    toks = [t for t in tokenize.generate_tokens(io.StringIO(src).readline)]
    depth, rows = 0, {}
    for t in toks:
        if t.type in (tokenize.NEWLINE, tokenize.NL, tokenize.ENDMARKER, tokenize.INDENT, tokenize.DEDENT):
            continue
        if t.start[0] != t.end[0]:
            return lines                                  # a token over several rows (multi-line string): leave the statement alone
        if t.type == tokenize.OP and t.string in ")]}":
            depth -= 1
        rows.setdefault(t.start[0], []).append((t.type, t.string, t.start[1], t.end[1], depth))
        if t.type == tokenize.OP and t.string in "([{":
            depth += 1
    out = []
    for r, text in enumerate(lines, 1):
        if len(text) <= WIDTH + 2 or r not in rows:
            out.append(text)
            continue
        new = wrap_row(text, rows[r], r == 1)
        out += new if new is not None else [text]
    return out


def process(path):
    src = open(path).read()
    lines = src.split("\n")
    toks = list(tokenize.generate_tokens(io.StringIO(src).readline))
    # logical lines: (first physical line, last physical line), 1-based
    spans, start = [], None
    for t in toks:
        if t.type in (tokenize.COMMENT, tokenize.NL, tokenize.INDENT, tokenize.DEDENT, tokenize.ENDMARKER) and start is None:
            continue
        if start is None:
            start = t.start[0]
        if t.type == tokenize.NEWLINE:
            spans.append((start, t.end[0] if t.end[1] else t.end[0] - 1))
            start = None
    in_logical = {}
    for a, b in spans:
        for i in range(a, b + 1):
            in_logical[i] = (a, b)
    multiline_strings = set()
    for t in toks:
        if t.type == tokenize.STRING and t.start[0] != t.end[0]:
            for i in range(t.start[0], t.end[0] + 1):
                multiline_strings.add(i)
    out, i, n = [], 1, len(lines)
    done_spans = set()
    while i <= n:
        line = lines[i - 1]
        span = in_logical.get(i)
        if span and span not in done_spans and any(len(lines[j - 1]) > WIDTH + 2 for j in range(span[0],
                                                                                                span[1] + 1)) and not any(j in multiline_strings
                                                                                                                                   for j in range(span[0],
                                                                                                                                                  span[1]
                                                                                                                                                  + 1)):
            done_spans.add(span)
            try:
                new = wrap_logical(lines[span[0] - 1:span[1]], span[0])
            except (tokenize.TokenError, IndentationError):
                new = lines[span[0] - 1:span[1]]
            out += new
            i = span[1] + 1
            continue
        if span is None and len(line) > WIDTH + 2 and line.lstrip().startswith("#") and i not in multiline_strings:
            ind = line[:len(line) - len(line.lstrip())]
            pieces = wrap_comment(ind, line.strip())
            nxt = lines[i] if i < n else ""
            # the overflow flows into the following comment line of the same block (same indentation, ordinary text), which is then looked at in its turn
            if len(pieces) > 1 and in_logical.get(i + 1) is None and nxt.startswith(ind + "# ") and not nxt.startswith(ind + "#  ") and (i + 1) not in multiline_strings \
                    and not nxt[len(ind) + 2:].startswith(("-", "*", "=")):
                lines[i] = ind + "# " + pieces[-1][len(ind) + 2:] + " " + nxt[len(ind) + 2:]
                pieces = pieces[:-1]
            out += pieces
        else:
            out.append(line)
        i += 1
    new_src = "\n".join(out)
    try:
        same = ast.dump(ast.parse(src)) == ast.dump(ast.parse(new_src))
    except SyntaxError as e:
        print("%s: result does not parse (%s): left alone" % (path, e))
        open("/tmp/wrap_failed.py", "w").write(new_src)
        return
    if not same:
        print("%s: AST changed: left alone" % path)
        return
    open(path, "w").write(new_src)
    print("%s: %d -> %d lines, longest %d" % (path, len(lines), len(out), max(len(l) for l in out)))


for f in sys.argv[1:]:
    process(f)
