#!/usr/bin/env python3
"""Keeps a markdown file within a line width (140 columns): a table one of whose rows is wider becomes a list -- one item per row, '**first cell**' followed by
'header: cell' for the other cells -- and items / paragraphs are re-wrapped.  Code blocks and tables that fit stay as they are.   tools/wrap_md.py FILE [width]"""
import re
import sys
import textwrap

path = sys.argv[1]; width = int(sys.argv[2]) if len(sys.argv) > 2 else 140
lines = open(path).read().split("\n")
out = []; i = 0; in_code = False


def cells(row):
    return [c.strip() for c in row.strip().strip("|").split("|")]


def wrap(text, first, rest):
    return textwrap.wrap(text, width=width, initial_indent=first, subsequent_indent=rest, break_long_words=False, break_on_hyphens=False) or [first.rstrip()]


while i < len(lines):
    l = lines[i]
    if l.startswith("```"):
        in_code = not in_code; out.append(l); i += 1; continue
    if in_code:
        out.append(l); i += 1; continue
    if l.startswith("|") and i + 1 < len(lines) and re.match(r"^\|[-| :]+\|?$", lines[i + 1]):
        j = i
        while j < len(lines) and lines[j].startswith("|"):
            j += 1
        block = lines[i:j]
        if max(len(x) for x in block) <= width:
            out += block
        else:
            head = cells(block[0])
            for row in block[2:]:
                c = cells(row)
                parts = ["**%s**" % c[0]] if c and c[0] else []
                for h, v in zip(head[1:], c[1:]):
                    if v:
                        parts.append("%s: %s" % (h, v) if h else v)
                out += wrap(" — ".join(parts[:1]) + (" — " + "; ".join(parts[1:]) if len(parts) > 1 else ""), "* ", "  ")
        i = j; continue
    if len(l) > width and not l.startswith("#"):
        m = re.match(r"^(\s*(?:[*-]|\d+\.)\s+)(.*)$", l)
        if m:
            out += wrap(m.group(2), m.group(1), " " * len(m.group(1)))
        else:
            out += wrap(l, "", "")
        i += 1; continue
    out.append(l); i += 1
open(path, "w").write("\n".join(out))
over = [k + 1 for k, x in enumerate(out) if len(x) > width]
print("%s: %d lines, %d over %d columns %s" % (path, len(out), len(over), width, over[:10]))
