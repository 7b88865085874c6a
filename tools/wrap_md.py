#!/usr/bin/env python
"""Re-flows a markdown file to a maximum line width: paragraphs and list items are wrapped, and a table with a row wider than the limit becomes a
bullet list (one item per row: **first cell** -- `header`: cell; ...), which can be wrapped.  Code fences and short tables stay as they are.

    python tools/wrap_md.py DESIGN.md [--width 160]
"""
import argparse
import re
import textwrap


def cells(row):
    return [c.strip() for c in re.split(r"(?<!\\)\|", row.strip().strip("|"))]


def wrap(text, width, first="", rest=""):
    return textwrap.fill(text, width=width, initial_indent=first, subsequent_indent=rest, break_long_words=False, break_on_hyphens=False)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("path")
    ap.add_argument("--width", type=int, default=160)
    a = ap.parse_args()
    lines = open(a.path, encoding="utf-8").read().split("\n")
    out, i, fence = [], 0, False
    while i < len(lines):
        l = lines[i]
        if l.startswith("```"):
            fence = not fence
            out.append(l)
            i += 1
            continue
        if fence:
            out.append(l)
            i += 1
            continue
        if l.startswith("|"):
            j = i
            while j < len(lines) and lines[j].startswith("|"):
                j += 1
            tab = lines[i:j]
            if max(len(t) for t in tab) <= a.width or len(tab) < 3:
                out += tab
            else:
                head = cells(tab[0])
                for row in tab[2:]:
                    c = cells(row)
                    parts = []
                    for h, v in zip(head[1:], c[1:]):
                        if v:
                            parts.append(("%s: %s" % (h, v)) if h else v)
                    first = c[0] if c[0].startswith(("`", "*")) else "**%s**" % c[0] if c[0] else ""
                    out.append(wrap((first + " -- " if first else "") + "; ".join(parts), a.width, "- ", "  "))
            i = j
            continue
        m = re.match(r"^(\s*(?:[-*]|\d+\.)\s+)(.*)$", l)
        if len(l) > a.width and not l.startswith("#"):
            if m:
                out.append(wrap(m.group(2), a.width, m.group(1), " " * len(m.group(1))))
            else:
                ind = re.match(r"^\s*", l).group(0)
                out.append(wrap(l.strip(), a.width, ind, ind))
        else:
            out.append(l)
        i += 1
    open(a.path, "w", encoding="utf-8").write("\n".join(out))


if __name__ == "__main__":
    main()
