#!/usr/bin/env python
"""Re-flows a markdown file to a maximum line width: paragraphs and list items are joined and wrapped again, and a table with a row wider than the
limit becomes a bullet list (one item per row: **first cell** -- `header`: cell; ...), which can be wrapped.  Code fences, headings and tables that fit
stay as they are.

    python tools/wrap_md.py DESIGN.md [--width 146]      (146 code points keep lines with a few 2-3 byte characters under 160 bytes)
"""
import argparse
import re
import textwrap

ITEM = re.compile(r"^(\s*)((?:[-*]|\d+\.)\s+)(.*)$")


def cells(row):
    return [c.strip() for c in re.split(r"(?<!\\)\|", row.strip().strip("|"))]


def wrap(text, width, first="", rest=""):
    return textwrap.fill(" ".join(text.split()), width=width, initial_indent=first, subsequent_indent=rest, break_long_words=False, break_on_hyphens=False)


def special(l):
    return (not l.strip()) or l.startswith(("#", "|", "```", "{", "@@")) or l.lstrip().startswith("```")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("path")
    ap.add_argument("--width", type=int, default=146)
    a = ap.parse_args()
    lines = open(a.path, encoding="utf-8").read().split("\n")
    out, i, fence = [], 0, False
    while i < len(lines):
        l = lines[i]
        if l.lstrip().startswith("```"):
            fence = not fence
            out.append(l)
            i += 1
            continue
        if fence or not l.strip() or l.startswith(("#", "{", "@@")):
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
                    parts = [("%s: %s" % (h, v)) if h else v for h, v in zip(head[1:], c[1:]) if v]
                    first = c[0] if c[0].startswith(("`", "*")) else "**%s**" % c[0] if c[0] else ""
                    out.append(wrap((first + " -- " if first else "") + "; ".join(parts), a.width, "- ", "  "))
            i = j
            continue
        m = ITEM.match(l)
        if m:  # a list item and its continuation lines (indented deeper than the marker, not items themselves)
            ind, marker, text = m.groups()
            j = i + 1
            cont = len(ind) + len(marker)
            while j < len(lines) and lines[j].strip() and not special(lines[j]) and not ITEM.match(lines[j]) and (len(lines[j]) - len(lines[j].lstrip())) >= min(cont, 2 + len(ind)):
                text += " " + lines[j].strip()
                j += 1
            out.append(wrap(text, a.width, ind + marker, " " * cont))
            i = j
            continue
        ind = re.match(r"^\s*", l).group(0)
        j, text = i + 1, l.strip()
        while j < len(lines) and not special(lines[j]) and not ITEM.match(lines[j]) and re.match(r"^\s*", lines[j]).group(0) == ind:
            text += " " + lines[j].strip()
            j += 1
        out.append(wrap(text, a.width, ind, ind))
        i = j
    open(a.path, "w", encoding="utf-8").write("\n".join(out))


if __name__ == "__main__":
    main()
