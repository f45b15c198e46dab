#!/usr/bin/env python3
"""Re-flow the prose of a Markdown file to <= WIDTH columns (default 118): paragraphs and list items are wrapped; tables, fenced code, headings and
blank lines are left alone.  `python tools/wrap_md.py FILE [WIDTH]` rewrites FILE in place and prints the lines that are still longer (tables)."""
import re
import sys
import textwrap


def main():
    path = sys.argv[1]; width = int(sys.argv[2]) if len(sys.argv) > 2 else 118
    lines = open(path, encoding="utf-8").read().split("\n")
    out = []; para = []; indent = ""; first = ""; fence = False

    def flush():
        nonlocal para, indent, first
        if para:
            text = " ".join(s.strip() for s in para)
            out.extend(textwrap.wrap(text, width=width, initial_indent=first, subsequent_indent=indent, break_long_words=False, break_on_hyphens=False))
            para = []

    for ln in lines:
        if ln.lstrip().startswith("```"):
            flush(); fence = not fence; out.append(ln); continue
        if fence or ln.startswith("|") or ln.startswith("#") or not ln.strip() or ln.startswith("<") or re.match(r"^\s*\|", ln):
            flush(); out.append(ln); continue
        m = re.match(r"^(\s*)([-*+]|\d+\.)\s+", ln)
        if m:   # a new list item
            flush(); first = m.group(0); indent = " " * len(m.group(0)); para = [ln[len(m.group(0)):]]; continue
        if not para:
            ws = re.match(r"^\s*", ln).group(0); first = ws; indent = ws
        para.append(ln)
    flush()
    open(path, "w", encoding="utf-8").write("\n".join(out))
    long = [(i + 1, len(l)) for i, l in enumerate(out) if len(l) > width + 2 and not l.startswith("|")]
    print(f"{path}: {len(out)} lines; {len(long)} prose / code lines longer than {width + 2}: {long[:8]}")


if __name__ == "__main__":
    main()
