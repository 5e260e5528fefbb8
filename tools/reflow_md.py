#!/usr/bin/env python3
"""Reflow the paragraphs and bullet items of a markdown file to <= WIDTH characters (tables, headings, code fences and blank
lines untouched).  usage: tools/reflow_md.py FILE... [--width 116] [--check]"""
import re
import sys
import textwrap


def reflow(text: str, width: int) -> str:
    out, buf, fence = [], [], False

    def flush():
        if buf:
            joined = " ".join(l.strip() for l in buf)
            m = re.match(r"^(\s*(?:\* |- |\d+\. ))", buf[0])
            ind = " " * len(m.group(1)) if m else ""
            first = m.group(1) if m else ""
            body = joined[len(first.strip()) + 1:] if m else joined
            out.extend(textwrap.wrap(body, width, initial_indent=first, subsequent_indent=ind, break_long_words=False,
                                     break_on_hyphens=False))
            buf.clear()

    for l in text.split("\n"):
        if l.startswith("```"):
            flush(); fence = not fence; out.append(l); continue
        if fence or l.startswith("|") or l.startswith("#") or not l.strip():
            flush(); out.append(l)
        elif re.match(r"^\s*(\* |- |\d+\. )", l):
            flush(); buf.append(l)
        else:
            buf.append(l)
    flush()
    return "\n".join(out)


if __name__ == "__main__":
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    width = int(sys.argv[sys.argv.index("--width") + 1]) if "--width" in sys.argv else 116
    args = [a for a in args if not a.isdigit()]
    bad = 0
    for f in args:
        text = open(f, encoding="utf8").read()
        if "--check" in sys.argv:
            for i, l in enumerate(text.split("\n"), 1):
                if len(l) > 120:
                    print(f"{f}:{i}: {len(l)} columns"); bad += 1
        else:
            open(f, "w", encoding="utf8").write(reflow(text, width))
    sys.exit(1 if bad else 0)
